"""bench.py starts its own ranks when it is asked for several GPUs outside torch.distributed.run (CPU, gloo)."""
import json
import os
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

RANK_SCRIPT = textwrap.dedent("""
    import json, os, sys
    import torch, torch.distributed as dist
    dist.init_process_group("gloo")
    t = torch.ones(1) * (dist.get_rank() + 1)
    dist.all_reduce(t)
    print("chatter from rank", dist.get_rank())
    if "--fail" in sys.argv and dist.get_rank() == 1:
        sys.exit(3)
    if dist.get_rank() == 0:
        print(json.dumps({"metric": "m", "sum": float(t), "world": dist.get_world_size(), "argv": sys.argv[1:]}))
    dist.destroy_process_group()
""")


def test_launcher_relays_rank0_line_and_status(tmp_path, capfd):
    import bench
    script = tmp_path / "rank.py"
    script.write_text(RANK_SCRIPT)
    rc = bench.launch_ranks(2, ["--gpus", "2", "--steps", "1"], script=str(script))
    out, err = capfd.readouterr()
    assert rc == 0
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1, out            # ONE JSON line on stdout; everything else went to stderr
    got = json.loads(lines[0])
    assert got["world"] == 2 and got["sum"] == 3.0 and got["argv"] == ["--gpus", "2", "--steps", "1"]
    assert "chatter from rank" in err
    rc = bench.launch_ranks(2, ["--fail"], script=str(script))
    capfd.readouterr()
    assert rc != 0


def test_main_self_launches_only_outside_a_rank(monkeypatch):
    import bench
    calls = []
    monkeypatch.setattr(bench, "launch_ranks", lambda n, argv, script=None: calls.append((n, list(argv))) or 7)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as exc:
        bench.main()
    assert exc.value.code == 7 and calls == [(4, ["--gpus", "4", "--steps", "2"])]
    # inside a rank whose world does not match --gpus: an error, not a second launch
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit) as exc:
        bench.main()
    assert "does not match" in str(exc.value.code) and len(calls) == 1


def test_pmc_traffic_file_is_consistent():
    import bench
    t = bench.pmc_traffic()
    for name in ("k_mlp_fwd_bf16x3", "k_mlp_fwd_x4"):
        e = t[name]
        per = (2 * e["fetch_kib_per_launch"] + e["write_kib_per_launch"]) * 1024 / e["points_per_launch"]
        assert abs(per - e["hbm_bytes_per_point"]) < 0.01 * per, (name, per)
        assert os.path.exists(os.path.join(ROOT, e["source"]))
