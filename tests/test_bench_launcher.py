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


def test_guard_prints_the_headline_once_and_only_once():
    import time
    import bench
    written, exits = [], []
    g = bench.Guard(0, write=written.append, exit_fn=exits.append, catch_sigterm=False, poll_s=0.02)
    g.abort("too early")                       # no headline yet: nothing to print, still leaves
    assert written == [] and exits == [1]
    out = {"metric": "m", "value": 1.0}
    g.set_headline(out)
    out["late_extra"] = {"x": 1}               # added after the snapshot: must not appear in an aborted line
    g.arm("slow_leg", 0.05)
    time.sleep(0.5)                            # the watchdog fires
    assert len(written) == 1 and exits == [1, 1]
    line = json.loads(written[0])
    assert line["value"] == 1.0 and "late_extra" not in line and line["extras_aborted"].startswith("slow_leg")
    g.finish(out)                              # the abort already printed: no second line
    assert len(written) == 1
    # the ordinary course: disarmed legs never fire, finish prints the complete line
    written2 = []
    g2 = bench.Guard(0, write=written2.append, exit_fn=exits.append, catch_sigterm=False, poll_s=0.02)
    g2.set_headline(out)
    g2.arm("quick_leg", 0.05)
    g2.disarm()
    time.sleep(0.2)
    g2.finish(out)
    assert len(written2) == 1 and json.loads(written2[0])["late_extra"] == {"x": 1} and exits == [1, 1]
    # ranks other than 0 never print
    written3 = []
    g3 = bench.Guard(1, write=written3.append, exit_fn=exits.append, catch_sigterm=False, rank_grace=0.0, poll_s=0.02)
    g3.set_headline(out)
    g3.abort("x")
    g3.finish(out)
    assert written3 == []


GUARDED_RANKS = textwrap.dedent("""
    import json, os, sys, time
    from datetime import timedelta
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    import bench
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    guard = bench.Guard(rank, rank_grace=2.0)
    dist.init_process_group("gloo", timeout=timedelta(seconds=60))
    legs = bench.LegSync(world, None, dist.distributed_c10d._get_default_store())
    out = {"metric": "m", "value": 42.0}
    guard.set_headline(out)
    mode = sys.argv[1]

    def summed():
        t = torch.ones(1) * (rank + 1)
        dist.all_reduce(t)
        return {"sum": float(t)}

    def cannot_prepare():
        if rank == 1:
            raise RuntimeError("no inputs on rank 1")
        return None

    def broken(_state=None):
        if rank == 1:
            if mode == "die":
                os._exit(9)
            raise RuntimeError("rank 1 fails inside the leg")   # and never joins the collective below
        return summed()

    legs.run(out, guard, "good", summed, collective=True, seconds=20)
    legs.run(out, guard, "unprepared", lambda s: summed(), prepare=cannot_prepare, collective=True, seconds=20)
    legs.run(out, guard, "good_again", summed, collective=True, seconds=20)
    if mode in ("raise", "die"):
        legs.run(out, guard, "stress", broken, collective=True, seconds=3)
        legs.run(out, guard, "never_reached", summed, collective=True, seconds=20)
    guard.finish(out)
    dist.destroy_process_group()
""") % ROOT


@pytest.mark.parametrize("mode", ["clean", "raise", "die"])
def test_a_rank_that_fails_inside_a_collective_leg_cannot_take_the_headline_with_it(tmp_path, capfd, mode):
    """Two gloo ranks: every rank agrees to skip a leg one of them cannot prepare; when rank 1 raises inside a leg (or
    dies) and leaves rank 0 alone in the collective, the run ends within the leg's deadline, rank 0's headline is on
    stdout with `extras_aborted`, and the status is not 0."""
    import time
    import bench
    script = tmp_path / "ranks.py"
    script.write_text(GUARDED_RANKS)
    t0 = time.monotonic()
    rc = bench.launch_ranks(2, [mode], script=str(script))
    took = time.monotonic() - t0
    out, err = capfd.readouterr()
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1, (out, err[-2000:])
    got = json.loads(lines[0])
    assert got["value"] == 42.0
    if mode == "clean":
        assert rc == 0 and got["good"] == {"sum": 3.0} and got["good_again"] == {"sum": 3.0}
        assert "skipped" in got["unprepared"] and "extras_aborted" not in got
    else:
        assert rc != 0 and got["extras_aborted"].startswith("stress") and "never_reached" not in got
        assert took < 50.0, took
