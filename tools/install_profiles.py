"""Copies the files a profile round left under gpurun_out/ into profiles/ (run from the repo root, after
tools/profile_round.sh r3, the two train traces, tools/trace_gaps.py, tools/pmc_round.sh and the two gloo rehearsals)."""
import json
import shutil
import subprocess

KEEP = ("k_mlp_", "k_dw_x3", "k_dw_scales", "k_composite_fwd", "k_composite_bwd", "k_sample_fine", "k_sample_coarse", "k_reduce_regions",
        "k_adam", "k_gather_split2")
out = []
fetch, write = {}, {}
for f in ("infer_fetch", "infer_write", "infer_sq", "train_fetch", "train_write", "train_sq"):
    out.append("== " + f)
    on, name = False, None
    for line in open(f"gpurun_out/pmcr/{f}.txt"):
        if not line.startswith(" "):
            on, name = any(k in line for k in KEEP), line
        if on:
            out.append(line.rstrip("\n"))
            for key, store in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
                if f.startswith("infer") and line.strip().startswith(key):
                    store[name] = float(line.split("per dispatch")[1])
open("profiles/r3_pmc_mlp.txt", "w").write("\n".join(out) + "\n")
for a, b in (("r3_bench.json",) * 2, ("r3_bench_kernel_stats.csv",) * 2, ("r3_train_kernel_stats.csv",) * 2,
             ("r3_train8192_kernel_stats.csv",) * 2, ("r3_train1024_kernel_stats.csv",) * 2,
             ("r3_n2.json", "r3_bench_n2_gloo_rehearsal.json"), ("r3_n2_fail.json", "r3_bench_n2_gloo_rehearsal_rank1_fails.json")):
    shutil.copy("gpurun_out/prof/" + a, "profiles/" + b)
busy = subprocess.run(["python3", "tools/trace_gaps.py", "gpurun_out/prof/r3_train1024_kernel_trace.csv", "0.6", "0.85"],
                      capture_output=True, text=True).stdout
open("profiles/r3_train1024_busy.txt", "w").write(busy)
t = json.load(open("profiles/pmc_traffic.json"))
pts = 4096000     # a 400x400 render: 5 launches of 32000 rays x 256 / 2 points
for kernel, match in (("k_mlp_fwd_x4", "k_mlp_fwd_x4<true, false>"), ("k_mlp_fwd_f16x2", "k_mlp_fwd_f16x2<true, false>")):
    fe = next(v for k, v in fetch.items() if match in k)
    wr = next(v for k, v in write.items() if match in k)
    t[kernel] = {"hbm_bytes_per_point": round((2 * fe + wr) * 1024 / pts, 2), "fetch_kib_per_launch": round(fe),
                 "write_kib_per_launch": round(wr), "points_per_launch": pts, "source": "profiles/r3_pmc_mlp.txt",
                 "formula": "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 / points"}
json.dump(t, open("profiles/pmc_traffic.json", "w"), indent=1)
print({k: v["hbm_bytes_per_point"] for k, v in t.items()})
