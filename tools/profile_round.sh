# Regenerates the files under profiles/ (run from the repo root on the GPU box): bash tools/profile_round.sh r1
set -e
tag=${1:-r1}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/prof
timeout -k 10 500 python3 bench.py > gpurun_out/prof/${tag}_bench.json 2> gpurun_out/prof/bench.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o ${tag}_bench -- python3 bench.py --steps 2 --warmup 1 --cpu-rays 0 > gpurun_out/prof/bench_prof.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o ${tag}_train -- python3 bench.py --steps 1 --warmup 0 --res 200 --cpu-rays 0 --no-bf16 > gpurun_out/prof/train_prof.log 2>&1
ls gpurun_out/prof
