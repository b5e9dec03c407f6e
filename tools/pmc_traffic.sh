# HBM traffic (FETCH_SIZE / WRITE_SIZE, one pass each) of every kernel of one 400x400 render: bash tools/pmc_traffic.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmct
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmct -o $c -- python3 bench.py --steps 1 --warmup 0 --res 400 --cpu-rays 0 --no-train > gpurun_out/pmct/$c.log 2>&1
  python3 profiles/summarize_pmc.py $(ls gpurun_out/pmct/${c}_counter_collection.csv gpurun_out/pmct/*/${c}_counter_collection.csv 2>/dev/null | head -1) k_ > gpurun_out/pmct/$c.txt
done
