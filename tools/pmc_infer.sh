# Counters of the inference kernels of one 400x400 render (three rocprofv3 passes: HBM reads, HBM writes, SQ/clock):
# bash tools/pmc_infer.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmci
run() {  # name, counters
  timeout -s KILL 200 rocprofv3 --pmc $2 --output-format csv -d gpurun_out/pmci -o $1 -- python3 bench.py --steps 1 --warmup 0 --res 400 --cpu-rays 0 --no-train > gpurun_out/pmci/$1.log 2>&1
  python3 profiles/summarize_pmc.py $(ls gpurun_out/pmci/${1}_counter_collection.csv gpurun_out/pmci/*/${1}_counter_collection.csv 2>/dev/null | head -1) k_mlp > gpurun_out/pmci/$1.txt
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
cat gpurun_out/pmci/fetch.txt gpurun_out/pmci/write.txt gpurun_out/pmci/sq.txt
