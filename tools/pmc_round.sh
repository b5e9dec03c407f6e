# Counter passes of one round (run from the repo root on the GPU box): bash tools/pmc_round.sh
# HBM reads / HBM writes / SQ+clock for the inference kernels of one 400x400 render, then the same three for the
# training kernels of two 8192-ray steps.  Every pass is a run of its own; a failing pass ends the script.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmcr
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
summ() { python3 profiles/summarize_pmc.py $(ls gpurun_out/pmcr/${1}_counter_collection.csv gpurun_out/pmcr/*/${1}_counter_collection.csv 2>/dev/null | head -1) k_ > gpurun_out/pmcr/$1.txt; }
infer() { timeout -s KILL 200 rocprofv3 --pmc $2 --output-format csv -d gpurun_out/pmcr -o $1 -- python3 bench.py --steps 1 --warmup 0 --res 400 --cpu-rays 0 --no-train > gpurun_out/pmcr/$1.log 2>&1; summ $1; }
train() { timeout -s KILL 200 rocprofv3 --pmc $2 --output-format csv -d gpurun_out/pmcr -o $1 -- python3 tools/trace_train.py 8192 2 > gpurun_out/pmcr/$1.log 2>&1; summ $1; }
infer infer_fetch FETCH_SIZE
infer infer_write WRITE_SIZE
infer infer_sq "$SQ"
train train_fetch FETCH_SIZE
train train_write WRITE_SIZE
train train_sq "$SQ"
for f in infer_fetch infer_write infer_sq train_fetch train_write train_sq; do echo "== $f"; grep -A12 "k_mlp\|k_dw" gpurun_out/pmcr/$f.txt | head -80; done
