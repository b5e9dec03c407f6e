# Counters of the weight-gradient kernel alone (tools/time_dw.py): bash tools/pmc_dw.sh [n_pts]
n=${1:-1572864}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmcdw && \
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmcdw -o p1 -- python3 tools/time_dw.py $n 4 > gpurun_out/pmcdw/p1.log 2>&1 && \
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmcdw -o p2 -- python3 tools/time_dw.py $n 4 > gpurun_out/pmcdw/p2.log 2>&1 && \
timeout -s KILL 120 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum --output-format csv -d gpurun_out/pmcdw -o p3 -- python3 tools/time_dw.py $n 4 > gpurun_out/pmcdw/p3.log 2>&1
for f in p1 p2 p3; do python3 profiles/summarize_pmc.py $(ls gpurun_out/pmcdw/${f}_counter_collection.csv gpurun_out/pmcdw/*/${f}_counter_collection.csv 2>/dev/null | head -1) k_dw > gpurun_out/pmcdw/$f.txt; done; cat gpurun_out/pmcdw/p?.txt; tail -2 gpurun_out/pmcdw/p3.log
