# L2 counters of the split-operand forward kernel (product library and variants): bash tools/pmc_fwd.sh [TAG ...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmcf
for t in "" "$@"; do
  if [ -n "$t" ]; then export NERF_HIP_LIB=$PWD/nerf_amd/libnerf_hip_$t.so; else unset NERF_HIP_LIB; fi
  name=${t:-product}
  i=0
  for ctr in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout -s KILL 120 rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/pmcf -o ${name}_$i -- python3 tools/time_fwd.py x3 2 > gpurun_out/pmcf/${name}_$i.log 2>&1 || { echo pass $i of $name failed; tail -3 gpurun_out/pmcf/${name}_$i.log; exit 1; }
    f=$(ls gpurun_out/pmcf/${name}_${i}_counter_collection.csv gpurun_out/pmcf/*/${name}_${i}_counter_collection.csv 2>/dev/null | head -1)
    echo "== $name: $ctr"; python3 profiles/summarize_pmc.py $f k_mlp_fwd_bf16x3 | tail -n +2
  done
done
