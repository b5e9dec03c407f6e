# Counter passes over one forward kernel timed by tools/time_fwd.py: bash tools/pmc_kernel.sh <which: x2only|x3|x4|save> <kernel name filter> [TAG ...]
# (TAG: variant libraries nerf_amd/libnerf_hip_TAG.so; the product library always runs first).  One rocprofv3 run per pass.
which=$1; filt=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmck
for t in "" "$@"; do
  if [ -n "$t" ]; then export NERF_HIP_LIB=$PWD/nerf_amd/libnerf_hip_$t.so; else unset NERF_HIP_LIB; fi
  name=${t:-product}_$which
  i=0
  for ctr in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
             "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    i=$((i+1))
    timeout -s KILL 120 rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/pmck -o ${name}_$i -- python3 tools/time_fwd.py $which 2 > gpurun_out/pmck/${name}_$i.log 2>&1 || { echo pass $i of $name failed; tail -3 gpurun_out/pmck/${name}_$i.log; exit 1; }
    f=$(ls gpurun_out/pmck/${name}_${i}_counter_collection.csv gpurun_out/pmck/*/${name}_${i}_counter_collection.csv 2>/dev/null | head -1)
    echo "== $name: $ctr"; python3 profiles/summarize_pmc.py $f $filt | tail -n +2
  done
done
