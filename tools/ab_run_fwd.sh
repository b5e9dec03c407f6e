# On the GPU box: forward kernel timings of the product library and variants, interleaved twice:
#   bash tools/ab_run_fwd.sh [x3|save] TAG [TAG2 ...]
which=${1:-x3}; shift
for rep in 1 2; do
  for t in "" "$@"; do
    if [ -n "$t" ]; then export NERF_HIP_LIB=$PWD/nerf_amd/libnerf_hip_$t.so; else unset NERF_HIP_LIB; fi
    timeout -k 10 100 python3 tools/time_fwd.py $which 10 2>/dev/null | tail -1 || exit 1
  done
done
