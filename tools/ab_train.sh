# On the GPU box: train steps with the product library and with the variant, alternating: bash tools/ab_train.sh [n_rand] [steps]
n=${1:-8192}; k=${2:-10}
for rep in 1 2; do
  for v in A B; do
    if [ $v = B ]; then export NERF_HIP_LIB=$PWD/nerf_amd/libnerf_hip_B.so; else unset NERF_HIP_LIB; fi
    timeout -k 10 120 python3 tools/trace_train.py $n $k 2>&1 | grep -o "'rays_per_s': [0-9.]*, 'ms_per_step': [0-9.]*\|mlp_tflops_algorithmic': [0-9.]*" | tr '\n' ' ' || exit 1
    echo " <- $v"
  done
done
