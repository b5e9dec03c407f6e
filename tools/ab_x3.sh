# Variant builds of the library that differ in the split-operand forward / backward-data kernels:
#   bash tools/ab_x3.sh TAG "-DSWITCH ..." [TAG2 "..."]...  -> nerf_amd/libnerf_hip_TAG.so (select with NERF_HIP_LIB)
set -e
cd "$(dirname "$0")/../nerf_amd/csrc"
make -s -j8
FLAGS="-O3 -std=c++20 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function -fno-gpu-flush-denormals-to-zero"
while [ $# -ge 2 ]; do
  tag=$1; defs=$2; shift 2
  for f in mlp_fwd_bf16x3 mlp_bwd_bf16x3; do /opt/rocm/bin/hipcc $FLAGS $defs -c $f.hip -o build/${f}_$tag.o & done; wait
  objs=$(ls build/*.o | grep -v "mlp_fwd_bf16x3\|mlp_bwd_bf16x3")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libnerf_hip_$tag.so $objs build/mlp_fwd_bf16x3_$tag.o build/mlp_bwd_bf16x3_$tag.o
  rm build/mlp_fwd_bf16x3_$tag.o build/mlp_bwd_bf16x3_$tag.o
  echo built ../libnerf_hip_$tag.so with $defs
done
