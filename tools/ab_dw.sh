# Variant builds of the library that differ only in mlp_dw_x3.hip: bash tools/ab_dw.sh TAG "-DSWITCH ..." [TAG2 "..."]...
# -> nerf_amd/libnerf_hip_TAG.so (select with NERF_HIP_LIB); the product library is left untouched.
set -e
cd "$(dirname "$0")/../nerf_amd/csrc"
make -s -j8
while [ $# -ge 2 ]; do
  tag=$1; defs=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function -fno-gpu-flush-denormals-to-zero $defs -c mlp_dw_x3.hip -o build/mlp_dw_x3_$tag.o
  objs=$(ls build/*.o | grep -v "mlp_dw_x3")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libnerf_hip_$tag.so $objs build/mlp_dw_x3_$tag.o
  rm build/mlp_dw_x3_$tag.o
  echo built ../libnerf_hip_$tag.so with $defs
done
