# A variant build of the library that differs in some source files:
#   bash tools/ab_lib.sh TAG "-DSWITCH ..." file1 [file2 ...]   (file names without .hip)
# -> nerf_amd/libnerf_hip_TAG.so (select with NERF_HIP_LIB); every other object comes from the product build.
set -e
cd "$(dirname "$0")/../nerf_amd/csrc"
make -s -j8
FLAGS="-O3 -std=c++20 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function -fno-gpu-flush-denormals-to-zero"
tag=$1; defs=$2; shift 2
pat=""
for f in "$@"; do /opt/rocm/bin/hipcc $FLAGS $defs -c $f.hip -o build/${f}__$tag.o & pat="$pat\|/$f\.o"; done; wait
objs=$(ls build/*.o | grep -v "__" | grep -v "XXXX$pat")
new=""; for f in "$@"; do new="$new build/${f}__$tag.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libnerf_hip_$tag.so $objs $new
rm -f $new
echo built ../libnerf_hip_$tag.so with $defs
