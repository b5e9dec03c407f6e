# Builds a variant of the library beside the product one: bash tools/ab_build.sh "-DSOME_SWITCH" -> nerf_amd/libnerf_hip_B.so
# (select it with NERF_HIP_LIB=nerf_amd/libnerf_hip_B.so; the product library is rebuilt without the switch afterwards).
set -e
cd "$(dirname "$0")/../nerf_amd/csrc"
make -s clean && make -s -j8 EXTRA="$1" && cp ../libnerf_hip.so ../libnerf_hip_B.so
make -s clean && make -s -j8
ls -la ../libnerf_hip.so ../libnerf_hip_B.so
