// Probes and the thread-local error string of libnerf_hip.so.
#include "common.hpp"
#include <cstring>

namespace nerfhip {
char* err_buf() {
    static thread_local char buf[512] = "";
    return buf;
}
}  // namespace nerfhip

extern "C" {

int nerf_version(void) { return 1; }

const char* nerf_arch(void) { return "gfx950"; }

const char* nerf_last_error(void) { return nerfhip::err_buf(); }

int nerf_check_device(void) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return nerfhip::fail(NERF_E_LAUNCH, "hipGetDevice: %s", hipGetErrorString(e));
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return nerfhip::fail(NERF_E_LAUNCH, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return nerfhip::fail(NERF_E_ARCH, "device %d is %s; this library holds gfx950 code only", dev, prop.gcnArchName);
    return 0;
}

}  // extern "C"
