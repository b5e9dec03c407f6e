// Probes and the thread-local error string of libnerf_hip.so.
#include "common.hpp"
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>

namespace nerfhip {
char* err_buf() {
    static thread_local char buf[512] = "";
    return buf;
}

hipError_t lds_limit_once(const void* kernel, size_t bytes) {
    static std::mutex mu;
    static std::map<std::tuple<int, const void*>, size_t> done;   // (device, kernel) -> the limit it was given
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    auto it = done.find({dev, kernel});
    if (it != done.end() && it->second == bytes) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) done[{dev, kernel}] = bytes;
    return e;
}

int compute_units() {
    static std::mutex mu;
    static std::map<int, int> known;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    std::lock_guard<std::mutex> lock(mu);
    auto it = known.find(dev);
    if (it != known.end()) return it->second;
    int n_cu = 256;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
    known[dev] = n_cu;
    return n_cu;
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
