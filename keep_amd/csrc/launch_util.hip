// Host-side launch helpers shared by the kernel files of libkeep_hip (gfx950).
#include "common.h"

#include <map>
#include <mutex>

namespace {
std::mutex g_mu;
struct OptIn { signed char st[64] = {}; size_t bytes[64] = {}; };      // per device: 0 not asked, 1 granted, -1 refused
std::map<const void*, OptIn> g_optin;
int g_cus[64] = {};
}  // namespace

// Dynamic LDS beyond 64 KiB has to be asked for per kernel AND per device (hipFuncSetAttribute).  The answer is cached per (kernel, device) --
// keyed by the kernel's address, not by its type: every instantiation of a kernel template has the same function type -- and a refusal clears
// the sticky HIP error, so that the fallback launch the caller then makes is not reported as failed by the next hipGetLastError() and the
// refused attribute call is not repeated on every launch.
bool keep_lds_opt_in(const void* kernel, size_t bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return false; }
    std::lock_guard<std::mutex> lock(g_mu);
    if (dev < 0 || dev >= 64) {
        const bool ok = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
        if (!ok) (void)hipGetLastError();
        return ok;
    }
    OptIn& o = g_optin[kernel];
    if (o.st[dev] != 0 && o.bytes[dev] >= bytes) return o.st[dev] > 0;
    const bool ok = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    o.st[dev] = ok ? 1 : -1;
    o.bytes[dev] = bytes;
    return ok;
}

// number of CUs of the current device (the grid of the persistent kernels)
int keep_num_cus() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 256; }
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g_cus[dev]) {
        int n = 0;
        g_cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return g_cus[dev];
}
