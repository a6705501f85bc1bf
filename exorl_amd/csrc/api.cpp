// Error plumbing and device queries of libexorl_hip.so.
#include <cstdarg>
#include <cstdio>

#include "common.h"

namespace exorl {
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace exorl

extern "C" const char* exorl_last_error(void) { return exorl::g_err; }
extern "C" int exorl_abi_version(void) { return EXORL_ABI_VERSION; }

extern "C" int exorl_device_info(char* name_out, int name_len, int* num_cus, int64_t* hbm_bytes) {
    int dev = 0;
    EXORL_CHECK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    EXORL_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    if (name_out && name_len > 0) snprintf(name_out, name_len, "%s (%s)", prop.name, prop.gcnArchName);
    if (num_cus) *num_cus = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return 0;
}
