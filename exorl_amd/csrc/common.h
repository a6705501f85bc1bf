// Shared helpers for libexorl_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/exorl_hip.h"

namespace exorl {

void set_error(const char* fmt, ...);

#define EXORL_CHECK_HIP(expr)                                                                  \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            ::exorl::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return 1;                                                                          \
        }                                                                                      \
    } while (0)

#define EXORL_REQUIRE(cond, ...)                         \
    do {                                                 \
        if (!(cond)) {                                   \
            ::exorl::set_error(__VA_ARGS__);             \
            return 2;                                    \
        }                                                \
    } while (0)

#define EXORL_TRY(expr)              \
    do {                             \
        int _rc = (expr);            \
        if (_rc != 0) return _rc;    \
    } while (0)

#define EXORL_LAUNCH_CHECK() EXORL_CHECK_HIP(hipGetLastError())

// The ABI carries hyper-parameters as fp32, the reference holds them as Python floats (doubles): 0.9f is 0.89999998, not 0.9.
// Recover the short decimal the caller meant (7 significant digits identify an fp32 value's intended literal).
inline double dec7(float x) {
    char buf[40];
    snprintf(buf, sizeof(buf), "%.7g", (double)x);
    return strtod(buf, nullptr);
}
inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int64_t round_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }
inline int cdiv(int64_t a, int64_t b) { return static_cast<int>((a + b - 1) / b); }

// ---- device helpers ---------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// Philox4x32-10 (the build's own counter-based stream; restated in oracle/replay.py for parity).
struct Philox {
    static constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    __host__ __device__ static inline void round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
        const uint64_t p0 = (uint64_t)M0 * c[0];
        const uint64_t p1 = (uint64_t)M1 * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    }
    __host__ __device__ static inline void gen(uint32_t (&c)[4], uint64_t key) {
        uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            round(c, k0, k1);
            k0 += W0; k1 += W1;
        }
    }
};

}  // namespace exorl
