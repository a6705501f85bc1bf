// Fused Adam step (+ optional Polyak target update) over a net's flat parameter buffer:
//   torch.optim.Adam defaults as constructed at td3_bc.py:96-97 (single-tensor CPU arithmetic)
//   utils.soft_update_params, utils/utils.py:44-47 — fused into the critic's Adam pass: the target is not
//   read between the critic step (td3_bc.py:142) and the soft update (td3_bc.py:186-187).
// HBM-bound: 7 words/param (p,g,m,v read; p,m,v written) + 2 for the target (read+write); float4 streams.
#include "kernels.h"

namespace exorl {

typedef __bf16 bf16s_t;
__device__ __forceinline__ unsigned short to_bf16(float x) {
    bf16s_t b = (bf16s_t)x;
    return __builtin_bit_cast(unsigned short, b);
}

// Writes the derived copies (ShadowSpec) of the 4 parameters at flat index 4*i4.
__device__ __forceinline__ unsigned short to_bf16_lo(float x) { return to_bf16(x - __uint_as_float((unsigned)to_bf16(x) << 16)); }

// w1l / w0l: lo planes (split-bf16 mode) or null
__device__ __forceinline__ void write_shadows(int64_t i4, const float4& pv, const ShadowSpec& sh, float* w0t,
                                              unsigned short* w1b, unsigned short* w0b, unsigned short* w1l, unsigned short* w0l) {
    const int64_t idx = 4 * i4;
    const int64_t hh = (int64_t)sh.H * sh.H, w0n = (int64_t)sh.H * sh.in_dim;
    for (int t = 0; t < sh.n_heads; ++t) {
        if (w1b && idx >= sh.w1_off[t] && idx < sh.w1_off[t] + hh) {          // W1 is 4-aligned, hh % 4 == 0 for H % 2 == 0
            unsigned short* d = w1b + t * hh + (idx - sh.w1_off[t]);
            const float e[4] = {pv.x, pv.y, pv.z, pv.w};
            for (int q = 0; q < 4; ++q)
                if (idx + q < sh.w1_off[t] + hh) {
                    d[q] = to_bf16(e[q]);
                    if (w1l) w1l[t * hh + (idx - sh.w1_off[t]) + q] = to_bf16_lo(e[q]);
                }
            return;
        }
    }
    for (int t = 0; t < sh.n_trunks; ++t) {
        if (idx + 3 >= sh.w0_off[t] && idx < sh.w0_off[t] + w0n) {
            const float e[4] = {pv.x, pv.y, pv.z, pv.w};
            for (int q = 0; q < 4; ++q) {
                const int64_t l = idx + q - sh.w0_off[t];
                if (l >= 0 && l < w0n) {
                    w0t[t * w0n + (l % sh.in_dim) * sh.H + l / sh.in_dim] = e[q];
                    if (w0b) {
                        const int64_t Kp = (sh.in_dim + 31) / 32 * 32;
                        w0b[t * (int64_t)sh.H * Kp + (l / sh.in_dim) * Kp + l % sh.in_dim] = to_bf16(e[q]);
                        if (w0l) w0l[t * (int64_t)sh.H * Kp + (l / sh.in_dim) * Kp + l % sh.in_dim] = to_bf16_lo(e[q]);
                    }
                }
            }
            return;
        }
    }
}

__global__ __launch_bounds__(256) void refresh_shadows_kernel(const float* __restrict__ p, int64_t n4, ShadowSpec sh,
                                                              int target) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
        write_shadows(i, reinterpret_cast<const float4*>(p)[i], sh, target ? sh.t_w0t : sh.w0t, target ? sh.t_w1b : sh.w1b,
                      target ? sh.t_w0b : sh.w0b, target ? sh.t_w1l : sh.w1l, target ? sh.t_w0l : sh.w0l);
}

int refresh_shadows(const float* p, int64_t n, const ShadowSpec& sh, bool target, hipStream_t s) {
    const int64_t n4 = n / 4;
    int blocks = cdiv(n4, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(refresh_shadows_kernel, dim3(blocks), dim3(256), 0, s, p, n4, sh, target ? 1 : 0);
    EXORL_LAUNCH_CHECK();
    return 0;
}

template <bool DEV>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   float* __restrict__ target, int64_t n4, AdamConst cv,
                                                   const AdamConst* __restrict__ cp, ShadowSpec sh, int has_shadows,
                                                   unsigned long long* bump) {
    const AdamConst c = DEV ? *cp : cv;
    if (bump && blockIdx.x == 0 && threadIdx.x == 0) *bump += 1ull;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 pv = reinterpret_cast<float4*>(p)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 mv = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        adam_elem(pv.x, gv.x, mv.x, vv.x, c);
        adam_elem(pv.y, gv.y, mv.y, vv.y, c);
        adam_elem(pv.z, gv.z, mv.z, vv.z, c);
        adam_elem(pv.w, gv.w, mv.w, vv.w, c);
        reinterpret_cast<float4*>(p)[i] = pv;
        reinterpret_cast<float4*>(m)[i] = mv;
        reinterpret_cast<float4*>(v)[i] = vv;
        if (target) {
            float4 tv = reinterpret_cast<float4*>(target)[i];
            tv.x = polyak(pv.x, tv.x, c.tau, c.one_minus_tau);
            tv.y = polyak(pv.y, tv.y, c.tau, c.one_minus_tau);
            tv.z = polyak(pv.z, tv.z, c.tau, c.one_minus_tau);
            tv.w = polyak(pv.w, tv.w, c.tau, c.one_minus_tau);
            reinterpret_cast<float4*>(target)[i] = tv;
            if (has_shadows) write_shadows(i, tv, sh, sh.t_w0t, sh.t_w1b, sh.t_w0b, sh.t_w1l, sh.t_w0l);
        }
        if (has_shadows) write_shadows(i, pv, sh, sh.w0t, sh.w1b, sh.w0b, sh.w1l, sh.w0l);
    }
}

int adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
              int64_t t, float* target, float tau, hipStream_t s) {
    EXORL_REQUIRE(n % 4 == 0, "adam_step: n=%lld must be a multiple of 4 (flat buffers are padded)", (long long)n);
    EXORL_REQUIRE(t >= 1, "adam_step: step count t=%lld must be >= 1", (long long)t);
    // Python-double scalar math of torch's _single_tensor_adam, then cast to fp32 like the tensor ops do
    const double b1d = dec7(b1), b2d = dec7(b2);
    AdamConst c;
    fill_adam_const(c, pow(b1d, (double)t), pow(b2d, (double)t), dec7(lr), b1d, b2d, dec7(eps), dec7(tau));
    const int64_t n4 = n / 4;
    int blocks = cdiv(n4, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL((adam_kernel<false>), dim3(blocks), dim3(256), 0, s, p, g, m, v, target, n4, c, (const AdamConst*)nullptr,
                       ShadowSpec{}, 0, (unsigned long long*)nullptr);
    EXORL_LAUNCH_CHECK();
    return 0;
}

int adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const AdamConst* c_dev, float* target,
                  const ShadowSpec* shadows, hipStream_t s, uint64_t* bump) {
    EXORL_REQUIRE(n % 4 == 0, "adam_step_dev: n must be a multiple of 4");
    const int64_t n4 = n / 4;
    int blocks = cdiv(n4, 256);
    if (blocks > 2048) blocks = 2048;
    AdamConst dummy{};
    hipLaunchKernelGGL((adam_kernel<true>), dim3(blocks), dim3(256), 0, s, p, g, m, v, target, n4, dummy, c_dev,
                       shadows ? *shadows : ShadowSpec{}, shadows ? 1 : 0, reinterpret_cast<unsigned long long*>(bump));
    EXORL_LAUNCH_CHECK();
    return 0;
}

__global__ void step_begin_kernel(StepState* st, int advance_replay) {
    if (threadIdx.x == 0 && blockIdx.x == 0) step_begin_device(st, advance_replay);
}

__global__ void set_device_float_kernel(float* dst, float value) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *dst = value;
}

// stream-ordered scalar write (kernel argument, no host buffer whose lifetime would matter)
int set_device_float(float* dst, float value, hipStream_t s) {
    hipLaunchKernelGGL(set_device_float_kernel, dim3(1), dim3(64), 0, s, dst, value);
    EXORL_LAUNCH_CHECK();
    return 0;
}

int step_begin(StepState* st, int advance_replay, hipStream_t s) {
    hipLaunchKernelGGL(step_begin_kernel, dim3(1), dim3(64), 0, s, st, advance_replay);
    EXORL_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void soft_update_kernel(const float* __restrict__ p, float* __restrict__ target,
                                                          int64_t n, float tau, float one_minus_tau) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        target[i] = polyak(p[i], target[i], tau, one_minus_tau);
}

int soft_update(const float* p, float* target, int64_t n, float tau, hipStream_t s) {
    int blocks = cdiv(n, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(soft_update_kernel, dim3(blocks), dim3(256), 0, s, p, target, n, (float)dec7(tau), (float)(1.0 - dec7(tau)));
    EXORL_LAUNCH_CHECK();
    return 0;
}

}  // namespace exorl

extern "C" int exorl_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                               float beta2, float eps, int64_t t, float* target, float tau, void* stream) {
    return exorl::adam_step(p, g, m, v, n, lr, beta1, beta2, eps, t, target, tau, exorl::as_stream(stream));
}
extern "C" int exorl_soft_update(const float* p, float* target, int64_t n, float tau, void* stream) {
    return exorl::soft_update(p, target, n, tau, exorl::as_stream(stream));
}
