// Fused layer kernels around the H x H MFMA GEMMs — second generation.
// They replace chains of generic kernels (small-K GEMM -> LayerNorm -> tanh; head dgrad -> column sums; LN
// backward -> column sums -> small-N GEMMs) whose 16-workgroup column reductions and 16-tile skinny GEMMs left
// the 256 CUs idle.  Reference arithmetic (file:line in /root/reference):
//   trunk  Linear(in,H)+LayerNorm(H)+Tanh   agents/offline_learning/td3_bc.py:16-17,38-39; unsupervised_learning/ddpg.py:48-49,91-93
//   head   Linear(H,n)                       td3_bc.py:20,41; ddpg.py:62,107
//
// Geometry: a trunk workgroup owns R=16 whole rows (all H<=1024 columns: 4 per thread), so LayerNorm statistics
// and the dX reduction stay on chip; the first-layer weight is read through a transposed shadow W0T[in][H]
// (coalesced: lane = column), x rows sit in LDS transposed so one ds_read_b128 feeds 4 rows.
// Parameter gradients that reduce over the batch are written as per-workgroup partial rows
// P[chunk][...] and summed in a fixed order by finalize_grads (deterministic, no atomics).
#include "kernels.h"

namespace exorl {

constexpr int MAX_IN = 256;       // first-layer fan-in limit of the fused trunk kernels
constexpr float LN_EPS2 = 1e-5f;
typedef __bf16 bf16_t;

__device__ __forceinline__ unsigned short f2bf(float x) {
    bf16_t b = (bf16_t)x;
    return __builtin_bit_cast(unsigned short, b);
}
// lo plane of the split-bf16 representation x = hi + lo, hi = bf16(x)
__device__ __forceinline__ unsigned short f2bf_lo(float x) { return f2bf(x - __uint_as_float((unsigned)f2bf(x) << 16)); }
__device__ __forceinline__ ushort4 f4_to_bf4(const float4& v) { ushort4 q; q.x = f2bf(v.x); q.y = f2bf(v.y); q.z = f2bf(v.z); q.w = f2bf(v.w); return q; }
__device__ __forceinline__ ushort4 f4_to_bf4_lo(const float4& v) { ushort4 q; q.x = f2bf_lo(v.x); q.y = f2bf_lo(v.y); q.z = f2bf_lo(v.z); q.w = f2bf_lo(v.w); return q; }

// ------------------------------------------------------------------------------------------------
// trunk forward: h = tanh(LN(x W0^T + b0) * g + beta).
// 512 threads = 8 waves, one row per wave, lane owns columns 4*lane + 256*i (+0..3), i < 4 (H <= 1024, H % 4 == 0).
// W0T is staged through LDS in 16-row k-chunks (64 KB: two workgroups per CU) and read back as conflict-free
// ds_read_b128; LayerNorm statistics are wave-local (no workgroup barrier on the critical path).
constexpr int TF_ROWS = 8;
constexpr int TF_LDS_FLOATS = 35 * 1024;      // 140 KB for the W0T chunk: all of a (O+A <= 35) x 1024 first layer at once

// tanh(x) = 1 - 2/(1 + e^{2x}) on the hardware exp2/rcp units (abs. error ~1e-7; used in the bf16 fast mode, where
// the 1024^2 tanh evaluations of libm quality would cost more VALU time than the layer's FMAs)
__device__ __forceinline__ float tanh_fast(float x) {
    const float e = __expf(2.0f * x);
    return 1.0f - 2.0f * __frcp_rn(1.0f + e);      // correctly rounded on purpose: v_rcp_f32's 1 ulp flipped one of CRR's indicator weights in the full-size parity run
}

template <bool FAST>
__global__ __launch_bounds__(512) void trunk_fwd_kernel(const float* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ W0T, const float* __restrict__ b0,
                                                        const float* __restrict__ gain, const float* __restrict__ beta,
                                                        float* __restrict__ h, float* __restrict__ xhat,
                                                        float* __restrict__ rstd, unsigned short* __restrict__ hb,
                                                        unsigned short* __restrict__ xhb, int rows, int in_dim, int H,
                                                        int64_t astride, int64_t pstride, int64_t tstride, int TF_KC) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ws = smem;                          // [TF_KC][H]
    float* xs = smem + TF_KC * H;              // [TF_ROWS][MAX_IN]
    const int net = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = blockIdx.x * TF_ROWS + wave;
    const bool live = row < rows;
    for (int i = tid; i < TF_ROWS * in_dim; i += 512) {
        const int r = i / in_dim, k = i % in_dim;
        const int rr = blockIdx.x * TF_ROWS + r;
        xs[r * MAX_IN + k] = rr < rows ? x[(int64_t)rr * ldx + k] : 0.f;
    }
    const float* Wt = W0T + net * tstride;
    const int H4 = H >> 2;
    float4 z[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c4 = lane + 64 * i;
        z[i] = c4 < H4 ? reinterpret_cast<const float4*>(b0 + net * pstride)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    auto accumulate = [&](int kbeg, int kend, int k0) {        // z += x[k0+k] * W0T[k0+k][:] for staged k in [kbeg, kend)
#pragma unroll 5
        for (int k = kbeg; k < kend; ++k) {
            const float xv = xs[wave * MAX_IN + k0 + k];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c4 = lane + 64 * i;
                if (c4 < H4) {
                    const float4 w = reinterpret_cast<const float4*>(ws + k * H)[c4];
                    z[i].x += xv * w.x; z[i].y += xv * w.y; z[i].z += xv * w.z; z[i].w += xv * w.w;
                }
            }
        }
    };
    for (int k0 = 0; k0 < in_dim; k0 += TF_KC) {
        const int kc = in_dim - k0 < TF_KC ? in_dim - k0 : TF_KC;
        // The chunk is staged in two halves: the global loads of the second half are in flight (8 x 16 B per thread)
        // while the first half is being consumed, so only one memory round trip is exposed per chunk.
        const float4* gsrc = reinterpret_cast<const float4*>(Wt + (int64_t)k0 * H);
        const int kh = (kc + 1) >> 1;
        const int n4a = kh * H4, n4 = kc * H4;
        __syncthreads();                                       // previous chunk fully consumed
        for (int i0 = tid; i0 < n4a; i0 += 8 * 512) {
            float4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = i0 + u * 512 < n4a ? gsrc[i0 + u * 512] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + u * 512 < n4a) reinterpret_cast<float4*>(ws)[i0 + u * 512] = t[u];
        }
        float4 t2[8];                                          // second half (fits one batch for kc <= 32 at H = 1024)
        const bool one_batch = n4 - n4a <= 8 * 512;
        if (one_batch) {
#pragma unroll
            for (int u = 0; u < 8; ++u) t2[u] = n4a + tid + u * 512 < n4 ? gsrc[n4a + tid + u * 512] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        accumulate(0, kh, k0);
        if (one_batch) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (n4a + tid + u * 512 < n4) reinterpret_cast<float4*>(ws)[n4a + tid + u * 512] = t2[u];
        } else {
            for (int i = n4a + tid; i < n4; i += 512) reinterpret_cast<float4*>(ws)[i] = gsrc[i];
        }
        __syncthreads();
        accumulate(kh, kc, k0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (lane + 64 * i < H4) s += (z[i].x + z[i].y) + (z[i].z + z[i].w);
    const float mean = wave_sum(s) / (float)H;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (lane + 64 * i < H4) {
            z[i].x -= mean; z[i].y -= mean; z[i].z -= mean; z[i].w -= mean;
            s2 += (z[i].x * z[i].x + z[i].y * z[i].y) + (z[i].z * z[i].z + z[i].w * z[i].w);
        }
    }
    const float rs = 1.0f / sqrtf(wave_sum(s2) / (float)H + LN_EPS2);
    if (!live) return;
    const int64_t o = net * astride + (int64_t)row * H;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c4 = lane + 64 * i;
        if (c4 >= H4) continue;
        const float4 g = reinterpret_cast<const float4*>(gain + net * pstride)[c4];
        const float4 be = reinterpret_cast<const float4*>(beta + net * pstride)[c4];
        const float4 xh = make_float4(z[i].x * rs, z[i].y * rs, z[i].z * rs, z[i].w * rs);
        float4 hv;
        if constexpr (FAST)
            hv = make_float4(tanh_fast(xh.x * g.x + be.x), tanh_fast(xh.y * g.y + be.y), tanh_fast(xh.z * g.z + be.z),
                             tanh_fast(xh.w * g.w + be.w));
        else
            hv = make_float4(tanhf(xh.x * g.x + be.x), tanhf(xh.y * g.y + be.y), tanhf(xh.z * g.z + be.z),
                             tanhf(xh.w * g.w + be.w));
        if (h) reinterpret_cast<float4*>(h + o)[c4] = hv;
        if (xhat) reinterpret_cast<float4*>(xhat + o)[c4] = xh;
        if (hb) {
            ushort4 q;
            q.x = f2bf(hv.x); q.y = f2bf(hv.y); q.z = f2bf(hv.z); q.w = f2bf(hv.w);
            reinterpret_cast<ushort4*>(hb + o)[c4] = q;
        }
        if (xhb) {
            ushort4 q;
            q.x = f2bf(xh.x); q.y = f2bf(xh.y); q.z = f2bf(xh.z); q.w = f2bf(xh.w);
            reinterpret_cast<ushort4*>(xhb + o)[c4] = q;
        }
    }
    if (rstd && lane == 0) rstd[net * (int64_t)rows + row] = rs;
}

// ------------------------------------------------------------------------------------------------
// trunk forward on the matrix cores (bf16 fast mode): z^T = W0 x^T as v_mfma_f32_16x16x32_bf16 tiles, then LayerNorm + tanh
// in registers. The wave-per-row kernel above re-stages the whole first-layer weight (120 KB) through LDS for every 8 rows and
// spends its time waiting on that; here a workgroup owns 16 rows, each of its 8 waves a 1/8 slab of the H columns, and the
// weight fragments go L2 -> registers directly (16 B per lane per tile, from a bf16 shadow W0b[H][Kp] kept by the optimiser).
//   A (16x32) = W0b rows: lane l holds W0b[col0 + (l&15)][8(l>>4) .. +7]        B (32x16) = x^T: lane l holds x[row0 + (l&15)][8(l>>4) .. +7]
//   C lane l, reg i = z[row0 + (l&15)][col0 + 4(l>>4) + i]  ->  4 consecutive columns of one row per lane (8-byte bf16 stores)
// LayerNorm statistics: lane-local over its T*4 columns, 2 shuffles across the 4 lanes of a row, 8 wave partials through LDS.
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4_t;

// X3 (split-bf16 mode): x and W0 enter as hi + lo pairs, z = hi*hi + hi*lo + lo*hi (fp32-grade first layer), and h / xhat
// leave as hi + lo planes.
// slab row stride in bf16 elements: + 8 (16 B) puts row r of the 8-byte transposing writes 4 banks after row r - 1, so the 32 lanes a
// write instruction retires together (16 rows x 2 column quads) hit 64 distinct banks for T = 8; + 16 made rows r and r + 8 collide
// (SQ_LDS_BANK_CONFLICT was 64 % of the kernel's LDS cycles)
#define TF16_LD(T) (16 * (T) + 8)
template <int T, bool X3>     // T = H / 128 column tiles per wave
__global__ __launch_bounds__(512) void trunk_fwd16_kernel(const TrunkBatch tb, int64_t ldx, int rows, int in_dim, int Kp) {
    constexpr int H = T * 128;
    __shared__ float red[2][8][16];
    __shared__ __attribute__((aligned(16))) unsigned short stage[8][16][TF16_LD(T)];      // per-wave 16 x 16T output slab (+16 B row pad)
    const TrunkItem& it = tb.it[blockIdx.y];
    const float* __restrict__ x = it.x;
    const float* __restrict__ b0 = it.b0;
    const float* __restrict__ gain = it.gain;
    const float* __restrict__ beta = it.beta;
    float* __restrict__ rstd = it.rstd;
    unsigned short* __restrict__ hb = it.hb;
    unsigned short* __restrict__ xhb = it.xhb;
    const unsigned short* __restrict__ W0b = it.W0b;
    constexpr int net = 0;
    constexpr int64_t astride = 0, pstride = 0, wstride = 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rn = lane & 15, kq = lane >> 4;                 // batch row within the tile / k-group (and column quad of C)
    const int row = blockIdx.x * 16 + rn;
    const bool live = row < rows;
    const unsigned short* Wb = W0b + net * wstride;
    const int col0 = wave * (H / 8);
    f32x4_t acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // bias / gain / beta of this lane's columns are fetched NOW, with the inputs and weight fragments, instead of behind the MFMAs and
    // behind the second LayerNorm barrier (two more L2 round trips on a kernel that is one latency chain)
    float4 bias4[T], gain4[T], beta4[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int c = col0 + t * 16 + 4 * kq;
        bias4[t] = *reinterpret_cast<const float4*>(b0 + net * pstride + c);
        gain4[t] = *reinterpret_cast<const float4*>(gain + net * pstride + c);
        beta4[t] = *reinterpret_cast<const float4*>(beta + net * pstride + c);
    }
    const int64_t xrow = (int64_t)(live ? row : 0) * ldx;          // unconditional loads (see head_fwd4): index clamped, value masked
    for (int k0 = 0; k0 < Kp; k0 += 32) {
        bf16x8_t bx, bxl;
        const int kb = k0 + 8 * kq;
        float xs[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) xs[j] = x[xrow + (kb + j < in_dim ? kb + j : 0)];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xv = (live && kb + j < in_dim) ? xs[j] : 0.f;
            bx[j] = (__bf16)xv;
            if constexpr (X3) bxl[j] = (__bf16)(xv - (float)bx[j]);
        }
        bf16x8_t aw[T];
#pragma unroll
        for (int t = 0; t < T; ++t)
            aw[t] = *reinterpret_cast<const bf16x8_t*>(Wb + (int64_t)(col0 + t * 16 + rn) * Kp + kb);
        if constexpr (X3) {
            bf16x8_t awl[T];                       // all lo-plane fragments in flight together (the compiler otherwise waits on each)
#pragma unroll
            for (int t = 0; t < T; ++t) awl[t] = *reinterpret_cast<const bf16x8_t*>(it.W0l + (int64_t)(col0 + t * 16 + rn) * Kp + kb);
            __builtin_amdgcn_sched_barrier(0);
            f32x4_t cross[T];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                cross[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[t], bxl, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                cross[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(awl[t], bx, cross[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < T; ++t) acc[t] += cross[t];
        }
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[t], bx, acc[t], 0, 0, 0);
    }
    // + bias; row statistics
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const float4 b = bias4[t];
        acc[t][0] += b.x; acc[t][1] += b.y; acc[t][2] += b.z; acc[t][3] += b.w;
        s += (acc[t][0] + acc[t][1]) + (acc[t][2] + acc[t][3]);
    }
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (kq == 0) red[0][wave][rn] = s;
    __syncthreads();
    float mean = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) mean += red[0][w][rn];
    mean *= 1.0f / (float)H;
    float s2 = 0.f;
#pragma unroll
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc[t][i] -= mean; s2 += acc[t][i] * acc[t][i]; }
    }
    s2 += __shfl_xor(s2, 16, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (kq == 0) red[1][wave][rn] = s2;
    __syncthreads();
    float var = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) var += red[1][w][rn];
    const float rs = 1.0f / sqrtf(var * (1.0f / (float)H) + LN_EPS2);
    float4 hv[T], xv[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const float4 g = gain4[t];
        const float4 be = beta4[t];
        xv[t] = make_float4(acc[t][0] * rs, acc[t][1] * rs, acc[t][2] * rs, acc[t][3] * rs);
        hv[t] = make_float4(tanh_fast(xv[t].x * g.x + be.x), tanh_fast(xv[t].y * g.y + be.y), tanh_fast(xv[t].z * g.z + be.z),
                            tanh_fast(xv[t].w * g.w + be.w));
    }
    // The MFMA C layout gives a lane 4 consecutive columns of one row (8 B of bf16); stored directly that is 32 B per row per
    // instruction. Each wave transposes its 16 x 16T slab through LDS instead, so a store instruction writes whole 32T-byte row slabs.
    unsigned short* mine = &stage[wave][0][0];
    auto flush = [&](unsigned short* __restrict__ plane, const float4 (&v)[T], bool lo) {
#pragma unroll
        for (int t = 0; t < T; ++t)
            *reinterpret_cast<ushort4*>(mine + rn * TF16_LD(T) + t * 16 + 4 * kq) = lo ? f4_to_bf4_lo(v[t]) : f4_to_bf4(v[t]);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int idx = lane; idx < 32 * T; idx += 64) {
            const int r = idx / (2 * T), ch = idx % (2 * T);
            const uint4 q = *reinterpret_cast<const uint4*>(mine + r * TF16_LD(T) + 8 * ch);
            const int grow = blockIdx.x * 16 + r;
            // (plain stores on purpose: streaming/non-temporal stores shortened this kernel by 4 us and lengthened its consumers,
            // the H x H GEMM and LayerNorm backward, by 5 + 2 us — the slab is wanted in L2 / Infinity Cache)
            if (grow < rows) *reinterpret_cast<uint4*>(plane + net * astride + (int64_t)grow * H + col0 + 8 * ch) = q;
        }
        __builtin_amdgcn_wave_barrier();
    };
    flush(hb, hv, false);
    if constexpr (X3) flush(it.hl, hv, true);
    if (xhb) {
        flush(xhb, xv, false);
        if constexpr (X3) flush(it.xhl, xv, true);
    }
    if (!live) return;
    if (rstd && wave == 0 && kq == 0) rstd[net * (int64_t)rows + row] = rs;
}

// 8 rows per workgroup. The per-element epilogue (LayerNorm affine, tanh, 2-4 bf16 plane conversions: ~40 VALU instructions) is what a
// trunk workgroup spends its time on, and with 16 rows a 2048-row launch is 128 workgroups — half the CUs idle while the rest each grind
// 16 x H elements. An MFMA tile still has 16 batch-row slots, so the 8 rows are entered twice, zeroed in opposite halves
//   B0[k][j] = x[j & 7][k] for j < 8, else 0       B1[k][j] = x[j & 7][k] for j >= 8, else 0
// and C = A_p B0 + A_(p+T/2) B1 gives lane (j, kq) row j & 7 of column tile p (j < 8) or p + T/2 (j >= 8): all 64 lanes hold
// distinct outputs, half as many as in the 16-row kernel.
template <int T, bool X3>     // T = H / 128 column tiles per wave, even
__global__ __launch_bounds__(512) void trunk_fwd8_kernel(const TrunkBatch tb, int64_t ldx, int rows, int in_dim, int Kp) {
    constexpr int H = T * 128, TP = T / 2;
    __shared__ float red[2][8][8];
    __shared__ __attribute__((aligned(16))) unsigned short stage[8][8][TF16_LD(T)];      // per-wave 8 x 16T output slab (+16 B row pad)
    const TrunkItem& it = tb.it[blockIdx.y];
    const float* __restrict__ x = it.x;
    float* __restrict__ rstd = it.rstd;
    unsigned short* __restrict__ hb = it.hb;
    unsigned short* __restrict__ xhb = it.xhb;
    const unsigned short* __restrict__ Wb = it.W0b;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rn = lane & 15, kq = lane >> 4, r8 = rn & 7, half = rn >> 3;
    const int row = blockIdx.x * 8 + r8;
    const bool live = row < rows;
    const int col0 = wave * (H / 8);
    f32x4_t acc[TP];
#pragma unroll
    for (int p = 0; p < TP; ++p) acc[p] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float4 bias4[TP], gain4[TP], beta4[TP];       // fetched with the inputs and weight fragments (one memory round trip for the kernel)
#pragma unroll
    for (int p = 0; p < TP; ++p) {
        const int c = col0 + (p + TP * half) * 16 + 4 * kq;
        bias4[p] = *reinterpret_cast<const float4*>(it.b0 + c);
        gain4[p] = *reinterpret_cast<const float4*>(it.gain + c);
        beta4[p] = *reinterpret_cast<const float4*>(it.beta + c);
    }
    const int64_t xrow = (int64_t)(live ? row : 0) * ldx;          // unconditional loads: index clamped, value masked
    for (int k0 = 0; k0 < Kp; k0 += 32) {
        const int kb = k0 + 8 * kq;
        float xs[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) xs[j] = x[xrow + (kb + j < in_dim ? kb + j : 0)];
        bf16x8_t aw[T];
#pragma unroll
        for (int t = 0; t < T; ++t) aw[t] = *reinterpret_cast<const bf16x8_t*>(Wb + (int64_t)(col0 + t * 16 + rn) * Kp + kb);
        bf16x8_t awl[X3 ? T : 1];
        if constexpr (X3) {
#pragma unroll
            for (int t = 0; t < T; ++t) awl[t] = *reinterpret_cast<const bf16x8_t*>(it.W0l + (int64_t)(col0 + t * 16 + rn) * Kp + kb);
        }
        bf16x8_t b0, b1, b0l, b1l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xv = (live && kb + j < in_dim) ? xs[j] : 0.f;
            const __bf16 hi = (__bf16)xv;
            const __bf16 lo = (__bf16)(xv - (float)hi);
            const __bf16 zero = (__bf16)0.f;
            b0[j] = half ? zero : hi; b1[j] = half ? hi : zero;
            b0l[j] = half ? zero : lo; b1l[j] = half ? lo : zero;
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (X3) {
            f32x4_t cross[TP];
#pragma unroll
            for (int p = 0; p < TP; ++p) {
                cross[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[p], b0l, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                cross[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[p + TP], b1l, cross[p], 0, 0, 0);
                cross[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(awl[p], b0, cross[p], 0, 0, 0);
                cross[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(awl[p + TP], b1, cross[p], 0, 0, 0);
            }
#pragma unroll
            for (int p = 0; p < TP; ++p) acc[p] += cross[p];
        }
#pragma unroll
        for (int p = 0; p < TP; ++p) {
            acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[p], b0, acc[p], 0, 0, 0);
            acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[p + TP], b1, acc[p], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < TP; ++p) {
        const float4 b = bias4[p];
        acc[p][0] += b.x; acc[p][1] += b.y; acc[p][2] += b.z; acc[p][3] += b.w;
        s += (acc[p][0] + acc[p][1]) + (acc[p][2] + acc[p][3]);
    }
    s += __shfl_xor(s, 8, 64);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (lane < 8) red[0][wave][r8] = s;
    __syncthreads();
    float mean = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) mean += red[0][w][r8];
    mean *= 1.0f / (float)H;
    float s2 = 0.f;
#pragma unroll
    for (int p = 0; p < TP; ++p) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc[p][i] -= mean; s2 += acc[p][i] * acc[p][i]; }
    }
    s2 += __shfl_xor(s2, 8, 64);
    s2 += __shfl_xor(s2, 16, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (lane < 8) red[1][wave][r8] = s2;
    __syncthreads();
    float var = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) var += red[1][w][r8];
    const float rs = 1.0f / sqrtf(var * (1.0f / (float)H) + LN_EPS2);
    float4 hv[TP], xv[TP];
#pragma unroll
    for (int p = 0; p < TP; ++p) {
        const float4 g = gain4[p];
        const float4 be = beta4[p];
        xv[p] = make_float4(acc[p][0] * rs, acc[p][1] * rs, acc[p][2] * rs, acc[p][3] * rs);
        hv[p] = make_float4(tanh_fast(xv[p].x * g.x + be.x), tanh_fast(xv[p].y * g.y + be.y), tanh_fast(xv[p].z * g.z + be.z),
                            tanh_fast(xv[p].w * g.w + be.w));
    }
    // each wave transposes its 8 x 16T slab through LDS so that a store instruction writes whole 32T-byte row slabs (see trunk_fwd16_kernel)
    unsigned short* mine = &stage[wave][0][0];
    auto flush = [&](unsigned short* __restrict__ plane, const float4 (&v)[TP], bool lo) {
#pragma unroll
        for (int p = 0; p < TP; ++p)
            *reinterpret_cast<ushort4*>(mine + r8 * TF16_LD(T) + (p + TP * half) * 16 + 4 * kq) = lo ? f4_to_bf4_lo(v[p]) : f4_to_bf4(v[p]);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int idx = lane; idx < 16 * T; idx += 64) {
            const int r = idx / (2 * T), ch = idx % (2 * T);
            const uint4 q = *reinterpret_cast<const uint4*>(mine + r * TF16_LD(T) + 8 * ch);
            const int grow = blockIdx.x * 8 + r;
            if (grow < rows) *reinterpret_cast<uint4*>(plane + (int64_t)grow * H + col0 + 8 * ch) = q;
        }
        __builtin_amdgcn_wave_barrier();
    };
    flush(hb, hv, false);
    if constexpr (X3) flush(it.hl, hv, true);
    if (xhb) {
        flush(xhb, xv, false);
        if constexpr (X3) flush(it.xhl, xv, true);
    }
    if (rstd && wave == 0 && lane < 8 && live) rstd[row] = rs;
}

bool trunk_fwd16_supported(int H) { return H >= 128 && H <= 1024 && H % 128 == 0; }

int trunk_fwd16_batch(const TrunkBatch& tb, int count, int64_t ldx, int rows, int in_dim, int H, hipStream_t s) {
    EXORL_REQUIRE(trunk_fwd16_supported(H) && in_dim >= 1 && in_dim <= MAX_IN && count >= 1 && count <= 4, "trunk_fwd16: unsupported H=%d in=%d n=%d",
                  H, in_dim, count);
    const int Kp = (int)round_up(in_dim, 32);
    const dim3 grid(cdiv(rows, 16), count);
    const bool x3 = tb.it[0].W0l != nullptr;
    for (int i = 0; i < count; ++i)
        EXORL_REQUIRE((tb.it[i].W0l != nullptr) == x3 && (!x3 || (tb.it[i].hl && (!tb.it[i].xhb || tb.it[i].xhl))), "trunk_fwd16: inconsistent lo planes");
    const int var = tune_variant();
    // 8-row workgroups where 16-row ones would leave CUs idle (measured, H = 1024: 2048 rows x 1 net 13.4 -> 10.3 us, 1024 x 2 nets 12.9 -> 10.0);
    // at 256 or more 16-row workgroups the 8-row grid is two rounds per CU and slower (1024 rows x 4 nets: 13.1 -> 16.2 us)
    const bool rows8 = H % 256 == 0 && !(var & 4194304) && ((var & 8388608) || cdiv(rows, 16) * count < 256);
    if (rows8) {
        const dim3 grid8(cdiv(rows, 8), count);
#define EXORL_TF8(T) do { if (x3) hipLaunchKernelGGL((trunk_fwd8_kernel<T, true>), grid8, dim3(512), 0, s, tb, ldx, rows, in_dim, Kp); \
                          else hipLaunchKernelGGL((trunk_fwd8_kernel<T, false>), grid8, dim3(512), 0, s, tb, ldx, rows, in_dim, Kp); } while (0)
        switch (H / 256) {
            case 1: EXORL_TF8(2); break;
            case 2: EXORL_TF8(4); break;
            case 3: EXORL_TF8(6); break;
            default: EXORL_TF8(8); break;
        }
#undef EXORL_TF8
        EXORL_LAUNCH_CHECK();
        return 0;
    }
#define EXORL_TF16(T) do { if (x3) hipLaunchKernelGGL((trunk_fwd16_kernel<T, true>), grid, dim3(512), 0, s, tb, ldx, rows, in_dim, Kp); \
                           else hipLaunchKernelGGL((trunk_fwd16_kernel<T, false>), grid, dim3(512), 0, s, tb, ldx, rows, in_dim, Kp); } while (0)
    switch (H / 128) {
        case 1: EXORL_TF16(1); break;
        case 2: EXORL_TF16(2); break;
        case 3: EXORL_TF16(3); break;
        case 4: EXORL_TF16(4); break;
        case 5: EXORL_TF16(5); break;
        case 6: EXORL_TF16(6); break;
        case 7: EXORL_TF16(7); break;
        default: EXORL_TF16(8); break;
    }
#undef EXORL_TF16
    EXORL_LAUNCH_CHECK();
    return 0;
}

int trunk_fwd16(const float* x, int64_t ldx, const unsigned short* W0b, const float* b0, const float* gain, const float* beta, float* rstd,
                unsigned short* h_bf16, unsigned short* xhat_bf16, int rows, int in_dim, int H, int nets, int64_t astride, int64_t pstride,
                hipStream_t s, const unsigned short* W0l, unsigned short* h_lo, unsigned short* xhat_lo) {
    EXORL_REQUIRE(h_bf16 && nets >= 1 && nets <= 4, "trunk_fwd16: bad arguments");
    TrunkBatch tb{};
    const int64_t wstride = (int64_t)H * round_up(in_dim, 32);
    for (int n = 0; n < nets; ++n)
        tb.it[n] = TrunkItem{x, W0b + n * wstride, b0 + n * pstride, gain + n * pstride, beta + n * pstride, rstd ? rstd + (int64_t)n * rows : nullptr,
                             h_bf16 + n * astride, xhat_bf16 ? xhat_bf16 + n * astride : nullptr, W0l ? W0l + n * wstride : nullptr,
                             h_lo ? h_lo + n * astride : nullptr, (xhat_bf16 && xhat_lo) ? xhat_lo + n * astride : nullptr};
    return trunk_fwd16_batch(tb, nets, ldx, rows, in_dim, H, s);
}

int trunk_fwd(const float* x, int64_t ldx, const float* W0T, const float* b0, const float* gain, const float* beta,
              float* h, float* xhat, float* rstd, unsigned short* h_bf16, unsigned short* xhat_bf16, int rows, int in_dim,
              int H, int nets, int64_t astride, int64_t pstride, int64_t tstride, hipStream_t s) {
    EXORL_REQUIRE(H >= 4 && H <= 1024 && H % 4 == 0 && in_dim >= 1 && in_dim <= MAX_IN, "trunk_fwd: unsupported H=%d in=%d", H, in_dim);
    int kc = TF_LDS_FLOATS / H;                // k-rows of W0T staged per pass
    if (kc > in_dim) kc = in_dim;
    const size_t lds = ((size_t)kc * H + TF_ROWS * MAX_IN) * sizeof(float);
    const dim3 grid(cdiv(rows, TF_ROWS), nets);
    static bool attr_set = false;
    if (!attr_set) {       // > 64 KB of dynamic LDS needs the opt-in on HIP
        EXORL_CHECK_HIP(hipFuncSetAttribute((const void*)trunk_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        EXORL_CHECK_HIP(hipFuncSetAttribute((const void*)trunk_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    if (h_bf16) hipLaunchKernelGGL((trunk_fwd_kernel<true>), grid, dim3(512), lds, s, x, ldx, W0T, b0, gain, beta, h, xhat, rstd, h_bf16,
                                   xhat_bf16, rows, in_dim, H, astride, pstride, tstride, kc);
    else        hipLaunchKernelGGL((trunk_fwd_kernel<false>), grid, dim3(512), lds, s, x, ldx, W0T, b0, gain, beta, h, xhat, rstd, h_bf16,
                                   xhat_bf16, rows, in_dim, H, astride, pstride, tstride, kc);
    EXORL_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// trunk backward, part 1: LayerNorm/tanh backward. One row per wave; 8 waves x 2 rows per workgroup.
//   dz0 = rstd * (dxh - mean(dxh) - xhat * mean(dxh*xhat)),  dxh = dh*(1-h^2)*gain      (written in place over dh)
// and per-workgroup partial column sums P[chunk] = [dgain H][dbeta H][db0 H] (PARAMS only).
constexpr int TB_ROWS = 8;
// rows per workgroup of the row-chunked backward kernels grow with the batch: the per-chunk partial gradients are summed by the
// optimiser launch, and CQL's 10 B-row critic pass would otherwise leave 1280 partial rows per parameter to it
// (measured on CQL, 10240 rows: optimiser launch 84 -> 50 us, the two producers +9 and +7 us)
static int tb_iters(int rows) { return rows >= 8192 ? 4 : 1; }
static int or_iters(int rows) { return rows >= 8192 ? 4 : 1; }
__device__ __forceinline__ float4 bf4_to_f4(ushort4 q) {
    return make_float4(__uint_as_float((unsigned)q.x << 16), __uint_as_float((unsigned)q.y << 16),
                       __uint_as_float((unsigned)q.z << 16), __uint_as_float((unsigned)q.w << 16));
}

__device__ __forceinline__ float wave_sum8(const float (&v)[8], int lane) {
    float a[4], b[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float keep = (lane & 1) ? v[2 * j + 1] : v[2 * j];
        const float send = (lane & 1) ? v[2 * j] : v[2 * j + 1];
        a[j] = keep + __shfl_xor(send, 1, 64);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float keep = (lane & 2) ? a[2 * j + 1] : a[2 * j];
        const float send = (lane & 2) ? a[2 * j] : a[2 * j + 1];
        b[j] = keep + __shfl_xor(send, 2, 64);
    }
    float r = ((lane & 4) ? b[1] : b[0]) + __shfl_xor((lane & 4) ? b[0] : b[1], 4, 64);
    r += __shfl_xor(r, 8, 64);
    r += __shfl_xor(r, 16, 64);
    r += __shfl_xor(r, 32, 64);
    return r;
}

// B16: h and xhat are read from their bf16 copies (fast mode keeps no fp32 copies of the trunk activations)
// dx != nullptr (dgrad-only pass): dx[net][row][j] = sum_c dz[row][c] W0T[j][c], j < dx_cols — the d/d(action) the actor
// step needs (td3_bc.py:152-155) formed from the dz row while it is still in registers; dz itself is then not stored.
// LO: the bf16 copies come as hi + lo planes (split-bf16); RECOMP: h = tanh(xhat g + beta) is recomputed instead of read. Both are template
// parameters, and every load below is unconditional (column index clamped, surplus lanes masked afterwards): a null check or a bounds
// guard around a load compiles to branch + load + s_waitcnt vmcnt(0), i.e. the twelve to sixteen loads of a row — and the 24 weight
// loads of the d/d(action) epilogue — one L2 round trip at a time.
template <bool PARAMS, bool B16, bool LO, bool RECOMP>
__global__ __launch_bounds__(512) void ln_bwd_kernel(float* dh, const float* __restrict__ h, const float* __restrict__ xhat,
                                                     const unsigned short* __restrict__ hb, const unsigned short* __restrict__ xhb,
                                                     const float* __restrict__ rstd, const float* __restrict__ gain,
                                                     float* __restrict__ P, int rows, int H, int64_t astride,
                                                     int64_t pstride, const float* __restrict__ w0t, int64_t tstride,
                                                     float* __restrict__ dx, int dx_cols, const unsigned short* __restrict__ hl,
                                                     const unsigned short* __restrict__ xhl, int iters, const float* __restrict__ beta) {
    __shared__ __attribute__((aligned(16))) float red[PARAMS ? 3 * 8 * 1024 : 4];      // 96 KB of the CU's 160: the three column vectors at once
    const int net = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H4 = H >> 2;
    float4 g[4], be[4], pg[4], pb[4], pb0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c4 = lane + 64 * i;
        g[i] = c4 < H4 ? reinterpret_cast<const float4*>(gain + net * pstride)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
        be[i] = (RECOMP && c4 < H4) ? reinterpret_cast<const float4*>(beta + net * pstride)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
        pg[i] = pb[i] = pb0[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int it = 0; it < iters; ++it) {           // 8 rows per pass; large batches take 4 passes per workgroup (4x fewer partial rows)
        const int row = blockIdx.x * TB_ROWS * iters + it * 8 + wave;
        if (row >= rows) continue;                      // wave-uniform
        const int64_t o = net * astride + (int64_t)row * H;
        float4 d[4], xh[4];
        float s1 = 0.f, s2 = 0.f;
        // every stream of the row in flight before the first use
        float4 dvs[4], hvs[4];
        ushort4 xb[4], xl[4], hbv[4], hlv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cc = lane + 64 * i < H4 ? lane + 64 * i : 0;
            dvs[i] = reinterpret_cast<const float4*>(dh + o)[cc];
            if constexpr (B16) {
                xb[i] = reinterpret_cast<const ushort4*>(xhb + o)[cc];
                if constexpr (LO) xl[i] = reinterpret_cast<const ushort4*>(xhl + o)[cc];
                if constexpr (!RECOMP) {
                    hbv[i] = reinterpret_cast<const ushort4*>(hb + o)[cc];
                    if constexpr (LO) hlv[i] = reinterpret_cast<const ushort4*>(hl + o)[cc];
                }
            } else {
                hvs[i] = reinterpret_cast<const float4*>(h + o)[cc];
                xh[i] = reinterpret_cast<const float4*>(xhat + o)[cc];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c4 = lane + 64 * i;
            d[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c4 >= H4) { xh[i] = make_float4(0.f, 0.f, 0.f, 0.f); }
            if (c4 < H4) {
                float4 hv;
                if constexpr (B16) {
                    xh[i] = bf4_to_f4(xb[i]);
                    if constexpr (LO) {                  // split-bf16 mode: hi + lo planes
                        const float4 l2 = bf4_to_f4(xl[i]);
                        xh[i].x += l2.x; xh[i].y += l2.y; xh[i].z += l2.z; xh[i].w += l2.w;
                    }
                    if constexpr (RECOMP) {              // h = tanh(xhat * g + beta) again (as the forward formed it) instead of reading its planes:
                        hv.x = tanh_fast(xh[i].x * g[i].x + be[i].x); hv.y = tanh_fast(xh[i].y * g[i].y + be[i].y);     // 4 B per element less
                        hv.z = tanh_fast(xh[i].z * g[i].z + be[i].z); hv.w = tanh_fast(xh[i].w * g[i].w + be[i].w);
                    } else {
                        hv = bf4_to_f4(hbv[i]);
                        if constexpr (LO) {
                            const float4 l1 = bf4_to_f4(hlv[i]);
                            hv.x += l1.x; hv.y += l1.y; hv.z += l1.z; hv.w += l1.w;
                        }
                    }
                } else {
                    hv = hvs[i];
                }
                const float4 dv = dvs[i];
                float4 dy;
                dy.x = dv.x * (1.0f - hv.x * hv.x); dy.y = dv.y * (1.0f - hv.y * hv.y);
                dy.z = dv.z * (1.0f - hv.z * hv.z); dy.w = dv.w * (1.0f - hv.w * hv.w);
                if constexpr (PARAMS) {
                    pg[i].x += dy.x * xh[i].x; pg[i].y += dy.y * xh[i].y; pg[i].z += dy.z * xh[i].z; pg[i].w += dy.w * xh[i].w;
                    pb[i].x += dy.x; pb[i].y += dy.y; pb[i].z += dy.z; pb[i].w += dy.w;
                }
                d[i].x = dy.x * g[i].x; d[i].y = dy.y * g[i].y; d[i].z = dy.z * g[i].z; d[i].w = dy.w * g[i].w;
                s1 += (d[i].x + d[i].y) + (d[i].z + d[i].w);
                s2 += (d[i].x * xh[i].x + d[i].y * xh[i].y) + (d[i].z * xh[i].z + d[i].w * xh[i].w);
            }
        }
        const float m1 = wave_sum(s1) / (float)H, m2 = wave_sum(s2) / (float)H;
        const float rs = rstd[net * (int64_t)rows + row];
        float4 dzv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c4 = lane + 64 * i;
            dzv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c4 < H4) {
                float4 v;
                v.x = rs * (d[i].x - m1 - xh[i].x * m2); v.y = rs * (d[i].y - m1 - xh[i].y * m2);
                v.z = rs * (d[i].z - m1 - xh[i].z * m2); v.w = rs * (d[i].w - m1 - xh[i].w * m2);
                if (PARAMS || !dx) reinterpret_cast<float4*>(dh + o)[c4] = v;
                dzv[i] = v;
                if constexpr (PARAMS) { pb0[i].x += v.x; pb0[i].y += v.y; pb0[i].z += v.z; pb0[i].w += v.w; }
            }
        }
        if constexpr (!PARAMS) {
            if (dx) {
                for (int j0 = 0; j0 < dx_cols; j0 += 8) {
                    float part[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) part[j] = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {        // 8 weight rows of a column chunk in flight at once (row index clamped: dzv is 0 past H/4,
                        const int cc = lane + 64 * i < H4 ? lane + 64 * i : 0;          // the surplus outputs are never stored)
                        float4 wv[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            wv[j] = reinterpret_cast<const float4*>(w0t + net * tstride + (int64_t)(j0 + j < dx_cols ? j0 + j : 0) * H)[cc];
#pragma unroll
                        for (int j = 0; j < 8; ++j) part[j] += dzv[i].x * wv[j].x + dzv[i].y * wv[j].y + dzv[i].z * wv[j].z + dzv[i].w * wv[j].w;
                    }
                    const float tot = wave_sum8(part, lane);
                    const int j = j0 + (lane & 7);
                    if (lane < 8 && j < dx_cols) dx[((int64_t)net * rows + row) * dx_cols + j] = tot;
                }
            }
        }
    }
    if constexpr (PARAMS) {        // cross-wave sums of the three column vectors: one barrier, then 3 x 8 LDS reads per column
        float* Pn = P + ((int64_t)net * gridDim.x + blockIdx.x) * 3 * H;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c4 = lane + 64 * i;
            if (c4 < H4) {
                reinterpret_cast<float4*>(red + wave * 1024)[c4] = pg[i];
                reinterpret_cast<float4*>(red + (8 + wave) * 1024)[c4] = pb[i];
                reinterpret_cast<float4*>(red + (16 + wave) * 1024)[c4] = pb0[i];
            }
        }
        __syncthreads();
        for (int c = tid; c < H; c += 512) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                float sacc = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) sacc += red[(q * 8 + w) * 1024 + c];
                Pn[q * H + c] = sacc;
            }
        }
    }
}

int ln_bwd(float* dh, const float* h, const float* xhat, const unsigned short* h_bf16, const unsigned short* xhat_bf16,
           const float* rstd, const float* gain, float* P, int rows, int H, int nets, int64_t astride, int64_t pstride,
           int want_params, hipStream_t s, const float* w0t, int64_t tstride, float* dx, int dx_cols, const unsigned short* h_lo,
           const unsigned short* xhat_lo, const float* beta) {
    EXORL_REQUIRE(H >= 4 && H <= 1024 && H % 4 == 0, "ln_bwd: unsupported H=%d", H);
    EXORL_REQUIRE(!dx || (!want_params && w0t && dx_cols >= 1), "ln_bwd: the dx epilogue belongs to the dgrad-only pass");
    const dim3 grid(trunk_chunks(rows), nets);
    const bool b16 = h_bf16 && xhat_bf16;
    const int iters = tb_iters(rows);
    EXORL_REQUIRE((h_lo != nullptr) == (xhat_lo != nullptr) && (!h_lo || b16), "ln_bwd: lo planes come in pairs, with the bf16 hi planes");
    const bool lo = h_lo != nullptr, rc = b16 && beta != nullptr;
#define EXORL_LNB(PA, BB, LL, RR) hipLaunchKernelGGL((ln_bwd_kernel<PA, BB, LL, RR>), grid, dim3(512), 0, s, dh, h, xhat, h_bf16, xhat_bf16, rstd, gain, P, rows, H, astride, pstride, w0t, tstride, dx, dx_cols, h_lo, xhat_lo, iters, beta)
#define EXORL_LNB2(PA) do { if (!b16) EXORL_LNB(PA, false, false, false); \
                             else if (lo && rc) EXORL_LNB(PA, true, true, true); else if (lo) EXORL_LNB(PA, true, true, false); \
                             else if (rc) EXORL_LNB(PA, true, false, true); else EXORL_LNB(PA, true, false, false); } while (0)
    if (want_params) EXORL_LNB2(true); else EXORL_LNB2(false);
#undef EXORL_LNB2
#undef EXORL_LNB
    EXORL_LAUNCH_CHECK();
    return 0;
}
int trunk_chunks(int rows) { return cdiv(rows, TB_ROWS * tb_iters(rows)); }

// ------------------------------------------------------------------------------------------------
// outer-product column reduction: P[chunk][j][c] = sum_{m in chunk} u[m][j] * v[m][c]   (first-layer wgrad:
// u = x (rows x J), v = dz0).  Thread = column, 16 rows per workgroup, u rows broadcast from LDS.
constexpr int OR_ROWS = 32;
__global__ __launch_bounds__(256) void outer_reduce_kernel(const float* __restrict__ u, int64_t ldu, int J,
                                                           const float* __restrict__ v, float* __restrict__ P, int rows,
                                                           int H, int64_t vstride, int iters) {
    __shared__ __attribute__((aligned(16))) float us[OR_ROWS][32];
    const int net = blockIdx.z;
    const int c = blockIdx.x * 256 + threadIdx.x;
    float* Pn = P + ((int64_t)net * gridDim.y + blockIdx.y) * (int64_t)J * H;
    for (int j0 = 0; j0 < J; j0 += 32) {
        float acc[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) acc[j] = 0.f;
        for (int sub = 0; sub < iters; ++sub) {
            const int row0 = (blockIdx.y * iters + sub) * OR_ROWS;
            const int nr = rows - row0 < OR_ROWS ? rows - row0 : OR_ROWS;       // <= 0 past the end: nothing staged, nothing added
            // all OR_ROWS rows of this thread's column and the staged u values are requested up front, unconditionally (row clamped; the
            // staged u of a row past the end is 0, so whatever the clamped v row holds contributes nothing): one memory round trip per
            // sub-block instead of four guarded ones
            float vall[OR_ROWS];
            const int cc = c < H ? c : 0;
#pragma unroll
            for (int r = 0; r < OR_ROWS; ++r) vall[r] = v[net * vstride + (int64_t)(row0 + (r < nr ? r : 0)) * H + cc];
            float ust[OR_ROWS * 32 / 256];
#pragma unroll
            for (int q = 0; q < OR_ROWS * 32 / 256; ++q) {
                const int i = threadIdx.x + 256 * q, r = i >> 5, j = i & 31;
                ust[q] = u[(int64_t)(row0 + (r < nr ? r : 0)) * ldu + (j0 + j < J ? j0 + j : 0)];
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < OR_ROWS * 32 / 256; ++q) {
                const int i = threadIdx.x + 256 * q, r = i >> 5, j = i & 31;
                us[r][j] = (r < nr && j0 + j < J) ? ust[q] : 0.f;
            }
            __syncthreads();
            if (c < H) {
#pragma unroll 1
                for (int rb = 0; rb < OR_ROWS; rb += 8) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const float vr = vall[rb + r];
#pragma unroll
                        for (int j4 = 0; j4 < 8; ++j4) {
                            const float4 uu = *reinterpret_cast<const float4*>(&us[rb + r][4 * j4]);
                            acc[4 * j4] += uu.x * vr; acc[4 * j4 + 1] += uu.y * vr; acc[4 * j4 + 2] += uu.z * vr; acc[4 * j4 + 3] += uu.w * vr;
                        }
                        __builtin_amdgcn_sched_barrier(0);     // keep one row's 8 LDS reads live at a time (VGPR budget)
                    }
                }
            }
        }
        if (c < H) {
#pragma unroll
            for (int j = 0; j < 32; ++j)
                if (j0 + j < J) Pn[(int64_t)(j0 + j) * H + c] = acc[j];
        }
    }
}

int outer_reduce(const float* u, int64_t ldu, int J, const float* v, float* P, int rows, int H, int nets, int64_t vstride,
                 hipStream_t s) {
    hipLaunchKernelGGL(outer_reduce_kernel, dim3(cdiv(H, 256), outer_chunks(rows), nets), dim3(256), 0, s, u, ldu, J, v, P, rows,
                       H, vstride, or_iters(rows));
    EXORL_LAUNCH_CHECK();
    return 0;
}
int outer_chunks(int rows) { return cdiv(rows, OR_ROWS * or_iters(rows)); }

// Sums 8 per-lane values over the wave with 10 cross-lane exchanges instead of 48: three butterfly steps that each
// halve the number of live values (lane l ends up owning value index l & 7), then three plain steps over the remaining
// 8-lane groups. Returns the wave total of value (lane & 7) in every lane.
// ------------------------------------------------------------------------------------------------
// head forward v2: out[m][j] = b[j] + sum_c a[m][c] W[j][c]; one wave per row, float4 streams (H % 4 == 0)
__device__ __forceinline__ float philox_normal_f(uint64_t seed, uint64_t counter, uint32_t elem) {
    uint32_t c[4] = {elem, 0u, (uint32_t)counter, (uint32_t)(counter >> 32)};
    Philox::gen(c, seed);
    const float u1 = ((float)c[0] + 1.0f) * 2.3283064365386963e-10f;
    const float u2 = (float)c[1] * 2.3283064365386963e-10f;
    return sqrtf(-2.0f * __logf(u1)) * __cosf(6.283185307179586f * u2);
}

template <int NO>
__global__ __launch_bounds__(256) void head_fwd4_kernel(const float* __restrict__ a, const float* __restrict__ W,
                                                        const float* __restrict__ b, float* __restrict__ out, int rows,
                                                        int H, int nout, int tanh_out, int64_t astride, int64_t pstride,
                                                        int64_t ostride, SampleSpec sp) {
    const int net = blockIdx.y;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float4* ar = reinterpret_cast<const float4*>(a + net * astride + (int64_t)row * H);
    const float* Wn = W + net * pstride;
    float acc[NO];
#pragma unroll
    for (int j = 0; j < NO; ++j) acc[j] = 0.f;
    const int H4 = H >> 2;
    float4 avs[4];                 // the whole row (H <= 1024) in flight before any use: one memory round trip, not four
#pragma unroll
    for (int i = 0; i < 4; ++i) avs[i] = lane + 64 * i < H4 ? ar[lane + 64 * i] : make_float4(0.f, 0.f, 0.f, 0.f);
    // Weight rows are loaded UNCONDITIONALLY (row index clamped, results of the surplus outputs never stored): a `j < nout` guard around
    // each load compiles to a branch + load + s_waitcnt vmcnt(0) per output and chunk, i.e. 24 serial L2 round trips per row at nout = 6
    // (7 of the kernel's 11.5 us, visible only in the ISA); all NO loads of a chunk are now in flight together
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c4 = lane + 64 * i < H4 ? lane + 64 * i : 0;         // lanes past H/4 hold av = 0
        const float4 av = avs[i];
        float4 wv[NO];
#pragma unroll
        for (int j = 0; j < NO; ++j) wv[j] = reinterpret_cast<const float4*>(Wn + (int64_t)(j < nout ? j : 0) * H)[c4];
#pragma unroll
        for (int j = 0; j < NO; ++j) acc[j] += av.x * wv[j].x + av.y * wv[j].y + av.z * wv[j].z + av.w * wv[j].w;
    }
    if constexpr (NO == 1) {
        float v = wave_sum(acc[0]) + (b ? b[net * pstride] : 0.f);
        if (tanh_out) v = tanhf(v);
        if (lane == 0) out[net * ostride + (int64_t)row * nout] = v;
    } else {
#pragma unroll
        for (int g8 = 0; g8 < NO / 8; ++g8) {
            float part[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) part[j] = acc[8 * g8 + j];
            const int j = 8 * g8 + (lane & 7);
            float v = wave_sum8(part, lane);
            if (lane < 8 && j < nout) {
                v += b ? b[net * pstride + j] : 0.f;
                if (tanh_out) v = tanhf(v);
                out[net * ostride + (int64_t)row * nout + j] = v;
                if (sp.dst_next) {                   // TruncatedNormal(mu, std).sample(clip) (utils.py:139-149)
                    const int half = row >= sp.B, m = half ? row - sp.B : row, e = m * nout + j;
                    const float* nb = half ? sp.noise_a : sp.noise_c;
                    const float z = nb ? nb[e] : philox_normal_f(sp.seed, (uint64_t)half + (sp.counter_ptr ? *sp.counter_ptr : 0ull), (uint32_t)e);
                    const float eps = fminf(fmaxf(z * (sp.stddev_ptr ? *sp.stddev_ptr : sp.stddev), -sp.clip), sp.clip);
                    const float x = fminf(fmaxf(v + eps, -1.0f + 1e-6f), 1.0f - 1e-6f);
                    (half ? sp.dst_pi : sp.dst_next)[(int64_t)m * sp.dst_ld + j] = x;
                }
            }
        }
    }
}

// scalar heads of several nets in one launch (critic Q1,Q2 and target Q1,Q2): per-net pointers instead of strides
__global__ __launch_bounds__(256) void head_fwd1_batch_kernel(const HeadBatch hb, int rows, int H) {
    const HeadItem& it = hb.it[blockIdx.y];
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float4* ar = reinterpret_cast<const float4*>(it.a + (int64_t)row * H);
    const int H4 = H >> 2;
    float4 avs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) avs[i] = lane + 64 * i < H4 ? ar[lane + 64 * i] : make_float4(0.f, 0.f, 0.f, 0.f);
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c4 = lane + 64 * i;
        if (c4 >= H4) break;
        const float4 wv = reinterpret_cast<const float4*>(it.W)[c4];
        acc += avs[i].x * wv.x + avs[i].y * wv.y + avs[i].z * wv.z + avs[i].w * wv.w;
    }
    const float v = wave_sum(acc) + it.b[0];
    if (lane == 0) it.out[row] = v;
}

int head_fwd1_batch(const HeadBatch& hb, int count, int rows, int H, hipStream_t s) {
    EXORL_REQUIRE(count >= 1 && count <= 4 && H % 4 == 0 && H <= 1024, "head_fwd1_batch: count=%d H=%d unsupported", count, H);
    hipLaunchKernelGGL(head_fwd1_batch_kernel, dim3(cdiv(rows, 4), count), dim3(256), 0, s, hb, rows, H);
    EXORL_LAUNCH_CHECK();
    return 0;
}

int head_fwd4(const float* a, const float* W, const float* b, float* out, int rows, int H, int nout, int tanh_out,
              int nets, int64_t astride, int64_t pstride, int64_t ostride, hipStream_t s, const SampleSpec* sample) {
    EXORL_REQUIRE(nout >= 1 && nout <= 32 && H % 4 == 0 && H <= 1024, "head_fwd4: nout=%d H=%d unsupported", nout, H);
    EXORL_REQUIRE(!sample || (nout > 1 && nets == 1 && tanh_out && rows == 2 * sample->B), "head_fwd4: sampling epilogue needs the stacked actor head");
    const SampleSpec sp = sample ? *sample : SampleSpec{};
    dim3 grid(cdiv(rows, 4), nets);
    if (nout == 1) hipLaunchKernelGGL((head_fwd4_kernel<1>), grid, dim3(256), 0, s, a, W, b, out, rows, H, nout, tanh_out, astride, pstride, ostride, sp);
    else if (nout <= 8) hipLaunchKernelGGL((head_fwd4_kernel<8>), grid, dim3(256), 0, s, a, W, b, out, rows, H, nout, tanh_out, astride, pstride, ostride, sp);
    else if (nout <= 16) hipLaunchKernelGGL((head_fwd4_kernel<16>), grid, dim3(256), 0, s, a, W, b, out, rows, H, nout, tanh_out, astride, pstride, ostride, sp);
    else hipLaunchKernelGGL((head_fwd4_kernel<32>), grid, dim3(256), 0, s, a, W, b, out, rows, H, nout, tanh_out, astride, pstride, ostride, sp);
    EXORL_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// head backward: dz[m][c] = (sum_j dout[m][j] W[j][c]) * (a[m][c] > 0), written fp32 and/or bf16, plus
// per-chunk partials of db_hidden[c] = sum_m dz and dW[j][c] = sum_m dout[m][j] a[m][c] and db_out[j].
// Thread = 4 consecutive columns (float4 streams), 8 rows per workgroup.
// P layout per (net, chunk): [dW nout*H][db_hidden H][db_out 16]
constexpr int HB_ROWS = 8;
// d(loss)/d(head output) for row m, output j of net `net` (see DoutSpec)
__device__ __forceinline__ float dout_value(const DoutSpec& d, int net, int m, int j, int rows, int nout, float lam = 1.0f) {
    if (d.mode == EXORL_DOUT_BUFFER) return d.buf[((int64_t)net * rows + m) * nout + j];
    if (d.mode == EXORL_DOUT_CQL_ACTOR) {
        // actor_loss = (alpha*log_pi - Q).mean() over (B,A), y = tanh(x), x = mu + std z  (cql.py:236-255). With
        // g = dQterm/dy * (1-y^2), c = alpha/(Bg*A):  dL/dmu = g + 2 c y;  dL/dstd = g z + c (-1/std + 2 y z)
        const int A = nout >> 1, jj = j < A ? j : j - A;
        const float* rr = d.raw + (int64_t)m * nout;
        const float mu = tanhf(rr[jj]);
        const float ls = rr[A + jj];
        const float sd = expf(fminf(fmaxf(ls, -10.0f), 2.0f));
        const int e = m * A + jj;
        const float z = d.z ? d.z[e] : philox_normal_f(d.seed, (d.counter_ptr ? *d.counter_ptr : 0ull) * 8 + d.counter, (uint32_t)e);
        const float y = tanhf(mu + sd * z);
        float dy = 0.f;
        for (int t = 0; t < d.da_nets; ++t) dy += d.da[((int64_t)t * rows + m) * A + jj];
        const float g = dy * (1.0f - y * y);
        const float c = *d.alpha_ptr * d.inv_bg / (float)A;
        if (j < A) return (g + 2.0f * c * y) * (1.0f - mu * mu);
        const float inside = (ls >= -10.0f && ls <= 2.0f) ? 1.0f : 0.0f;
        return (g * z + c * (-1.0f / sd + 2.0f * y * z)) * sd * inside;
    }
    if (d.mode == EXORL_DOUT_TD) {
        const float y = d.reward[m] + d.discount[m] * fminf(d.tq[m], d.tq[rows + m]);
        const float g = 2.0f * (d.q[net * rows + m] - y) * d.inv_bg;
        return d.task ? g * d.task[(int64_t)m * d.task_ld + j] : g;
    }
    if (d.mode == EXORL_DOUT_ACTOR_Q) {
        const float lambda = d.use_lambda ? d.alpha / (d.stats[0] * d.inv_bg) : 1.0f;
        const float q1 = d.q[m], q2 = d.q[rows + m];
        const float w1 = q1 < q2 ? 1.0f : (q1 == q2 ? 0.5f : 0.0f);      // torch.min backward: ties split evenly
        const float g = -lambda * d.inv_bg * (net == 0 ? w1 : 1.0f - w1);
        return d.task ? g * d.task[(int64_t)m * d.task_ld + j] : g;
    }
    // EXORL_DOUT_ACTOR_MU: gradient at the actor's pre-tanh output
    const int i = m * nout + j;
    const float mv = d.mu[i];
    float dmu;
    if (d.kind == EXORL_AGENT_BC) {
        const float sd = d.stddev_ptr ? *d.stddev_ptr : d.stddev;
        dmu = -(d.a_data[i] - mv) / (sd * sd) * d.inv_bg;
    } else if (d.kind == EXORL_AGENT_CRR) {
        const float sd = d.stddev_ptr ? *d.stddev_ptr : d.stddev;
        dmu = -d.w[m] * (d.a_data[i] - mv) / (sd * sd) * d.inv_bg;     // -(log_prob * w).mean(), crr.py:185-186
    } else {
        dmu = 0.f;
        for (int t = 0; t < d.da_nets; ++t) dmu += d.da[((int64_t)t * rows + m) * nout + j];
        dmu *= lam;
        if (d.kind == EXORL_AGENT_TD3_BC) dmu += 2.0f * d.inv_bg / (float)nout * (mv - d.a_data[i]);
    }
    return dmu * (1.0f - mv * mv);
}

template <int NO>
__global__ __launch_bounds__(256) void head_bwd_kernel(const DoutSpec dspec, const float* __restrict__ W,
                                                       const float* __restrict__ a, float* __restrict__ dz,
                                                       unsigned short* __restrict__ dzb, float* __restrict__ P, int rows,
                                                       int H, int nout, int64_t astride, int64_t pstride, int want_params,
                                                       unsigned short* __restrict__ dzl) {
    __shared__ float ds[HB_ROWS * 16];
    __shared__ float lam_s[5];
    const int net = blockIdx.y;
    const int row0 = blockIdx.x * HB_ROWS;
    const int c4 = threadIdx.x;
    const int H4 = H >> 2;
    // Every global read of the kernel is issued here, before the first barrier, so that the weight rows, the activation rows and the
    // loss-gradient inputs share one memory round trip with the lambda partials (a load behind a barrier or a branch costs its own).
    const int c4c = c4 < H4 ? c4 : 0;
    const int nr = rows - row0 < HB_ROWS ? rows - row0 : HB_ROWS;
    float4 w[NO], pw[NO], avs[HB_ROWS];
#pragma unroll
    for (int j = 0; j < NO; ++j) {
        w[j] = reinterpret_cast<const float4*>(W + net * pstride + (int64_t)(j < nout ? j : 0) * H)[c4c];     // ds is 0 for j >= nout
        pw[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int r = 0; r < HB_ROWS; ++r)          // row clamped; rows past nr are skipped below
        avs[r] = reinterpret_cast<const float4*>(a + net * astride + (int64_t)(row0 + (r < nr ? r : 0)) * H)[c4c];
    // deterministic-actor gradient (TD3 / TD3+BC / DDPG): its inputs are read now, lambda is applied after the reduction below
    const bool mu_path = dspec.mode == EXORL_DOUT_ACTOR_MU && dspec.kind != EXORL_AGENT_BC && dspec.kind != EXORL_AGENT_CRR;
    const int dr = (threadIdx.x >> 4) & (HB_ROWS - 1), dj = threadIdx.x & 15;
    const bool dlive = threadIdx.x < HB_ROWS * 16 && row0 + dr < rows && dj < nout;
    float mu_v = 0.f, da_v = 0.f, ad_v = 0.f;
    if (mu_path) {
        const int i = (dlive ? row0 + dr : 0) * nout + (dlive ? dj : 0);
        mu_v = dspec.mu[i];
        for (int t = 0; t < dspec.da_nets; ++t) da_v += dspec.da[(int64_t)t * rows * nout + i];
        if (dspec.kind == EXORL_AGENT_TD3_BC) ad_v = dspec.a_data[i];
    }
    float lam = 1.0f;
    if (dspec.mode == EXORL_DOUT_ACTOR_MU && dspec.lam_parts && dspec.use_lambda) {     // lambda = alpha / mean |min(Q1,Q2)| (td3_bc.py:154)
        float sabs = 0.f;
        for (int i = threadIdx.x; i < dspec.lam_chunks; i += blockDim.x) sabs += dspec.lam_parts[2 * i];
        sabs = wave_sum(sabs);
        if ((threadIdx.x & 63) == 0) lam_s[threadIdx.x >> 6] = sabs;
        __syncthreads();
        if (threadIdx.x == 0) lam_s[4] = dspec.alpha / (((lam_s[0] + lam_s[1]) + (lam_s[2] + lam_s[3])) * dspec.inv_bg);
        __syncthreads();
        lam = lam_s[4];
    }
    if (threadIdx.x < HB_ROWS * 16) {
        float dv = 0.f;
        if (mu_path) {            // same arithmetic as dout_value's last branch
            float dmu = da_v * lam;
            if (dspec.kind == EXORL_AGENT_TD3_BC) dmu += 2.0f * dspec.inv_bg / (float)nout * (mu_v - ad_v);
            dv = dlive ? dmu * (1.0f - mu_v * mu_v) : 0.f;
        } else if (dlive) {
            dv = dout_value(dspec, net, row0 + dr, dj, rows, nout, lam);
        }
        ds[threadIdx.x] = dv;
    }
    __syncthreads();
    const int64_t nh = (int64_t)(nout + 1) * H + 16;
    float* Pn = P ? P + ((int64_t)net * gridDim.x + blockIdx.x) * nh : nullptr;
    if (want_params && threadIdx.x < nout) {      // db_out[j] partial over this chunk's rows
        float sj = 0.f;
        for (int r = 0; r < HB_ROWS; ++r) sj += ds[r * 16 + threadIdx.x];
        Pn[(int64_t)(nout + 1) * H + threadIdx.x] = sj;
    }
    if (c4 >= H4) return;
    float4 pb = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r = 0; r < HB_ROWS; ++r) {
        if (r >= nr) break;
        const int64_t o = net * astride + (int64_t)(row0 + r) * H;
        const float4 av = avs[r];
        float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < NO; ++j) {
            const float d = ds[r * 16 + j];
            sacc.x += d * w[j].x; sacc.y += d * w[j].y; sacc.z += d * w[j].z; sacc.w += d * w[j].w;
            pw[j].x += d * av.x; pw[j].y += d * av.y; pw[j].z += d * av.z; pw[j].w += d * av.w;
        }
        float4 v;
        v.x = av.x > 0.f ? sacc.x : 0.f; v.y = av.y > 0.f ? sacc.y : 0.f;
        v.z = av.z > 0.f ? sacc.z : 0.f; v.w = av.w > 0.f ? sacc.w : 0.f;
        if (dz) reinterpret_cast<float4*>(dz + o)[c4] = v;
        if (dzb) reinterpret_cast<ushort4*>(dzb + o)[c4] = f4_to_bf4(v);
        if (dzl) reinterpret_cast<ushort4*>(dzl + o)[c4] = f4_to_bf4_lo(v);
        pb.x += v.x; pb.y += v.y; pb.z += v.z; pb.w += v.w;
    }
    if (want_params) {
#pragma unroll
        for (int j = 0; j < NO; ++j)
            if (j < nout) reinterpret_cast<float4*>(Pn + (int64_t)j * H)[c4] = pw[j];
        reinterpret_cast<float4*>(Pn + (int64_t)nout * H)[c4] = pb;
    }
}

int head_bwd(const DoutSpec& dspec, const float* W, const float* a, float* dz, unsigned short* dz_bf16, float* P, int rows,
             int H, int nout, int nets, int64_t astride, int64_t pstride, int want_params, hipStream_t s, unsigned short* dz_lo) {
    EXORL_REQUIRE(nout >= 1 && nout <= 16 && H % 4 == 0 && H <= 1024, "head_bwd: nout=%d H=%d unsupported", nout, H);
    dim3 grid(cdiv(rows, HB_ROWS), nets);
    if (nout == 1) hipLaunchKernelGGL((head_bwd_kernel<1>), grid, dim3(256), 0, s, dspec, W, a, dz, dz_bf16, P, rows, H, nout, astride, pstride, want_params, dz_lo);
    else if (nout <= 8) hipLaunchKernelGGL((head_bwd_kernel<8>), grid, dim3(256), 0, s, dspec, W, a, dz, dz_bf16, P, rows, H, nout, astride, pstride, want_params, dz_lo);
    else hipLaunchKernelGGL((head_bwd_kernel<16>), grid, dim3(256), 0, s, dspec, W, a, dz, dz_bf16, P, rows, H, nout, astride, pstride, want_params, dz_lo);
    EXORL_LAUNCH_CHECK();
    return 0;
}
// 16 < nout <= 32 (CQL's 2A-wide actor head): one column per thread keeps the 2 x 32 per-output accumulators in registers.
// Same partial layout as head_bwd_kernel, with 32 db_out slots.
__global__ __launch_bounds__(256) void head_bwd_wide_kernel(const DoutSpec dspec, const float* __restrict__ W,
                                                            const float* __restrict__ a, float* __restrict__ dz,
                                                            unsigned short* __restrict__ dzb, float* __restrict__ P, int rows, int H,
                                                            int nout, int64_t astride, int64_t pstride, int want_params,
                                                            unsigned short* __restrict__ dzl) {
    __shared__ float ds[HB_ROWS * 32];
    const int net = blockIdx.z;
    const int row0 = blockIdx.y * HB_ROWS;
    const int c = blockIdx.x * 256 + threadIdx.x;
    {
        const int r = threadIdx.x >> 5, j = threadIdx.x & 31;        // 8 rows x 32 outputs = 256 threads
        ds[threadIdx.x] = (row0 + r < rows && j < nout) ? dout_value(dspec, net, row0 + r, j, rows, nout) : 0.f;
    }
    __syncthreads();
    const int64_t nh = (int64_t)(nout + 1) * H + 32;
    float* Pn = P ? P + ((int64_t)net * gridDim.y + blockIdx.y) * nh : nullptr;
    if (want_params && blockIdx.x == 0 && threadIdx.x < nout) {
        float sj = 0.f;
        for (int r = 0; r < HB_ROWS; ++r) sj += ds[r * 32 + threadIdx.x];
        Pn[(int64_t)(nout + 1) * H + threadIdx.x] = sj;
    }
    if (c >= H) return;
    float w[32], pw[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        w[j] = j < nout ? W[net * pstride + (int64_t)j * H + c] : 0.f;
        pw[j] = 0.f;
    }
    float pb = 0.f;
    const int nr = rows - row0 < HB_ROWS ? rows - row0 : HB_ROWS;
    for (int r = 0; r < nr; ++r) {
        const int64_t o = net * astride + (int64_t)(row0 + r) * H + c;
        const float av = a[o];
        float sacc = 0.f;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const float d = ds[r * 32 + j];
            sacc += d * w[j];
            pw[j] += d * av;
        }
        const float v = av > 0.f ? sacc : 0.f;
        if (dz) dz[o] = v;
        if (dzb) dzb[o] = f2bf(v);
        if (dzl) dzl[o] = f2bf_lo(v);
        pb += v;
    }
    if (want_params) {
#pragma unroll
        for (int j = 0; j < 32; ++j)
            if (j < nout) Pn[(int64_t)j * H + c] = pw[j];
        Pn[(int64_t)nout * H + c] = pb;
    }
}

int head_bwd_wide(const DoutSpec& dspec, const float* W, const float* a, float* dz, unsigned short* dz_bf16, float* P, int rows, int H,
                  int nout, int64_t astride, int64_t pstride, int want_params, hipStream_t s, unsigned short* dz_lo) {
    EXORL_REQUIRE(nout > 16 && nout <= 32, "head_bwd_wide: nout=%d out of range", nout);
    hipLaunchKernelGGL(head_bwd_wide_kernel, dim3(cdiv(H, 256), cdiv(rows, HB_ROWS), 1), dim3(256), 0, s, dspec, W, a, dz, dz_bf16, P, rows,
                       H, nout, astride, pstride, want_params, dz_lo);
    EXORL_LAUNCH_CHECK();
    return 0;
}

int head_chunks(int rows) { return cdiv(rows, HB_ROWS); }

// ------------------------------------------------------------------------------------------------
constexpr int QH_ROWS = 4;      // rows per workgroup: 256 workgroups at B = 1024 (8 rows left half the CUs idle and cost 17.6 us)
// scalar critic heads forward + backward (see QHeadArgs). One workgroup per chunk of QH_ROWS rows, thread = 4 consecutive
// columns of both critic nets; the h2 rows read for the dot products stay in registers for the dz2 pass.
// TP (mode 0): the target critic's two Q values arrive as per-row partial dots from the forward GEMM's epilogue (QHeadArgs::tpart, Gemm16Problem::
// head_part) — its hidden activations were never written: 8 MB less stored by the GEMM and 8 MB less read here at H = B = 1024
template <int MODE, bool TP>
__global__ __launch_bounds__(256) void qhead_kernel(const QHeadArgs g) {
    __shared__ float part[4][4 * QH_ROWS];      // [wave][net * QH_ROWS + r]
    __shared__ float dq[2 * QH_ROWS];
    const int row0 = blockIdx.x * QH_ROWS;
    const int c4 = threadIdx.x, H = g.H, H4 = H >> 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nr = g.rows - row0 < QH_ROWS ? g.rows - row0 : QH_ROWS;
    constexpr int nn = (MODE == 0 && !TP) ? 4 : 2;
    const bool on = c4 < H4;
    float4 avs[2][QH_ROWS];
    float4 w[2];
    float dots[4 * QH_ROWS];
    // all nn weight rows and nn x QH_ROWS activation rows in flight before the first use: the loads are unconditional (indices clamped,
    // surplus lanes / rows masked below) — guarded loads compile to one L2 round trip each (see head_fwd4)
    float4 wvs[nn], ts[nn][QH_ROWS];
    {
        const int cc = on ? c4 : 0;
#pragma unroll
        for (int n = 0; n < nn; ++n) {
            wvs[n] = reinterpret_cast<const float4*>(g.W[n])[cc];
#pragma unroll
            for (int r = 0; r < QH_ROWS; ++r) ts[n][r] = reinterpret_cast<const float4*>(g.a[n] + (int64_t)(row0 + (r < nr ? r : 0)) * H)[cc];
        }
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        if (n >= nn) {
#pragma unroll
            for (int r = 0; r < QH_ROWS; ++r) dots[n * QH_ROWS + r] = 0.f;
            continue;
        }
        const float4 wv = on ? wvs[n < nn ? n : 0] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < 2) w[n] = wv;
#pragma unroll
        for (int r = 0; r < QH_ROWS; ++r) {
            const float4 t = (on && r < nr) ? ts[n < nn ? n : 0][r] : make_float4(0.f, 0.f, 0.f, 0.f);
            dots[n * QH_ROWS + r] = (t.x * wv.x + t.y * wv.y) + (t.z * wv.z + t.w * wv.w);
            if (n < 2) avs[n][r] = t;
        }
    }
#pragma unroll
    for (int i = 0; i < 4 * QH_ROWS; ++i) {
        const float v = wave_sum(dots[i]);
        if (lane == 0) part[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4 * QH_ROWS) {             // thread i owns (net, row) = (i / QH_ROWS, i % QH_ROWS)
        const int n = threadIdx.x / QH_ROWS, r = threadIdx.x % QH_ROWS;
        if (n < nn && r < nr) {
            const float v = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x])) + g.b[n][0];
            part[0][threadIdx.x] = v;
            (n < 2 ? g.q : g.tq)[(int64_t)(n & 1) * g.rows + row0 + r] = v;
        } else if (TP && MODE == 0 && n >= 2 && r < nr) {          // folded target heads: the GEMM's column-block dots, in block order
            const float* tp = g.tpart[n - 2] + (int64_t)(row0 + r) * g.tslots;
            float v = 0.f;
            for (int j = 0; j < g.tslots; ++j) v += tp[j];
            v += g.b[n][0];
            part[0][threadIdx.x] = v;
            g.tq[(int64_t)(n & 1) * g.rows + row0 + r] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * QH_ROWS) {
        const int n = threadIdx.x / QH_ROWS, r = threadIdx.x % QH_ROWS;
        float d = 0.f;
        if (r < nr) {
            const float q1 = part[0][r], q2 = part[0][QH_ROWS + r];
            if (MODE == 0) {
                const float y = g.reward[row0 + r] + g.discount[row0 + r] * fminf(part[0][2 * QH_ROWS + r], part[0][3 * QH_ROWS + r]);
                d = 2.0f * ((n == 0 ? q1 : q2) - y) * g.inv_bg;
            } else {
                const float w1 = q1 < q2 ? 1.0f : (q1 == q2 ? 0.5f : 0.0f);
                d = -g.inv_bg * (n == 0 ? w1 : 1.0f - w1);
            }
        }
        dq[threadIdx.x] = d;
    }
    if (MODE == 1 && threadIdx.x == 0) {
        float sa = 0.f, sm = 0.f;
        for (int r = 0; r < nr; ++r) { const float m = fminf(part[0][r], part[0][QH_ROWS + r]); sa += fabsf(m); sm += m; }
        g.abs_part[2 * blockIdx.x] = sa;
        g.abs_part[2 * blockIdx.x + 1] = sm;
    }
    __syncthreads();
    const int64_t nh = 2 * (int64_t)H + 16;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        float* Pn = g.P ? g.P + ((int64_t)n * gridDim.x + blockIdx.x) * nh : nullptr;
        if (Pn && threadIdx.x == 0) {
            float sj = 0.f;
            for (int r = 0; r < QH_ROWS; ++r) sj += dq[n * QH_ROWS + r];
            Pn[2 * (int64_t)H] = sj;
        }
        if (!on) continue;
        float4 pw = make_float4(0.f, 0.f, 0.f, 0.f), pb = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < QH_ROWS; ++r) {
            if (r >= nr) break;
            const float d = dq[n * QH_ROWS + r];
            const float4 av = avs[n][r];
            pw.x += d * av.x; pw.y += d * av.y; pw.z += d * av.z; pw.w += d * av.w;
            float4 v;
            v.x = av.x > 0.f ? d * w[n].x : 0.f; v.y = av.y > 0.f ? d * w[n].y : 0.f;
            v.z = av.z > 0.f ? d * w[n].z : 0.f; v.w = av.w > 0.f ? d * w[n].w : 0.f;
            const int64_t o = n * g.act + (int64_t)(row0 + r) * H;
            if (g.dz) reinterpret_cast<float4*>(g.dz + o)[c4] = v;
            if (g.dzb) reinterpret_cast<ushort4*>(g.dzb + o)[c4] = f4_to_bf4(v);
            if (g.dzl) reinterpret_cast<ushort4*>(g.dzl + o)[c4] = f4_to_bf4_lo(v);
            pb.x += v.x; pb.y += v.y; pb.z += v.z; pb.w += v.w;
        }
        if (Pn) {
            reinterpret_cast<float4*>(Pn)[c4] = pw;
            reinterpret_cast<float4*>(Pn + H)[c4] = pb;
        }
    }
}

int qhead_chunks(int rows) { return cdiv(rows, QH_ROWS); }

int qhead(const QHeadArgs& q, hipStream_t s) {
    EXORL_REQUIRE(q.H % 4 == 0 && q.H <= 1024 && q.rows > 0 && (q.mode == 0 || q.mode == 1), "qhead: unsupported H=%d rows=%d", q.H, q.rows);
    if (q.mode == 0 && q.tpart[0]) hipLaunchKernelGGL((qhead_kernel<0, true>), dim3(cdiv(q.rows, QH_ROWS)), dim3(256), 0, s, q);
    else if (q.mode == 0) hipLaunchKernelGGL((qhead_kernel<0, false>), dim3(cdiv(q.rows, QH_ROWS)), dim3(256), 0, s, q);
    else hipLaunchKernelGGL((qhead_kernel<1, false>), dim3(cdiv(q.rows, QH_ROWS)), dim3(256), 0, s, q);
    EXORL_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// finalize: sums the per-chunk partials in chunk order and scatters into the flat gradient buffer.
// sum of n partials spaced `stride` apart, 16 loads in flight
__device__ __forceinline__ float chunk_sum(const float* __restrict__ p, int n, int64_t stride) {
    float acc = 0.f;
    int ch = 0;
    for (; ch + 32 <= n; ch += 32) {            // 32 in flight: the 256 qhead chunks are 8 dependent rounds instead of 16 (same order of adds)
        float t[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) t[q] = p[(int64_t)(ch + q) * stride];
#pragma unroll
        for (int q = 0; q < 32; ++q) acc += t[q];
    }
    for (; ch + 16 <= n; ch += 16) {
        float t[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) t[q] = p[(int64_t)(ch + q) * stride];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc += t[q];
    }
    for (; ch < n; ++ch) acc += p[(int64_t)ch * stride];
    return acc;
}

__global__ __launch_bounds__(256) void finalize_grads_kernel(FinalizeArgs f) {
    const int H = f.H;
    const int64_t nh = (int64_t)(f.nout + 1) * H + (f.nout > 16 ? 32 : 16);   // head elements per net (head_bwd[_wide]_kernel)
    const int64_t nt = 3 * (int64_t)H;                       // LN/bias column sums      (ln_bwd_kernel)
    const int64_t nw = (int64_t)f.in_dim * H;                // first-layer weight grad  (outer_reduce_kernel)
    const int64_t total = f.n_heads * nh + f.n_trunks * (nt + nw);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < f.n_heads * nh) {
            const int net = (int)(i / nh);
            const int64_t e = i % nh;
            const int64_t j = e / H;
            if (j > f.nout && e - (int64_t)(f.nout + 1) * H >= f.nout) continue;      // padding of the db_out slot
            const float sacc = chunk_sum(f.Ph + (int64_t)net * f.head_chunks * nh + e, f.head_chunks, nh);
            float* Gn = f.G + net * f.head_stride;
            if (j < f.nout) Gn[f.gW2 + e] = sacc;
            else if (j == f.nout) Gn[f.gb1 + (e - (int64_t)f.nout * H)] = sacc;
            else Gn[f.gb2 + (e - (int64_t)(f.nout + 1) * H)] = sacc;
        } else {
            const int64_t ii = i - f.n_heads * nh;
            const int net = (int)(ii / (nt + nw));
            const int64_t e = ii % (nt + nw);
            float* Gn = f.G + net * f.trunk_stride;
            if (e < nt) {
                const float sacc = chunk_sum(f.Pt + (int64_t)net * f.trunk_chunks * nt + e, f.trunk_chunks, nt);
                const int seg = (int)(e / H), c = (int)(e % H);
                if (seg == 0) Gn[f.gg + c] = sacc;
                else if (seg == 1) Gn[f.gbeta + c] = sacc;
                else Gn[f.gb0 + c] = sacc;
            } else {
                const int64_t ew = e - nt;
                const float sacc = chunk_sum(f.Pw + (int64_t)net * f.w_chunks * nw + ew, f.w_chunks, nw);
                const int k = (int)(ew / H), c = (int)(ew % H);
                Gn[f.gW0 + (int64_t)c * f.in_dim + k] = sacc;
            }
        }
    }
}

// ---- finalize + optimiser step in one launch (see FusedAdamArgs) ---------------------------------------------------
__device__ __forceinline__ void step_small(const FusedAdamArgs& a, const AdamConst& c, int64_t gi, float grad, float& pnew, float& tnew) {
    a.g[gi] = grad;
    float p = a.p[gi], m = a.m[gi], v = a.v[gi];
    adam_elem(p, grad, m, v, c);
    a.p[gi] = p; a.m[gi] = m; a.v[gi] = v;
    pnew = p; tnew = 0.f;
    if (a.target) {
        tnew = polyak(p, a.target[gi], c.tau, c.one_minus_tau);
        a.target[gi] = tnew;
    }
}

typedef float v4f_nt __attribute__((ext_vector_type(4)));
typedef unsigned v4u_nt __attribute__((ext_vector_type(4)));
// Measured and NOT adopted (round 3; exorl_gemm_tune bit 4 switches it on): non-temporal loads and stores for the optimiser pass's streams, on
// the theory that 117 MB touched once per step should not churn a 32 MB L2. The step got SLOWER — 0.3035 ms against 0.2868 ms with plain
// accesses (bench.py, same box): the gradient the wgrad GEMM has just written and the weights the next forward reads are served from the L2 /
// Infinity Cache when they are left there.
template <bool NT>
__device__ __forceinline__ float4 ld4(const float* p, int64_t i4) {
    if constexpr (NT) { const v4f_nt v = __builtin_nontemporal_load(reinterpret_cast<const v4f_nt*>(p) + i4); return make_float4(v.x, v.y, v.z, v.w); }
    else return reinterpret_cast<const float4*>(p)[i4];
}
template <bool NT>
__device__ __forceinline__ void st4(float* p, int64_t i4, const float4& v) {
    if constexpr (NT) { const v4f_nt w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, reinterpret_cast<v4f_nt*>(p) + i4); }
    else reinterpret_cast<float4*>(p)[i4] = v;
}
template <bool NT>
__device__ __forceinline__ void st4u(unsigned short* p, int64_t i8, const uint4& v) {
    if constexpr (NT) { const v4u_nt w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, reinterpret_cast<v4u_nt*>(p) + i8); }
    else reinterpret_cast<uint4*>(p)[i8] = v;
}

template <bool NT>
__global__ __launch_bounds__(256) void finalize_adam_kernel(FinalizeArgs f, FusedAdamArgs a, ShadowSpec sh, int nb_fin) {
    const AdamConst c = *a.c;
    if (a.bump && blockIdx.x == 0 && threadIdx.x == 0) *a.bump += 1ull;
    const int H = f.H;
    if ((int)blockIdx.x >= nb_fin) {             // H x H weights: gradient straight from the wgrad GEMM. 8 consecutive elements per thread
        // and pass: ten 16-byte loads in flight, and the bf16 hi/lo planes leave as 16-byte stores too (8-byte stores run at ~0.6 of the
        // 16-byte rate, MI355X_MICROARCH.md) — the pass is HBM-bound (44 B per element with a target)
        const int64_t hh8 = (int64_t)H * H / 8, n8 = a.n_heads * hh8;
        for (int64_t i = (int64_t)(blockIdx.x - nb_fin) * blockDim.x + threadIdx.x; i < n8; i += (int64_t)(gridDim.x - nb_fin) * blockDim.x) {
            const int t = (int)(i / hh8);
            const int64_t e8 = i % hh8, gi4 = a.w1_off[t] / 4 + 2 * e8;
            float4 pv[2], gv[2], mv[2], vv[2], tv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                pv[u] = ld4<NT>(a.p, gi4 + u);
                gv[u] = ld4<NT>(a.g, gi4 + u);
                mv[u] = ld4<NT>(a.m, gi4 + u);
                vv[u] = ld4<NT>(a.v, gi4 + u);
                tv[u] = a.target ? ld4<NT>(a.target, gi4 + u) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            ushort4 bh[2], bl[2], th[2], tl[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                adam_elem(pv[u].x, gv[u].x, mv[u].x, vv[u].x, c); adam_elem(pv[u].y, gv[u].y, mv[u].y, vv[u].y, c);
                adam_elem(pv[u].z, gv[u].z, mv[u].z, vv[u].z, c); adam_elem(pv[u].w, gv[u].w, mv[u].w, vv[u].w, c);
                st4<NT>(a.p, gi4 + u, pv[u]);
                st4<NT>(a.m, gi4 + u, mv[u]);
                st4<NT>(a.v, gi4 + u, vv[u]);
                bh[u] = f4_to_bf4(pv[u]); bl[u] = f4_to_bf4_lo(pv[u]);
                if (a.target) {
                    tv[u].x = polyak(pv[u].x, tv[u].x, c.tau, c.one_minus_tau); tv[u].y = polyak(pv[u].y, tv[u].y, c.tau, c.one_minus_tau);
                    tv[u].z = polyak(pv[u].z, tv[u].z, c.tau, c.one_minus_tau); tv[u].w = polyak(pv[u].w, tv[u].w, c.tau, c.one_minus_tau);
                    st4<NT>(a.target, gi4 + u, tv[u]);
                    th[u] = f4_to_bf4(tv[u]); tl[u] = f4_to_bf4_lo(tv[u]);
                }
            }
            auto pack = [](const ushort4& x, const ushort4& y) {
                uint4 r;
                r.x = x.x | ((unsigned)x.y << 16); r.y = x.z | ((unsigned)x.w << 16); r.z = y.x | ((unsigned)y.y << 16); r.w = y.z | ((unsigned)y.w << 16);
                return r;
            };
            // the bf16 planes are the next GEMMs' operands: plain stores (they should stay in cache)
            if (sh.w1b) reinterpret_cast<uint4*>(sh.w1b + (int64_t)t * H * H)[e8] = pack(bh[0], bh[1]);
            if (sh.w1l) reinterpret_cast<uint4*>(sh.w1l + (int64_t)t * H * H)[e8] = pack(bl[0], bl[1]);
            if (a.target) {
                if (sh.t_w1b) reinterpret_cast<uint4*>(sh.t_w1b + (int64_t)t * H * H)[e8] = pack(th[0], th[1]);
                if (sh.t_w1l) reinterpret_cast<uint4*>(sh.t_w1l + (int64_t)t * H * H)[e8] = pack(tl[0], tl[1]);
            }
        }
        return;
    }
    const int64_t nh = (int64_t)(f.nout + 1) * H + (f.nout > 16 ? 32 : 16);
    const int64_t nt = 3 * (int64_t)H;
    const int64_t nw = (int64_t)f.in_dim * H;
    const int64_t total = f.n_heads * nh + f.n_trunks * (nt + nw);
    const int64_t Kp = (f.in_dim + 31) / 32 * 32;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)nb_fin * blockDim.x) {
        float pn, tn;
        if (i < f.n_heads * nh) {
            const int net = (int)(i / nh);
            const int64_t e = i % nh;
            const int64_t j = e / H;
            if (j > f.nout && e - (int64_t)(f.nout + 1) * H >= f.nout) continue;
            const float sacc = chunk_sum(f.Ph + (int64_t)net * f.head_chunks * nh + e, f.head_chunks, nh);
            const int64_t base = net * f.head_stride;
            const int64_t gi = base + (j < f.nout ? f.gW2 + e : (j == f.nout ? f.gb1 + (e - (int64_t)f.nout * H) : f.gb2 + (e - (int64_t)(f.nout + 1) * H)));
            step_small(a, c, gi, sacc, pn, tn);
        } else {
            const int64_t ii = i - f.n_heads * nh;
            const int net = (int)(ii / (nt + nw));
            const int64_t e = ii % (nt + nw);
            const int64_t base = net * f.trunk_stride;
            if (e < nt) {
                const float sacc = chunk_sum(f.Pt + (int64_t)net * f.trunk_chunks * nt + e, f.trunk_chunks, nt);
                const int seg = (int)(e / H), cc = (int)(e % H);
                step_small(a, c, base + (seg == 0 ? f.gg : (seg == 1 ? f.gbeta : f.gb0)) + cc, sacc, pn, tn);
            } else {
                const int64_t ew = e - nt;
                const float sacc = chunk_sum(f.Pw + (int64_t)net * f.w_chunks * nw + ew, f.w_chunks, nw);
                const int k = (int)(ew / H), cc = (int)(ew % H);
                step_small(a, c, base + f.gW0 + (int64_t)cc * f.in_dim + k, sacc, pn, tn);
                // derived copies of the first-layer weight (ShadowSpec): transposed fp32 and K-padded bf16
                sh.w0t[net * nw + (int64_t)k * H + cc] = pn;
                if (sh.w0b) sh.w0b[(int64_t)net * H * Kp + (int64_t)cc * Kp + k] = f2bf(pn);
                if (sh.w0l) sh.w0l[(int64_t)net * H * Kp + (int64_t)cc * Kp + k] = f2bf_lo(pn);
                if (a.target && sh.t_w0t) {
                    sh.t_w0t[net * nw + (int64_t)k * H + cc] = tn;
                    if (sh.t_w0b) sh.t_w0b[(int64_t)net * H * Kp + (int64_t)cc * Kp + k] = f2bf(tn);
                    if (sh.t_w0l) sh.t_w0l[(int64_t)net * H * Kp + (int64_t)cc * Kp + k] = f2bf_lo(tn);
                }
            }
        }
    }
}

// part 0: the whole optimiser pass; 1: everything but the H x H weights; 2: the H x H weights only (their gradient is complete as soon
// as the wgrad GEMM is, so that ~80 % of the pass — the HBM-bound part — can run beside the LayerNorm-backward / first-layer-wgrad chain
// that produces the other gradients; `f` is not read then)
int finalize_adam(const FinalizeArgs& f, const FusedAdamArgs& a, const ShadowSpec& sh, hipStream_t s, int part) {
    EXORL_REQUIRE(sh.H % 4 == 0 && a.n_heads >= 1 && a.n_heads <= 2 && part >= 0 && part <= 2, "finalize_adam: unsupported geometry");
    const int H = sh.H;
    const int64_t total = part == 2 ? 0 : f.n_heads * ((int64_t)(f.nout + 1) * f.H + (f.nout > 16 ? 32 : 16)) + f.n_trunks * (int64_t)(3 + f.in_dim) * f.H;
    const int nb_fin = cdiv(total, 256);
    int nb_w1 = part == 1 ? 0 : cdiv((int64_t)a.n_heads * H * H / 8, 256);
    if (nb_w1 > 2048) nb_w1 = 2048;
    FusedAdamArgs aa = a;
    if (part == 2) aa.bump = nullptr;
    FinalizeArgs ff = f;
    if (part == 2) ff.H = H;
    if (tune_variant() & 4) hipLaunchKernelGGL(finalize_adam_kernel<true>, dim3(nb_fin + nb_w1), dim3(256), 0, s, ff, aa, sh, nb_fin);
    else hipLaunchKernelGGL(finalize_adam_kernel<false>, dim3(nb_fin + nb_w1), dim3(256), 0, s, ff, aa, sh, nb_fin);
    EXORL_LAUNCH_CHECK();
    return 0;
}

int finalize_grads(const FinalizeArgs& f, hipStream_t s) {
    const int64_t total = f.n_heads * ((int64_t)(f.nout + 1) * f.H + (f.nout > 16 ? 32 : 16)) + f.n_trunks * (int64_t)(3 + f.in_dim) * f.H;
    hipLaunchKernelGGL(finalize_grads_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, f);
    EXORL_LAUNCH_CHECK();
    return 0;
}

}  // namespace exorl
