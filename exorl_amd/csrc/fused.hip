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

constexpr int TR = 16;            // rows per trunk workgroup
constexpr int CPT = 4;            // columns per thread (H <= 1024)
constexpr int MAX_IN = 256;       // first-layer fan-in limit of the fused trunk kernels
constexpr float LN_EPS2 = 1e-5f;
typedef __bf16 bf16_t;

__device__ __forceinline__ unsigned short f2bf(float x) {
    bf16_t b = (bf16_t)x;
    return __builtin_bit_cast(unsigned short, b);
}

// sums v[0..N) over the 256-thread block; result valid in all threads
template <int N>
__device__ __forceinline__ void block_sum256(float (&v)[N], float* red /* [N][4] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float s = wave_sum(v[i]);
        if (lane == 0) red[i * 4 + wave] = s;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = red[i * 4 + 0] + red[i * 4 + 1] + red[i * 4 + 2] + red[i * 4 + 3];
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// trunk forward: h = tanh(LN(x W0^T + b0) * g + beta)
__global__ __launch_bounds__(256) void trunk_fwd_kernel(const float* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ W0T, const float* __restrict__ b0,
                                                        const float* __restrict__ gain, const float* __restrict__ beta,
                                                        float* __restrict__ h, float* __restrict__ xhat,
                                                        float* __restrict__ rstd, unsigned short* __restrict__ hb,
                                                        int rows, int in_dim, int H, int64_t astride, int64_t pstride,
                                                        int64_t tstride) {
    __shared__ __attribute__((aligned(16))) float xs[MAX_IN * TR];     // [k][r]
    __shared__ float red[TR * 4];
    const int net = blockIdx.y;
    const int row0 = blockIdx.x * TR;
    const int tid = threadIdx.x;
    for (int i = tid; i < TR * in_dim; i += 256) {
        const int r = i / in_dim, k = i % in_dim;
        xs[k * TR + r] = (row0 + r < rows) ? x[(int64_t)(row0 + r) * ldx + k] : 0.f;
    }
    __syncthreads();
    const float* Wt = W0T + net * tstride;
    float z[CPT][TR];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = tid + 256 * i;
        const float b = c < H ? b0[net * pstride + c] : 0.f;
#pragma unroll
        for (int r = 0; r < TR; ++r) z[i][r] = b;
    }
    for (int k = 0; k < in_dim; ++k) {
        float w[CPT];
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = tid + 256 * i;
            w[i] = c < H ? Wt[(int64_t)k * H + c] : 0.f;
        }
#pragma unroll
        for (int r4 = 0; r4 < TR / 4; ++r4) {
            const float4 xv = *reinterpret_cast<const float4*>(&xs[k * TR + 4 * r4]);
#pragma unroll
            for (int i = 0; i < CPT; ++i) {
                z[i][4 * r4 + 0] += w[i] * xv.x;
                z[i][4 * r4 + 1] += w[i] * xv.y;
                z[i][4 * r4 + 2] += w[i] * xv.z;
                z[i][4 * r4 + 3] += w[i] * xv.w;
            }
        }
    }
    // LayerNorm statistics (two-pass), biased variance
    float s[TR];
#pragma unroll
    for (int r = 0; r < TR; ++r) {
        s[r] = 0.f;
#pragma unroll
        for (int i = 0; i < CPT; ++i) s[r] += (tid + 256 * i < H) ? z[i][r] : 0.f;
    }
    block_sum256<TR>(s, red);
    float mean[TR];
#pragma unroll
    for (int r = 0; r < TR; ++r) {
        mean[r] = s[r] / (float)H;
        s[r] = 0.f;
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const float d = (tid + 256 * i < H) ? z[i][r] - mean[r] : 0.f;
            z[i][r] = d;
            s[r] += d * d;
        }
    }
    block_sum256<TR>(s, red);
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = tid + 256 * i;
        if (c >= H) continue;
        const float g = gain[net * pstride + c], be = beta[net * pstride + c];
#pragma unroll
        for (int r = 0; r < TR; ++r) {
            if (row0 + r >= rows) continue;
            const float rs = 1.0f / sqrtf(s[r] / (float)H + LN_EPS2);
            const float xh = z[i][r] * rs;
            const float hv = tanhf(xh * g + be);
            const int64_t o = net * astride + (int64_t)(row0 + r) * H + c;
            h[o] = hv;
            if (xhat) xhat[o] = xh;
            if (hb) hb[o] = f2bf(hv);
        }
    }
    if (rstd && tid < TR && row0 + tid < rows) rstd[net * (int64_t)rows + row0 + tid] = 1.0f / sqrtf(s[tid] / (float)H + LN_EPS2);
}

int trunk_fwd(const float* x, int64_t ldx, const float* W0T, const float* b0, const float* gain, const float* beta,
              float* h, float* xhat, float* rstd, unsigned short* h_bf16, int rows, int in_dim, int H, int nets,
              int64_t astride, int64_t pstride, int64_t tstride, hipStream_t s) {
    EXORL_REQUIRE(H >= 1 && H <= 256 * CPT && in_dim >= 1 && in_dim <= MAX_IN, "trunk_fwd: unsupported H=%d in=%d", H, in_dim);
    hipLaunchKernelGGL(trunk_fwd_kernel, dim3(cdiv(rows, TR), nets), dim3(256), 0, s, x, ldx, W0T, b0, gain, beta, h, xhat, rstd,
                       h_bf16, rows, in_dim, H, astride, pstride, tstride);
    EXORL_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// trunk backward: dz0 = LN/tanh backward of dh; per-workgroup partials of dgain, dbeta, db0, dW0 (transposed
// [k][c]); optional dx[:, col0:col0+ncols] partial per net.  P layout per (net, chunk): [dg H][dbeta H][db0 H][dW0T in*H]
template <bool PARAMS, bool DX>
__global__ __launch_bounds__(256) void trunk_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ h,
                                                        const float* __restrict__ xhat, const float* __restrict__ rstd,
                                                        const float* __restrict__ gain, const float* __restrict__ x,
                                                        int64_t ldx, const float* __restrict__ W0T, float* __restrict__ P,
                                                        float* __restrict__ dx, int dx_col0, int dx_cols, int rows,
                                                        int in_dim, int H, int64_t astride, int64_t pstride,
                                                        int64_t tstride) {
    __shared__ __attribute__((aligned(16))) float xs[PARAMS ? MAX_IN * TR : 4];
    __shared__ float red[2 * TR * 4];
    __shared__ float dxs[DX ? 4 : 1][TR * 16];
    const int net = blockIdx.y;
    const int row0 = blockIdx.x * TR;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if constexpr (PARAMS) {
        for (int i = tid; i < TR * in_dim; i += 256) {
            const int r = i / in_dim, k = i % in_dim;
            xs[k * TR + r] = (row0 + r < rows) ? x[(int64_t)(row0 + r) * ldx + k] : 0.f;
        }
    }
    // pass 1 (row-major so only one row's loads are in flight: VGPR budget): dxh = dh*(1-h^2)*gain, row sums
    float dz[CPT][TR];
    float pg[CPT], pb[CPT], pb0[CPT], g[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = tid + 256 * i;
        pg[i] = pb[i] = pb0[i] = 0.f;
        g[i] = c < H ? gain[net * pstride + c] : 0.f;
    }
    const int64_t base = net * astride + (int64_t)row0 * H;
#pragma unroll
    for (int r = 0; r < TR; ++r) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = tid + 256 * i;
            float d = 0.f;
            if (c < H && row0 + r < rows) {
                const int64_t o = base + (int64_t)r * H + c;
                const float hv = h[o], xh = xhat[o];
                const float dy = dh[o] * (1.0f - hv * hv);
                pg[i] += dy * xh;
                pb[i] += dy;
                d = dy * g[i];
                s1 += d;
                s2 += d * xh;
            }
            dz[i][r] = d;
        }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if (lane == 0) { red[(2 * r) * 4 + wave] = s1; red[(2 * r + 1) * 4 + wave] = s2; }
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    // pass 2: dz0 = rstd * (dxh - mean(dxh) - xhat * mean(dxh*xhat))
#pragma unroll
    for (int r = 0; r < TR; ++r) {
        const float m1 = (red[(2 * r) * 4] + red[(2 * r) * 4 + 1] + red[(2 * r) * 4 + 2] + red[(2 * r) * 4 + 3]) / (float)H;
        const float m2 = (red[(2 * r + 1) * 4] + red[(2 * r + 1) * 4 + 1] + red[(2 * r + 1) * 4 + 2] + red[(2 * r + 1) * 4 + 3]) / (float)H;
        const float rs = (row0 + r < rows) ? rstd[net * (int64_t)rows + row0 + r] : 0.f;
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = tid + 256 * i;
            float d = 0.f;
            if (c < H && row0 + r < rows) d = rs * (dz[i][r] - m1 - xhat[base + (int64_t)r * H + c] * m2);
            dz[i][r] = d;
            pb0[i] += d;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (PARAMS) {
        const int64_t psz = (int64_t)(3 + in_dim) * H;
        float* Pn = P + ((int64_t)net * gridDim.x + blockIdx.x) * psz;
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = tid + 256 * i;
            if (c < H) { Pn[c] = pg[i]; Pn[H + c] = pb[i]; Pn[2 * H + c] = pb0[i]; }
        }
        for (int k = 0; k < in_dim; ++k) {
            float acc[CPT] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r4 = 0; r4 < TR / 4; ++r4) {
                const float4 xv = *reinterpret_cast<const float4*>(&xs[k * TR + 4 * r4]);
#pragma unroll
                for (int i = 0; i < CPT; ++i)
                    acc[i] += dz[i][4 * r4] * xv.x + dz[i][4 * r4 + 1] * xv.y + dz[i][4 * r4 + 2] * xv.z + dz[i][4 * r4 + 3] * xv.w;
            }
#pragma unroll
            for (int i = 0; i < CPT; ++i) {
                const int c = tid + 256 * i;
                if (c < H) Pn[(int64_t)(3 + k) * H + c] = acc[i];
            }
        }
    }
    if constexpr (DX) {       // dx[r][j] = sum_c dz0[r][c] W0[c][col0+j]  (reduction over the columns held by all threads)
        const float* Wt = W0T + net * tstride;
        for (int j = 0; j < dx_cols; ++j) {
            float w[CPT];
#pragma unroll
            for (int i = 0; i < CPT; ++i) {
                const int c = tid + 256 * i;
                w[i] = c < H ? Wt[(int64_t)(dx_col0 + j) * H + c] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < TR; ++r) {
                float v = 0.f;
#pragma unroll
                for (int i = 0; i < CPT; ++i) v += dz[i][r] * w[i];
                v = wave_sum(v);
                if (lane == 0) dxs[wave][r * 16 + j] = v;
            }
        }
        __syncthreads();
        for (int i = tid; i < TR * dx_cols; i += 256) {
            const int r = i / dx_cols, j = i % dx_cols;
            if (row0 + r < rows)
                dx[((int64_t)net * rows + row0 + r) * dx_cols + j] = dxs[0][r * 16 + j] + dxs[1][r * 16 + j] + dxs[2][r * 16 + j] + dxs[3][r * 16 + j];
        }
    }
}

int trunk_bwd(const float* dh, const float* h, const float* xhat, const float* rstd, const float* gain, const float* x,
              int64_t ldx, const float* W0T, float* P, float* dx, int dx_col0, int dx_cols, int rows, int in_dim, int H,
              int nets, int64_t astride, int64_t pstride, int64_t tstride, int want_params, hipStream_t s) {
    EXORL_REQUIRE(H >= 1 && H <= 256 * CPT && in_dim >= 1 && in_dim <= MAX_IN && dx_cols <= 16, "trunk_bwd: unsupported H=%d in=%d", H, in_dim);
    const dim3 grid(cdiv(rows, TR), nets), block(256);
#define EXORL_TB(PA, DXX) hipLaunchKernelGGL((trunk_bwd_kernel<PA, DXX>), grid, block, 0, s, dh, h, xhat, rstd, gain, x, ldx, W0T, P, dx, \
                                             dx_col0, dx_cols, rows, in_dim, H, astride, pstride, tstride)
    if (want_params && dx) EXORL_TB(true, true);
    else if (want_params) EXORL_TB(true, false);
    else if (dx) EXORL_TB(false, true);
    else { set_error("trunk_bwd: nothing to compute"); return 2; }
#undef EXORL_TB
    EXORL_LAUNCH_CHECK();
    return 0;
}
int trunk_chunks(int rows) { return cdiv(rows, TR); }

// ------------------------------------------------------------------------------------------------
// head forward v2: out[m][j] = b[j] + sum_c a[m][c] W[j][c]; one wave per row, float4 streams (H % 4 == 0)
template <int NO>
__global__ __launch_bounds__(256) void head_fwd4_kernel(const float* __restrict__ a, const float* __restrict__ W,
                                                        const float* __restrict__ b, float* __restrict__ out, int rows,
                                                        int H, int nout, int tanh_out, int64_t astride, int64_t pstride,
                                                        int64_t ostride) {
    const int net = blockIdx.y;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float4* ar = reinterpret_cast<const float4*>(a + net * astride + (int64_t)row * H);
    const float* Wn = W + net * pstride;
    float acc[NO];
#pragma unroll
    for (int j = 0; j < NO; ++j) acc[j] = 0.f;
    for (int c4 = lane; c4 < H / 4; c4 += 64) {
        const float4 av = ar[c4];
#pragma unroll
        for (int j = 0; j < NO; ++j) {
            if (j < nout) {
                const float4 wv = reinterpret_cast<const float4*>(Wn + (int64_t)j * H)[c4];
                acc[j] += av.x * wv.x + av.y * wv.y + av.z * wv.z + av.w * wv.w;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NO; ++j) {
        if (j < nout) {
            float v = wave_sum(acc[j]) + b[net * pstride + j];
            if (tanh_out) v = tanhf(v);
            if (lane == 0) out[net * ostride + (int64_t)row * nout + j] = v;
        }
    }
}

int head_fwd4(const float* a, const float* W, const float* b, float* out, int rows, int H, int nout, int tanh_out,
              int nets, int64_t astride, int64_t pstride, int64_t ostride, hipStream_t s) {
    EXORL_REQUIRE(nout >= 1 && nout <= 16 && H % 4 == 0, "head_fwd4: nout=%d H=%d unsupported", nout, H);
    dim3 grid(cdiv(rows, 4), nets);
    if (nout == 1) hipLaunchKernelGGL((head_fwd4_kernel<1>), grid, dim3(256), 0, s, a, W, b, out, rows, H, nout, tanh_out, astride, pstride, ostride);
    else if (nout <= 8) hipLaunchKernelGGL((head_fwd4_kernel<8>), grid, dim3(256), 0, s, a, W, b, out, rows, H, nout, tanh_out, astride, pstride, ostride);
    else hipLaunchKernelGGL((head_fwd4_kernel<16>), grid, dim3(256), 0, s, a, W, b, out, rows, H, nout, tanh_out, astride, pstride, ostride);
    EXORL_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// head backward: dz[m][c] = (sum_j dout[m][j] W[j][c]) * (a[m][c] > 0), written fp32 and/or bf16, plus
// per-chunk partials of db_hidden[c] = sum_m dz and dW[j][c] = sum_m dout[m][j] a[m][c].
// P layout per (net, chunk): [dW nout*H][db_hidden H][db_out 16]
constexpr int HB_ROWS = 32;
template <int NO>
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ W,
                                                       const float* __restrict__ a, float* __restrict__ dz,
                                                       unsigned short* __restrict__ dzb, float* __restrict__ P, int rows,
                                                       int H, int nout, int64_t astride, int64_t pstride, int64_t dstride,
                                                       int want_params) {
    __shared__ float ds[HB_ROWS * 16];
    const int net = blockIdx.z;
    const int row0 = blockIdx.y * HB_ROWS;
    const int c = blockIdx.x * 256 + threadIdx.x;
    for (int i = threadIdx.x; i < HB_ROWS * nout; i += 256) {
        const int r = i / nout, j = i % nout;
        ds[r * 16 + j] = (row0 + r < rows) ? dout[net * dstride + (int64_t)(row0 + r) * nout + j] : 0.f;
    }
    __syncthreads();
    const int64_t nh = (int64_t)(nout + 1) * H + 16;
    float* Pn = P ? P + ((int64_t)net * gridDim.y + blockIdx.y) * nh : nullptr;
    if (want_params && blockIdx.x == 0 && threadIdx.x < nout) {      // db_out[j] partial over this chunk's rows
        float sj = 0.f;
        for (int r = 0; r < HB_ROWS; ++r) sj += ds[r * 16 + threadIdx.x];
        Pn[(int64_t)(nout + 1) * H + threadIdx.x] = sj;
    }
    if (c >= H) return;
    float w[NO], pw[NO];
#pragma unroll
    for (int j = 0; j < NO; ++j) {
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
        for (int j = 0; j < NO; ++j) {
            const float d = ds[r * 16 + j];
            sacc += d * w[j];
            pw[j] += d * av;
        }
        const float v = av > 0.f ? sacc : 0.f;
        if (dz) dz[o] = v;
        if (dzb) dzb[o] = f2bf(v);
        pb += v;
    }
    if (want_params) {
#pragma unroll
        for (int j = 0; j < NO; ++j)
            if (j < nout) Pn[(int64_t)j * H + c] = pw[j];
        Pn[(int64_t)nout * H + c] = pb;
    }
}

int head_bwd(const float* dout, const float* W, const float* a, float* dz, unsigned short* dz_bf16, float* P, int rows,
             int H, int nout, int nets, int64_t astride, int64_t pstride, int64_t dstride, int want_params, hipStream_t s) {
    EXORL_REQUIRE(nout >= 1 && nout <= 16, "head_bwd: nout=%d unsupported", nout);
    dim3 grid(cdiv(H, 256), cdiv(rows, HB_ROWS), nets);
    if (nout == 1) hipLaunchKernelGGL((head_bwd_kernel<1>), grid, dim3(256), 0, s, dout, W, a, dz, dz_bf16, P, rows, H, nout, astride, pstride, dstride, want_params);
    else if (nout <= 8) hipLaunchKernelGGL((head_bwd_kernel<8>), grid, dim3(256), 0, s, dout, W, a, dz, dz_bf16, P, rows, H, nout, astride, pstride, dstride, want_params);
    else hipLaunchKernelGGL((head_bwd_kernel<16>), grid, dim3(256), 0, s, dout, W, a, dz, dz_bf16, P, rows, H, nout, astride, pstride, dstride, want_params);
    EXORL_LAUNCH_CHECK();
    return 0;
}
int head_chunks(int rows) { return cdiv(rows, HB_ROWS); }

// ------------------------------------------------------------------------------------------------
// finalize: sums the per-chunk partials in chunk order and scatters into the flat gradient buffer.
__global__ __launch_bounds__(256) void finalize_grads_kernel(FinalizeArgs f) {
    const int H = f.H;
    const int64_t nh = (int64_t)(f.nout + 1) * H + 16;       // head elements per net (see head_bwd_kernel)
    const int64_t nt = (int64_t)(3 + f.in_dim) * H;          // trunk elements per net (see trunk_bwd_kernel)
    const int64_t total = f.n_heads * nh + f.n_trunks * nt;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < f.n_heads * nh) {
            const int net = (int)(i / nh);
            const int64_t e = i % nh;
            const int64_t j = e / H;
            if (j > f.nout && e - (int64_t)(f.nout + 1) * H >= f.nout) continue;      // padding of the db_out slot
            const float* p = f.Ph + (int64_t)net * f.head_chunks * nh + e;
            float sacc = 0.f;
            for (int ch = 0; ch < f.head_chunks; ++ch) sacc += p[(int64_t)ch * nh];
            float* Gn = f.G + net * f.head_stride;
            if (j < f.nout) Gn[f.gW2 + e] = sacc;
            else if (j == f.nout) Gn[f.gb1 + (e - (int64_t)f.nout * H)] = sacc;
            else Gn[f.gb2 + (e - (int64_t)(f.nout + 1) * H)] = sacc;
        } else {
            const int64_t ii = i - f.n_heads * nh;
            const int net = (int)(ii / nt);
            const int64_t e = ii % nt;
            const float* p = f.Pt + (int64_t)net * f.trunk_chunks * nt + e;
            float sacc = 0.f;
            for (int ch = 0; ch < f.trunk_chunks; ++ch) sacc += p[(int64_t)ch * nt];
            const int seg = (int)(e / H), c = (int)(e % H);
            float* Gn = f.G + net * f.trunk_stride;
            if (seg == 0) Gn[f.gg + c] = sacc;
            else if (seg == 1) Gn[f.gbeta + c] = sacc;
            else if (seg == 2) Gn[f.gb0 + c] = sacc;
            else Gn[f.gW0 + (int64_t)c * f.in_dim + (seg - 3)] = sacc;
        }
    }
}

int finalize_grads(const FinalizeArgs& f, hipStream_t s) {
    const int64_t total = f.n_heads * ((int64_t)(f.nout + 1) * f.H + 16) + f.n_trunks * (int64_t)(3 + f.in_dim) * f.H;
    hipLaunchKernelGGL(finalize_grads_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, f);
    EXORL_LAUNCH_CHECK();
    return 0;
}

}  // namespace exorl
