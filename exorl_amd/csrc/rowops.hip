// Row-wise and column-wise layer kernels around the MFMA GEMMs:
//   LayerNorm+Tanh forward/backward  (nn.LayerNorm(H), nn.Tanh: td3_bc.py:17,39; ddpg.py:49,93)
//   output heads Linear(H, n<=16)    (td3_bc.py:20,26,41; ddpg.py:62,107) forward / dgrad(+ReLU mask) / wgrad
//   bias / LayerNorm-affine gradients (column sums over the batch)
// All HBM/L2-bound byte movers: one wave per row (64 lanes stride the 1024 columns -> 256 B coalesced
// segments), column reductions as 64 columns x 16 row-groups per 1024-thread workgroup (deterministic,
// no atomics). `nets` independent nets per launch on blockIdx.y.
#include "kernels.h"

namespace exorl {

constexpr int MAX_PER_LANE = 16;   // H <= 1024
constexpr float LN_EPS = 1e-5f;

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ln_tanh_fwd_kernel(const float* z, const float* __restrict__ gain,
                                                          const float* __restrict__ beta, float* h,
                                                          float* __restrict__ xhat, float* __restrict__ rstd,
                                                          int rows, int H, int64_t astride, int64_t pstride) {
    const int net = blockIdx.y;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* zr = z + net * astride + (int64_t)row * H;
    const float* g = gain + net * pstride;
    const float* b = beta + net * pstride;
    float v[MAX_PER_LANE];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < H ? zr[c] : 0.f;
        s += v[i];
    }
    const float mean = wave_sum(s) / (float)H;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        const float d = c < H ? v[i] - mean : 0.f;
        v[i] = d;
        s2 += d * d;
    }
    const float var = wave_sum(s2) / (float)H;
    const float rs = 1.0f / sqrtf(var + LN_EPS);
    float* hr = h + net * astride + (int64_t)row * H;
    float* xr = xhat ? xhat + net * astride + (int64_t)row * H : nullptr;
#pragma unroll
    for (int i = 0; i < MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        if (c < H) {
            const float xh = v[i] * rs;
            if (xr) xr[c] = xh;
            hr[c] = tanhf(xh * g[c] + b[c]);
        }
    }
    if (rstd && lane == 0) rstd[net * (int64_t)rows + row] = rs;
}

int ln_tanh_fwd(const float* z, const float* gain, const float* beta, float* h, float* xhat, float* rstd,
                int rows, int H, int nets, int64_t astride, int64_t pstride, hipStream_t s) {
    EXORL_REQUIRE(H >= 1 && H <= 64 * MAX_PER_LANE, "ln_tanh_fwd: H=%d unsupported (max %d)", H, 64 * MAX_PER_LANE);
    dim3 grid(cdiv(rows, 4), nets);
    hipLaunchKernelGGL(ln_tanh_fwd_kernel, grid, dim3(256), 0, s, z, gain, beta, h, xhat, rstd, rows, H, astride, pstride);
    EXORL_LAUNCH_CHECK();
    return 0;
}

// dz = rstd * (dxh - mean(dxh) - xhat * mean(dxh*xhat)),  dxh = dh*(1-h^2)*gain
__global__ __launch_bounds__(256) void ln_tanh_bwd_kernel(const float* dh, const float* __restrict__ h,
                                                          const float* __restrict__ xhat, const float* __restrict__ rstd,
                                                          const float* __restrict__ gain, float* dz,
                                                          int rows, int H, int64_t astride, int64_t pstride) {
    const int net = blockIdx.y;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int64_t base = net * astride + (int64_t)row * H;
    const float* g = gain + net * pstride;
    float dx[MAX_PER_LANE], xh[MAX_PER_LANE];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        if (c < H) {
            const float hv = h[base + c];
            const float d = dh[base + c] * (1.0f - hv * hv) * g[c];
            dx[i] = d;
            xh[i] = xhat[base + c];
            s1 += d;
            s2 += d * xh[i];
        } else {
            dx[i] = 0.f; xh[i] = 0.f;
        }
    }
    const float m1 = wave_sum(s1) / (float)H;
    const float m2 = wave_sum(s2) / (float)H;
    const float rs = rstd[net * (int64_t)rows + row];
#pragma unroll
    for (int i = 0; i < MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        if (c < H) dz[base + c] = rs * (dx[i] - m1 - xh[i] * m2);
    }
}

int ln_tanh_bwd(const float* dh, const float* h, const float* xhat, const float* rstd, const float* gain, float* dz,
                int rows, int H, int nets, int64_t astride, int64_t pstride, hipStream_t s) {
    EXORL_REQUIRE(H >= 1 && H <= 64 * MAX_PER_LANE, "ln_tanh_bwd: H=%d unsupported", H);
    dim3 grid(cdiv(rows, 4), nets);
    hipLaunchKernelGGL(ln_tanh_bwd_kernel, grid, dim3(256), 0, s, dh, h, xhat, rstd, gain, dz, rows, H, astride, pstride);
    EXORL_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Column reductions: 64 columns x 16 row-groups per workgroup.
__device__ __forceinline__ float block_colreduce(float v, float (*red)[64], int cx, int ry) {
    __syncthreads();
    red[ry][cx] = v;
    __syncthreads();
    float s = 0.f;
    if (ry == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s += red[i][cx];
    }
    return s;
}

// dgain[c] = sum_m dy*xhat, dbeta[c] = sum_m dy,  dy = dh*(1-h^2)
__global__ __launch_bounds__(1024) void ln_param_grad_kernel(const float* __restrict__ dh, const float* __restrict__ h,
                                                             const float* __restrict__ xhat, float* __restrict__ dgain,
                                                             float* __restrict__ dbeta, int rows, int H,
                                                             int64_t astride, int64_t pstride) {
    __shared__ float red[16][64];
    const int net = blockIdx.y;
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    float sg = 0.f, sb = 0.f;
    if (c < H) {
        for (int r = ry; r < rows; r += 16) {
            const int64_t i = net * astride + (int64_t)r * H + c;
            const float hv = h[i];
            const float dy = dh[i] * (1.0f - hv * hv);
            sg += dy * xhat[i];
            sb += dy;
        }
    }
    const float tg = block_colreduce(sg, red, cx, ry);
    const float tb = block_colreduce(sb, red, cx, ry);
    if (ry == 0 && c < H) {
        dgain[net * pstride + c] = tg;
        dbeta[net * pstride + c] = tb;
    }
}

// vector form (H % 4 == 0): 32 columns per workgroup (8 float4 lanes x 32 row lanes), as colsum4_kernel
__global__ __launch_bounds__(256) void ln_param_grad4_kernel(const float* __restrict__ dh, const float* __restrict__ h,
                                                              const float* __restrict__ xhat, float* __restrict__ dgain,
                                                              float* __restrict__ dbeta, int rows, int H, int64_t astride, int64_t pstride) {
    __shared__ float4 red[2][32][8];
    const int net = blockIdx.y;
    const int cg = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int c = blockIdx.x * 32 + 4 * cg;
    float4 sg = make_float4(0.f, 0.f, 0.f, 0.f), sb = sg;
    if (c < H) {
        for (int r0 = rl; r0 < rows; r0 += 64) {
            float4 hv[2], dv[2], xv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int r = r0 + 32 * u;
                if (r < rows) {
                    const int64_t i = net * astride + (int64_t)r * H + c;
                    hv[u] = *reinterpret_cast<const float4*>(h + i);
                    dv[u] = *reinterpret_cast<const float4*>(dh + i);
                    xv[u] = *reinterpret_cast<const float4*>(xhat + i);
                } else {
                    hv[u] = dv[u] = xv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float4 dy = make_float4(dv[u].x * (1.0f - hv[u].x * hv[u].x), dv[u].y * (1.0f - hv[u].y * hv[u].y),
                                              dv[u].z * (1.0f - hv[u].z * hv[u].z), dv[u].w * (1.0f - hv[u].w * hv[u].w));
                sg.x += dy.x * xv[u].x; sg.y += dy.y * xv[u].y; sg.z += dy.z * xv[u].z; sg.w += dy.w * xv[u].w;
                sb.x += dy.x; sb.y += dy.y; sb.z += dy.z; sb.w += dy.w;
            }
        }
    }
    red[0][rl][cg] = sg;
    red[1][rl][cg] = sb;
    __syncthreads();
    if (rl < 2 && c < H) {
        float4 t = red[rl][0][cg];
#pragma unroll
        for (int i = 1; i < 32; ++i) { t.x += red[rl][i][cg].x; t.y += red[rl][i][cg].y; t.z += red[rl][i][cg].z; t.w += red[rl][i][cg].w; }
        *reinterpret_cast<float4*>((rl == 0 ? dgain : dbeta) + net * pstride + c) = t;
    }
}

int ln_param_grad(const float* dh, const float* h, const float* xhat, float* dgain, float* dbeta,
                  int rows, int H, int nets, int64_t astride, int64_t pstride, hipStream_t s) {
    auto al = [](const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
    if (H % 4 == 0 && astride % 4 == 0 && pstride % 4 == 0 && al(dh) && al(h) && al(xhat) && al(dgain) && al(dbeta)) {
        hipLaunchKernelGGL(ln_param_grad4_kernel, dim3(cdiv(H, 32), nets), dim3(256), 0, s, dh, h, xhat, dgain, dbeta, rows, H, astride, pstride);
        EXORL_LAUNCH_CHECK();
        return 0;
    }
    dim3 grid(cdiv(H, 64), nets);
    hipLaunchKernelGGL(ln_param_grad_kernel, grid, dim3(1024), 0, s, dh, h, xhat, dgain, dbeta, rows, H, astride, pstride);
    EXORL_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, int rows,
                                                      int cols, int64_t astride, int64_t pstride) {
    __shared__ float red[16][64];
    const int net = blockIdx.y;
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    float sum = 0.f;
    if (c < cols)
        for (int r = ry; r < rows; r += 16) sum += x[net * astride + (int64_t)r * cols + c];
    const float t = block_colreduce(sum, red, cx, ry);
    if (ry == 0 && c < cols) out[net * pstride + c] = t;
}

// Vector form (cols % 4 == 0, 16-byte aligned): a workgroup owns 32 columns (8 float4 lanes x 32 row lanes) -> cols/32 workgroups
// instead of cols/64, four independent row loads in flight per thread, fixed-order sum of the 32 row lanes through LDS.
// MASK: the ReLU backward of the layer is applied on the way (x = act > 0 ? x : 0, written back) — mlp_backward's
// relu_bwd + bias-gradient pair in one pass over dZ.
template <bool MASK>
__global__ __launch_bounds__(256) void colsum4_kernel(float* __restrict__ x, const float* __restrict__ act, float* __restrict__ out, int rows,
                                                      int cols, int64_t astride, int64_t pstride) {
    __shared__ float4 red[32][8];
    const int net = blockIdx.y;
    const int cg = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int c = blockIdx.x * 32 + 4 * cg;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < cols) {
        float* xb = x + net * astride + c;
        const float* ab = MASK ? act + net * astride + c : nullptr;
        int r = rl;
        for (; r + 96 < rows; r += 128) {
            float4 v[4], a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = *reinterpret_cast<const float4*>(xb + (int64_t)(r + 32 * u) * cols);
                if constexpr (MASK) a[u] = *reinterpret_cast<const float4*>(ab + (int64_t)(r + 32 * u) * cols);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if constexpr (MASK) {
                    v[u].x = a[u].x > 0.f ? v[u].x : 0.f; v[u].y = a[u].y > 0.f ? v[u].y : 0.f;
                    v[u].z = a[u].z > 0.f ? v[u].z : 0.f; v[u].w = a[u].w > 0.f ? v[u].w : 0.f;
                    *reinterpret_cast<float4*>(xb + (int64_t)(r + 32 * u) * cols) = v[u];
                }
                acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
            }
        }
        for (; r < rows; r += 32) {
            float4 v = *reinterpret_cast<const float4*>(xb + (int64_t)r * cols);
            if constexpr (MASK) {
                const float4 a = *reinterpret_cast<const float4*>(ab + (int64_t)r * cols);
                v.x = a.x > 0.f ? v.x : 0.f; v.y = a.y > 0.f ? v.y : 0.f; v.z = a.z > 0.f ? v.z : 0.f; v.w = a.w > 0.f ? v.w : 0.f;
                *reinterpret_cast<float4*>(xb + (int64_t)r * cols) = v;
            }
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    red[rl][cg] = acc;
    __syncthreads();
    if (rl == 0 && c < cols) {
        float4 t = red[0][cg];
#pragma unroll
        for (int i = 1; i < 32; ++i) { t.x += red[i][cg].x; t.y += red[i][cg].y; t.z += red[i][cg].z; t.w += red[i][cg].w; }
        *reinterpret_cast<float4*>(out + net * pstride + c) = t;
    }
}

static bool colsum_vec_ok(const float* x, const float* out, int cols, int64_t astride, int64_t pstride) {
    return cols % 4 == 0 && astride % 4 == 0 && pstride % 4 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0;
}

int colsum(const float* x, float* out, int rows, int cols, int nets, int64_t astride, int64_t pstride, hipStream_t s) {
    if (colsum_vec_ok(x, out, cols, astride, pstride)) {
        hipLaunchKernelGGL((colsum4_kernel<false>), dim3(cdiv(cols, 32), nets), dim3(256), 0, s, const_cast<float*>(x), (const float*)nullptr, out,
                           rows, cols, astride, pstride);
        EXORL_LAUNCH_CHECK();
        return 0;
    }
    dim3 grid(cdiv(cols, 64), nets);
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(1024), 0, s, x, out, rows, cols, astride, pstride);
    EXORL_LAUNCH_CHECK();
    return 0;
}

// x = act > 0 ? x : 0 in place, out[c] = sum_r x[r][c]; returns 1 (and does nothing) when the shapes need the two-kernel form
int relu_bwd_colsum(float* x, const float* act, float* out, int rows, int cols, hipStream_t s) {
    if (!colsum_vec_ok(x, out, cols, 0, 0) || reinterpret_cast<uintptr_t>(act) % 16 != 0) return 1;
    hipLaunchKernelGGL((colsum4_kernel<true>), dim3(cdiv(cols, 32), 1), dim3(256), 0, s, x, act, out, rows, cols, (int64_t)0, (int64_t)0);
    EXORL_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// out = x W^T + b for a layer whose fan-in dwarfs its width (the 39200-wide Linear layers on pixel features): a rows x F output is
// only rows/64 x F/64 tiles, so the reduction is cut into `splits` slabs that run as one grouped launch (splits x tiles workgroups)
// into `scratch` (splits x rows x F floats) and are summed in slab order.
__global__ __launch_bounds__(256) void splitk_sum_kernel(const float* __restrict__ P, const float* __restrict__ bias, float* __restrict__ z,
                                                         int64_t n, int F, int splits) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float acc = bias ? bias[i % F] : 0.f;
        for (int s = 0; s < splits; ++s) acc += P[(int64_t)s * n + i];
        z[i] = acc;
    }
}
int linear_splitk(int prec, const float* x, int64_t ldx, const float* W, const float* b, float* out, int rows, int F, int K, float* scratch,
                  int splits, hipStream_t s) {
    EXORL_REQUIRE(splits >= 1 && splits <= 32 && scratch, "linear_splitk: bad arguments");
    const int kc = (int)round_up(cdiv(K, splits), 4);
    GemmProblem p[32];
    int cnt = 0;
    for (int i = 0; i < splits; ++i) {
        const int k0 = i * kc;
        if (k0 >= K) break;
        p[cnt++] = GemmProblem{x + k0, W + k0, scratch + (int64_t)i * rows * F, nullptr, rows, F, K - k0 < kc ? K - k0 : kc, ldx, K, F};
    }
    EXORL_TRY(gemm_grouped(prec, 0, 0, p, cnt, false, false, s));
    const int64_t n = (int64_t)rows * F;
    const int64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(splitk_sum_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, s, scratch, b, out, n, F, cnt);
    EXORL_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Heads: out[m][j] = b[j] + sum_c a[m][c] W[j][c], j < nout <= 16 (optionally tanh) — one wave per row.
constexpr int MAX_NOUT = 16;

__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ a, const float* __restrict__ W,
                                                       const float* __restrict__ b, float* __restrict__ out, int rows,
                                                       int H, int nout, int tanh_out, int64_t astride, int64_t pstride,
                                                       int64_t ostride) {
    const int net = blockIdx.y;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* ar = a + net * astride + (int64_t)row * H;
    const float* Wn = W + net * pstride;
    float acc[MAX_NOUT];
#pragma unroll
    for (int j = 0; j < MAX_NOUT; ++j) acc[j] = 0.f;
    for (int c = lane; c < H; c += 64) {
        const float av = ar[c];
#pragma unroll
        for (int j = 0; j < MAX_NOUT; ++j)
            if (j < nout) acc[j] += av * Wn[(int64_t)j * H + c];
    }
#pragma unroll
    for (int j = 0; j < MAX_NOUT; ++j) {
        if (j < nout) {
            float v = wave_sum(acc[j]) + b[net * pstride + j];
            if (tanh_out) v = tanhf(v);
            if (lane == 0) out[net * ostride + (int64_t)row * nout + j] = v;
        }
    }
}

int head_fwd(const float* a, const float* W, const float* b, float* out, int rows, int H, int nout, int tanh_out,
             int nets, int64_t astride, int64_t pstride, int64_t ostride, hipStream_t s) {
    EXORL_REQUIRE(nout >= 1 && nout <= MAX_NOUT, "head_fwd: nout=%d unsupported (max %d)", nout, MAX_NOUT);
    dim3 grid(cdiv(rows, 4), nets);
    hipLaunchKernelGGL(head_fwd_kernel, grid, dim3(256), 0, s, a, W, b, out, rows, H, nout, tanh_out, astride, pstride, ostride);
    EXORL_LAUNCH_CHECK();
    return 0;
}

// dz[m][c] = (sum_j dout[m][j] W[j][c]) * (a[m][c] > 0)     (Linear(H,n) dgrad fused with the ReLU mask)
__global__ __launch_bounds__(256) void head_bwd_dx_kernel(const float* __restrict__ dout, const float* __restrict__ W,
                                                          const float* __restrict__ a, float* __restrict__ dz, int rows,
                                                          int H, int nout, int64_t astride, int64_t pstride,
                                                          int64_t dstride) {
    const int net = blockIdx.z;
    const int row = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= H) return;
    const float* d = dout + net * dstride + (int64_t)row * nout;
    const float* Wn = W + net * pstride;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < MAX_NOUT; ++j)
        if (j < nout) s += d[j] * Wn[(int64_t)j * H + c];
    const int64_t i = net * astride + (int64_t)row * H + c;
    dz[i] = a[i] > 0.f ? s : 0.f;
}

int head_bwd_dx(const float* dout, const float* W, const float* a, float* dz, int rows, int H, int nout,
                int nets, int64_t astride, int64_t pstride, int64_t dstride, hipStream_t s) {
    EXORL_REQUIRE(nout >= 1 && nout <= MAX_NOUT, "head_bwd_dx: nout=%d unsupported", nout);
    dim3 grid(cdiv(H, 256), rows, nets);
    hipLaunchKernelGGL(head_bwd_dx_kernel, grid, dim3(256), 0, s, dout, W, a, dz, rows, H, nout, astride, pstride, dstride);
    EXORL_LAUNCH_CHECK();
    return 0;
}

// dW[j][c] = sum_m dout[m][j] a[m][c];  db_hidden[c] = sum_m dz[m][c];  db_out[j] = sum_m dout[m][j]
__global__ __launch_bounds__(1024) void head_bwd_params_kernel(const float* __restrict__ dout, const float* __restrict__ a,
                                                               const float* __restrict__ dz, float* __restrict__ dW,
                                                               float* __restrict__ db_hidden, float* __restrict__ db_out,
                                                               int rows, int H, int nout, int64_t astride,
                                                               int64_t pstride, int64_t dstride) {
    __shared__ float red[16][64];
    const int net = blockIdx.y;
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const float* d = dout + net * dstride;
    float acc[MAX_NOUT];
#pragma unroll
    for (int j = 0; j < MAX_NOUT; ++j) acc[j] = 0.f;
    float sb = 0.f;
    if (c < H) {
        for (int r = ry; r < rows; r += 16) {
            const int64_t i = net * astride + (int64_t)r * H + c;
            const float av = a[i];
            sb += dz[i];
#pragma unroll
            for (int j = 0; j < MAX_NOUT; ++j)
                if (j < nout) acc[j] += d[(int64_t)r * nout + j] * av;
        }
    }
#pragma unroll
    for (int j = 0; j < MAX_NOUT; ++j) {
        if (j < nout) {
            const float t = block_colreduce(acc[j], red, cx, ry);
            if (ry == 0 && c < H) dW[net * pstride + (int64_t)j * H + c] = t;
        }
    }
    const float tb = block_colreduce(sb, red, cx, ry);
    if (ry == 0 && c < H) db_hidden[net * pstride + c] = tb;
    if (blockIdx.x == 0 && ry == 0) {     // wave 0: db_out
        for (int j = 0; j < nout; ++j) {
            float sj = 0.f;
            for (int r = cx; r < rows; r += 64) sj += d[(int64_t)r * nout + j];
            sj = wave_sum(sj);
            if (cx == 0) db_out[net * pstride + j] = sj;
        }
    }
}

int head_bwd_params(const float* dout, const float* a, const float* dz, float* dW, float* db_hidden, float* db_out,
                    int rows, int H, int nout, int nets, int64_t astride, int64_t pstride, int64_t dstride, hipStream_t s) {
    EXORL_REQUIRE(nout >= 1 && nout <= MAX_NOUT, "head_bwd_params: nout=%d unsupported", nout);
    dim3 grid(cdiv(H, 64), nets);
    hipLaunchKernelGGL(head_bwd_params_kernel, grid, dim3(1024), 0, s, dout, a, dz, dW, db_hidden, db_out, rows, H, nout,
                       astride, pstride, dstride);
    EXORL_LAUNCH_CHECK();
    return 0;
}

}  // namespace exorl

extern "C" int exorl_ln_tanh_fwd(const float* z, const float* gain, const float* beta, float* h, float* xhat,
                                 float* rstd, int32_t rows, int32_t H, void* stream) {
    return exorl::ln_tanh_fwd(z, gain, beta, h, xhat, rstd, rows, H, 1, 0, 0, exorl::as_stream(stream));
}
