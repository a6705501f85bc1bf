// kNN particle-entropy building block: for every source row the k smallest L2 distances to
// a set of target rows, sorted ascending. Replaces the broadcast (b1,b2,c) temporaries + topk of
//   /root/reference/utils/utils.py:289-300 (PBE, APT reward, 2 GiB temp at B=1024,c=512) and
//   /root/reference/agents/unsupervised_learning/proto.py:114-119 (Proto reward vs the 2048-row queue).
// Two launches:
//   pairdist2_kernel  squared distances in the difference form the reference uses, sum_c (s_c - t_c)^2 (not the
//     |s|^2 + |t|^2 - 2 s.t GEMM form, which loses the exact zeros on the diagonal that PBE relies on), as a register-blocked
//     tile kernel: a workgroup owns 64 source x 64 target rows, a thread a 4 x 4 block of pairs (rows ty + 16 i, tx + 16 j),
//     both operands staged through LDS in 32-column chunks and read back as ds_read_b128 along c (row stride 36 floats:
//     conflict-free for 16 consecutive rows). 2 LDS reads per 32 VALU operations; every operand byte is read from L2 once per
//     64-row block of the other operand instead of once per 4 rows.
//   knn_select_kernel one wave per source row: its row of squared distances -> LDS, k rounds of wave-wide arg-min extraction
//     (wavefront-level top-k, no sort of the row), sqrt on the way out.
#include "kernels.h"

namespace exorl {

constexpr int KNN_MAX_TGT = 4096;
constexpr int KNN_MAX_K = 64;
constexpr int PD_T = 64, PD_K = 32, PD_LD = PD_K + 4;

__global__ __launch_bounds__(256) void pairdist2_kernel(const float* __restrict__ src, int n_src, const float* __restrict__ tgt,
                                                        int n_tgt, int dim, float* __restrict__ d2, int ld, int vec) {
    __shared__ __attribute__((aligned(16))) float S[PD_T * PD_LD];
    __shared__ __attribute__((aligned(16))) float T[PD_T * PD_LD];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int s0 = blockIdx.y * PD_T, t0 = blockIdx.x * PD_T;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int c0 = 0; c0 < dim; c0 += PD_K) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int idx = tid + 256 * u, r = idx >> 3, c = c0 + 4 * (idx & 7);
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
            if (vec && c + 4 <= dim) {
                if (s0 + r < n_src) a = *reinterpret_cast<const float4*>(src + (int64_t)(s0 + r) * dim + c);
                if (t0 + r < n_tgt) b = *reinterpret_cast<const float4*>(tgt + (int64_t)(t0 + r) * dim + c);
            } else {
                float av[4] = {0.f, 0.f, 0.f, 0.f}, bv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (c + q < dim && s0 + r < n_src) av[q] = src[(int64_t)(s0 + r) * dim + c + q];
                    if (c + q < dim && t0 + r < n_tgt) bv[q] = tgt[(int64_t)(t0 + r) * dim + c + q];
                }
                a = make_float4(av[0], av[1], av[2], av[3]);
                b = make_float4(bv[0], bv[1], bv[2], bv[3]);
            }
            *reinterpret_cast<float4*>(S + r * PD_LD + 4 * (idx & 7)) = a;
            *reinterpret_cast<float4*>(T + r * PD_LD + 4 * (idx & 7)) = b;
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < PD_K; c += 4) {
            float4 sv[4], tv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sv[i] = *reinterpret_cast<const float4*>(S + (ty + 16 * i) * PD_LD + c);
                tv[i] = *reinterpret_cast<const float4*>(T + (tx + 16 * i) * PD_LD + c);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float d = sv[i].x - tv[j].x; acc[i][j] += d * d;
                    d = sv[i].y - tv[j].y; acc[i][j] += d * d;
                    d = sv[i].z - tv[j].z; acc[i][j] += d * d;
                    d = sv[i].w - tv[j].w; acc[i][j] += d * d;
                }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = s0 + ty + 16 * i;
        if (r >= n_src) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = t0 + tx + 16 * j;
            if (t < ld) d2[(int64_t)r * ld + t] = t < n_tgt ? acc[i][j] : INFINITY;
        }
    }
}

__global__ __launch_bounds__(256) void knn_select_kernel(const float* __restrict__ d2, int n_src, int n_pad, int k,
                                                         float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    if (row >= n_src) return;                      // wave-uniform; no workgroup barriers below
    float* mydist = smem + wave * n_pad;
    for (int t = lane; t < n_pad; t += 64) mydist[t] = d2[(int64_t)row * n_pad + t];
    __builtin_amdgcn_wave_barrier();
    for (int j = 0; j < k; ++j) {
        float best = INFINITY;
        int bi = 0x7fffffff;
        for (int t = lane; t < n_pad; t += 64) {
            const float d = mydist[t];
            if (d < best) { best = d; bi = t; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ob = __shfl_xor(best, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (lane == 0) {
            out[(int64_t)row * k + j] = sqrtf(best);
            if (bi < n_pad) mydist[bi] = INFINITY;
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
    }
}

// d2: caller's scratch of n_src x round_up(n_tgt, 64) floats (the engines carve it from their workspace)
int knn_topk(const float* src, int n_src, const float* tgt, int n_tgt, int dim, int k, float* out, float* d2, hipStream_t s) {
    const int n_pad = (n_tgt + 63) & ~63;
    float* g_d2 = d2;
    const int vec = (dim % 4 == 0 && reinterpret_cast<uintptr_t>(src) % 16 == 0 && reinterpret_cast<uintptr_t>(tgt) % 16 == 0) ? 1 : 0;
    hipLaunchKernelGGL(pairdist2_kernel, dim3(n_pad / PD_T, cdiv(n_src, PD_T)), dim3(256), 0, s, src, n_src, tgt, n_tgt, dim, g_d2, n_pad, vec);
    EXORL_LAUNCH_CHECK();
    hipLaunchKernelGGL(knn_select_kernel, dim3(cdiv(n_src, 4)), dim3(256), 4 * (size_t)n_pad * sizeof(float), s, g_d2, n_src, n_pad, k, out);
    EXORL_LAUNCH_CHECK();
    return 0;
}

}  // namespace exorl

extern "C" int exorl_knn_topk(const float* src, int32_t n_src, const float* tgt, int32_t n_tgt, int32_t dim, int32_t k,
                              float* out, void* stream) {
    using namespace exorl;
    EXORL_REQUIRE(src && tgt && out, "knn_topk: null argument");
    EXORL_REQUIRE(n_src > 0 && n_tgt > 0 && n_tgt <= KNN_MAX_TGT && dim > 0, "knn_topk: unsupported sizes n_src=%d n_tgt=%d (max %d) dim=%d",
                  n_src, n_tgt, KNN_MAX_TGT, dim);
    EXORL_REQUIRE(k >= 1 && k <= KNN_MAX_K && k <= n_tgt, "knn_topk: k=%d out of range (<= %d, <= n_tgt)", k, KNN_MAX_K);
    // stand-alone entry (tests, callers without a workspace): library-owned scratch, grown on demand outside graph capture
    static float* scratch = nullptr;
    static size_t scratch_floats = 0;
    const size_t need = (size_t)n_src * ((n_tgt + 63) & ~63);
    hipStream_t s = as_stream(stream);
    if (need > scratch_floats) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        EXORL_CHECK_HIP(hipStreamIsCapturing(s, &cs));
        EXORL_REQUIRE(cs == hipStreamCaptureStatusNone, "knn_topk: the stand-alone entry sizes its scratch on first use; call it once before capturing");
        EXORL_CHECK_HIP(hipDeviceSynchronize());
        if (scratch) EXORL_CHECK_HIP(hipFree(scratch));
        scratch = nullptr; scratch_floats = 0;
        EXORL_CHECK_HIP(hipMalloc(&scratch, need * sizeof(float)));
        scratch_floats = need;
    }
    return knn_topk(src, n_src, tgt, n_tgt, dim, k, out, scratch, s);
}