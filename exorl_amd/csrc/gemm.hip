// Grouped MFMA GEMM for the 1024-wide actor/critic layers: forward (x W^T),
// dgrad (dZ W) and wgrad (dZ^T H) of nn.Linear as used at
//   /root/reference/agents/offline_learning/td3_bc.py:16-20,37-47 and unsupervised_learning/ddpg.py:48-62,86-108.
//
// gfx950 design:
//  * 64x64 output tile per 256-thread workgroup (4 waves, one 32x32 MFMA accumulator each) so a
//    1024x1024 layer yields 256 workgroups per net — twin critics / stacked actor batches fill 256 CUs twice.
//  * fp32 parity mode: v_mfma_f32_32x32x2_f32 (exact fp32 products, k-ordered fmaf chain);
//    fast mode: v_mfma_f32_32x32x16_bf16 on operands rounded fp32->bf16 while staging (fp32 accumulate).
//  * LDS tile = 64 rows x 128 B, 16-byte units XOR-swizzled by (row>>1)&7 so the ds_read_b128 fragment
//    reads (16-lane groups on distinct rows) are bank-conflict free; one unit = 4 fp32 k's or 8 bf16 k's.
//    The k order inside a step is permuted identically for A and B (unit 2q+h feeds lane-half h).
//  * register-staged double buffering: global loads for tile t+1 are issued before the MFMAs of tile t
//    and written to the other LDS buffer after them — one barrier per k-tile.
//  * operands whose reduction index is the slow dimension (dgrad B, wgrad A and B) are transposed in
//    registers on the way to LDS (two rows x KU k's per thread), so all three GEMM forms share one inner loop.
#include <vector>

#include "kernels.h"

namespace exorl {

// Optional per-launch timing with HIP events on the launch stream (bench.py's roofline leg): off by default,
// the product path never pays for it.
struct GemmProfile {
    bool on = false;
    std::vector<hipEvent_t> ev;      // pairs
    std::vector<double> flops;
    size_t used = 0;
};
static GemmProfile g_prof;
constexpr size_t PROF_MAX_LAUNCHES = 1 << 15;

struct GemmBatch {
    GemmProblem p[4];
    int relu;
    int accumulate;
};

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int TILE = 64;        // BM = BN
constexpr int ROWB = 128;       // bytes per LDS tile row
constexpr int TILEB = TILE * ROWB;

__device__ __forceinline__ int lds_off(int row, int unit) { return row * ROWB + ((unit ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, v);
}

// L == 0: element (r,k) at ptr[r*ld + k]   (k contiguous)
// L == 1: element (r,k) at ptr[k*ld + r]   (r contiguous)
template <int L, bool VEC, int KU>
__device__ __forceinline__ void load_tile(const float* __restrict__ ptr, int64_t ld, int R, int K, int r0, int k0,
                                          int tid, float (&reg)[2][KU]) {
    if constexpr (L == 0) {
        const int unit = tid & 7;
        const int k = k0 + unit * KU;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int row = r0 + (tid >> 3) + 32 * u;
            const float* src = ptr + (int64_t)row * ld + k;
            if constexpr (VEC) {
#pragma unroll
                for (int c = 0; c < KU / 4; ++c) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (row < R && k + 4 * c + 4 <= K) v = *reinterpret_cast<const float4*>(src + 4 * c);
                    reg[u][4 * c + 0] = v.x; reg[u][4 * c + 1] = v.y; reg[u][4 * c + 2] = v.z; reg[u][4 * c + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int j = 0; j < KU; ++j) reg[u][j] = (row < R && k + j < K) ? src[j] : 0.f;
            }
        }
    } else {
        const int row = r0 + 2 * (tid & 31);
        const int k = k0 + (tid >> 5) * KU;
#pragma unroll
        for (int j = 0; j < KU; ++j) {
            const float* src = ptr + (int64_t)(k + j) * ld + row;
            if constexpr (VEC) {
                float2 v = make_float2(0.f, 0.f);
                if (k + j < K && row + 2 <= R) v = *reinterpret_cast<const float2*>(src);
                reg[0][j] = v.x; reg[1][j] = v.y;
            } else {
                reg[0][j] = (k + j < K && row < R) ? src[0] : 0.f;
                reg[1][j] = (k + j < K && row + 1 < R) ? src[1] : 0.f;
            }
        }
    }
}

template <int L, int PREC, int KU>
__device__ __forceinline__ void store_tile(unsigned char* lds, int tid, const float (&reg)[2][KU]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        int row, unit;
        if constexpr (L == 0) { row = (tid >> 3) + 32 * u; unit = tid & 7; }
        else                  { row = 2 * (tid & 31) + u;  unit = tid >> 5; }
        uint4 w;
        if constexpr (PREC == EXORL_PREC_F32) {
            w.x = __float_as_uint(reg[u][0]); w.y = __float_as_uint(reg[u][1]);
            w.z = __float_as_uint(reg[u][2]); w.w = __float_as_uint(reg[u][3]);
        } else {
            w.x = pack_bf16(reg[u][0], reg[u][1]); w.y = pack_bf16(reg[u][2], reg[u][3]);
            w.z = pack_bf16(reg[u][4], reg[u][5]); w.w = pack_bf16(reg[u][6], reg[u][7]);
        }
        *reinterpret_cast<uint4*>(lds + lds_off(row, unit)) = w;
    }
}

template <int PREC, int AL, int BL, bool VEC>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmBatch gb) {
    constexpr int KU = (PREC == EXORL_PREC_F32) ? 4 : 8;   // k elements per 16-byte unit
    constexpr int KPT = KU * 8;                            // k elements per LDS tile
    __shared__ __attribute__((aligned(16))) unsigned char smem[2][2][TILEB];

    const GemmProblem& P = gb.p[blockIdx.z];
    const int M = P.M, N = P.N, K = P.K;
    const int tiles_n = (N + TILE - 1) / TILE;
    const int tiles_m = (M + TILE - 1) / TILE;
    if ((int)blockIdx.x >= tiles_n * tiles_m) return;
    const int m0 = ((int)blockIdx.x / tiles_n) * TILE;
    const int n0 = ((int)blockIdx.x % tiles_n) * TILE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5;

    float ra[2][KU], rb[2][KU];
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    const int nk = (K + KPT - 1) / KPT;
    load_tile<AL, VEC, KU>(P.A, P.lda, M, K, m0, 0, tid, ra);
    load_tile<BL, VEC, KU>(P.B, P.ldb, N, K, n0, 0, tid, rb);
    store_tile<AL, PREC, KU>(smem[0][0], tid, ra);
    store_tile<BL, PREC, KU>(smem[0][1], tid, rb);
    __syncthreads();

    const int arow = wm * 32 + (lane & 31);
    const int brow = wn * 32 + (lane & 31);

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            load_tile<AL, VEC, KU>(P.A, P.lda, M, K, m0, (kt + 1) * KPT, tid, ra);
            load_tile<BL, VEC, KU>(P.B, P.ldb, N, K, n0, (kt + 1) * KPT, tid, rb);
        }
        const unsigned char* As = smem[cur][0];
        const unsigned char* Bs = smem[cur][1];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 a = *reinterpret_cast<const uint4*>(As + lds_off(arow, 2 * q + h));
            const uint4 b = *reinterpret_cast<const uint4*>(Bs + lds_off(brow, 2 * q + h));
            if constexpr (PREC == EXORL_PREC_F32) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                              acc, 0, 0, 0);
            }
        }
        if (kt + 1 < nk) {
            store_tile<AL, PREC, KU>(smem[cur ^ 1][0], tid, ra);
            store_tile<BL, PREC, KU>(smem[cur ^ 1][1], tid, rb);
        }
        __syncthreads();
    }

    // epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int n = n0 + wn * 32 + (lane & 31);
    if (n < N) {
        const float bias = P.bias ? P.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m < M) {
                float v = acc[r] + bias;
                if (gb.relu) v = fmaxf(v, 0.f);
                float* dst = P.C + (int64_t)m * P.ldc + n;
                if (gb.accumulate) v += *dst;
                *dst = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// bf16-operand variant (fast mode): A and B already live in memory as bf16 (written by the producing kernels:
// trunk_fwd -> h1, head_bwd -> dz2, Adam -> W1 shadow), so staging moves half the bytes and does no conversion.
// The per-CU L2->LDS path (~64 B/clk) bounds these 1024^3 layers, so the tile is BM x 64 with BM = 128 when that
// still yields >= 256 workgroups: (128+64) rows of 128 B per k-tile instead of 2 x (64+64).
// Layout-1 operands (reduction index slow) are transposed in registers: 8 k-rows x 2 columns per thread as
// 4-byte loads, v_perm_b32 splits the low/high bf16 into two 16-byte k-contiguous units.
template <int L, int NU>
__device__ __forceinline__ void load_tile16(const unsigned short* __restrict__ ptr, int64_t ld, int R, int K, int r0, int k0,
                                            int tid, uint4 (&reg)[NU]) {
    if constexpr (L == 0) {
        const int unit = tid & 7;
        const int k = k0 + unit * 8;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int row = r0 + (tid >> 3) + 32 * u;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (row < R && k + 8 <= K) v = *reinterpret_cast<const uint4*>(ptr + (int64_t)row * ld + k);
            reg[u] = v;
        }
    } else {
#pragma unroll
        for (int g = 0; g < NU / 2; ++g) {
            const int row = r0 + 2 * (tid & 31) + 64 * g;
            const int k = k0 + (tid >> 5) * 8;
            uint32_t d[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                d[j] = 0u;
                if (k + j < K && row + 2 <= R) d[j] = *reinterpret_cast<const uint32_t*>(ptr + (int64_t)(k + j) * ld + row);
            }
            // low halves -> row, high halves -> row+1
            reg[2 * g].x = __builtin_amdgcn_perm(d[1], d[0], 0x05040100u);
            reg[2 * g].y = __builtin_amdgcn_perm(d[3], d[2], 0x05040100u);
            reg[2 * g].z = __builtin_amdgcn_perm(d[5], d[4], 0x05040100u);
            reg[2 * g].w = __builtin_amdgcn_perm(d[7], d[6], 0x05040100u);
            reg[2 * g + 1].x = __builtin_amdgcn_perm(d[1], d[0], 0x07060302u);
            reg[2 * g + 1].y = __builtin_amdgcn_perm(d[3], d[2], 0x07060302u);
            reg[2 * g + 1].z = __builtin_amdgcn_perm(d[5], d[4], 0x07060302u);
            reg[2 * g + 1].w = __builtin_amdgcn_perm(d[7], d[6], 0x07060302u);
        }
    }
}

template <int L, int NU>
__device__ __forceinline__ void store_tile16(unsigned char* lds, int tid, const uint4 (&reg)[NU]) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        int row, unit;
        if constexpr (L == 0) { row = (tid >> 3) + 32 * u; unit = tid & 7; }
        else                  { row = 2 * (tid & 31) + (u & 1) + 64 * (u >> 1); unit = tid >> 5; }
        *reinterpret_cast<uint4*>(lds + lds_off(row, unit)) = reg[u];
    }
}

struct Gemm16Batch {
    Gemm16Problem p[4];
    int relu;
    int accumulate;
};

template <int AL, int BL, int BM>
__global__ __launch_bounds__(256) void gemm16_kernel(const Gemm16Batch gb) {
    constexpr int KPT = 64;                   // bf16 k per LDS tile (128 B rows)
    constexpr int NUA = BM / 32, NUB = 2;     // 16-byte units per thread
    constexpr int MT = BM / 64;               // 32x32 accumulator tiles per wave along M
    __shared__ __attribute__((aligned(16))) unsigned char smem[2][(BM + 64) * ROWB];

    const Gemm16Problem& P = gb.p[blockIdx.z];
    const int M = P.M, N = P.N, K = P.K;
    const int tiles_n = (N + 63) / 64;
    const int tiles_m = (M + BM - 1) / BM;
    if ((int)blockIdx.x >= tiles_n * tiles_m) return;
    const int m0 = ((int)blockIdx.x / tiles_n) * BM;
    const int n0 = ((int)blockIdx.x % tiles_n) * 64;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5;

    uint4 ra[NUA], rb[NUB];
    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    const int nk = (K + KPT - 1) / KPT;
    load_tile16<AL, NUA>(P.A, P.lda, M, K, m0, 0, tid, ra);
    load_tile16<BL, NUB>(P.B, P.ldb, N, K, n0, 0, tid, rb);
    store_tile16<AL, NUA>(smem[0], tid, ra);
    store_tile16<BL, NUB>(smem[0] + BM * ROWB, tid, rb);
    __syncthreads();

    const int arow = wm * (BM / 2) + (lane & 31);
    const int brow = wn * 32 + (lane & 31);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            load_tile16<AL, NUA>(P.A, P.lda, M, K, m0, (kt + 1) * KPT, tid, ra);
            load_tile16<BL, NUB>(P.B, P.ldb, N, K, n0, (kt + 1) * KPT, tid, rb);
        }
        const unsigned char* As = smem[cur];
        const unsigned char* Bs = smem[cur] + BM * ROWB;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 b = *reinterpret_cast<const uint4*>(Bs + lds_off(brow, 2 * q + h));
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const uint4 a = *reinterpret_cast<const uint4*>(As + lds_off(arow + 32 * t, 2 * q + h));
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                                 acc[t], 0, 0, 0);
            }
        }
        if (kt + 1 < nk) {
            store_tile16<AL, NUA>(smem[cur ^ 1], tid, ra);
            store_tile16<BL, NUB>(smem[cur ^ 1] + BM * ROWB, tid, rb);
        }
        __syncthreads();
    }

    const int n = n0 + wn * 32 + (lane & 31);
    if (n < N) {
        const float bias = P.bias ? P.bias[n] : 0.f;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * (BM / 2) + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < M) {
                    float v = acc[t][r] + bias;
                    if (gb.relu) v = fmaxf(v, 0.f);
                    float* dst = P.C + (int64_t)m * P.ldc + n;
                    if (gb.accumulate) v += *dst;
                    *dst = v;
                }
            }
        }
    }
}

template <int AL, int BL>
static int launch16(const Gemm16Batch& gb, int count, int tiles64, int tiles128, hipStream_t s) {
    const bool prof = g_prof.on && g_prof.used < PROF_MAX_LAUNCHES;
    if (prof) {
        if (g_prof.ev.size() < 2 * (g_prof.used + 1)) {
            hipEvent_t a, b;
            EXORL_CHECK_HIP(hipEventCreate(&a));
            EXORL_CHECK_HIP(hipEventCreate(&b));
            g_prof.ev.push_back(a);
            g_prof.ev.push_back(b);
        }
        double f = 0;
        for (int i = 0; i < count; ++i) f += 2.0 * gb.p[i].M * (double)gb.p[i].N * gb.p[i].K;
        g_prof.flops.push_back(f);
        EXORL_CHECK_HIP(hipEventRecord(g_prof.ev[2 * g_prof.used], s));
    }
    if (tiles128 * count >= 256) hipLaunchKernelGGL((gemm16_kernel<AL, BL, 128>), dim3(tiles128, 1, count), dim3(256), 0, s, gb);
    else                         hipLaunchKernelGGL((gemm16_kernel<AL, BL, 64>), dim3(tiles64, 1, count), dim3(256), 0, s, gb);
    EXORL_LAUNCH_CHECK();
    if (prof) {
        EXORL_CHECK_HIP(hipEventRecord(g_prof.ev[2 * g_prof.used + 1], s));
        g_prof.used += 1;
    }
    return 0;
}

// bf16-in-memory operands, fp32 output. Requirements (checked): 16-byte aligned rows for layout 0 (ld % 8 == 0,
// K % 8 == 0), 4-byte aligned pairs for layout 1 (ld % 2 == 0, R % 2 == 0).
int gemm16_grouped(int a_layout, int b_layout, const Gemm16Problem* probs, int count, bool relu, bool accumulate, hipStream_t s) {
    EXORL_REQUIRE(count >= 1 && count <= 4, "gemm16_grouped: count %d out of range", count);
    Gemm16Batch gb;
    memset(&gb, 0, sizeof(gb));
    int t64 = 0, t128 = 0;
    for (int i = 0; i < count; ++i) {
        const Gemm16Problem& p = probs[i];
        gb.p[i] = p;
        EXORL_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "gemm16_grouped: empty problem %d", i);
        auto ok = [](const unsigned short* ptr, int64_t ld, int R, int K, int layout) {
            const uintptr_t a = reinterpret_cast<uintptr_t>(ptr);
            if (layout == 0) return (a % 16 == 0) && (ld % 8 == 0) && (K % 8 == 0);
            return (a % 4 == 0) && (ld % 2 == 0) && (R % 2 == 0);
        };
        EXORL_REQUIRE(ok(p.A, p.lda, p.M, p.K, a_layout) && ok(p.B, p.ldb, p.N, p.K, b_layout),
                      "gemm16_grouped: problem %d (M=%d N=%d K=%d lda=%lld ldb=%lld) violates the bf16 path's alignment rules "
                      "(hidden_dim and batch must be multiples of 8 in bf16 precision)", i, p.M, p.N, p.K, (long long)p.lda, (long long)p.ldb);
        const int a64 = cdiv(p.M, 64) * cdiv(p.N, 64), a128 = cdiv(p.M, 128) * cdiv(p.N, 64);
        t64 = a64 > t64 ? a64 : t64;
        t128 = a128 > t128 ? a128 : t128;
    }
    gb.relu = relu ? 1 : 0;
    gb.accumulate = accumulate ? 1 : 0;
    if (a_layout == 0 && b_layout == 0) return launch16<0, 0>(gb, count, t64, t128, s);
    if (a_layout == 0 && b_layout == 1) return launch16<0, 1>(gb, count, t64, t128, s);
    if (a_layout == 1 && b_layout == 1) return launch16<1, 1>(gb, count, t64, t128, s);
    set_error("gemm16_grouped: unsupported layout combination %d %d", a_layout, b_layout);
    return 2;
}

template <int PREC, int AL, int BL>
static int launch_layout(const GemmBatch& gb, int count, int max_tiles, bool vec, hipStream_t s) {
    dim3 grid(max_tiles, 1, count), block(256);
    const bool prof = g_prof.on && g_prof.used < PROF_MAX_LAUNCHES;
    if (prof) {
        if (g_prof.ev.size() < 2 * (g_prof.used + 1)) {
            hipEvent_t a, b;
            EXORL_CHECK_HIP(hipEventCreate(&a));
            EXORL_CHECK_HIP(hipEventCreate(&b));
            g_prof.ev.push_back(a);
            g_prof.ev.push_back(b);
        }
        double f = 0;
        for (int i = 0; i < count; ++i) f += 2.0 * gb.p[i].M * (double)gb.p[i].N * gb.p[i].K;
        g_prof.flops.push_back(f);
        EXORL_CHECK_HIP(hipEventRecord(g_prof.ev[2 * g_prof.used], s));
    }
    if (vec) hipLaunchKernelGGL((gemm_kernel<PREC, AL, BL, true>), grid, block, 0, s, gb);
    else     hipLaunchKernelGGL((gemm_kernel<PREC, AL, BL, false>), grid, block, 0, s, gb);
    EXORL_LAUNCH_CHECK();
    if (prof) {
        EXORL_CHECK_HIP(hipEventRecord(g_prof.ev[2 * g_prof.used + 1], s));
        g_prof.used += 1;
    }
    return 0;
}

template <int PREC>
static int launch_prec(const GemmBatch& gb, int count, int al, int bl, int max_tiles, bool vec, hipStream_t s) {
    if (al == 0 && bl == 0) return launch_layout<PREC, 0, 0>(gb, count, max_tiles, vec, s);
    if (al == 0 && bl == 1) return launch_layout<PREC, 0, 1>(gb, count, max_tiles, vec, s);
    if (al == 1 && bl == 1) return launch_layout<PREC, 1, 1>(gb, count, max_tiles, vec, s);
    if (al == 1 && bl == 0) return launch_layout<PREC, 1, 0>(gb, count, max_tiles, vec, s);
    set_error("gemm: bad layout %d %d", al, bl);
    return 2;
}

static bool aligned_for_vec(const GemmProblem& p, int al, int bl) {
    auto ok = [](const float* ptr, int64_t ld, int R, int K, int layout) {
        const uintptr_t a = reinterpret_cast<uintptr_t>(ptr);
        if (layout == 0) return (a % 16 == 0) && (ld % 4 == 0) && (K % 4 == 0);
        return (a % 8 == 0) && (ld % 2 == 0) && (R % 2 == 0);
    };
    return ok(p.A, p.lda, p.M, p.K, al) && ok(p.B, p.ldb, p.N, p.K, bl);
}

// Launches up to 4 independent problems (same layouts / epilogue flags) as one grid.
int gemm_grouped(int precision, int a_layout, int b_layout, const GemmProblem* probs, int count, bool relu,
                 bool accumulate, hipStream_t s) {
    EXORL_REQUIRE(count >= 1 && count <= 4, "gemm_grouped: count %d out of range", count);
    GemmBatch gb;
    memset(&gb, 0, sizeof(gb));
    int max_tiles = 0;
    bool vec = true;
    for (int i = 0; i < count; ++i) {
        gb.p[i] = probs[i];
        EXORL_REQUIRE(probs[i].M > 0 && probs[i].N > 0 && probs[i].K > 0, "gemm_grouped: empty problem %d", i);
        const int t = cdiv(probs[i].M, TILE) * cdiv(probs[i].N, TILE);
        max_tiles = t > max_tiles ? t : max_tiles;
        vec = vec && aligned_for_vec(probs[i], a_layout, b_layout);
    }
    gb.relu = relu ? 1 : 0;
    gb.accumulate = accumulate ? 1 : 0;
    if (precision == EXORL_PREC_F32) return launch_prec<EXORL_PREC_F32>(gb, count, a_layout, b_layout, max_tiles, vec, s);
    if (precision == EXORL_PREC_BF16) return launch_prec<EXORL_PREC_BF16>(gb, count, a_layout, b_layout, max_tiles, vec, s);
    set_error("gemm_grouped: unknown precision %d", precision);
    return 2;
}

}  // namespace exorl

extern "C" int exorl_profile_gemm(int32_t enable) {
    exorl::g_prof.on = enable != 0;
    if (enable) { exorl::g_prof.used = 0; exorl::g_prof.flops.clear(); }
    return 0;
}

// Synchronises, then returns per-launch (algorithmic FLOPs, milliseconds) of the GEMM launches recorded
// since exorl_profile_gemm(1); n_out = number of launches written (<= cap).
extern "C" int exorl_profile_gemm_read(double* flops_out, float* ms_out, int32_t cap, int32_t* n_out) {
    using namespace exorl;
    EXORL_REQUIRE(flops_out && ms_out && n_out, "profile_gemm_read: null argument");
    EXORL_CHECK_HIP(hipDeviceSynchronize());
    int n = 0;
    for (size_t i = 0; i < g_prof.used && n < cap; ++i, ++n) {
        float ms = 0.f;
        EXORL_CHECK_HIP(hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]));
        flops_out[n] = g_prof.flops[i];
        ms_out[n] = ms;
    }
    *n_out = n;
    return 0;
}

extern "C" int exorl_gemm_bf16(int32_t a_layout, int32_t b_layout, int32_t M, int32_t N, int32_t K, const uint16_t* A,
                               int64_t lda, const uint16_t* B, int64_t ldb, float* C, int64_t ldc, const float* bias,
                               int32_t relu, int32_t accumulate, void* stream) {
    exorl::Gemm16Problem p{A, B, C, bias, M, N, K, lda, ldb, ldc};
    return exorl::gemm16_grouped(a_layout, b_layout, &p, 1, relu != 0, accumulate != 0, exorl::as_stream(stream));
}

extern "C" int exorl_gemm(int32_t precision, int32_t a_layout, int32_t b_layout, int32_t M, int32_t N, int32_t K,
                          const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                          const float* bias, int32_t relu, int32_t accumulate, void* stream) {
    exorl::GemmProblem p{A, B, C, bias, M, N, K, lda, ldb, ldc};
    return exorl::gemm_grouped(precision, a_layout, b_layout, &p, 1, relu != 0, accumulate != 0, exorl::as_stream(stream));
}
