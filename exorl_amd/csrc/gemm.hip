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
#include <algorithm>
#include <type_traits>
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

constexpr int GEMM_MAX_GROUP = 32;     // problems per launch of the generic kernel (split-K slabs of one layer share a launch)
struct GemmBatch {
    GemmProblem p[GEMM_MAX_GROUP];
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

// x = hi + lo with hi = bf16(x), lo = bf16(x - hi): 16 significand bits in two bf16 MFMA operands
__device__ __forceinline__ float bf16_residual(float x) { return x - (float)(__bf16)x; }

template <int L, int PREC, int KU>
__device__ __forceinline__ void store_tile(unsigned char* lds, int tid, const float (&reg)[2][KU]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        int row, unit;
        if constexpr (L == 0) { row = (tid >> 3) + 32 * u; unit = tid & 7; }
        else                  { row = 2 * (tid & 31) + u;  unit = tid >> 5; }
        uint4 w;
        if constexpr (PREC == EXORL_PREC_BF16X6) {      // three planes: hi, mid = bf16(x - hi), lo = bf16(x - hi - mid): images at +0, +2, +4 tiles
            float md[8], lw[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float r1 = bf16_residual(reg[u][j]);
                md[j] = r1;
                lw[j] = bf16_residual(r1);
            }
            uint4 m2, l2;
            m2.x = pack_bf16(md[0], md[1]); m2.y = pack_bf16(md[2], md[3]); m2.z = pack_bf16(md[4], md[5]); m2.w = pack_bf16(md[6], md[7]);
            l2.x = pack_bf16(lw[0], lw[1]); l2.y = pack_bf16(lw[2], lw[3]); l2.z = pack_bf16(lw[4], lw[5]); l2.w = pack_bf16(lw[6], lw[7]);
            *reinterpret_cast<uint4*>(lds + 2 * TILEB + lds_off(row, unit)) = m2;
            *reinterpret_cast<uint4*>(lds + 4 * TILEB + lds_off(row, unit)) = l2;
        }
        if constexpr (PREC == EXORL_PREC_BF16X3) {      // lo image two tiles after the hi image (A_hi B_hi A_lo B_lo)
            uint4 l;
            l.x = pack_bf16(bf16_residual(reg[u][0]), bf16_residual(reg[u][1])); l.y = pack_bf16(bf16_residual(reg[u][2]), bf16_residual(reg[u][3]));
            l.z = pack_bf16(bf16_residual(reg[u][4]), bf16_residual(reg[u][5])); l.w = pack_bf16(bf16_residual(reg[u][6]), bf16_residual(reg[u][7]));
            *reinterpret_cast<uint4*>(lds + 2 * TILEB + lds_off(row, unit)) = l;
        }
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

// measured (round 3): one LDS stage pays for three planes (Proto pixels 9.8 -> 9.5 ms, ICM pixels 27.2 -> 22.8 ms per update in bf16x6) and is
// neutral to slightly worse for two (rnd 1528 -> 1514, icm_apt 1186 -> 1172 update()/s): two planes keep the double buffer
#ifndef EXORL_GEMM_X3_SINGLE_STAGE
#define EXORL_GEMM_X3_SINGLE_STAGE 0
#endif
template <int PREC, int AL, int BL, bool VEC>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmBatch gb) {
    constexpr int KU = (PREC == EXORL_PREC_F32) ? 4 : 8;   // k elements per 16-byte unit
    constexpr int KPT = KU * 8;                            // k elements per LDS tile
    constexpr bool X3 = PREC == EXORL_PREC_BF16X3, X6 = PREC == EXORL_PREC_BF16X6;
    constexpr int NT = X6 ? 6 : (X3 ? 4 : 2);              // LDS tiles per stage: [A p0][B p0][A p1][B p1][A p2][B p2]
    // two stages; the three-plane mode needs 96 KB, past the 64 KB a static array may have: dynamic there
    extern __shared__ __attribute__((aligned(16))) unsigned char gk_dyn[];
    __shared__ __attribute__((aligned(16))) unsigned char gk_static[X6 ? 16 : ((X3 && EXORL_GEMM_X3_SINGLE_STAGE) ? 1 : 2) * NT * TILEB];
    unsigned char* const smem_base = X6 ? gk_dyn : gk_static;
    auto smem = [&](int stage, int tile) { return smem_base + (stage * NT + tile) * TILEB; };

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
    f32x16 acc, acc2, acc3;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; acc2[i] = 0.f; acc3[i] = 0.f; }

    const int nk = (K + KPT - 1) / KPT;
    load_tile<AL, VEC, KU>(P.A, P.lda, M, K, m0, 0, tid, ra);
    load_tile<BL, VEC, KU>(P.B, P.ldb, N, K, n0, 0, tid, rb);
    store_tile<AL, PREC, KU>(smem(0, 0), tid, ra);
    store_tile<BL, PREC, KU>(smem(0, 1), tid, rb);
    __syncthreads();

    const int arow = wm * 32 + (lane & 31);
    const int brow = wn * 32 + (lane & 31);

    // three planes: ONE LDS stage (48 KB, three workgroups per CU) instead of two (96 KB, one workgroup of four waves per CU — one wave per SIMD
    // with nothing to hide a barrier or an LDS round trip behind); the next tile still travels in registers while this one is multiplied
    constexpr bool SS = X6 || (X3 && EXORL_GEMM_X3_SINGLE_STAGE);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = SS ? 0 : (kt & 1);
        if (kt + 1 < nk) {
            load_tile<AL, VEC, KU>(P.A, P.lda, M, K, m0, (kt + 1) * KPT, tid, ra);
            load_tile<BL, VEC, KU>(P.B, P.ldb, N, K, n0, (kt + 1) * KPT, tid, rb);
        }
        const unsigned char* As = smem(cur, 0);
        const unsigned char* Bs = smem(cur, 1);
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
                if constexpr (X3) {
                    const uint4 al = *reinterpret_cast<const uint4*>(As + 2 * TILEB + lds_off(arow, 2 * q + h));
                    const uint4 bl = *reinterpret_cast<const uint4*>(Bs + 2 * TILEB + lds_off(brow, 2 * q + h));
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bl), acc2, 0, 0, 0);
                    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al), __builtin_bit_cast(bf16x8, b), acc3, 0, 0, 0);
                }
                if constexpr (X6) {       // acc: hi*hi; acc2: hi*mid + mid*hi (~2^-8 of it); acc3: hi*lo + lo*hi + mid*mid (~2^-16)
                    const uint4 am = *reinterpret_cast<const uint4*>(As + 2 * TILEB + lds_off(arow, 2 * q + h));
                    const uint4 bm = *reinterpret_cast<const uint4*>(Bs + 2 * TILEB + lds_off(brow, 2 * q + h));
                    const uint4 al = *reinterpret_cast<const uint4*>(As + 4 * TILEB + lds_off(arow, 2 * q + h));
                    const uint4 bl = *reinterpret_cast<const uint4*>(Bs + 4 * TILEB + lds_off(brow, 2 * q + h));
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bm), acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, am), __builtin_bit_cast(bf16x8, b), acc2, 0, 0, 0);
                    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bl), acc3, 0, 0, 0);
                    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al), __builtin_bit_cast(bf16x8, b), acc3, 0, 0, 0);
                    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, am), __builtin_bit_cast(bf16x8, bm), acc3, 0, 0, 0);
                }
            }
        }
        if constexpr (SS) __syncthreads();             // every wave is done reading the stage before it is overwritten
        if (kt + 1 < nk) {
            store_tile<AL, PREC, KU>(smem(SS ? 0 : cur ^ 1, 0), tid, ra);
            store_tile<BL, PREC, KU>(smem(SS ? 0 : cur ^ 1, 1), tid, rb);
        }
        __syncthreads();
    }

    // epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    if constexpr (X3) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = (acc2[i] + acc3[i]) + acc[i];      // cross terms first
    }
    if constexpr (X6) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = (acc3[i] + acc2[i]) + acc[i];      // smallest class first
    }
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
template <int L, int NU, bool GUARD>
__device__ __forceinline__ void load_tile16(const unsigned short* __restrict__ ptr, int64_t ld, int R, int K, int r0, int k0,
                                            int tid, uint4 (&reg)[NU]) {
    if constexpr (L == 0) {
        const int unit = tid & 7;
        const int k = k0 + unit * 8;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int row = r0 + (tid >> 3) + 32 * u;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (!GUARD || (row < R && k + 8 <= K)) v = *reinterpret_cast<const uint4*>(ptr + (int64_t)row * ld + k);
            reg[u] = v;
        }
    } else {       // raw: reg[2g] = k rows 0..3, reg[2g+1] = k rows 4..7, each dword = {col, col+1}; permuted at store time
#pragma unroll
        for (int g = 0; g < NU / 2; ++g) {
            const int row = r0 + 2 * (tid & 31) + 64 * g;
            const int k = k0 + (tid >> 5) * 8;
            uint32_t d[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                d[j] = 0u;
                if (!GUARD || (k + j < K && row + 2 <= R)) d[j] = *reinterpret_cast<const uint32_t*>(ptr + (int64_t)(k + j) * ld + row);
            }
            reg[2 * g] = make_uint4(d[0], d[1], d[2], d[3]);
            reg[2 * g + 1] = make_uint4(d[4], d[5], d[6], d[7]);
        }
    }
}

template <int L, int NU>
__device__ __forceinline__ void store_tile16(unsigned char* lds, int tid, const uint4 (&reg)[NU]) {
    if constexpr (L == 0) {
#pragma unroll
        for (int u = 0; u < NU; ++u)
            *reinterpret_cast<uint4*>(lds + lds_off((tid >> 3) + 32 * u, tid & 7)) = reg[u];
    } else {
#pragma unroll
        for (int g = 0; g < NU / 2; ++g) {
            const uint4 lo = reg[2 * g], hi = reg[2 * g + 1];
            uint4 e, o;      // low halves -> column `row`, high halves -> column `row+1`
            e.x = __builtin_amdgcn_perm(lo.y, lo.x, 0x05040100u); e.y = __builtin_amdgcn_perm(lo.w, lo.z, 0x05040100u);
            e.z = __builtin_amdgcn_perm(hi.y, hi.x, 0x05040100u); e.w = __builtin_amdgcn_perm(hi.w, hi.z, 0x05040100u);
            o.x = __builtin_amdgcn_perm(lo.y, lo.x, 0x07060302u); o.y = __builtin_amdgcn_perm(lo.w, lo.z, 0x07060302u);
            o.z = __builtin_amdgcn_perm(hi.y, hi.x, 0x07060302u); o.w = __builtin_amdgcn_perm(hi.w, hi.z, 0x07060302u);
            const int row = 2 * (tid & 31) + 64 * g, unit = tid >> 5;
            *reinterpret_cast<uint4*>(lds + lds_off(row, unit)) = e;
            *reinterpret_cast<uint4*>(lds + lds_off(row + 1, unit)) = o;
        }
    }
}

struct Gemm16Batch {
    Gemm16Problem p[4];
    int relu;
    int accumulate;
    int swizzle;      // 1: XCD-aware tile order (blocks that share an XCD take neighbouring tiles)
    int a_t[4];       // mixed-layout launch (gemm16g_mixed_kernel): problem i reads A as a k image (layout 1)
    int xcd_map;      // 1: 1-D grid, workgroup id -> (problem, tile) so that each XCD owns a compact block of one problem's output
    int count;
};

// Workgroups are dealt to the 8 XCDs round-robin by linear id and every XCD has its own L2, so what an XCD's tiles touch is
// fetched once per XCD over the fabric. With tile order following the id, an XCD ends up needing every problem's whole B
// operand (PMC: 39 MB of fabric reads for a launch whose operands total 12 MB). This mapping gives XCD x = id & 7 the problem
// x / (8/count) and, inside it, one block of an (sm x sn) split of the output, walked row-major by id >> 3: per XCD only a
// 1/sm slice of A and a 1/sn slice of B of ONE problem.
struct XcdTile { int p, tm, tn; bool ok; };
__device__ __forceinline__ XcdTile xcd_tile(int id, int count, int tiles_m, int tiles_n) {
    const int X = 8 / count;                       // XCDs per problem (count in {1, 2, 4})
    const int sm = X == 8 ? 4 : 2, sn = X == 2 ? 1 : 2;
    const int xcd = id & 7, slot = id >> 3;
    const int xq = xcd % X;
    const int bm = tiles_m / sm, bn = tiles_n / sn;
    XcdTile t;
    t.p = xcd / X;
    t.tm = (xq / sn) * bm + slot / bn;
    t.tn = (xq % sn) * bn + slot % bn;
    t.ok = slot < bm * bn;
    return t;
}
static bool xcd_map_ok(const Gemm16Batch& gb, int count, int tile) {
    if (!(count == 1 || count == 2 || count == 4)) return false;
    const int X = 8 / count, sm = X == 8 ? 4 : 2, sn = X == 2 ? 1 : 2;
    for (int i = 0; i < count; ++i) {
        if (gb.p[i].M != gb.p[0].M || gb.p[i].N != gb.p[0].N) return false;
        if ((gb.p[i].M / tile) % sm != 0 || (gb.p[i].N / tile) % sn != 0) return false;
    }
    return true;
}
static int g_gemm16_variant = -1;    // experiment switch (exorl_gemm_tune): -1 = default heuristics

template <int AL, int BL, int BM, int NS, bool GUARD>
__global__ __launch_bounds__(256) void gemm16_kernel(const Gemm16Batch gb) {
    constexpr int KPT = 64;                   // bf16 k per LDS tile (128 B rows)
    constexpr int NUA = BM / 32, NUB = 2;     // 16-byte units per thread
    constexpr int MT = BM / 64;               // 32x32 accumulator tiles per wave along M
    __shared__ __attribute__((aligned(16))) unsigned char smem[2][(BM + 64) * ROWB];

    const Gemm16Problem& P = gb.p[blockIdx.z];
    const int M = P.M, N = P.N, K = P.K;
    const int tiles_n = (N + 63) / 64;
    const int tiles_m = (M + BM - 1) / BM;
    const int ntiles = tiles_n * tiles_m;
    if ((int)blockIdx.x >= ntiles) return;
    int tile = blockIdx.x;
    if (gb.swizzle && (ntiles & 7) == 0) {
        // blocks b, b+8, b+16.. share an XCD (round-robin dispatch): give each XCD a contiguous run of tiles so its
        // private L2 holds one A row-panel set and the B panels instead of the whole of A
        const int per = ntiles >> 3;
        tile = (tile & 7) * per + (tile >> 3);
    }
    const int m0 = (tile / tiles_n) * BM;
    const int n0 = (tile % tiles_n) * 64;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5;

    uint4 ra[NS][NUA], rb[NS][NUB];      // NS register stages: loads run NS-1 k-tiles ahead of their LDS store
    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    const int nk = (K + KPT - 1) / KPT;
    const int arow = wm * (BM / 2) + (lane & 31);
    const int brow = wn * 32 + (lane & 31);
    auto compute = [&](int cur) {
        const unsigned char* As = smem[cur];
        const unsigned char* Bs = smem[cur] + BM * ROWB;
        uint4 af[MT][4], bfr[4];          // all fragment reads of the k-tile first (one exposed LDS latency), then MFMAs
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bfr[q] = *reinterpret_cast<const uint4*>(Bs + lds_off(brow, 2 * q + h));
#pragma unroll
            for (int t = 0; t < MT; ++t) af[t][q] = *reinterpret_cast<const uint4*>(As + lds_off(arow + 32 * t, 2 * q + h));
        }
        __builtin_amdgcn_sched_barrier(0);      // keep the reads batched: hipcc otherwise sinks each pair to its MFMA
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int t = 0; t < MT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[t][q]),
                                                                 __builtin_bit_cast(bf16x8, bfr[q]), acc[t], 0, 0, 0);
    };
    auto issue = [&](auto sc, int kt) {          // loads of k-tile kt into register stage sc (clamped: never past K)
        constexpr int st = decltype(sc)::value;
        const int k0 = (kt < nk ? kt : nk - 1) * KPT;
        load_tile16<AL, NUA, GUARD>(P.A, P.lda, M, K, m0, k0, tid, ra[st]);
        load_tile16<BL, NUB, GUARD>(P.B, P.ldb, N, K, n0, k0, tid, rb[st]);
    };
    auto commit = [&](auto sc, int buf) {        // register stage sc -> LDS buffer buf
        constexpr int st = decltype(sc)::value;
        store_tile16<AL, NUA>(smem[buf], tid, ra[st]);
        store_tile16<BL, NUB>(smem[buf] + BM * ROWB, tid, rb[st]);
    };
    auto step = [&](auto sc, int kt) {           // k-tile kt lives in LDS[kt&1]; tile kt+1 is in stage (sc+1)%NS
        constexpr int st = decltype(sc)::value;
        issue(sc, kt + NS);                      // stage sc was committed last step: refill it NS tiles ahead
        compute(kt & 1);
        if (kt + 1 < nk) commit(std::integral_constant<int, (st + 1) % NS>{}, (kt & 1) ^ 1);
        __syncthreads();
    };
    // prologue: tiles 0..NS-1 in flight, tile 0 committed
    issue(std::integral_constant<int, 0>{}, 0);
    if constexpr (NS > 1) issue(std::integral_constant<int, 1>{}, 1);
    if constexpr (NS > 2) issue(std::integral_constant<int, 2>{}, 2);
    if constexpr (NS > 3) issue(std::integral_constant<int, 3>{}, 3);
    commit(std::integral_constant<int, 0>{}, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += NS) {
        step(std::integral_constant<int, 0>{}, kt);
        if constexpr (NS > 1) { if (kt + 1 >= nk) break; step(std::integral_constant<int, 1>{}, kt + 1); }
        if constexpr (NS > 2) { if (kt + 2 >= nk) break; step(std::integral_constant<int, 2>{}, kt + 2); }
        if constexpr (NS > 3) { if (kt + 3 >= nk) break; step(std::integral_constant<int, 3>{}, kt + 3); }
    }

    const int n = n0 + wn * 32 + (lane & 31);
    if (n < N) {
        const float bias = P.bias ? P.bias[n] : 0.f;
        const bool relu = gb.relu != 0;
        if (gb.accumulate) {
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * (BM / 2) + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (m < M) {
                        float* dst = P.C + (int64_t)m * P.ldc + n;
                        float v = acc[t][r] + bias;
                        if (relu) v = fmaxf(v, 0.f);
                        *dst = v + *dst;
                    }
                }
        } else {
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * (BM / 2) + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (m < M) {
                        float v = acc[t][r] + bias;
                        if (relu) v = fmaxf(v, 0.f);
                        P.C[(int64_t)m * P.ldc + n] = v;
                    }
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// LDS-DMA variant for exactly-tiled problems (M, N, K multiples of 64): the production path of the 1024-wide layers.
// Operand tiles go global -> LDS with `global_load_lds_dwordx4` (no VGPR round trip, no ds_write), four stages deep,
// paced by counted s_waitcnt vmcnt + one raw s_barrier per k-tile (the loads of tiles t+1, t+2 stay in flight
// across the barrier). LDS images are lane-linear per wave-instruction; the bank swizzles are applied to the per-lane
// SOURCE address and again on the fragment read:
//   k-contiguous operand ("row image", [row][64 k]):   unit' = unit ^ ((row>>1)&7), fragments by ds_read_b128
//   reduction-slow operand ("k image", [k][64 rows], a straight copy of memory): unit' = unit ^ 4*((k>>1)&1),
//     fragments by ds_read_b64_tr_b16 (hardware transpose: 4 k-rows x 16 columns per 16-lane group), so dgrad and
//     wgrad need no transposed copies of W1 / dZ / H in memory and no register shuffles.
typedef short v4s16 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int G16G_NSTG = 4;
constexpr int G16G_IMG = 64 * ROWB;                // 8 KB per operand image

template <bool AT, bool BT, int NSTG = G16G_NSTG, bool X3 = false>
__device__ __forceinline__ void gemm16g_body(const Gemm16Batch& gb, unsigned char* smem) {
    constexpr int IMG = G16G_IMG;

    int pidx = blockIdx.z, m0, n0;
    if (gb.xcd_map) {
        const XcdTile xt = xcd_tile(blockIdx.x, gb.count, gb.p[0].M >> 6, gb.p[0].N >> 6);
        if (!xt.ok) return;
        pidx = xt.p; m0 = xt.tm << 6; n0 = xt.tn << 6;
    } else {
        const int tiles_n = gb.p[pidx].N >> 6, ntiles = tiles_n * (gb.p[pidx].M >> 6);
        if ((int)blockIdx.x >= ntiles) return;
        int tile = blockIdx.x;
        if (gb.swizzle && (ntiles & 7) == 0) tile = (tile & 7) * (ntiles >> 3) + (tile >> 3);
        m0 = (tile / tiles_n) << 6;
        n0 = (tile % tiles_n) << 6;
    }
    const Gemm16Problem& P = gb.p[pidx];
    const int K = P.K;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5;
    const int nk = K >> 6;                         // multiple of NSTG (checked by the launcher)

    constexpr int NIMG = X3 ? 4 : 2;               // images per stage: A_hi B_hi [A_lo B_lo]
    f32x16 acc, acc2, acc3;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; acc2[i] = 0.f; acc3[i] = 0.f; }

    // ---- LDS-DMA source pointers: this wave's two 1-KB pieces per operand image (image rows 16*wave+8j + lane/8);
    // everything per-lane is computed once, the k-loop only adds a constant stride
    const unsigned short* src[8];
    int64_t kstep[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rr = 16 * wave + 8 * j + (lane >> 3), p = lane & 7;
        const int64_t oa = !AT ? (int64_t)(m0 + rr) * P.lda + 8 * (p ^ ((rr >> 1) & 7)) : (int64_t)rr * P.lda + m0 + 8 * (p ^ (4 * ((rr >> 1) & 1)));
        const int64_t ob = !BT ? (int64_t)(n0 + rr) * P.ldb + 8 * (p ^ ((rr >> 1) & 7)) : (int64_t)rr * P.ldb + n0 + 8 * (p ^ (4 * ((rr >> 1) & 1)));
        src[j] = P.A + oa;
        src[2 + j] = P.B + ob;
        if constexpr (X3) { src[4 + j] = P.A_lo + oa; src[6 + j] = P.B_lo + ob; }
    }
    kstep[0] = AT ? 64 * P.lda : 64;
    kstep[1] = BT ? 64 * P.ldb : 64;
    const int piece = 16 * wave * ROWB;            // wave-uniform LDS offset of this wave's pieces inside an image

    // ---- fragment read offsets (per lane, stage-relative; stage and q enter as immediates)
    const int arow = wm * 32 + (lane & 31), brow = wn * 32 + (lane & 31);
    const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
    auto tr_off = [&](int rbase) {      // k image: k-row 8*(g>>1)+qq, columns rbase + 16*(g&1) + 4*pp ..+3
        const int c = rbase + 16 * (g & 1) + 4 * pp;
        return (8 * (g >> 1) + qq) * ROWB + (((c >> 3) ^ (4 * ((qq >> 1) & 1))) << 4) + ((c & 7) << 1);
    };
    int aoff[4], boff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        aoff[q] = AT ? tr_off(wm * 32) + q * 16 * ROWB : lds_off(arow, 2 * q + h);
        boff[q] = BT ? tr_off(wn * 32) + q * 16 * ROWB : lds_off(brow, 2 * q + h);
    }

    auto fill = [&](auto sc) {                     // issue the 4 DMA pieces of the next k-tile into stage sc
        constexpr int st = decltype(sc)::value;
        unsigned char* base = smem + st * NIMG * IMG + piece;
        __builtin_amdgcn_global_load_lds((const void*)src[0], (lds_void*)(base), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const void*)src[1], (lds_void*)(base + 8 * ROWB), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const void*)src[2], (lds_void*)(base + IMG), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const void*)src[3], (lds_void*)(base + IMG + 8 * ROWB), 16, 0, 0);
        src[0] += kstep[0]; src[1] += kstep[0]; src[2] += kstep[1]; src[3] += kstep[1];
        if constexpr (X3) {
            __builtin_amdgcn_global_load_lds((const void*)src[4], (lds_void*)(base + 2 * IMG), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const void*)src[5], (lds_void*)(base + 2 * IMG + 8 * ROWB), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const void*)src[6], (lds_void*)(base + 3 * IMG), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const void*)src[7], (lds_void*)(base + 3 * IMG + 8 * ROWB), 16, 0, 0);
            src[4] += kstep[0]; src[5] += kstep[0]; src[6] += kstep[1]; src[7] += kstep[1];
        }
    };
    auto frag = [&](const unsigned char* img, bool tr, int off) -> bf16x8 {
        if (!tr) return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(img + off));
        const v4s16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16*)(img + off));
        const v4s16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16*)(img + off + 4 * ROWB));
        typedef short v8s16 __attribute__((ext_vector_type(8)));
        const v8s16 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        return __builtin_bit_cast(bf16x8, v);
    };
    auto step = [&](auto sc, int t) {              // k-tile t sits in stage sc
        constexpr int st = decltype(sc)::value;
        // tile t has landed once at most the fills of the two younger tiles remain outstanding (4 DMA pieces per tile per wave)
        const int younger = nk - 1 - t;            // NSTG - 2 younger tiles may still be in flight
        const int allowed = younger < NSTG - 2 ? younger : NSTG - 2;
#define EXORL_WAITC(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
        if constexpr (X3) {
            switch (allowed) {
                case 0: EXORL_WAITC(0); break;
                case 1: EXORL_WAITC(8); break;
                case 2: EXORL_WAITC(16); break;
                case 3: EXORL_WAITC(24); break;
                case 4: EXORL_WAITC(32); break;
                case 5: EXORL_WAITC(40); break;
                default: EXORL_WAITC(48); break;
            }
        } else {
            switch (allowed) {
                case 0: EXORL_WAITC(0); break;
                case 1: EXORL_WAITC(4); break;
                case 2: EXORL_WAITC(8); break;
                case 3: EXORL_WAITC(12); break;
                case 4: EXORL_WAITC(16); break;
                case 5: EXORL_WAITC(20); break;
                case 6: EXORL_WAITC(24); break;
                case 7: EXORL_WAITC(28); break;
                default: EXORL_WAITC(32); break;
            }
        }
#undef EXORL_WAITC
        __builtin_amdgcn_s_barrier();              // every wave's pieces of tile t are in LDS; stage st-1 is no longer being read
        asm volatile("" ::: "memory");
        if (t + NSTG - 1 < nk) fill(std::integral_constant<int, (st + NSTG - 1) % NSTG>{});
        const unsigned char* As = smem + st * NIMG * IMG;
        const unsigned char* Bs = As + IMG;
        bf16x8 af[4], bfr[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            af[q] = frag(As, AT, aoff[q]);
            bfr[q] = frag(Bs, BT, boff[q]);
        }
        if constexpr (X3) {                        // hi*hi + hi*lo + lo*hi, three independent accumulators
            bf16x8 al[4], bl[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                al[q] = frag(As + 2 * IMG, AT, aoff[q]);
                bl[q] = frag(Bs + 2 * IMG, BT, boff[q]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[q], bfr[q], acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[q], bl[q], acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[q], bfr[q], acc3, 0, 0, 0);
            }
            return;
        }
        __builtin_amdgcn_sched_barrier(0);
        // two independent accumulation chains: PMC shows ~36 % of wave cycles as MFMA issue stalls (SQ_WAIT_INST_ANY)
        // when all four MFMAs of a k-tile chain through one accumulator
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bfr[0], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bfr[1], acc2, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bfr[2], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[3], bfr[3], acc2, 0, 0, 0);
    };

    static_assert(NSTG >= 2 && NSTG <= 10 && (X3 ? NSTG <= 8 : true), "stage count out of range of the vmcnt table");
#define EXORL_FILL0(I) if constexpr (NSTG > I + 1) { if (I < nk) fill(std::integral_constant<int, I>{}); }
    EXORL_FILL0(0) EXORL_FILL0(1) EXORL_FILL0(2) EXORL_FILL0(3) EXORL_FILL0(4) EXORL_FILL0(5) EXORL_FILL0(6) EXORL_FILL0(7) EXORL_FILL0(8)
#undef EXORL_FILL0
    int t = 0;
#define EXORL_STEP(I) if constexpr (NSTG > I) step(std::integral_constant<int, I>{}, t + I);
    for (; t + NSTG <= nk; t += NSTG) {
        EXORL_STEP(0) EXORL_STEP(1) EXORL_STEP(2) EXORL_STEP(3) EXORL_STEP(4) EXORL_STEP(5) EXORL_STEP(6) EXORL_STEP(7) EXORL_STEP(8) EXORL_STEP(9)
    }
#undef EXORL_STEP
#define EXORL_TAIL(I) if constexpr (NSTG > I + 1) { if (t + I < nk) step(std::integral_constant<int, I>{}, t + I); }     // nk % NSTG tail
    EXORL_TAIL(0) EXORL_TAIL(1) EXORL_TAIL(2) EXORL_TAIL(3) EXORL_TAIL(4) EXORL_TAIL(5) EXORL_TAIL(6) EXORL_TAIL(7) EXORL_TAIL(8)
#undef EXORL_TAIL

#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = X3 ? (acc2[i] + acc3[i]) + acc[i] : acc[i] + acc2[i];      // small terms first
    const int n = n0 + wn * 32 + (lane & 31);
    const float bias = P.bias ? P.bias[n] : 0.f;
    const bool relu = gb.relu != 0;
    float* crow = P.C + (int64_t)(m0 + wm * 32 + 4 * h) * P.ldc + n;
    if (gb.accumulate) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float* dst = crow + (int64_t)((r & 3) + 8 * (r >> 2)) * P.ldc;
            float v = acc[r] + bias;
            if (relu) v = fmaxf(v, 0.f);
            *dst = v + *dst;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[r] + bias;
            if (relu) v = fmaxf(v, 0.f);
            crow[(int64_t)((r & 3) + 8 * (r >> 2)) * P.ldc] = v;
        }
    }
}

// ---- 128 x 128 workgroup tile (4 waves, each 64 x 64 = 2 x 2 MFMA sub-tiles) ------------------------------------------
// The 64 x 64 kernel above re-reads every operand byte from LDS for 32 x 32 of output per wave: 8 KB of ds_read per 4 MFMAs,
// twice what the CU's 128 B/clk LDS port can feed while the MFMAs run (PMC: the pipe is LDS-bound). A 64 x 64 wave tile reuses
// each fragment twice (16 KB per 16 MFMAs: LDS and MFMA time balance) and halves the L2->LDS bytes per FLOP. An operand tile is
// two of the 64-row images of the kernel above side by side (same swizzles, same fragment reads); 4 stages x 32 KB = 128 KB of
// LDS, one workgroup per CU.
constexpr int G16H_STAGE = 4 * G16G_IMG;           // A0 A1 B0 B1

template <bool AT, bool BT, int NSTG, bool X3 = false>
__device__ __forceinline__ void gemm16h_body(const Gemm16Batch& gb, unsigned char* smem) {
    constexpr int IMG = G16G_IMG;
    const Gemm16Problem& P = gb.p[blockIdx.z];
    const int M = P.M, N = P.N, K = P.K;
    const int tiles_n = N >> 7, tiles_m = M >> 7;
    const int ntiles = tiles_n * tiles_m;
    if ((int)blockIdx.x >= ntiles) return;
    int tile = blockIdx.x;
    if (gb.swizzle && (ntiles & 7) == 0) tile = (tile & 7) * (ntiles >> 3) + (tile >> 3);
    const int m0 = (tile / tiles_n) << 7;
    const int n0 = (tile % tiles_n) << 7;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5;
    const int nk = K >> 6;                         // multiple of NSTG (checked by the launcher)

    constexpr int STAGE = (X3 ? 8 : 4) * IMG;      // A0 A1 B0 B1 [A0l A1l B0l B1l]
    f32x16 acc[2][2], accx[X3 ? 2 : 1][X3 ? 2 : 1];      // accx: the two cross terms hi*lo + lo*hi of the split-bf16 product, summed
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc[a][b][i] = 0.f; if constexpr (X3) accx[a][b][i] = 0.f; }

    // DMA source pointers: for each of the 4 images this wave's two 1-KB pieces (image rows 16*wave + 8j + lane/8)
    const unsigned short* src[X3 ? 16 : 8];
    int64_t kstep[2];
#pragma unroll
    for (int img = 0; img < 2; ++img)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int rr = 16 * wave + 8 * j + (lane >> 3), p = lane & 7;
            const int ma = m0 + 64 * img, nb = n0 + 64 * img;
            const int64_t oa = !AT ? (int64_t)(ma + rr) * P.lda + 8 * (p ^ ((rr >> 1) & 7)) : (int64_t)rr * P.lda + ma + 8 * (p ^ (4 * ((rr >> 1) & 1)));
            const int64_t ob = !BT ? (int64_t)(nb + rr) * P.ldb + 8 * (p ^ ((rr >> 1) & 7)) : (int64_t)rr * P.ldb + nb + 8 * (p ^ (4 * ((rr >> 1) & 1)));
            src[2 * img + j] = P.A + oa;
            src[4 + 2 * img + j] = P.B + ob;
            if constexpr (X3) { src[8 + 2 * img + j] = P.A_lo + oa; src[12 + 2 * img + j] = P.B_lo + ob; }
        }
    kstep[0] = AT ? 64 * P.lda : 64;
    kstep[1] = BT ? 64 * P.ldb : 64;
    const int piece = 16 * wave * ROWB;

    const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
    auto tr_off = [&](int rbase) {
        const int c = rbase + 16 * (g & 1) + 4 * pp;
        return (8 * (g >> 1) + qq) * ROWB + (((c >> 3) ^ (4 * ((qq >> 1) & 1))) << 4) + ((c & 7) << 1);
    };
    int aoff[2][4], boff[2][4];                    // [sub-tile][q]: fragment offsets inside this wave's A / B image
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            aoff[t][q] = AT ? tr_off(t * 32) + q * 16 * ROWB : lds_off(t * 32 + (lane & 31), 2 * q + h);
            boff[t][q] = BT ? tr_off(t * 32) + q * 16 * ROWB : lds_off(t * 32 + (lane & 31), 2 * q + h);
        }

    auto fill = [&](auto sc) {
        constexpr int st = decltype(sc)::value;
        unsigned char* base = smem + st * STAGE + piece;
#pragma unroll
        for (int i = 0; i < (X3 ? 8 : 4); ++i) {   // images A0 A1 B0 B1 [A0l A1l B0l B1l]
            __builtin_amdgcn_global_load_lds((const void*)src[2 * i], (lds_void*)(base + i * IMG), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const void*)src[2 * i + 1], (lds_void*)(base + i * IMG + 8 * ROWB), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            src[i] += kstep[0]; src[4 + i] += kstep[1];
            if constexpr (X3) { src[8 + i] += kstep[0]; src[12 + i] += kstep[1]; }
        }
    };
    auto frag = [&](const unsigned char* img, bool tr, int off) -> bf16x8 {
        if (!tr) return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(img + off));
        const v4s16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16*)(img + off));
        const v4s16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16*)(img + off + 4 * ROWB));
        typedef short v8s16 __attribute__((ext_vector_type(8)));
        const v8s16 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        return __builtin_bit_cast(bf16x8, v);
    };
    auto step = [&](auto sc, int t) {
        constexpr int st = decltype(sc)::value;
        // tile t has landed once at most the fills of the two younger tiles remain outstanding (8 DMA pieces per tile per wave)
        const int younger = nk - 1 - t;            // NSTG - 2 younger tiles may still be in flight
        static_assert(!X3 || NSTG == 2, "split-bf16 128 x 128 tiles: 2 stages of 64 KB");
        if (NSTG >= 4 && younger >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (NSTG >= 3 && younger >= 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + NSTG - 1 < nk) fill(std::integral_constant<int, (st + NSTG - 1) % NSTG>{});
        const unsigned char* As = smem + st * STAGE + wm * IMG;
        const unsigned char* Bs = smem + st * STAGE + (2 + wn) * IMG;
        if constexpr (X3) {                        // one q (k16) at a time: 8 fragments live instead of 32
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    ah[u] = frag(As, AT, aoff[u][q]);
                    bh[u] = frag(Bs, BT, boff[u][q]);
                    al[u] = frag(As + 4 * IMG, AT, aoff[u][q]);
                    bl[u] = frag(Bs + 4 * IMG, BT, boff[u][q]);
                }
#pragma unroll
                for (int ua = 0; ua < 2; ++ua)
#pragma unroll
                    for (int ub = 0; ub < 2; ++ub) {
                        acc[ua][ub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ua], bh[ub], acc[ua][ub], 0, 0, 0);
                        accx[ua][ub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ua], bl[ub], accx[ua][ub], 0, 0, 0);
                        accx[ua][ub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ua], bh[ub], accx[ua][ub], 0, 0, 0);
                    }
            }
            return;
        }
        bf16x8 af[2][4], bfr[2][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {              // q-major: the first MFMAs only wait for the first reads
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                af[u][q] = frag(As, AT, aoff[u][q]);
                bfr[u][q] = frag(Bs, BT, boff[u][q]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][q], bfr[0][q], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][q], bfr[1][q], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][q], bfr[0][q], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][q], bfr[1][q], acc[1][1], 0, 0, 0);
        }
    };

    fill(std::integral_constant<int, 0>{});
    if constexpr (NSTG >= 3) fill(std::integral_constant<int, 1>{});
    if constexpr (NSTG >= 4) fill(std::integral_constant<int, 2>{});
    for (int t = 0; t < nk; t += NSTG) {
        step(std::integral_constant<int, 0>{}, t);
        step(std::integral_constant<int, 1>{}, t + 1);
        if constexpr (NSTG >= 4) {
            step(std::integral_constant<int, 2>{}, t + 2);
            step(std::integral_constant<int, 3>{}, t + 3);
        }
    }

    const bool relu = gb.relu != 0;
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            const int n = n0 + wn * 64 + tb * 32 + (lane & 31);
            const float bias = P.bias ? P.bias[n] : 0.f;
            float* crow = P.C + (int64_t)(m0 + wm * 64 + ta * 32 + 4 * h) * P.ldc + n;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* dst = crow + (int64_t)((r & 3) + 8 * (r >> 2)) * P.ldc;
                float v = (X3 ? accx[ta][tb][r] + acc[ta][tb][r] : acc[ta][tb][r]) + bias;
                if (relu) v = fmaxf(v, 0.f);
                *dst = gb.accumulate ? v + *dst : v;
            }
        }
}

template <bool AT, bool BT, int NSTG>
__global__ __launch_bounds__(256) void gemm16h_kernel(const Gemm16Batch gb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_h[];
    gemm16h_body<AT, BT, NSTG>(gb, smem_h);
}

template <int NSTG>
__global__ __launch_bounds__(256) void gemm16h_mixed_kernel(const Gemm16Batch gb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_h[];
    if (gb.a_t[blockIdx.z]) gemm16h_body<true, true, NSTG>(gb, smem_h);
    else gemm16h_body<false, true, NSTG>(gb, smem_h);
}

template <bool AT, bool BT>
__global__ __launch_bounds__(256) void gemm16hx3_kernel(const Gemm16Batch gb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_h[];
    gemm16h_body<AT, BT, 2, true>(gb, smem_h);
}
__global__ __launch_bounds__(256) void gemm16hx3_mixed_kernel(const Gemm16Batch gb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_h[];
    if (gb.a_t[blockIdx.z]) gemm16h_body<true, true, 2, true>(gb, smem_h);
    else gemm16h_body<false, true, 2, true>(gb, smem_h);
}
constexpr int G16HX3_LDS = 2 * 8 * G16G_IMG;       // 128 KB: one workgroup per CU
static int g16hx3_enable() {
    static bool done = false;
    if (done) return 0;
    EXORL_CHECK_HIP(hipFuncSetAttribute((const void*)gemm16hx3_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, G16HX3_LDS));
    EXORL_CHECK_HIP(hipFuncSetAttribute((const void*)gemm16hx3_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G16HX3_LDS));
    EXORL_CHECK_HIP(hipFuncSetAttribute((const void*)gemm16hx3_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G16HX3_LDS));
    EXORL_CHECK_HIP(hipFuncSetAttribute((const void*)gemm16hx3_mixed_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, G16HX3_LDS));
    done = true;
    return 0;
}
// 128 x 128 tiles pay off when they still fill the chip: 4-problem launches of 1024^2 outputs are 256 workgroups
static bool g16hx3_fits(const Gemm16Batch& gb, int count) {
    if (g_gemm16_variant >= 0 && (g_gemm16_variant & 65536)) return false;        // experiment switch: off
    int tiles = 0;
    for (int i = 0; i < count; ++i) {
        if (gb.p[i].M % 128 != 0 || gb.p[i].N % 128 != 0 || gb.p[i].K % 128 != 0) return false;
        tiles += (gb.p[i].M >> 7) * (gb.p[i].N >> 7);
    }
    return tiles >= 256 || (g_gemm16_variant >= 0 && (g_gemm16_variant & 131072));
}
constexpr size_t G16H_LDS = (size_t)G16G_NSTG * G16H_STAGE;      // 128 KB at 4 stages; 64 KB at 2 (two workgroups per CU)
static int g16h_stages() { return (g_gemm16_variant >= 0 && (g_gemm16_variant & 512)) ? 2 : 4; }
template <int NSTG>
static int g16h_enable_n() {          // > 64 KB of dynamic LDS needs the opt-in, once per kernel
    static bool done = false;
    if (done) return 0;
    const int lds = NSTG * G16H_STAGE;
    EXORL_CHECK_HIP(hipFuncSetAttribute((const void*)gemm16h_kernel<false, false, NSTG>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    EXORL_CHECK_HIP(hipFuncSetAttribute((const void*)gemm16h_kernel<false, true, NSTG>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    EXORL_CHECK_HIP(hipFuncSetAttribute((const void*)gemm16h_kernel<true, true, NSTG>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    EXORL_CHECK_HIP(hipFuncSetAttribute((const void*)gemm16h_mixed_kernel<NSTG>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    done = true;
    return 0;
}
static int g16h_enable() { return g16h_stages() == 2 ? g16h_enable_n<2>() : g16h_enable_n<4>(); }
static bool g16h_fits(const Gemm16Batch& gb, int count) {
    // measured slower than the 64 x 64 tiles on the 1024-wide layers (15-17 us vs 9-12 us per problem: one workgroup per CU leaves
    // the k-tile chain wait -> barrier -> DMA issue -> LDS reads -> MFMA exposed): opt-in through tuning bit 1024
    if (g_gemm16_variant >= 0 && (g_gemm16_variant & 65536)) return false;
    int tiles = 0;
    for (int i = 0; i < count; ++i) {
        if (gb.p[i].M % 128 != 0 || gb.p[i].N % 128 != 0) return false;
        tiles += (gb.p[i].M >> 7) * (gb.p[i].N >> 7);
    }
    // by default only where 128 x 128 tiles still give every CU a workgroup (the 4-problem launches); bit 1024 forces them everywhere
    return tiles >= 256 || (g_gemm16_variant >= 0 && (g_gemm16_variant & 1024));
}

// ---- "p" kernels: 128 x TN workgroup tile (TN = 128 or 64), k32 stages, 4-deep LDS-DMA ring, XCD-local tile blocks -----------
// What round 2's measurements changed (tools/micro/fill_bench.hip): a CU pulls ~115 GB/s from its XCD's L2 on EVERY path (LDS-DMA,
// registers, both) but only 30-40 GB/s from the Infinity Cache, so the ~75 GB/s the kernels above sustain is an L2 miss rate, not a
// DMA limit: with tiles dealt in id order an XCD walks one tile-row of every problem and re-fetches each B strip once per tile.
// Here (a) workgroup id -> tile goes through xcd_tile(): the 32 workgroups of an XCD (one per CU, all resident) form a compact block
// of ONE problem's output (512 x 512 or 512 x 1024), so every operand strip an XCD touches is fetched once and reused 4-8 times while
// the CUs walk k together; (b) the stage is 32 k wide (32 KB for a split-bf16 128 x 128 tile), four stages deep: two to three
// stages are always in flight, which a single workgroup per CU needs to cover the DMA latency (the 2 x 64 KB ring above could keep
// only one); (c) the k-loop is software-pipelined across the barrier: the fragments of the next k16 are read while the MFMAs of the
// current one issue, the barrier that certifies stage t+1 sits in the middle of step t.
// LDS images per 64-row block and stage (4 KB): "row image" [64 rows][32 k] with 64-byte rows, 16-byte units swizzled by
// (row >> 2) & 3 (the 16 lanes of a ds_read_b128 group then hit 16 distinct bank quads); "k image" [32 k][64 rows] = the first half
// of the 64-k image above (same swizzle, same ds_read_b64_tr_b16 fragments).

template <int I, int N, typename F>
__device__ __forceinline__ void g16p_static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); g16p_static_for<I + 1, N>(f); }
}

// ds_read_b64_tr_b16 through inline asm: the builtin carries no memory operand, so the compiler assumes it may read what an LDS-DMA
// still in flight is writing and puts an s_waitcnt vmcnt(0) in front of it — which also waits for the stage that was issued a moment
// ago, i.e. no prefetch at all for the transposed operands. The asm form is invisible to that pass; its results are fenced by
// g16p_lds_fence() (an lgkmcnt(0) that names them) before the MFMAs of the next region read them. The two 64-bit halves stay separate
// variables until that fence: a register move the compiler makes to pack them earlier would copy registers the LDS has not written yet.
template <int IMM>
__device__ __forceinline__ v4s16 g16p_read_tr(unsigned addr) {
    v4s16 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(IMM));
    return r;
}
struct G16pFrag {
    bf16x8 v;            // MFMA operand (row-image fragments are read straight into it)
    v4s16 lo, hi;        // k-image fragments: the two transposed 64-bit reads, packed into v by the fence
};
__device__ __forceinline__ void g16p_lds_fence(G16pFrag& f) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.lo), "+v"(f.hi));
    typedef short v8s16 __attribute__((ext_vector_type(8)));
    const v8s16 v = {f.lo.x, f.lo.y, f.lo.z, f.lo.w, f.hi.x, f.hi.y, f.hi.z, f.hi.w};
    f.v = __builtin_bit_cast(bf16x8, v);
}

// KS = k per stage (32: 64-byte row-image rows = half cache lines, four stages fit the 128 x 128 split-bf16 tile; 64: whole lines, the
// 64-wide images and swizzle of the gemm16g kernels above), NSTG = ring depth. A stage's slot is refilled with the stage NSTG ahead as soon
// as its last fragments have been read, i.e. behind the barrier that opens the stage's last k16.
// MS = MFMA shape: 32 -> v_mfma_f32_32x32x16_bf16 (a region = one k16), 16 -> v_mfma_f32_16x16x32_bf16 (a region = one k32; row images on
// 64-wide stages only). Same wave tile, same LDS images, same fragment bytes and MFMA cycles per region pair; the chip holds a higher clock
// on the 16 x 16 shape under a dense bf16 load (MI355X_MICROARCH.md, DVFS give-back (7)), so the faster one is picked by wall time.
// WS = wave-specialised workgroup of 512 threads: waves 0-3 (one per SIMD) run the MFMA + fragment-read stream only, waves 4-7 — their SIMD
// partners — issue the LDS-DMA fills and certify them (vmcnt) in front of the stage barrier. An LDS-DMA instruction costs its wave ~60-180
// issue cycles (MI355X_MICROARCH.md constants), eight of them per k32 stage oversubscribe the gaps of the 24 MFMAs of that stage; in a
// partner wave they cost the computing wave next to nothing (same guide, "Two waves per SIMD" item 7).
// STAMP (diagnostic build, tuning bit 33554432; no product launch takes it): wave 0 of every workgroup records s_memtime (shader clock) and
// s_memrealtime (100 MHz) at entry, at the first k-step, after the last k-step and after its stores have drained, into g16p_stamps —
// the in-kernel clock and the split prologue / k-loop / epilogue (MI355X_MICROARCH.md, DVFS give-back (6)). Nothing is computed from them.
__device__ unsigned long long g16p_stamps[8 * 1024];
// ABL (stamped builds only): 1 = no DMA after the prologue, 2 = no fragment reads, 3 = no MFMAs — what each part costs in CYCLES
// SPREAD = number of consecutive k-regions over which the LDS-DMA pieces of one stage refill are issued (1: all of them in the region behind
// the barrier that frees the slot). Stamps (tools/micro/stamp_bench.py) put the k-loop at MFMA cycles + ~60 cycles per DMA piece of the
// wave: the four waves issue their pieces together, the CU's vector-memory path takes them one at a time, and a wave blocked on an issue
// cannot feed its matrix pipe. Spread over two regions the same pieces have twice the MFMAs to hide behind.
// STAG = 0 | 2 | 4: the waves of a workgroup run in lockstep between barriers, so with one instruction stream they issue their DMA pieces
// in the same MFMA gaps. With STAG phases, wave w places its pieces in the gaps g = w mod STAG (mod STAG): STAG copies of the k-loop that
// differ only in that placement, chosen once per wave in front of the loop.
// NW = waves per workgroup, all of them loading and computing (4, or 8 = 512 threads on the same 128 x TN tile with half-size wave tiles).
// tools/micro/fill_bench.hip: ONE wave issues an LDS-DMA piece about every 100 cycles whatever it keeps in flight (4 waves per CU deliver
// 85 GB/s at 8, 16 or 32 pieces in flight each; 8 waves 117 GB/s, where the CU's L2 path saturates) — a 128 x 64 tile needs 24 pieces per
// 384 MFMA-cycles, i.e. more issue slots than four waves have.
// PFD > 0: one more wave (id NW) that computes nothing: it walks PFD stages ahead of the ring and touches this workgroup's 1/32 share of the
// lines the XCD's tile block will need (one dword per 128-byte line into a dead register), so that the fabric round trip of the FIRST touch of
// every operand line — each XCD fetches each of its lines exactly once per launch — is taken ahead of the LDS-DMA stream instead of inside it.
// It joins every stage barrier (that is its pacing: far-ahead lines would push unread ones out of a 4 MB L2) and exits after the last one.
template <bool AT, bool BT, bool X3, int TN, int KS, int NSTG, int MS = 32, bool WS = false, bool STAMP = false, int ABL = 0, int SPREAD = 1, int STAG = 0,
          int NW = 4, int PFD = 0>
__device__ __forceinline__ void gemm16p_body(const Gemm16Batch& gb, unsigned char* smem, int pidx, int m0, int n0) {
    constexpr int NBA = 2, NBB = TN / 64, NPL = X3 ? 2 : 1;
    constexpr int WR = (NW == 8 && TN == 64) ? 32 : 64;     // wave tile: WR rows x WC columns
    constexpr int WC = NW == 8 ? 32 : TN / 2;
    constexpr int WN = TN / WC;                             // waves along N (the rest along M)
    static_assert((NW == 4 || (NW == 8 && MS == 32 && !WS && KS == 64)) && (128 / WR) * WN == NW, "wave grid");
    constexpr int BLK = 64 * 2 * KS;                        // one 64-row block of one plane and stage (either image kind)
    constexpr int PLANE = (NBA + NBB) * BLK;                // [A blk0][A blk1][B blk0][B blk1]
    constexpr int STAGE = NPL * PLANE;                      // hi plane, then lo plane
    constexpr int PPW = KS / 8 / NW;                        // 1-KB DMA pieces per wave and block
    constexpr int NP = (NBA + NBB) * NPL * PPW;             // DMA pieces per wave and stage
    constexpr int NQ = MS == 16 ? KS / 32 : KS / 16;        // regions (k16 / k32) per stage
    constexpr int SA = WR / MS;                             // MS-row sub-tiles of a wave along M
    constexpr int SB = WC / MS;                             // MS-column sub-tiles of a wave along N
    constexpr int NPAIR = SA * SB;                          // MS x MS accumulators of a wave
    constexpr int NM = NPAIR * (X3 ? 3 : 1);                // MFMAs per region
    constexpr int NF = (SA + SB) * NPL;                     // fragments per region
    constexpr int SMIN = SA < SB ? SA : SB;
    static_assert(MS == 32 || (MS == 16 && !AT && !BT && KS == 64), "16 x 16 x 32 fragments are built for row images on 64-wide stages");
    typedef float acc_t __attribute__((ext_vector_type(MS == 16 ? 4 : 16)));
    constexpr int FPG = (NF + NM - 2) / (NM - 1);           // fragments read per MFMA gap: all of them behind the first NM-1 MFMAs
    static_assert(NQ % 2 == 0 && NP * (NSTG - 1) <= 63 && NSTG >= 2, "stage geometry");
    const Gemm16Problem& P = gb.p[pidx];
    const int nst = P.K / KS;                               // multiple of NSTG, >= NSTG (launcher)
    unsigned long long st_t[4] = {0, 0, 0, 0}, st_r[4] = {0, 0, 0, 0}, wacc_v = 0, wacc_b = 0;      // wacc: cycles in the stage waits (own DMA | barrier)
    auto stamp = [&](int i) {
        if constexpr (STAMP) { st_t[i] = __builtin_amdgcn_s_memtime(); st_r[i] = __builtin_amdgcn_s_memrealtime(); }
    };
    stamp(0);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = WS && wave_id >= 4;                 // wave-uniform role
    if constexpr (PFD > 0) {
        static_assert(!AT && !BT && KS == 64 && !WS, "the prefetch wave is written for row images on 64-wide stages (one line per row and stage)");
        if (wave_id == NW) {
            int nA = 0, nB = 0, rA = 0, rB = 0;
            if (gb.xcd_map) {                   // the XCD's tile block (xcd_tile): rows rA .. rA + nA of A, rows rB .. rB + nB of B
                const int X = 8 / gb.count, sm = X == 8 ? 4 : 2, sn = X == 2 ? 1 : 2, xq = (blockIdx.x & 7) % X;
                const int bm = (P.M >> 7) / sm, bn = (P.N / TN) / sn;
                nA = bm * 128; rA = (xq / sn) * nA;
                nB = bn * TN;  rB = (xq % sn) * nB;
            }
            const int nl = (nA + nB) * NPL, slot = blockIdx.x >> 3;
            // every touch loads into the SAME register, which stays live (read-write operand) until the loads have landed: an asm output the
            // compiler believes dead would be handed to the next value and overwritten when the data arrives
            unsigned dead = 0;
            auto touch = [&](int t) {           // this workgroup's share of stage t's lines: line l = (operand, plane, row)
                for (int l = slot * 64 + lane; l < nl; l += 32 * 64) {
                    const bool isB = l >= nA * NPL;
                    const int ll = isB ? l - nA * NPL : l, n = isB ? nB : nA, pl = ll / n, r = ll % n;
                    const unsigned short* base = isB ? (pl ? P.B_lo : P.B) : (pl ? P.A_lo : P.A);
                    const unsigned short* ptr = base + (int64_t)((isB ? rB : rA) + r) * (isB ? P.ldb : P.lda) + (int64_t)t * KS;
                    asm volatile("global_load_dword %0, %1, off" : "+v"(dead) : "v"(ptr) : "memory");
                }
            };
            for (int t = NSTG; t < PFD && t < nst; ++t) touch(t);
            __builtin_amdgcn_s_barrier();                               // B_0
            for (int t = 0; t + 1 < nst; ++t) {
                if (t + PFD < nst) touch(t + PFD);
                __builtin_amdgcn_s_barrier();                           // B_{t+1}
            }
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(dead) :: "memory");
            return;
        }
    }
    const int wave = WS ? (wave_id & 3) : wave_id;          // consumer: its 64 x TN/2 quadrant; loader: which pieces of a block it fetches
    const int wm = wave / WN, wn = wave % WN;
    const int h = lane >> 5;
    const int arow0 = (wm * WR) & 63, ablk = (wm * WR) >> 6;        // this wave's rows inside A block ablk
    const int bcol0 = (wn * WC) & 63, bblk = (wn * WC) >> 6;        // this wave's columns inside B block bblk

    acc_t acc[NPAIR], accx[X3 ? NPAIR : 1];                 // [ua * SB + ub]; accx: the two cross terms hi*lo + lo*hi
#pragma unroll
    for (int a = 0; a < NPAIR; ++a)
#pragma unroll
        for (int i = 0; i < (MS == 16 ? 4 : 16); ++i) { acc[a][i] = 0.f; if constexpr (X3) accx[a][i] = 0.f; }

    // row image: 2*KS-byte rows, 16-byte units swizzled so that the 16 lanes of a ds_read_b128 group hit 16 distinct bank quads
    auto row_off = [](int row, int unit) {
        if constexpr (KS == 32) return row * 64 + ((unit ^ ((row >> 2) & 3)) << 4);
        else return lds_off(row, unit);
    };
    // ---- DMA sources: this wave's pieces (index wave + 4 j) of every block; per-lane addresses carry the LDS swizzle
    const unsigned short* src[NP];
    {
        constexpr int U = KS / 8, RP = 64 / U;              // row image: 16-byte units per row, rows per 1-KB piece
        const int rr = lane / U, ur = lane % U;
        const int r8 = lane >> 3, u8 = lane & 7;            // k image piece: 8 k-rows x 8 units
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int b = 0; b < NBA + NBB; ++b)
#pragma unroll
                for (int j = 0; j < PPW; ++j) {
                    const int pc = wave + NW * j;
                    const bool isA = b < NBA;
                    const bool tr = isA ? AT : BT;
                    const int64_t ld = isA ? P.lda : P.ldb;
                    const int r0 = (isA ? m0 : n0) + 64 * (isA ? b : b - NBA);
                    const int irow = pc * RP + rr;          // row inside the 64-row block
                    const int usrc = KS == 32 ? (ur ^ ((irow >> 2) & 3)) : (ur ^ ((irow >> 1) & 7));
                    const int64_t o = !tr ? (int64_t)(r0 + irow) * ld + 8 * usrc
                                          : (int64_t)(8 * pc + r8) * ld + r0 + 8 * (u8 ^ (4 * ((r8 >> 1) & 1)));
                    const unsigned short* base = isA ? (pl ? P.A_lo : P.A) : (pl ? P.B_lo : P.B);
                    src[(pl * (NBA + NBB) + b) * PPW + j] = base + o;
                }
    }
    const int64_t kstepA = AT ? KS * P.lda : KS, kstepB = BT ? KS * P.ldb : KS;

    // ---- fragment offsets inside a stage's hi plane (lo plane: + PLANE)
    const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
    auto tr_off = [&](int rbase) {      // k image: k-row 8*(g>>1)+qq (+4 for the second half), columns rbase + 16*(g&1) + 4*pp ..+3
        const int c = rbase + 16 * (g & 1) + 4 * pp;
        return (8 * (g >> 1) + qq) * ROWB + (((c >> 3) ^ (4 * ((qq >> 1) & 1))) << 4) + ((c & 7) << 1);
    };
    int aoff[SA][NQ], boff[SB][NQ];                 // [sub-tile][q]
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
#pragma unroll
        for (int u = 0; u < SA; ++u) {
            if constexpr (MS == 16) aoff[u][q] = ablk * BLK + row_off(arow0 + u * 16 + (lane & 15), 4 * q + (lane >> 4));    // lane: row l & 15, k = 8 (l >> 4) ..+7
            else aoff[u][q] = ablk * BLK + (AT ? tr_off(arow0 + u * 32) + q * 16 * ROWB : row_off(arow0 + u * 32 + (lane & 31), 2 * q + h));
        }
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int blk = bblk, r0 = bcol0 + u * MS;
            if constexpr (MS == 16) boff[u][q] = (NBA + blk) * BLK + row_off(r0 + (lane & 15), 4 * q + (lane >> 4));
            else boff[u][q] = (NBA + blk) * BLK + (BT ? tr_off(r0) + q * 16 * ROWB : row_off(r0 + (lane & 31), 2 * q + h));
        }
    }

    auto fill_one = [&](int st, auto ic) {          // DMA piece I of stage slot st
        constexpr int I = decltype(ic)::value;
        constexpr int j = I % PPW, b = (I / PPW) % (NBA + NBB), pl = I / (PPW * (NBA + NBB));
        __builtin_amdgcn_global_load_lds((const void*)src[I], (lds_void*)(smem + st * STAGE + pl * PLANE + b * BLK + (wave + NW * j) * 1024), 16, 0, 0);
        src[I] += b < NBA ? kstepA : kstepB;
    };
    auto fill = [&](int st) { g16p_static_for<0, NP>([&](auto ic) { fill_one(st, ic); }); };

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;       // LDS byte address of the ring
    auto frag = [&](G16pFrag& f, auto plane, bool tr, int st, int off) {     // plane 0 = hi, 1 = lo
        constexpr int PL = decltype(plane)::value * PLANE;
        if (!tr) { f.v = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(smem + st * STAGE + PL + off)); return; }
        const unsigned a = lds0 + st * STAGE + off;
        f.lo = g16p_read_tr<PL>(a);
        f.hi = g16p_read_tr<PL + 4 * ROWB>(a);
    };
    struct Frags { G16pFrag ah[SA], bh[SB], al[X3 ? SA : 1], bl[X3 ? SB : 1]; };
    // fragment J of a region, in the order the MFMAs want them: per plane A0 B0 A1 B1 ... interleaved, then the rest of the longer list
    auto read_one = [&](Frags& f, int st, auto qc, auto jc) {
        constexpr int J = decltype(jc)::value, q = decltype(qc)::value;
        constexpr int pl = X3 ? J % 2 : 0, k = X3 ? J / 2 : J;           // k: 0 = A0, 1 = B0, 2 = A1, 3 = B1, ...
        constexpr bool isA = k < 2 * SMIN ? k % 2 == 0 : SA > SB;
        constexpr int u = k < 2 * SMIN ? k / 2 : k - SMIN;
        if constexpr (isA) {
            if constexpr (pl == 0) frag(f.ah[u], std::integral_constant<int, 0>{}, AT, st, aoff[u][q]);
            else frag(f.al[u], std::integral_constant<int, 1>{}, AT, st, aoff[u][q]);
        } else {
            if constexpr (pl == 0) frag(f.bh[u], std::integral_constant<int, 0>{}, BT, st, boff[u][q]);
            else frag(f.bl[u], std::integral_constant<int, 1>{}, BT, st, boff[u][q]);
        }
    };
    auto fence = [&](Frags& f) {               // the asm-issued transposed reads of f have landed (see g16p_read_tr)
        if constexpr (AT) {
#pragma unroll
            for (int u = 0; u < SA; ++u) { g16p_lds_fence(f.ah[u]); if constexpr (X3) g16p_lds_fence(f.al[u]); }
        }
        if constexpr (BT) {
#pragma unroll
            for (int u = 0; u < SB; ++u) { g16p_lds_fence(f.bh[u]); if constexpr (X3) g16p_lds_fence(f.bl[u]); }
        }
    };
    // MFMA I of a k16. Operands swapped: the accumulator is the TRANSPOSED 32 x 32 block, i.e. lane = output row, registers r..r+3 =
    // four consecutive output columns -> the epilogue stores 16 bytes per lane (16 stores per wave instead of 64)
    auto mfma_one = [&](const Frags& f, auto ic) {
        constexpr int I = decltype(ic)::value;
        constexpr int pair = X3 ? I / 3 : I, term = X3 ? I % 3 : 0, ua = pair / SB, ub = pair % SB;
        if constexpr (MS == 16) {       // D[n][m]: lane = output row (lane & 15), registers 0..3 = columns 4 (lane >> 4) ..+3
            if constexpr (term == 0) acc[pair] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.bh[ub].v, f.ah[ua].v, acc[pair], 0, 0, 0);
            else if constexpr (term == 1) accx[pair] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.bl[ub].v, f.ah[ua].v, accx[pair], 0, 0, 0);
            else accx[pair] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.bh[ub].v, f.al[ua].v, accx[pair], 0, 0, 0);
        } else {
        if constexpr (term == 0) acc[pair] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.bh[ub].v, f.ah[ua].v, acc[pair], 0, 0, 0);
        else if constexpr (term == 1) accx[pair] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.bl[ub].v, f.ah[ua].v, accx[pair], 0, 0, 0);
        else accx[pair] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.bh[ub].v, f.al[ua].v, accx[pair], 0, 0, 0);
        }
    };
    // One scheduling region = the NM MFMAs of a k16 on `cur`, with the reads of the NEXT k16's fragments (stage slot nst_, region NQn) into
    // `nxt` (FPG per MFMA gap) and, when NFILL > 0, the DMA issues of stage slot `fst` written out between them; sched_barrier(0)
    // after every piece keeps the machine scheduler from regrouping them (left alone it sinks the reads to just before their use; issued
    // as one block they exceed the 4-bit lgkmcnt and the compiler waits for all of them). The reads complete in the shadow of the matrix
    // pipe, the next region opens with waits that cost nothing.
    auto region = [&](Frags& cur, Frags& nxt, auto has_next, int nst_, auto nq, auto nfill, int fst, auto chunk, auto phase) {
        constexpr int PH = decltype(phase)::value;           // -1: pieces in the first gaps; >= 0: in the gaps congruent to PH mod STAG
        constexpr bool HN = decltype(has_next)::value;
        constexpr int NFILL = decltype(nfill)::value;
        constexpr int F0 = decltype(chunk)::value * NP / SPREAD, F1 = (decltype(chunk)::value + 1) * NP / SPREAD;      // this region's share of the refill
        g16p_static_for<0, NM>([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            if constexpr (ABL != 3) mfma_one(cur, ic);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (HN && ABL != 2) g16p_static_for<I * FPG, ((I + 1) * FPG < NF ? (I + 1) * FPG : NF)>([&](auto jc) { read_one(nxt, nst_, nq, jc); });
            constexpr int PPG = (F1 - F0 + NM - 1) / NM;     // DMA pieces per gap
            if constexpr (NFILL > 0 && ABL != 1) {
                if constexpr (PH < 0) g16p_static_for<F0 + I * PPG, (F0 + (I + 1) * PPG < F1 ? F0 + (I + 1) * PPG : F1)>([&](auto pc) { fill_one(fst, pc); });
                else g16p_static_for<0, F1 - F0>([&](auto ii) {
                    constexpr int i = decltype(ii)::value;
                    if constexpr ((PH + STAG * i) % NM == I) fill_one(fst, std::integral_constant<int, F0 + i>{});
                });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        // The fence of the fragments just requested closes the region that issued them — the same place in the instruction stream as the head
        // of the next region, but inside the same basic block. Round 2 had it at the head of the consuming region: wherever a branch or a loop
        // back-edge lay between the two (the nst == NSTG test behind the prologue reads, the group loop) the register allocator was free to
        // resolve the join with v_mov copies of the asm outputs BEFORE the wait, i.e. copies of registers the LDS had not written yet
        // (tools/check_async_reads.py shows them in the round-2 ISA of every SPREAD = 2 kernel with a k-image operand, split-bf16 included;
        // the plain-bf16 launches — 2-4 MFMAs between the reads and the copy — lost that race visibly: DESIGN 4, "the SPREAD = 2 anomaly").
        if constexpr (HN && ABL != 2) fence(nxt);
    };
    // a stage has landed for this wave once at most `younger` younger stages' pieces are outstanding (NP per stage)
    auto wait_landed = [&](auto younger) {
        constexpr int y = decltype(younger)::value;
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(y * NP) : "memory");
    };
    Frags fr[2];
    using T = std::true_type;
    using F = std::false_type;
    using NoFill = std::integral_constant<int, 0>;
    using Fill = std::integral_constant<int, NP>;
    // NSTG stages on slots 0..NSTG-1; LAST = the final group (nothing left to refill, no barrier after the last stage). Every condition is
    // a compile-time constant: a run-time branch in here makes the compiler fall back to lgkmcnt(0) / vmcnt(0)
    using C0 = std::integral_constant<int, 0>;
    static_assert(SPREAD >= 1 && SPREAD <= NQ && (SPREAD == 1 || !WS), "refill spread");
    auto group = [&](auto first, auto last, auto phase) {
        constexpr bool FIRST = decltype(first)::value, LAST = decltype(last)::value;
        g16p_static_for<0, NSTG>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            // the refill the previous stage began behind its barrier (chunk 0) continues in this stage's first SPREAD-1 regions
            constexpr bool PENDING = SPREAD > 1 && (s > 0 ? !LAST : !FIRST);
            g16p_static_for<0, NQ - 1>([&](auto qc) {        // all but the last k16 of the stage: read the next k16 of the same stage
                constexpr int q = decltype(qc)::value;
                if constexpr (PENDING && q < SPREAD - 1) {
                    region(fr[q & 1], fr[(q + 1) & 1], T{}, s, std::integral_constant<int, q + 1>{}, Fill{}, (s + NSTG - 1) % NSTG, std::integral_constant<int, q + 1>{}, phase);
                }
                else
                    region(fr[q & 1], fr[(q + 1) & 1], T{}, s, std::integral_constant<int, q + 1>{}, NoFill{}, 0, C0{}, phase);
            });
            Frags& cur = fr[(NQ - 1) & 1];
            Frags& nxt = fr[NQ & 1];
            if constexpr (!LAST || s < NSTG - 1) {
                // the next stage has landed for this wave when only the stages younger than it are outstanding: NSTG-2 of them in the steady
                // state, NSTG-2-s in the last group (nothing is issued there any more)
                unsigned long long w0 = 0, w1 = 0;
                if constexpr (STAMP) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); w0 = __builtin_amdgcn_s_memtime(); }
                if constexpr (!WS && ABL != 1) wait_landed(std::integral_constant<int, LAST ? NSTG - 2 - s : NSTG - 2>{});
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's reads of stage s are done: its slot may be refilled
                if constexpr (STAMP) w1 = __builtin_amdgcn_s_memtime();
                __builtin_amdgcn_s_barrier();                           // ... by anyone; and the next stage has landed for everyone
                asm volatile("" ::: "memory");
                if constexpr (STAMP) { const unsigned long long w2 = __builtin_amdgcn_s_memtime(); wacc_v += w1 - w0; wacc_b += w2 - w1; }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!LAST && !WS) region(cur, nxt, T{}, (s + 1) % NSTG, std::integral_constant<int, 0>{}, Fill{}, s, C0{}, phase);
                else region(cur, nxt, T{}, (s + 1) % NSTG, std::integral_constant<int, 0>{}, NoFill{}, 0, C0{}, phase);
            } else {
                region(cur, nxt, F{}, 0, std::integral_constant<int, 0>{}, NoFill{}, 0, C0{}, phase);
            }
        });
    };

    if constexpr (WS) {
        if (loader) {
            // the same barriers as the consumers, in the same order: B_0 = stage 0 has landed; B_{s+1} = stage s + 1 has landed (this wave's
            // pieces: vmcnt; everyone's: the barrier) and every consumer has finished reading stage s, whose slot takes stage s + NSTG
            g16p_static_for<0, NSTG>([&](auto sc) { fill(decltype(sc)::value); });
            wait_landed(std::integral_constant<int, NSTG - 1>{});
            __builtin_amdgcn_s_barrier();
            auto lgroup = [&](auto last) {
                constexpr bool LAST = decltype(last)::value;
                g16p_static_for<0, NSTG>([&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    if constexpr (!LAST || s < NSTG - 1) {
                        wait_landed(std::integral_constant<int, LAST ? NSTG - 2 - s : NSTG - 2>{});
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                        if constexpr (!LAST) fill(s);
                    }
                });
            };
            for (int t = 0; t + NSTG < nst; t += NSTG) lgroup(F{});
            lgroup(T{});
            return;
        }
    } else {
        g16p_static_for<0, NSTG>([&](auto sc) { fill(decltype(sc)::value); });
        wait_landed(std::integral_constant<int, NSTG - 1>{});
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    stamp(1);
    g16p_static_for<0, NF>([&](auto jc) { read_one(fr[0], 0, std::integral_constant<int, 0>{}, jc); });
    fence(fr[0]);                               // before any control flow (see region())
    auto kloop = [&](auto phase) {
        if constexpr (SPREAD == 1) {
            for (int t = 0; t + NSTG < nst; t += NSTG) group(F{}, F{}, phase);
            group(F{}, T{}, phase);
        } else if (nst == NSTG) {
            group(T{}, T{}, phase);
        } else {
            group(T{}, F{}, phase);
            for (int t = NSTG; t + NSTG < nst; t += NSTG) group(F{}, F{}, phase);
            group(F{}, T{}, phase);
        }
    };
    static_assert(STAG <= 0 || ((STAG == 2 || STAG == 4) && NM % STAG == 0 && !WS), "stagger");
    if constexpr (STAG <= 0) kloop(std::integral_constant<int, -1>{});
    else if constexpr (STAG == 2) { if (wave & 1) kloop(std::integral_constant<int, 1>{}); else kloop(C0{}); }
    else {
        if (wave == 0) kloop(C0{});
        else if (wave == 1) kloop(std::integral_constant<int, 1>{});
        else if (wave == 2) kloop(std::integral_constant<int, 2>{});
        else kloop(std::integral_constant<int, 3>{});
    }
    stamp(2);

    const bool relu = gb.relu != 0;
    if constexpr (MS == 16) {
#pragma unroll
        for (int ta = 0; ta < SA; ++ta)
#pragma unroll
            for (int tb = 0; tb < SB; ++tb) {
                const int nb = n0 + wn * WC + tb * 16 + 4 * (lane >> 4);
                if (P.n_store && nb >= P.n_store) continue;
                float4* dst = reinterpret_cast<float4*>(P.C + (int64_t)(m0 + wm * WR + ta * 16 + (lane & 15)) * P.ldc + nb);
                const float4 bias = P.bias ? *reinterpret_cast<const float4*>(P.bias + nb) : make_float4(0.f, 0.f, 0.f, 0.f);
                const acc_t& a0 = acc[ta * SB + tb];
                const acc_t& ax = accx[X3 ? ta * SB + tb : 0];
                float4 v;
                v.x = (X3 ? ax[0] + a0[0] : a0[0]) + bias.x;
                v.y = (X3 ? ax[1] + a0[1] : a0[1]) + bias.y;
                v.z = (X3 ? ax[2] + a0[2] : a0[2]) + bias.z;
                v.w = (X3 ? ax[3] + a0[3] : a0[3]) + bias.w;
                if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (gb.accumulate) { const float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                *dst = v;
            }
    } else if (P.head_part) {
        // folded scalar head: this wave's share of relu(acc + bias) . head_w for each of its rows; lanes l and l + 32 hold the two column
        // interleaves of one row. Nothing of C is stored.
#pragma unroll
        for (int ta = 0; ta < SA; ++ta) {
            float dot = 0.f;
#pragma unroll
            for (int tb = 0; tb < SB; ++tb) {
                const int nb = n0 + wn * WC + tb * 32 + 4 * h;
                const acc_t& a0 = acc[ta * SB + tb];
                const acc_t& ax = accx[X3 ? ta * SB + tb : 0];
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const float4 bias = P.bias ? *reinterpret_cast<const float4*>(P.bias + nb + 8 * g4) : make_float4(0.f, 0.f, 0.f, 0.f);
                    const float4 w = *reinterpret_cast<const float4*>(P.head_w + nb + 8 * g4);
                    float4 v;
                    v.x = (X3 ? ax[4 * g4 + 0] + a0[4 * g4 + 0] : a0[4 * g4 + 0]) + bias.x;
                    v.y = (X3 ? ax[4 * g4 + 1] + a0[4 * g4 + 1] : a0[4 * g4 + 1]) + bias.y;
                    v.z = (X3 ? ax[4 * g4 + 2] + a0[4 * g4 + 2] : a0[4 * g4 + 2]) + bias.z;
                    v.w = (X3 ? ax[4 * g4 + 3] + a0[4 * g4 + 3] : a0[4 * g4 + 3]) + bias.w;
                    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    dot += (v.x * w.x + v.y * w.y) + (v.z * w.z + v.w * w.w);
                }
            }
            dot += __shfl_xor(dot, 32);
            if (h == 0) P.head_part[(int64_t)(m0 + wm * WR + ta * 32 + (lane & 31)) * (P.N / WC) + (n0 / WC + wn)] = dot;
        }
    } else {
#pragma unroll
    for (int ta = 0; ta < SA; ++ta)
#pragma unroll
        for (int tb = 0; tb < SB; ++tb) {
            // lane: output row m0 + .. + (lane & 31); registers 4g..4g+3: columns nb + 8g + 4h .. +3
            const int nb = n0 + wn * WC + tb * 32 + 4 * h;
            float* crow = P.C + (int64_t)(m0 + wm * WR + ta * 32 + (lane & 31)) * P.ldc + nb;
            const acc_t& a0 = acc[ta * SB + tb];
            const acc_t& ax = accx[X3 ? ta * SB + tb : 0];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                if (P.n_store && nb + 8 * g4 >= P.n_store) continue;        // columns of the padded operand planes that C does not have
                float4* dst = reinterpret_cast<float4*>(crow + 8 * g4);
                const float4 bias = P.bias ? *reinterpret_cast<const float4*>(P.bias + nb + 8 * g4) : make_float4(0.f, 0.f, 0.f, 0.f);
                float4 v;
                v.x = (X3 ? ax[4 * g4 + 0] + a0[4 * g4 + 0] : a0[4 * g4 + 0]) + bias.x;
                v.y = (X3 ? ax[4 * g4 + 1] + a0[4 * g4 + 1] : a0[4 * g4 + 1]) + bias.y;
                v.z = (X3 ? ax[4 * g4 + 2] + a0[4 * g4 + 2] : a0[4 * g4 + 2]) + bias.z;
                v.w = (X3 ? ax[4 * g4 + 3] + a0[4 * g4 + 3] : a0[4 * g4 + 3]) + bias.w;
                if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (gb.accumulate) { const float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                *dst = v;
            }
        }
    }
    if constexpr (STAMP) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(3);
        if (lane == 0 && blockIdx.x < 256 && wave_id < 4) {             // 32 words per workgroup: wave w (< 4) at [8 w .. 8 w + 7]
            unsigned long long* o = g16p_stamps + blockIdx.x * 32 + wave_id * 8;
            o[0] = st_t[0]; o[1] = st_t[1]; o[2] = st_t[2]; o[3] = st_t[3];
            o[4] = st_r[1]; o[5] = st_r[2]; o[6] = wacc_v; o[7] = wacc_b;
        }
    }
}
// workgroup id -> (problem, tile origin): XCD-local blocks when the launcher found the problems uniform, id order otherwise
template <int TN>
__device__ __forceinline__ bool g16p_tile(const Gemm16Batch& gb, int& pidx, int& m0, int& n0) {
    constexpr int SH = TN == 128 ? 7 : 6;
    if (gb.xcd_map) {
        const XcdTile xt = xcd_tile(blockIdx.x, gb.count, gb.p[0].M >> 7, gb.p[0].N >> SH);
        if (!xt.ok) return false;
        pidx = xt.p; m0 = xt.tm << 7; n0 = xt.tn << SH;
        return true;
    }
    pidx = blockIdx.z;
    const int tiles_n = gb.p[pidx].N >> SH, ntiles = tiles_n * (gb.p[pidx].M >> 7);
    if ((int)blockIdx.x >= ntiles) return false;
    m0 = ((int)blockIdx.x / tiles_n) << 7;
    n0 = ((int)blockIdx.x % tiles_n) << SH;
    return true;
}

template <bool AT, bool BT, bool X3, int TN, int KS = 32, int NSTG = 4, int MS = 32, bool STAMP = false, int ABL = 0, int SPREAD = 1, int STAG = 0>
__global__ __launch_bounds__(256) void gemm16p_kernel(const Gemm16Batch gb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_p[];
    int pidx, m0, n0;
    if (!g16p_tile<TN>(gb, pidx, m0, n0)) return;
    gemm16p_body<AT, BT, X3, TN, KS, NSTG, MS, false, STAMP, ABL, SPREAD, STAG>(gb, smem_p, pidx, m0, n0);
}

#ifdef EXORL_GEMM_EXPERIMENTS
template <bool X3, int TN, int NWV, int PFD, bool STAMP = false>      // NWV computing waves + one prefetch wave PFD stages ahead (forward launches)
__global__ __launch_bounds__(64 * (NWV + 1)) void gemm16pf_kernel(const Gemm16Batch gb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_p[];
    int pidx, m0, n0;
    if (!g16p_tile<TN>(gb, pidx, m0, n0)) return;
    gemm16p_body<false, false, X3, TN, 64, 2, 32, false, STAMP, 0, 2, 0, NWV, PFD>(gb, smem_p, pidx, m0, n0);
}
template <bool AT, bool BT, bool X3, int TN, bool STAMP = false>      // 8 waves per workgroup, all loading and computing; 64-wide stages x 2
__global__ __launch_bounds__(512) void gemm16p8_kernel(const Gemm16Batch gb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_p[];
    int pidx, m0, n0;
    if (!g16p_tile<TN>(gb, pidx, m0, n0)) return;
    gemm16p_body<AT, BT, X3, TN, 64, 2, 32, false, STAMP, 0, 2, 0, 8>(gb, smem_p, pidx, m0, n0);
}
template <bool AT, bool BT, bool X3, int TN>      // wave-specialised (512 threads: 4 MFMA waves + 4 loader waves), k32 stages x 4
__global__ __launch_bounds__(512) void gemm16w_kernel(const Gemm16Batch gb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_p[];
    int pidx, m0, n0;
    if (!g16p_tile<TN>(gb, pidx, m0, n0)) return;
    gemm16p_body<AT, BT, X3, TN, 32, 4, 32, true>(gb, smem_p, pidx, m0, n0);
}
template <bool X3, int TN>
__global__ __launch_bounds__(512) void gemm16w_mixed_kernel(const Gemm16Batch gb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_p[];
    int pidx, m0, n0;
    if (!g16p_tile<TN>(gb, pidx, m0, n0)) return;
    if (gb.a_t[pidx]) gemm16p_body<true, true, X3, TN, 32, 4, 32, true>(gb, smem_p, pidx, m0, n0);
    else gemm16p_body<false, true, X3, TN, 32, 4, 32, true>(gb, smem_p, pidx, m0, n0);
}
#endif

template <bool X3, int TN, int SPREAD = 1>      // wgrad (A as a k image) and dgrad (A as a row image) of one Linear(H,H) in one launch; B is a k image in both
__global__ __launch_bounds__(256) void gemm16p_mixed_kernel(const Gemm16Batch gb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_p[];
    int pidx, m0, n0;
    if (!g16p_tile<TN>(gb, pidx, m0, n0)) return;
    if (gb.a_t[pidx]) gemm16p_body<true, true, X3, TN, 32, 4, 32, false, false, 0, SPREAD>(gb, smem_p, pidx, m0, n0);
    else gemm16p_body<false, true, X3, TN, 32, 4, 32, false, false, 0, SPREAD>(gb, smem_p, pidx, m0, n0);
}
constexpr int g16p_lds(bool x3, int tn, int ks = 32, int nstg = 4) { return nstg * (x3 ? 2 : 1) * (2 + tn / 64) * 64 * 2 * ks; }

// Tile width for a launch (0 = these kernels do not apply): 128 x 128 when that still gives every CU a workgroup, else 128 x 64.
static int g16p_pick(const Gemm16Batch& gb, int count, bool x3) {
    if (g_gemm16_variant >= 0 && (g_gemm16_variant & 262144)) return 0;          // experiment switch: off
    int t128 = 0;
    for (int i = 0; i < count; ++i) {
        const Gemm16Problem& p = gb.p[i];
        const bool al = p.lda % 8 == 0 && p.ldb % 8 == 0 && reinterpret_cast<uintptr_t>(p.A) % 16 == 0 && reinterpret_cast<uintptr_t>(p.B) % 16 == 0 &&
                        (!x3 || (p.A_lo && p.B_lo && reinterpret_cast<uintptr_t>(p.A_lo) % 16 == 0 && reinterpret_cast<uintptr_t>(p.B_lo) % 16 == 0));
        const bool cal = p.ldc % 4 == 0 && reinterpret_cast<uintptr_t>(p.C) % 16 == 0 && reinterpret_cast<uintptr_t>(p.bias) % 16 == 0;     // float4 epilogue
        if (!al || !cal || p.M % 128 != 0 || p.N % 64 != 0 || p.K % 128 != 0 || p.K < 128) return 0;
        t128 += p.N % 128 == 0 ? (p.M >> 7) * (p.N >> 7) : 0;
    }
    bool n128 = true;
    for (int i = 0; i < count; ++i) n128 = n128 && gb.p[i].N % 128 == 0;
    if (g_gemm16_variant >= 0 && (g_gemm16_variant & 524288)) return n128 ? 128 : 64;      // experiment: 128 x 128 wherever it tiles
    return (n128 && t128 >= 256) ? 128 : 64;
}
static bool g16p_uniform(const Gemm16Batch& gb, int count, int tn) {       // xcd_tile()'s preconditions
    if (g_gemm16_variant >= 0 && (g_gemm16_variant & 1048576)) return false;     // experiment: id order
    if (!(count == 1 || count == 2 || count == 4)) return false;
    const int X = 8 / count, sm = X == 8 ? 4 : 2, sn = X == 2 ? 1 : 2;
    for (int i = 0; i < count; ++i) {
        if (gb.p[i].M != gb.p[0].M || gb.p[i].N != gb.p[0].N) return false;
        if ((gb.p[i].M / 128) % sm != 0 || (gb.p[i].N / tn) % sn != 0) return false;
    }
    return true;
}
template <typename K>
static int g16p_launch(K kernel, Gemm16Batch& gb, int count, bool x3, int tn, hipStream_t s, int ks = 32, int nstg = 4, int threads = 256) {
    const int lds = g16p_lds(x3, tn, ks, nstg);
    static std::vector<const void*> enabled;         // > 64 KB of dynamic LDS needs the opt-in, once per kernel
    if (std::find(enabled.begin(), enabled.end(), (const void*)kernel) == enabled.end()) {
        EXORL_CHECK_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        enabled.push_back((const void*)kernel);
    }
    gb.count = count;
    gb.xcd_map = g16p_uniform(gb, count, tn) ? 1 : 0;
    int tmax = 0, ttot = 0;
    for (int i = 0; i < count; ++i) { const int t = (gb.p[i].M >> 7) * (gb.p[i].N / tn); tmax = t > tmax ? t : tmax; ttot += t; }
    hipLaunchKernelGGL(kernel, gb.xcd_map ? dim3(ttot, 1, 1) : dim3(tmax, 1, count), dim3(threads), lds, s, gb);
    EXORL_LAUNCH_CHECK();
    return 0;
}

template <bool AT, bool BT, int NSTG = G16G_NSTG>
__global__ __launch_bounds__(256) void gemm16g_kernel(const Gemm16Batch gb) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[NSTG * 2 * G16G_IMG];
    gemm16g_body<AT, BT, NSTG>(gb, smem);
}

// split-bf16 operands (Gemm16Problem::A_lo / B_lo): 2 stages x 4 images = 64 KB of LDS, two workgroups per CU
template <bool AT, bool BT>
__global__ __launch_bounds__(256) void gemm16x3_kernel(const Gemm16Batch gb) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 4 * G16G_IMG];
    gemm16g_body<AT, BT, 2, true>(gb, smem);
}
__global__ __launch_bounds__(256) void gemm16x3_mixed_kernel(const Gemm16Batch gb) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 4 * G16G_IMG];
    const int p = gb.xcd_map ? (int)(blockIdx.x & 7) / (8 / gb.count) : (int)blockIdx.z;
    if (gb.a_t[p]) gemm16g_body<true, true, 2, true>(gb, smem);
    else gemm16g_body<false, true, 2, true>(gb, smem);
}

// Deep-pipeline variants (experiment, tuning bits 16384 / 32768): dynamic LDS up to the full 160 KB of a CU, to test whether the
// k-loop is bound by bytes in flight (in-flight bytes <= LDS bytes). It is not: 5 x 32 KB split-bf16 stages with one workgroup per CU
// run 35.8 us per launch against 24.6 us for 2 stages x 2 workgroups; plain bf16 10 stages x 1 workgroup 23.8 us against 13.7 us, and
// 5 stages x 2 workgroups 13.75 us (no change). The XCD-block mapping changes nothing either (23.3 vs 23.4 us), so it is not fabric
// traffic, and the LDS array is at ~half its rate (128 KB of ds_read_b128 at 256 B/clk + 64 KB of DMA writes per CU and k-tile pair
// = ~1000 of the ~2100 cycles). What is left is the operand delivery rate of a CU through the LDS-DMA path: 64 KB per 2100 cycles =
// 73 GB/s, the same ~75 GB/s the plain bf16 kernel sustains and the 68-90 GB/s per CU MI355X_MICROARCH.md lists for LDS-DMA fills
// ("ldsdma-fill", "ring-gemm"); it needs >= 2 workgroups per CU to be reached. Split-bf16 moves twice the bytes and takes twice the
// time. Fewer operand bytes per FLOP (64 x 32 / 64 x 64 wave tiles) is the lever, at the price of half as many workgroups on a
// batch-1024 problem (the 128 x 128 experiment above).
template <bool AT, bool BT, int NSTG, bool X3>
__global__ __launch_bounds__(256) void gemm16d_kernel(const Gemm16Batch gb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_d[];
    gemm16g_body<AT, BT, NSTG, X3>(gb, smem_d);
}
template <int NSTG, bool X3>
__global__ __launch_bounds__(256) void gemm16d_mixed_kernel(const Gemm16Batch gb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_d[];
    const int p = gb.xcd_map ? (int)(blockIdx.x & 7) / (8 / gb.count) : (int)blockIdx.z;
    if (gb.a_t[p]) gemm16g_body<true, true, NSTG, X3>(gb, smem_d);
    else gemm16g_body<false, true, NSTG, X3>(gb, smem_d);
}
template <int NSTG, bool X3>
static int g16d_enable() {
    static bool done = false;
    if (done) return 0;
    const int lds = NSTG * (X3 ? 4 : 2) * G16G_IMG;
    EXORL_CHECK_HIP(hipFuncSetAttribute((const void*)gemm16d_kernel<false, false, NSTG, X3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    EXORL_CHECK_HIP(hipFuncSetAttribute((const void*)gemm16d_kernel<false, true, NSTG, X3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    EXORL_CHECK_HIP(hipFuncSetAttribute((const void*)gemm16d_kernel<true, true, NSTG, X3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    EXORL_CHECK_HIP(hipFuncSetAttribute((const void*)gemm16d_mixed_kernel<NSTG, X3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    done = true;
    return 0;
}
constexpr int G16D_X3_STAGES = 5;      // 5 x 32 KB = 160 KB: one workgroup per CU, four k-tiles (128 KB) in flight

// One launch for the wgrad and dgrad GEMMs of a Linear(H,H) backward: both read dZ (wgrad as a k image, dgrad as a row image)
// and are independent, so 2 x 512 tiles fill the 256 CUs four deep instead of two launches two deep, and one kernel boundary
// (~4.5 us of drain + cache write-back + ramp on this part) disappears. B is a k image in both.
__global__ __launch_bounds__(256) void gemm16g_mixed_kernel(const Gemm16Batch gb) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[G16G_NSTG * 2 * G16G_IMG];
    const int p = gb.xcd_map ? (int)(blockIdx.x & 7) / (8 / gb.count) : (int)blockIdx.z;
    if (gb.a_t[p]) gemm16g_body<true, true>(gb, smem);
    else gemm16g_body<false, true>(gb, smem);
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
    // variant bits (tuning): 1 = force BM 64, 2 = force BM 128, 8 = no XCD swizzle, 16 = keep guards,
    // 32 = 2 register stages, 64 = 4 register stages (default 3)
    const int var = g_gemm16_variant < 0 ? 0 : g_gemm16_variant;
    const bool big = (var & 2) ? true : ((var & 1) ? false : tiles128 * count >= 256);
    Gemm16Batch g2 = gb;
    g2.swizzle = (var & 8) ? 0 : 1;
    bool exact = true;      // every problem tiles exactly: loads need no bounds guards
    for (int i = 0; i < count; ++i)
        exact = exact && gb.p[i].M % 128 == 0 && gb.p[i].N % 64 == 0 && gb.p[i].K % 64 == 0;
    if (var & 16) exact = false;
    bool exact64 = true;
    for (int i = 0; i < count; ++i)
        exact64 = exact64 && gb.p[i].M % 64 == 0 && gb.p[i].N % 64 == 0 && gb.p[i].K % 256 == 0 && gb.p[i].lda % 8 == 0 &&
                  gb.p[i].ldb % 8 == 0 && reinterpret_cast<uintptr_t>(gb.p[i].A) % 16 == 0 && reinterpret_cast<uintptr_t>(gb.p[i].B) % 16 == 0;
    bool x3 = false;
    for (int i = 0; i < count; ++i) x3 = x3 || gb.p[i].A_lo || gb.p[i].B_lo;
    if (const int tn = g16p_pick(g2, count, x3)) {          // 128 x TN tiles, k32 stages, XCD-local blocks (see gemm16p_body)
        // both operands row images (the forward launches): 64-wide stages, two deep — whole cache lines per DMA row instead of halves, which
        // halves the requests the XCD L2s serve (critic fwd 21.8 -> 19.5 us, actor fwd 21.1 -> 18.9, critic+target fwd 33.7 -> 32.4); bit
        // 2097152 of the tuning variant switches back to the 32-wide stages
        const bool k64 = AL == 0 && BL == 0 && !(g_gemm16_variant >= 0 && (g_gemm16_variant & 2097152));
        const int var_ = g_gemm16_variant < 0 ? 0 : g_gemm16_variant;
        const bool sp1 = (var_ & 536870912) != 0;            // the previous schedule: a stage's refill issued in one region (SPREAD = 1)
        const bool stamped = (var_ & 33554432) != 0;        // diagnostic build with in-kernel clock stamps (tools/micro/stamp_bench.py)
        bool done = false;
#ifdef EXORL_GEMM_EXPERIMENTS      // measured and not adopted ("what was tried on the GEMM"): kept reproducible, not in the default build
        if (x3 && !done) {
            done = true;
            if (k64 && (var_ & 134217728) && !(var_ & 67108864) && !stamped) {        // + a prefetch wave (bit 27): 8 + 1 waves on 128 x 64 tiles (with bit
                if constexpr (AL == 0 && BL == 0) {                                  // 28), 4 + 1 otherwise; bit 30 = 10 instead of 6 stages ahead
                    const bool far = (var_ & 1073741824) != 0;
                    if (tn == 64 && (var_ & 268435456) && far) EXORL_TRY(g16p_launch(gemm16pf_kernel<true, 64, 8, 10>, g2, count, true, 64, s, 64, 2, 576));
                    else if (tn == 64 && (var_ & 268435456)) EXORL_TRY(g16p_launch(gemm16pf_kernel<true, 64, 8, 6>, g2, count, true, 64, s, 64, 2, 576));
                    else if (tn == 64 && far) EXORL_TRY(g16p_launch(gemm16pf_kernel<true, 64, 4, 10>, g2, count, true, 64, s, 64, 2, 320));
                    else if (tn == 64) EXORL_TRY(g16p_launch(gemm16pf_kernel<true, 64, 4, 6>, g2, count, true, 64, s, 64, 2, 320));
                    else if (var_ & 268435456) EXORL_TRY(g16p_launch(gemm16p8_kernel<false, false, true, 128>, g2, count, true, 128, s, 64, 2, 512));
                    else EXORL_TRY(g16p_launch(gemm16p_kernel<false, false, true, 128, 64, 2, 32, false, 0, 2>, g2, count, true, 128, s, 64, 2));
                }
            } else if (k64 && (var_ & 268435456)) {        // 8 waves per workgroup, all loading and computing (forward launches)
                if constexpr (AL == 0 && BL == 0) {
                    if (tn == 128 && stamped) EXORL_TRY(g16p_launch(gemm16p8_kernel<false, false, true, 128, true>, g2, count, true, 128, s, 64, 2, 512));
                    else if (tn == 128) EXORL_TRY(g16p_launch(gemm16p8_kernel<false, false, true, 128>, g2, count, true, 128, s, 64, 2, 512));
                    else if (stamped) EXORL_TRY(g16p_launch(gemm16p8_kernel<false, false, true, 64, true>, g2, count, true, 64, s, 64, 2, 512));
                    else EXORL_TRY(g16p_launch(gemm16p8_kernel<false, false, true, 64>, g2, count, true, 64, s, 64, 2, 512));
                }
            } else if (var_ & 8388608) {          // wave-specialised workgroups
                if (tn == 128) EXORL_TRY(g16p_launch(gemm16w_kernel<AL != 0, BL != 0, true, 128>, g2, count, true, 128, s, 32, 4, 512));
                else EXORL_TRY(g16p_launch(gemm16w_kernel<AL != 0, BL != 0, true, 64>, g2, count, true, 64, s, 32, 4, 512));
            } else if (k64 && (var_ & 4194304) && !(var_ & 1073741824)) {          // 16 x 16 x 32 MFMAs
                if constexpr (AL == 0 && BL == 0) {
                    if (tn == 128) EXORL_TRY(g16p_launch(gemm16p_kernel<false, false, true, 128, 64, 2, 16>, g2, count, true, 128, s, 64, 2));
                    else EXORL_TRY(g16p_launch(gemm16p_kernel<false, false, true, 64, 64, 2, 16>, g2, count, true, 64, s, 64, 2));
                }
            } else if (k64 && (var_ & 1073741824)) {                              // per-wave DMA placement (2 phases; 4 with bit 4194304)
                if constexpr (AL == 0 && BL == 0) {
                    if (tn == 128 && (var_ & 4194304)) EXORL_TRY(g16p_launch(gemm16p_kernel<false, false, true, 128, 64, 2, 32, false, 0, 2, 4>, g2, count, true, 128, s, 64, 2));
                    else if (tn == 128) EXORL_TRY(g16p_launch(gemm16p_kernel<false, false, true, 128, 64, 2, 32, false, 0, 2, 2>, g2, count, true, 128, s, 64, 2));
                    else EXORL_TRY(g16p_launch(gemm16p_kernel<false, false, true, 64, 64, 2, 32, false, 0, 2, 2>, g2, count, true, 64, s, 64, 2));
                }
            } else done = false;
        }
#endif
        if (done) {
        } else if (x3 && k64) {
            if constexpr (AL == 0 && BL == 0) {
                if (stamped) {
                    const int abl = (var_ >> 26) & 3;
                    if (tn == 128 && abl == 1) EXORL_TRY(g16p_launch(gemm16p_kernel<false, false, true, 128, 64, 2, 32, true, 1>, g2, count, true, 128, s, 64, 2));
                    else if (tn == 128 && abl == 2) EXORL_TRY(g16p_launch(gemm16p_kernel<false, false, true, 128, 64, 2, 32, true, 2>, g2, count, true, 128, s, 64, 2));
                    else if (tn == 128 && abl == 3) EXORL_TRY(g16p_launch(gemm16p_kernel<false, false, true, 128, 64, 2, 32, true, 3>, g2, count, true, 128, s, 64, 2));
                    else if (tn == 128 && sp1) EXORL_TRY(g16p_launch(gemm16p_kernel<false, false, true, 128, 64, 2, 32, true>, g2, count, true, 128, s, 64, 2));
                    else if (tn == 128) EXORL_TRY(g16p_launch(gemm16p_kernel<false, false, true, 128, 64, 2, 32, true, 0, 2>, g2, count, true, 128, s, 64, 2));
                    else EXORL_TRY(g16p_launch(gemm16p_kernel<false, false, true, 64, 64, 2, 32, true, 0, 2>, g2, count, true, 64, s, 64, 2));
                } else if (sp1) {
                    if (tn == 128) EXORL_TRY(g16p_launch(gemm16p_kernel<false, false, true, 128, 64, 2>, g2, count, true, 128, s, 64, 2));
                    else EXORL_TRY(g16p_launch(gemm16p_kernel<false, false, true, 64, 64, 2>, g2, count, true, 64, s, 64, 2));
                } else if (tn == 128) EXORL_TRY(g16p_launch(gemm16p_kernel<false, false, true, 128, 64, 2, 32, false, 0, 2>, g2, count, true, 128, s, 64, 2));
                else EXORL_TRY(g16p_launch(gemm16p_kernel<false, false, true, 64, 64, 2, 32, false, 0, 2>, g2, count, true, 64, s, 64, 2));
            }
        } else if (x3 && stamped) {
            if (tn == 128) EXORL_TRY(g16p_launch(gemm16p_kernel<AL != 0, BL != 0, true, 128, 32, 4, 32, true, 0, 2>, g2, count, true, 128, s));
            else EXORL_TRY(g16p_launch(gemm16p_kernel<AL != 0, BL != 0, true, 64, 32, 4, 32, true, 0, 2>, g2, count, true, 64, s));
        } else if (x3 && sp1) {
            if (tn == 128) EXORL_TRY(g16p_launch(gemm16p_kernel<AL != 0, BL != 0, true, 128>, g2, count, true, 128, s));
            else EXORL_TRY(g16p_launch(gemm16p_kernel<AL != 0, BL != 0, true, 64>, g2, count, true, 64, s));
        } else if (x3) {
            if (tn == 128) EXORL_TRY(g16p_launch(gemm16p_kernel<AL != 0, BL != 0, true, 128, 32, 4, 32, false, 0, 2>, g2, count, true, 128, s));
            else EXORL_TRY(g16p_launch(gemm16p_kernel<AL != 0, BL != 0, true, 64, 32, 4, 32, false, 0, 2>, g2, count, true, 64, s));
        } else if (sp1) {
            if (tn == 128) EXORL_TRY(g16p_launch(gemm16p_kernel<AL != 0, BL != 0, false, 128>, g2, count, false, 128, s));
            else EXORL_TRY(g16p_launch(gemm16p_kernel<AL != 0, BL != 0, false, 64>, g2, count, false, 64, s));
        } else {
            // plain bf16 planes take the two-region refill as well (round 3). Round 2 kept them on SPREAD = 1 because the k-image B operand came out
            // wrong, run-to-run different, under SPREAD = 2: that was the pre-fence register copy described in region() — a compiler-placed v_mov of
            // an asm-issued LDS read's destination — not the schedule (tests/test_gpu_ops.py::test_gemm_plain_bf16_k_image_regression).
            if (tn == 128) EXORL_TRY(g16p_launch(gemm16p_kernel<AL != 0, BL != 0, false, 128, 32, 4, 32, false, 0, 2>, g2, count, false, 128, s));
            else EXORL_TRY(g16p_launch(gemm16p_kernel<AL != 0, BL != 0, false, 64, 32, 4, 32, false, 0, 2>, g2, count, false, 64, s));
        }
        if (prof) {
            EXORL_CHECK_HIP(hipEventRecord(g_prof.ev[2 * g_prof.used + 1], s));
            g_prof.used += 1;
        }
        return 0;
    }
    if (x3) {
        bool okx = true;
        for (int i = 0; i < count; ++i)
            okx = okx && gb.p[i].A_lo && gb.p[i].B_lo && gb.p[i].M % 64 == 0 && gb.p[i].N % 64 == 0 && gb.p[i].K % 64 == 0 && gb.p[i].lda % 8 == 0 &&
                  gb.p[i].ldb % 8 == 0 && reinterpret_cast<uintptr_t>(gb.p[i].A_lo) % 16 == 0 && reinterpret_cast<uintptr_t>(gb.p[i].B_lo) % 16 == 0 &&
                  reinterpret_cast<uintptr_t>(gb.p[i].A) % 16 == 0 && reinterpret_cast<uintptr_t>(gb.p[i].B) % 16 == 0;
        EXORL_REQUIRE(okx, "gemm16_grouped: split-bf16 operands need M, N, K multiples of 64 and 16-byte aligned hi/lo planes");
        g2.count = count;
        g2.xcd_map = ((var & 2048) && xcd_map_ok(g2, count, 64)) ? 1 : 0;
        if (g16hx3_fits(g2, count)) {
            EXORL_TRY(g16hx3_enable());
            int t128 = 0;
            for (int i = 0; i < count; ++i) { const int t = (gb.p[i].M >> 7) * (gb.p[i].N >> 7); t128 = t > t128 ? t : t128; }
            g2.xcd_map = 0;
            hipLaunchKernelGGL((gemm16hx3_kernel<AL != 0, BL != 0>), dim3(t128, 1, count), dim3(256), G16HX3_LDS, s, g2);
        } else if (var & 16384) {
            EXORL_TRY((g16d_enable<G16D_X3_STAGES, true>()));
            hipLaunchKernelGGL((gemm16d_kernel<AL != 0, BL != 0, G16D_X3_STAGES, true>), g2.xcd_map ? dim3(tiles64 * count, 1, 1) : dim3(tiles64, 1, count),
                               dim3(256), G16D_X3_STAGES * 4 * G16G_IMG, s, g2);
        } else if (g2.xcd_map) hipLaunchKernelGGL((gemm16x3_kernel<AL != 0, BL != 0>), dim3(tiles64 * count, 1, 1), dim3(256), 0, s, g2);
        else hipLaunchKernelGGL((gemm16x3_kernel<AL != 0, BL != 0>), dim3(tiles64, 1, count), dim3(256), 0, s, g2);
        EXORL_LAUNCH_CHECK();
        if (prof) {
            EXORL_CHECK_HIP(hipEventRecord(g_prof.ev[2 * g_prof.used + 1], s));
            g_prof.used += 1;
        }
        return 0;
    }
    if (exact64 && !(var & 128)) {         // LDS-DMA pipeline (bit 128 of the tuning variant forces the register-staged kernels)
        if (g16h_fits(g2, count)) {
            EXORL_TRY(g16h_enable());
            int t128 = 0;
            for (int i = 0; i < count; ++i) { const int t = (gb.p[i].M >> 7) * (gb.p[i].N >> 7); t128 = t > t128 ? t : t128; }
            if (g16h_stages() == 2) hipLaunchKernelGGL((gemm16h_kernel<AL != 0, BL != 0, 2>), dim3(t128, 1, count), dim3(256), 2 * G16H_STAGE, s, g2);
            else hipLaunchKernelGGL((gemm16h_kernel<AL != 0, BL != 0, 4>), dim3(t128, 1, count), dim3(256), G16H_LDS, s, g2);
        } else {
            g2.count = count;
            g2.xcd_map = ((var & 2048) && xcd_map_ok(g2, count, 64)) ? 1 : 0;     // measured: no gain over id order (12.8 vs 12.5 us) -> opt-in
            if (var & 16384) {            // 10 x 16 KB = 160 KB, one workgroup per CU
                EXORL_TRY((g16d_enable<10, false>()));
                hipLaunchKernelGGL((gemm16d_kernel<AL != 0, BL != 0, 10, false>), g2.xcd_map ? dim3(tiles64 * count, 1, 1) : dim3(tiles64, 1, count),
                                   dim3(256), 10 * 2 * G16G_IMG, s, g2);
            } else if (var & 32768) {     // 5 x 16 KB = 80 KB, two workgroups per CU
                EXORL_TRY((g16d_enable<5, false>()));
                hipLaunchKernelGGL((gemm16d_kernel<AL != 0, BL != 0, 5, false>), g2.xcd_map ? dim3(tiles64 * count, 1, 1) : dim3(tiles64, 1, count),
                                   dim3(256), 5 * 2 * G16G_IMG, s, g2);
            } else if (g2.xcd_map) hipLaunchKernelGGL((gemm16g_kernel<AL != 0, BL != 0>), dim3(tiles64 * count, 1, 1), dim3(256), 0, s, g2);
            else if (var & 4096) hipLaunchKernelGGL((gemm16g_kernel<AL != 0, BL != 0, 3>), dim3(tiles64, 1, count), dim3(256), 0, s, g2);   // 3 WGs/CU
            else if (var & 8192) hipLaunchKernelGGL((gemm16g_kernel<AL != 0, BL != 0, 2>), dim3(tiles64, 1, count), dim3(256), 0, s, g2);   // 5 WGs/CU
            else hipLaunchKernelGGL((gemm16g_kernel<AL != 0, BL != 0>), dim3(tiles64, 1, count), dim3(256), 0, s, g2);
        }
        EXORL_LAUNCH_CHECK();
        if (prof) {
            EXORL_CHECK_HIP(hipEventRecord(g_prof.ev[2 * g_prof.used + 1], s));
            g_prof.used += 1;
        }
        return 0;
    }
    const int ns = (var & 32) ? 2 : ((var & 64) ? 4 : 3);
#define EXORL_G16(BMv, NSv, Gv) hipLaunchKernelGGL((gemm16_kernel<AL, BL, BMv, NSv, Gv>), dim3(big ? tiles128 : tiles64, 1, count), dim3(256), 0, s, g2)
    if (exact) {
        if (big) { if (ns == 2) EXORL_G16(128, 2, false); else if (ns == 3) EXORL_G16(128, 3, false); else EXORL_G16(128, 4, false); }
        else     { if (ns == 2) EXORL_G16(64, 2, false); else if (ns == 3) EXORL_G16(64, 3, false); else EXORL_G16(64, 4, false); }
    } else {
        if (big) EXORL_G16(128, 2, true); else EXORL_G16(64, 2, true);
    }
#undef EXORL_G16
    EXORL_LAUNCH_CHECK();
    if (prof) {
        EXORL_CHECK_HIP(hipEventRecord(g_prof.ev[2 * g_prof.used + 1], s));
        g_prof.used += 1;
    }
    return 0;
}

// probs[i] with a_layouts[i] in {0,1}, b_layout 1 for all, no bias/relu/accumulate. Falls back to one launch per layout when a
// problem does not meet the LDS-DMA kernel's tiling rules.
int gemm16_grouped_mixed(const int* a_layouts, const Gemm16Problem* probs, int count, hipStream_t s) {
    EXORL_REQUIRE(count >= 1 && count <= 4, "gemm16_grouped_mixed: count %d out of range", count);
    Gemm16Batch gb;
    memset(&gb, 0, sizeof(gb));
    bool ok = g_gemm16_variant < 0 || !(g_gemm16_variant & 128);
    int t64 = 0;
    double flops = 0;
    for (int i = 0; i < count; ++i) {
        const Gemm16Problem& p = probs[i];
        gb.p[i] = p;
        gb.a_t[i] = a_layouts[i] != 0;
        ok = ok && p.M > 0 && p.M % 64 == 0 && p.N % 64 == 0 && p.K % 256 == 0 && p.lda % 8 == 0 && p.ldb % 8 == 0 && !p.bias &&
             reinterpret_cast<uintptr_t>(p.A) % 16 == 0 && reinterpret_cast<uintptr_t>(p.B) % 16 == 0;
        const int a64 = cdiv(p.M, 64) * cdiv(p.N, 64);
        t64 = a64 > t64 ? a64 : t64;
        flops += 2.0 * p.M * (double)p.N * p.K;
    }
    bool x3 = false;
    for (int i = 0; i < count; ++i) x3 = x3 || probs[i].A_lo;
    if (!ok || (x3 && count > 0 && !probs[0].B_lo)) {
        for (int i = 0; i < count; ++i) EXORL_TRY(gemm16_grouped(a_layouts[i], 1, probs + i, 1, false, false, s));
        return 0;
    }
    gb.swizzle = (g_gemm16_variant >= 0 && (g_gemm16_variant & 8)) ? 0 : 1;
    const bool prof = g_prof.on && g_prof.used < PROF_MAX_LAUNCHES;
    if (prof) {
        if (g_prof.ev.size() < 2 * (g_prof.used + 1)) {
            hipEvent_t a, b;
            EXORL_CHECK_HIP(hipEventCreate(&a));
            EXORL_CHECK_HIP(hipEventCreate(&b));
            g_prof.ev.push_back(a);
            g_prof.ev.push_back(b);
        }
        g_prof.flops.push_back(flops);
        EXORL_CHECK_HIP(hipEventRecord(g_prof.ev[2 * g_prof.used], s));
    }
    if (const int tn = g16p_pick(gb, count, x3)) {
        const int var_ = g_gemm16_variant < 0 ? 0 : g_gemm16_variant;
        const bool sp1 = (var_ & 536870912) != 0;            // the previous schedule (SPREAD = 1)
        bool done = false;
#ifdef EXORL_GEMM_EXPERIMENTS
        if (x3 && (var_ & 8388608)) {
            done = true;
            if (tn == 128) EXORL_TRY(g16p_launch(gemm16w_mixed_kernel<true, 128>, gb, count, true, 128, s, 32, 4, 512));
            else EXORL_TRY(g16p_launch(gemm16w_mixed_kernel<true, 64>, gb, count, true, 64, s, 32, 4, 512));
        }
#endif
        if (done) {
        } else if (x3 && sp1) {
            if (tn == 128) EXORL_TRY(g16p_launch(gemm16p_mixed_kernel<true, 128>, gb, count, true, 128, s));
            else EXORL_TRY(g16p_launch(gemm16p_mixed_kernel<true, 64>, gb, count, true, 64, s));
        } else if (x3) {
            if (tn == 128) EXORL_TRY(g16p_launch(gemm16p_mixed_kernel<true, 128, 2>, gb, count, true, 128, s));
            else EXORL_TRY(g16p_launch(gemm16p_mixed_kernel<true, 64, 2>, gb, count, true, 64, s));
        } else if (sp1) {
            if (tn == 128) EXORL_TRY(g16p_launch(gemm16p_mixed_kernel<false, 128>, gb, count, false, 128, s));
            else EXORL_TRY(g16p_launch(gemm16p_mixed_kernel<false, 64>, gb, count, false, 64, s));
        } else {
            if (tn == 128) EXORL_TRY(g16p_launch(gemm16p_mixed_kernel<false, 128, 2>, gb, count, false, 128, s));
            else EXORL_TRY(g16p_launch(gemm16p_mixed_kernel<false, 64, 2>, gb, count, false, 64, s));
        }
    } else if (x3) {
        gb.count = count;
        gb.xcd_map = (g_gemm16_variant >= 0 && (g_gemm16_variant & 2048) && xcd_map_ok(gb, count, 64)) ? 1 : 0;
        if (g16hx3_fits(gb, count)) {
            EXORL_TRY(g16hx3_enable());
            int t128 = 0;
            for (int i = 0; i < count; ++i) { const int t = (gb.p[i].M >> 7) * (gb.p[i].N >> 7); t128 = t > t128 ? t : t128; }
            gb.xcd_map = 0;
            hipLaunchKernelGGL(gemm16hx3_mixed_kernel, dim3(t128, 1, count), dim3(256), G16HX3_LDS, s, gb);
        } else if (g_gemm16_variant >= 0 && (g_gemm16_variant & 16384)) {
            EXORL_TRY((g16d_enable<G16D_X3_STAGES, true>()));
            hipLaunchKernelGGL((gemm16d_mixed_kernel<G16D_X3_STAGES, true>), gb.xcd_map ? dim3(t64 * count, 1, 1) : dim3(t64, 1, count), dim3(256),
                               G16D_X3_STAGES * 4 * G16G_IMG, s, gb);
        } else if (gb.xcd_map) hipLaunchKernelGGL(gemm16x3_mixed_kernel, dim3(t64 * count, 1, 1), dim3(256), 0, s, gb);
        else hipLaunchKernelGGL(gemm16x3_mixed_kernel, dim3(t64, 1, count), dim3(256), 0, s, gb);
    } else if (g16h_fits(gb, count)) {
        EXORL_TRY(g16h_enable());
        int t128 = 0;
        for (int i = 0; i < count; ++i) { const int t = (gb.p[i].M >> 7) * (gb.p[i].N >> 7); t128 = t > t128 ? t : t128; }
        if (g16h_stages() == 2) hipLaunchKernelGGL(gemm16h_mixed_kernel<2>, dim3(t128, 1, count), dim3(256), 2 * G16H_STAGE, s, gb);
        else hipLaunchKernelGGL(gemm16h_mixed_kernel<4>, dim3(t128, 1, count), dim3(256), G16H_LDS, s, gb);
    } else {
        gb.count = count;
        gb.xcd_map = (g_gemm16_variant >= 0 && (g_gemm16_variant & 2048) && xcd_map_ok(gb, count, 64)) ? 1 : 0;
        if (g_gemm16_variant >= 0 && (g_gemm16_variant & 16384)) {
            EXORL_TRY((g16d_enable<10, false>()));
            hipLaunchKernelGGL((gemm16d_mixed_kernel<10, false>), gb.xcd_map ? dim3(t64 * count, 1, 1) : dim3(t64, 1, count), dim3(256), 10 * 2 * G16G_IMG, s, gb);
        } else if (g_gemm16_variant >= 0 && (g_gemm16_variant & 32768)) {
            EXORL_TRY((g16d_enable<5, false>()));
            hipLaunchKernelGGL((gemm16d_mixed_kernel<5, false>), gb.xcd_map ? dim3(t64 * count, 1, 1) : dim3(t64, 1, count), dim3(256), 5 * 2 * G16G_IMG, s, gb);
        } else if (gb.xcd_map) hipLaunchKernelGGL(gemm16g_mixed_kernel, dim3(t64 * count, 1, 1), dim3(256), 0, s, gb);
        else hipLaunchKernelGGL(gemm16g_mixed_kernel, dim3(t64, 1, count), dim3(256), 0, s, gb);
    }
    EXORL_LAUNCH_CHECK();
    if (prof) {
        EXORL_CHECK_HIP(hipEventRecord(g_prof.ev[2 * g_prof.used + 1], s));
        g_prof.used += 1;
    }
    return 0;
}

// bf16-in-memory operands, fp32 output. Requirements (checked): 16-byte aligned rows for layout 0 (ld % 8 == 0,
// K % 8 == 0), 4-byte aligned pairs for layout 1 (ld % 2 == 0, R % 2 == 0).
int gemm16_head_slots(const Gemm16Problem* probs, int count) {
    if (count < 1 || count > 4) return 0;
    Gemm16Batch gb;
    memset(&gb, 0, sizeof(gb));
    bool x3 = true;
    for (int i = 0; i < count; ++i) { gb.p[i] = probs[i]; x3 = x3 && probs[i].A_lo && probs[i].B_lo; }
    const int tn = g16p_pick(gb, count, x3);
    if (!tn) return 0;
    for (int i = 0; i < count; ++i)
        if (probs[i].N != probs[0].N) return 0;
    return probs[0].N / (tn / 2);               // four waves: 2 x 2, each TN / 2 columns wide
}

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
    for (int i = 0; i < count; ++i)
        if (probs[i].head_part)
            EXORL_REQUIRE(probs[i].head_w && a_layout == 0 && b_layout == 0 && !accumulate && gemm16_head_slots(probs, count) > 0 &&
                          !(g_gemm16_variant >= 0 && (g_gemm16_variant & 4194304)),
                          "gemm16_grouped: a folded head needs a forward launch on the 128 x TN kernels (ask gemm16_head_slots first)");
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
    constexpr size_t dyn = PREC == EXORL_PREC_BF16X6 ? 6 * TILEB : 0;          // 48 KB: one stage of three planes of A and B
    if constexpr (PREC == EXORL_PREC_BF16X6) {
        static bool attr = false;
        if (!attr) {
            EXORL_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_kernel<PREC, AL, BL, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
            EXORL_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_kernel<PREC, AL, BL, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
            attr = true;
        }
    }
    if (vec) hipLaunchKernelGGL((gemm_kernel<PREC, AL, BL, true>), grid, block, dyn, s, gb);
    else     hipLaunchKernelGGL((gemm_kernel<PREC, AL, BL, false>), grid, block, dyn, s, gb);
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

// ---- fp32-source operands onto the hi/lo-plane kernels (round 3) ------------------------------------------------------------------------
// The callers of gemm_grouped keep fp32 activations and weights (intrinsic modules, the pixel agents' heads and 39200-wide module layers); the
// generic kernel above splits them into hi/lo planes on their way into LDS — 64 x 64 tiles, register staging: ~62 TFLOP/s at 1024^3. The plane
// kernels (gemm16p_*) run the SAME arithmetic (hi*hi + hi*lo + lo*hi, three fp32 accumulators summed small-first) at 230-300 TFLOP/s but want
// bf16 planes in memory, M % 128 = 0, N % 64 = 0, K % 128 = 0. For problems large enough to pay for it this adapter writes zero-padded hi/lo
// planes of both operands into a scratch arena (one elementwise pass each: 8 B per element moved), runs the plane kernels, and — when C does
// not tile — copies the padded result back (bias, ReLU in the GEMM epilogue as always; accumulate in the copy). Groups of more than four
// problems (Disagreement's five models) go in chunks. exorl_gemm_tune bit 65536 switches it off (A/B).
struct PlaneArena { unsigned char* buf = nullptr; size_t bytes = 0; };
static PlaneArena g_plane_arena;

__global__ __launch_bounds__(256) void to_planes_kernel(const float* __restrict__ src, int64_t ld, int rows, int cols, unsigned short* __restrict__ hi,
                                                        unsigned short* __restrict__ lo, int rows_p, int cols_p) {
    // 8 consecutive columns per thread: two 16-byte loads (when aligned and in range), one 16-byte store per plane; padding is written as zeros
    const int64_t groups = (int64_t)rows_p * (cols_p / 8);
    const bool vec = (ld % 4 == 0) && (reinterpret_cast<uintptr_t>(src) % 16 == 0);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < groups; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / (cols_p / 8)), c0 = (int)(i % (cols_p / 8)) * 8;
        float v[8];
        if (r < rows && vec && c0 + 8 <= cols) {
            const float4 a = *reinterpret_cast<const float4*>(src + (int64_t)r * ld + c0), b = *reinterpret_cast<const float4*>(src + (int64_t)r * ld + c0 + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (r < rows && c0 + j < cols) ? src[(int64_t)r * ld + c0 + j] : 0.f;
        }
        uint4 h, l;
        h.x = pack_bf16(v[0], v[1]); h.y = pack_bf16(v[2], v[3]); h.z = pack_bf16(v[4], v[5]); h.w = pack_bf16(v[6], v[7]);
        l.x = pack_bf16(bf16_residual(v[0]), bf16_residual(v[1])); l.y = pack_bf16(bf16_residual(v[2]), bf16_residual(v[3]));
        l.z = pack_bf16(bf16_residual(v[4]), bf16_residual(v[5])); l.w = pack_bf16(bf16_residual(v[6]), bf16_residual(v[7]));
        *reinterpret_cast<uint4*>(hi + (int64_t)r * cols_p + c0) = h;
        *reinterpret_cast<uint4*>(lo + (int64_t)r * cols_p + c0) = l;
    }
}
// C[m][n] (+)= Cp[m][n] for the unpadded block
__global__ __launch_bounds__(256) void from_padded_kernel(const float* __restrict__ cp, int64_t ldp, float* __restrict__ c, int64_t ldc, int M, int N, int accumulate) {
    const int64_t n = (int64_t)M * N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / N), j = (int)(i % N);
        const float v = cp[(int64_t)m * ldp + j];
        float* d = c + (int64_t)m * ldc + j;
        *d = accumulate ? v + *d : v;
    }
}
__global__ void pad_bias_kernel(const float* __restrict__ b, float* __restrict__ out, int N, int Np) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Np) out[i] = i < N ? b[i] : 0.f;
}

static bool planes_adapter_wants(int al, int bl, const GemmProblem* probs, int count) {
    if (g_gemm16_variant >= 0 && (g_gemm16_variant & (65536 | 262144))) return false;
    if (!((al == 0 && bl == 0) || (al == 0 && bl == 1) || (al == 1 && bl == 1))) return false;
    for (int i = 0; i < count; ++i)
        if (probs[i].M < 256 || probs[i].N < 256 || probs[i].K < 256) return false;       // conversion passes + padded tiles must be worth it
    return true;
}

// 0 = done; -1 = not taken (the caller runs the generic kernel); > 0 = error
// C written in place by the plane kernels: whole row tiles, 16-byte rows and bias; a width that is not a multiple of the 128-column tile is
// handled by the epilogue's column guard (n_store) as long as it is a multiple of 4 (round 3: the padded copy + from_padded_kernel pass of the
// 39200-wide module layers was 0.7 ms of an ICM update on pixels and 2.0 ms of a Disagreement update; exorl_gemm_tune bit 134217728 brings it back)
static bool adapter_direct(const GemmProblem& p) {
    const bool wide_ok = p.N % 128 == 0 || (p.N % 4 == 0 && !(tune_variant() & 134217728));
    return p.M % 128 == 0 && wide_ok && p.ldc % 4 == 0 && reinterpret_cast<uintptr_t>(p.C) % 16 == 0 &&
           (!p.bias || reinterpret_cast<uintptr_t>(p.bias) % 16 == 0);
}
static int planes_adapter(int al, int bl, const GemmProblem* probs, int count, bool relu, bool accumulate, hipStream_t s) {
    struct Plan { int Mp, Np, Kp; size_t a_hi, a_lo, b_hi, b_lo, cp, bias; bool direct; };
    {   // never inside a stream capture: the arena may be re-allocated later, a captured graph would keep the old addresses
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        EXORL_CHECK_HIP(hipStreamIsCapturing(s, &cs));
        if (cs != hipStreamCaptureStatusNone) return -1;
    }
    for (int c0 = 0; c0 < count; c0 += 4) {
        const int nc = count - c0 < 4 ? count - c0 : 4;
        Plan pl[4];
        size_t need = 0;
        auto take = [&](size_t bytes) { const size_t o = need; need += (bytes + 255) & ~(size_t)255; return o; };
        for (int i = 0; i < nc; ++i) {
            const GemmProblem& p = probs[c0 + i];
            Plan& q = pl[i];
            q.Mp = (int)round_up(p.M, 128); q.Np = (int)round_up(p.N, 128); q.Kp = (int)round_up(p.K, 128);
            q.a_hi = take((size_t)q.Mp * q.Kp * 2); q.a_lo = take((size_t)q.Mp * q.Kp * 2);
            q.b_hi = take((size_t)q.Np * q.Kp * 2); q.b_lo = take((size_t)q.Np * q.Kp * 2);
            q.direct = adapter_direct(p);
            q.cp = q.direct ? 0 : take((size_t)q.Mp * q.Np * 4);
            q.bias = (q.direct || !p.bias) ? 0 : take((size_t)q.Np * 4);
        }
        if (need > g_plane_arena.bytes) {
            EXORL_CHECK_HIP(hipDeviceSynchronize());                                // nothing in flight may still read the old arena
            if (g_plane_arena.buf) EXORL_CHECK_HIP(hipFree(g_plane_arena.buf));
            g_plane_arena.buf = nullptr; g_plane_arena.bytes = 0;
            const size_t want = need + need / 8;
            if (hipMalloc((void**)&g_plane_arena.buf, want) != hipSuccess) {
                (void)hipGetLastError();
                if (c0 == 0) return -1;
                set_error("planes_adapter: could not grow the plane arena to %zu bytes", want);
                return 3;
            }
            g_plane_arena.bytes = want;
        }
        unsigned char* base = g_plane_arena.buf;
        Gemm16Problem q16[4];
        for (int i = 0; i < nc; ++i) {
            const GemmProblem& p = probs[c0 + i];
            const Plan& q = pl[i];
            auto u16 = [&](size_t o) { return reinterpret_cast<unsigned short*>(base + o); };
            // storage shapes: layout 0 = [rows = M or N][K]; layout 1 = [K][M or N]
            const int ar = al == 0 ? p.M : p.K, acol = al == 0 ? p.K : p.M, arp = al == 0 ? q.Mp : q.Kp, acp = al == 0 ? q.Kp : q.Mp;
            const int br = bl == 0 ? p.N : p.K, bcol = bl == 0 ? p.K : p.N, brp = bl == 0 ? q.Np : q.Kp, bcp = bl == 0 ? q.Kp : q.Np;
            auto grid = [](int64_t groups) { const int64_t b = (groups + 255) / 256; return (unsigned)(b > 8192 ? 8192 : (b < 1 ? 1 : b)); };
            hipLaunchKernelGGL(to_planes_kernel, dim3(grid((int64_t)arp * (acp / 8))), dim3(256), 0, s, p.A, p.lda, ar, acol, u16(q.a_hi), u16(q.a_lo), arp, acp);
            hipLaunchKernelGGL(to_planes_kernel, dim3(grid((int64_t)brp * (bcp / 8))), dim3(256), 0, s, p.B, p.ldb, br, bcol, u16(q.b_hi), u16(q.b_lo), brp, bcp);
            const float* bias = p.bias;
            if (!q.direct && p.bias) {
                hipLaunchKernelGGL(pad_bias_kernel, dim3(cdiv(q.Np, 256)), dim3(256), 0, s, p.bias, reinterpret_cast<float*>(base + q.bias), p.N, q.Np);
                bias = reinterpret_cast<float*>(base + q.bias);
            }
            EXORL_LAUNCH_CHECK();
            Gemm16Problem g{u16(q.a_hi), u16(q.b_hi), q.direct ? p.C : reinterpret_cast<float*>(base + q.cp), bias, q.Mp, q.Np, q.Kp, acp, bcp,
                            q.direct ? p.ldc : (int64_t)q.Np};
            g.A_lo = u16(q.a_lo); g.B_lo = u16(q.b_lo);
            g.n_store = q.direct && p.N % 128 != 0 ? p.N : 0;          // written in place, the tile columns past N skipped in the epilogue
            q16[i] = g;
        }
        bool all_direct = true;
        for (int i = 0; i < nc; ++i) all_direct = all_direct && pl[i].direct;
        // accumulate rides in the GEMM epilogue only when every C of the chunk is written in place; otherwise the copy-back adds
        EXORL_TRY(gemm16_grouped(al, bl, q16, nc, relu, accumulate && all_direct, s));
        for (int i = 0; i < nc; ++i) {
            if (pl[i].direct) {
                EXORL_REQUIRE(!accumulate || all_direct, "planes_adapter: mixed in-place / padded outputs with accumulate");
                continue;
            }
            const GemmProblem& p = probs[c0 + i];
            const int64_t n = (int64_t)p.M * p.N;
            hipLaunchKernelGGL(from_padded_kernel, dim3((unsigned)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256)), dim3(256), 0, s,
                               reinterpret_cast<const float*>(base + pl[i].cp), (int64_t)pl[i].Np, p.C, p.ldc, p.M, p.N, accumulate ? 1 : 0);
            EXORL_LAUNCH_CHECK();
        }
    }
    return 0;
}

// Launches up to 4 independent problems (same layouts / epilogue flags) as one grid.
// Diagnostic (tools/debug/config4_ablation.py; no product path sets it): which split-bf16 products run with exact fp32 products instead.
// bits: 1 / 2 forward (row-image A, row-image B) narrow / wide; 4 / 8 wgrad (k-image A and B); 16 / 32 dgrad (row-image A, k-image B);
// "wide" = a problem dimension >= 8192 (the 39200-wide layers of the pixel agents); 64 / 128 / 256 = conv forward / dgrad / wgrad (pixels.hip)
static int g_prec_override = 0;
int prec_override_mask() { return g_prec_override; }

int gemm_grouped(int precision, int a_layout, int b_layout, const GemmProblem* probs, int count, bool relu,
                 bool accumulate, hipStream_t s) {
    EXORL_REQUIRE(count >= 1 && count <= GEMM_MAX_GROUP, "gemm_grouped: count %d out of range", count);
    if (g_prec_override && precision == EXORL_PREC_BF16X3) {
        bool wide = false;
        for (int i = 0; i < count; ++i) wide = wide || probs[i].M >= 8192 || probs[i].N >= 8192 || probs[i].K >= 8192;
        const int form = (a_layout == 0 && b_layout == 0) ? 0 : (a_layout == 1 && b_layout == 1) ? 1 : (a_layout == 0 && b_layout == 1) ? 2 : -1;
        if (form >= 0 && (g_prec_override >> (2 * form + (wide ? 1 : 0))) & 1) precision = EXORL_PREC_F32;
    }
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
    if (precision == EXORL_PREC_BF16X3 && planes_adapter_wants(a_layout, b_layout, probs, count)) {
        bool mixed = false;          // accumulate with some outputs in place and some padded: keep it simple, generic kernel
        if (accumulate) {
            int direct = 0;
            for (int i = 0; i < count; ++i)
                direct += adapter_direct(probs[i]);
            mixed = direct != 0 && direct != count;
        }
        if (!mixed) {
            const int rc = planes_adapter(a_layout, b_layout, probs, count, relu, accumulate, s);
            if (rc >= 0) return rc;
        }
    }
    if (precision == EXORL_PREC_F32) return launch_prec<EXORL_PREC_F32>(gb, count, a_layout, b_layout, max_tiles, vec, s);
    if (precision == EXORL_PREC_BF16) return launch_prec<EXORL_PREC_BF16>(gb, count, a_layout, b_layout, max_tiles, vec, s);
    if (precision == EXORL_PREC_BF16X3) return launch_prec<EXORL_PREC_BF16X3>(gb, count, a_layout, b_layout, max_tiles, vec, s);
    if (precision == EXORL_PREC_BF16X6) return launch_prec<EXORL_PREC_BF16X6>(gb, count, a_layout, b_layout, max_tiles, vec, s);
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

// Median duration of an EMPTY start/stop event bracket on `stream`: the fixed cost hipEvent timing adds to every
// bracketed launch (subtracted by bench.py so its per-launch figure is comparable with rocprofv3's kernel durations).
extern "C" int exorl_profile_event_overhead(float* ms_out, void* stream) {
    using namespace exorl;
    EXORL_REQUIRE(ms_out, "profile_event_overhead: null argument");
    hipStream_t s = as_stream(stream);
    const int n = 31;
    hipEvent_t ev[2 * n];
    for (int i = 0; i < 2 * n; ++i) EXORL_CHECK_HIP(hipEventCreate(&ev[i]));
    for (int i = 0; i < n; ++i) {
        EXORL_CHECK_HIP(hipEventRecord(ev[2 * i], s));
        EXORL_CHECK_HIP(hipEventRecord(ev[2 * i + 1], s));
    }
    EXORL_CHECK_HIP(hipStreamSynchronize(s));
    std::vector<float> t(n);
    for (int i = 0; i < n; ++i) EXORL_CHECK_HIP(hipEventElapsedTime(&t[i], ev[2 * i], ev[2 * i + 1]));
    for (int i = 0; i < 2 * n; ++i) (void)hipEventDestroy(ev[i]);
    std::sort(t.begin(), t.end());
    *ms_out = t[n / 2];
    return 0;
}

namespace exorl { int tune_variant() { return g_gemm16_variant < 0 ? 0 : g_gemm16_variant; } }

extern "C" int exorl_debug_gemm_stamps(uint64_t* out_host, int32_t n_words) {
    EXORL_REQUIRE(out_host && n_words > 0 && n_words <= 8 * 1024, "debug_gemm_stamps: bad arguments");
    EXORL_CHECK_HIP(hipDeviceSynchronize());
    EXORL_CHECK_HIP(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(exorl::g16p_stamps), (size_t)n_words * sizeof(uint64_t)));
    return 0;
}

extern "C" int exorl_debug_precision_override(int32_t mask) {
    exorl::g_prec_override = mask;
    return 0;
}

extern "C" int exorl_gemm_tune(int32_t variant) {
    exorl::g_gemm16_variant = variant;
    return 0;
}

extern "C" int exorl_gemm_bf16(int32_t a_layout, int32_t b_layout, int32_t M, int32_t N, int32_t K, const uint16_t* A,
                               int64_t lda, const uint16_t* B, int64_t ldb, float* C, int64_t ldc, const float* bias,
                               int32_t relu, int32_t accumulate, void* stream) {
    exorl::Gemm16Problem p{A, B, C, bias, M, N, K, lda, ldb, ldc};
    return exorl::gemm16_grouped(a_layout, b_layout, &p, 1, relu != 0, accumulate != 0, exorl::as_stream(stream));
}

extern "C" int exorl_gemm_planes(int32_t count, const int32_t* a_layouts, int32_t b_layout, int32_t M, int32_t N, int32_t K,
                                 const uint16_t* const* A_hi, const uint16_t* const* A_lo, int64_t lda, const uint16_t* const* B_hi,
                                 const uint16_t* const* B_lo, int64_t ldb, float* const* C, int64_t ldc, int32_t relu, void* stream) {
    using namespace exorl;
    EXORL_REQUIRE(count >= 1 && count <= 4 && a_layouts && A_hi && B_hi && C, "gemm_planes: bad arguments");
    Gemm16Problem p[4];
    bool mixed = false;
    for (int i = 0; i < count; ++i) {
        p[i] = Gemm16Problem{A_hi[i], B_hi[i], C[i], nullptr, M, N, K, lda, ldb, ldc};
        if (A_lo && A_lo[i]) { p[i].A_lo = A_lo[i]; p[i].B_lo = B_lo ? B_lo[i] : nullptr; }
        mixed = mixed || a_layouts[i] != a_layouts[0];
    }
    if (mixed) {
        EXORL_REQUIRE(b_layout == 1 && !relu, "gemm_planes: mixed A layouts go with B as a k image and no epilogue (wgrad + dgrad)");
        int at[4];
        for (int i = 0; i < count; ++i) at[i] = a_layouts[i];
        return gemm16_grouped_mixed(at, p, count, as_stream(stream));
    }
    return gemm16_grouped(a_layouts[0], b_layout, p, count, relu != 0, false, as_stream(stream));
}

extern "C" int exorl_gemm(int32_t precision, int32_t a_layout, int32_t b_layout, int32_t M, int32_t N, int32_t K,
                          const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                          const float* bias, int32_t relu, int32_t accumulate, void* stream) {
    exorl::GemmProblem p{A, B, C, bias, M, N, K, lda, ldb, ldc};
    return exorl::gemm_grouped(precision, a_layout, b_layout, &p, 1, relu != 0, accumulate != 0, exorl::as_stream(stream));
}
