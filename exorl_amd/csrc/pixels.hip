// Pixel front end of the DDPG-backbone agents. Replaces (file:line in the reference repo):
//   utils.RandomShiftsAug   utils/utils.py:222-254   replicate pad 4 + random integer shift through F.grid_sample
//   ddpg.Encoder            agents/unsupervised_learning/ddpg.py:12-39   obs/255 - 0.5, 4 x [Conv2d 3x3 (stride 2,1,1,1) + ReLU], flatten
// Direct fp32 convolutions in the reference's NCHW layout (the flatten order feeds Linear(39200, feature_dim) unchanged):
//   forward / dgrad  one workgroup = one image x a 16x16 output tile x all 32 output channels; the input tile of every input
//                    channel and the layer's weights sit in LDS, a thread owns one output pixel and 32 accumulators
//   wgrad            one workgroup = one image, thread = (co, ci) pair with its 9 taps in registers; per-image partials, summed in
//                    image order by the column-sum kernel (deterministic, no atomics)
// The ReLU mask of layer l is applied where d(a_l) is produced (dgrad epilogue), so no separate masking pass touches the maps.
// In the bf16 / split-bf16 / three-plane precision modes the three 32 -> 32 stride-1 layers run as MFMA implicit GEMMs instead. Product path
// (round 3): conv3x3_ws_kernel (forward, dgrad; persistent form for batches) and conv_wgrad_ws_kernel — wave-specialised workgroups over a
// circular row buffer in LDS; the round-2 kernels conv3x3_mfma_kernel / conv_wgrad_mfma_kernel remain as the fallback for geometries those
// decline and for A/B (exorl_gemm_tune bits 1073741824 / 64). The first layer: forward over row-major strips (conv1_strip_kernel, every mode),
// weight gradients on MFMA with operands built in registers (conv1_wgrad_mfma_kernel); the tile kernels stay for precision fp32's other
// layers and as fallbacks.
#include <type_traits>

#include "kernels.h"

namespace exorl {

constexpr int CONV_CO = 32;          // every layer of the encoder has 32 output channels (ddpg.py:27-31)
constexpr int CONV_TILE = 16;

// ---- RandomShiftsAug: out[n][c][i][j] = bilinear tap of the replicate-padded image at (i + sy, j + sx) ---------------------------
__device__ __forceinline__ float lin_f32(float start, float end, float step, int i, int steps) {      // torch.linspace, fp32 CPU
    return i < steps / 2 ? start + step * (float)i : end - step * (float)(steps - i - 1);
}
__global__ __launch_bounds__(256) void aug_shift_kernel(const unsigned char* __restrict__ x, const int* __restrict__ shifts, uint64_t seed,
                                                        uint64_t counter, float* __restrict__ out, int n, int c, int h, int pad) {
#pragma clang fp contract(off)
    const int P = h + 2 * pad;
    const float eps = (float)(1.0 / P), start = -1.0f + eps, end = 1.0f - eps, step = (end - start) / (float)(P - 1);
    const float unit = (float)(2.0 / P);
    const int64_t total = (int64_t)n * c * h * h;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(idx % h), i = (int)((idx / h) % h), ch = (int)((idx / ((int64_t)h * h)) % c), b = (int)(idx / ((int64_t)h * h * c));
        int sx, sy;
        if (shifts) { sx = shifts[2 * b]; sy = shifts[2 * b + 1]; }
        else {      // torch.randint(0, 2 pad + 1, (n,1,1,2)): one Philox draw per image
            uint32_t r[4] = {(uint32_t)b, 13u, (uint32_t)counter, (uint32_t)(counter >> 32)};
            Philox::gen(r, seed);
            sx = (int)(((uint64_t)r[0] * (uint64_t)(2 * pad + 1)) >> 32);
            sy = (int)(((uint64_t)r[1] * (uint64_t)(2 * pad + 1)) >> 32);
        }
        const float gx = lin_f32(start, end, step, j, P) + (float)sx * unit;
        const float gy = lin_f32(start, end, step, i, P) + (float)sy * unit;
        const float ix = ((gx + 1.0f) * (float)P - 1.0f) / 2.0f, iy = ((gy + 1.0f) * (float)P - 1.0f) / 2.0f;
        const float fx = floorf(ix), fy = floorf(iy);
        const int x0 = (int)fx, y0 = (int)fy;
        const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
        const unsigned char* img = x + ((int64_t)b * c + ch) * h * h;
        auto tap = [&](int yy, int xx) -> float {
            if (yy < 0 || yy >= P || xx < 0 || xx >= P) return 0.f;      // padding_mode='zeros' outside the padded image
            const int sy2 = min(max(yy - pad, 0), h - 1), sx2 = min(max(xx - pad, 0), h - 1);
            return (float)img[sy2 * h + sx2];
        };
        out[idx] = tap(y0, x0) * (wy0 * wx0) + tap(y0, x0 + 1) * (wy0 * wx1) + tap(y0 + 1, x0) * (wy1 * wx0) + tap(y0 + 1, x0 + 1) * (wy1 * wx1);
    }
}

// Same arithmetic, one workgroup per (image, channel): the shift is drawn once per workgroup instead of once per output element (a
// Philox block per pixel was most of the elementwise kernel's time), and the per-column / per-row source indices and bilinear weights —
// they depend on (j, sx) and (i, sy) only — are tabulated in LDS. Products and sums per element are formed exactly as above.
constexpr int AUG_MAXH = 128;
__global__ __launch_bounds__(256) void aug_shift_rows_kernel(const unsigned char* __restrict__ x, const int* __restrict__ shifts, uint64_t seed,
                                                             uint64_t counter, float* __restrict__ out, int c, int h, int pad) {
#pragma clang fp contract(off)
    __shared__ int c0[2][AUG_MAXH], c1[2][AUG_MAXH];         // [0]: columns (x), [1]: rows (y): clamped source index of tap 0 / tap 1, -1 = zero padding
    __shared__ float w0[2][AUG_MAXH], w1[2][AUG_MAXH];
    const int b = blockIdx.x / c, ch = blockIdx.x % c;
    const int P = h + 2 * pad;
    const float eps = (float)(1.0 / P), start = -1.0f + eps, end = 1.0f - eps, step = (end - start) / (float)(P - 1);
    const float unit = (float)(2.0 / P);
    int sxy[2];
    if (shifts) { sxy[0] = shifts[2 * b]; sxy[1] = shifts[2 * b + 1]; }
    else {      // torch.randint(0, 2 pad + 1, (n,1,1,2)): one Philox draw per image
        uint32_t r[4] = {(uint32_t)b, 13u, (uint32_t)counter, (uint32_t)(counter >> 32)};
        Philox::gen(r, seed);
        sxy[0] = (int)(((uint64_t)r[0] * (uint64_t)(2 * pad + 1)) >> 32);
        sxy[1] = (int)(((uint64_t)r[1] * (uint64_t)(2 * pad + 1)) >> 32);
    }
    for (int t = threadIdx.x; t < 2 * h; t += 256) {
        const int d = t / h, k = t % h;
        const float g = lin_f32(start, end, step, k, P) + (float)sxy[d] * unit;
        const float ik = ((g + 1.0f) * (float)P - 1.0f) / 2.0f;
        const float fk = floorf(ik);
        const int k0 = (int)fk;
        w1[d][k] = ik - fk;
        w0[d][k] = 1.0f - (ik - fk);
        c0[d][k] = (k0 < 0 || k0 >= P) ? -1 : min(max(k0 - pad, 0), h - 1);
        c1[d][k] = (k0 + 1 < 0 || k0 + 1 >= P) ? -1 : min(max(k0 + 1 - pad, 0), h - 1);
    }
    __syncthreads();
    const unsigned char* img = x + ((int64_t)b * c + ch) * h * h;
    float* o = out + ((int64_t)b * c + ch) * h * h;
    // (row, column) of element e kept incrementally (a division and a remainder per element were a third of this kernel's instructions)
    int i = (int)threadIdx.x / h, j = (int)threadIdx.x - i * h;
    const int di = 256 / h, dj = 256 - di * h;
    for (int e = threadIdx.x; e < h * h; e += 256, i += di, j += dj) {
        if (j >= h) { j -= h; ++i; }
        const int y0 = c0[1][i], y1 = c1[1][i], x0 = c0[0][j], x1 = c1[0][j];
        const float wy0 = w0[1][i], wy1 = w1[1][i], wx0 = w0[0][j], wx1 = w1[0][j];
        const float t00 = (y0 >= 0 && x0 >= 0) ? (float)img[y0 * h + x0] : 0.f, t01 = (y0 >= 0 && x1 >= 0) ? (float)img[y0 * h + x1] : 0.f;
        const float t10 = (y1 >= 0 && x0 >= 0) ? (float)img[y1 * h + x0] : 0.f, t11 = (y1 >= 0 && x1 >= 0) ? (float)img[y1 * h + x1] : 0.f;
        o[e] = t00 * (wy0 * wx0) + t01 * (wy0 * wx1) + t10 * (wy1 * wx0) + t11 * (wy1 * wx1);
    }
}

// ---- weight shadows: Wf[ci][tap][co] for the forward pass, Wb[co][tap flipped][ci] for dgrad -----------------------------------
// ff / fb (32 -> 32 layers in the bf16 modes): the MFMA kernel's B fragments, hi plane then lo plane, lane-linear — k-step s (tap = s >> 1,
// ci = (s & 1) * 16 + 8 * (lane >> 5) + j), column lane & 31 — so that a convolution workgroup copies 36 KB into LDS with 16-byte loads
// instead of converting the layer's 9216 weights itself (1024 workgroups per launch did, ~4.5 us each, ahead of their first pass).
constexpr int CM_FRAG = 18 * 64;           // 16-byte fragments per plane
__global__ void conv_weight_shadow_kernel(const float* __restrict__ W, float* __restrict__ Wf, float* __restrict__ Wb, int ci_n,
                                          uint4* __restrict__ ff, uint4* __restrict__ fb) {
    const int total = CONV_CO * ci_n * 9;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int tap = i % 9, ci = (i / 9) % ci_n, co = i / (9 * ci_n);
        const float w = W[i];                                    // torch layout [co][ci][ky][kx]
        Wf[(ci * 9 + tap) * CONV_CO + co] = w;
        if (Wb) Wb[(co * 9 + (8 - tap)) * ci_n + ci] = w;        // dx = full correlation of dy with the flipped kernel
    }
    if (!ff || ci_n != CONV_CO) return;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 2 * CM_FRAG; i += gridDim.x * blockDim.x) {
        const int dir = i / CM_FRAG, r = i % CM_FRAG, st = r >> 6, lane = r & 63, kg = lane >> 5, col = lane & 31, tap = st >> 1;
        unsigned hi[4], lo[4], l3[4];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int cin = (st & 1) * 16 + 8 * kg + j;
            // forward: Wt[ci][tap][co] = W[co][ci][tap]; dgrad: the flipped shadow read with the roles of ci and co exchanged
            const float w = dir == 0 ? W[(col * CONV_CO + cin) * 9 + tap] : W[(cin * CONV_CO + col) * 9 + (8 - tap)];
            const __bf16 h = (__bf16)w;
            const float r1 = w - (float)h;
            const __bf16 l = (__bf16)r1;                         // second plane: the lo plane of the two-plane mode = the mid plane of the three-plane mode
            const __bf16 t = (__bf16)(r1 - (float)l);            // third plane (EXORL_PREC_BF16X6)
            const unsigned hb = __builtin_bit_cast(unsigned short, h), lb = __builtin_bit_cast(unsigned short, l), tb = __builtin_bit_cast(unsigned short, t);
            if (j & 1) { hi[j >> 1] |= hb << 16; lo[j >> 1] |= lb << 16; l3[j >> 1] |= tb << 16; } else { hi[j >> 1] = hb; lo[j >> 1] = lb; l3[j >> 1] = tb; }
        }
        uint4* dst = dir == 0 ? ff : fb;
        dst[r] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        dst[CM_FRAG + r] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        dst[2 * CM_FRAG + r] = make_uint4(l3[0], l3[1], l3[2], l3[3]);
    }
}

// ---- forward conv / dgrad: out[n][co][y][x] = bias[co] + sum_{ci,ky,kx} Wt[ci][tap][co] * in[n][ci][y*s + ky - pad][x*s + kx - pad] ----
// in_scale: in = in / 255 - 0.5 (Encoder.forward, ddpg.py:36). relu: ReLU epilogue. mask: out *= (mask > 0) (dgrad through the ReLU of
// the layer below). Zero padding `pad` (0 forward, 2 for dgrad). co_n <= 32 output channels, ci_n input channels.
__global__ __launch_bounds__(256) void conv3x3_kernel(const float* __restrict__ in, const float* __restrict__ Wt, const float* __restrict__ bias,
                                                      const float* __restrict__ mask, float* __restrict__ out, int ci_n, int ih, int iw,
                                                      int oh, int ow, int stride, int pad, int in_scale, int relu) {
    extern __shared__ float lds[];
    const int tin = (CONV_TILE - 1) * stride + 3;                 // input tile edge
    float* tile = lds;                                            // [ci][tin][tin]
    const int tiles_x = (ow + CONV_TILE - 1) / CONV_TILE;
    const int ty0 = (blockIdx.x / tiles_x) * CONV_TILE, tx0 = (blockIdx.x % tiles_x) * CONV_TILE;
    const int n = blockIdx.y;
    const int iy0 = ty0 * stride - pad, ix0 = tx0 * stride - pad;
    const float* inn = in + (int64_t)n * ci_n * ih * iw;
    for (int i = threadIdx.x; i < ci_n * tin * tin; i += 256) {
        const int xx = i % tin, yy = (i / tin) % tin, ci = i / (tin * tin);
        const int gy = iy0 + yy, gx = ix0 + xx;
        float v = 0.f;
        if (gy >= 0 && gy < ih && gx >= 0 && gx < iw) {
            v = inn[((int64_t)ci * ih + gy) * iw + gx];
            if (in_scale) v = v / 255.0f - 0.5f;
        }
        tile[i] = v;
    }
    __syncthreads();
    const int ly = threadIdx.x / CONV_TILE, lx = threadIdx.x % CONV_TILE;
    const int oy = ty0 + ly, ox = tx0 + lx;
    float acc[CONV_CO];
#pragma unroll
    for (int co = 0; co < CONV_CO; ++co) acc[co] = 0.f;
    // The 32 weights of one (ci, tap) are wave-uniform and contiguous in the shadow: the compiler fetches them with scalar loads
    // (s_load_dwordx16 x 2) and the FMAs take them as SGPR operands — one LDS read (the pixel) per 32 FMAs instead of one per FMA.
    for (int ci = 0; ci < ci_n; ++ci) {
        const float* t = tile + ci * tin * tin + (ly * stride) * tin + lx * stride;
        const float* wci = Wt + (int64_t)ci * 9 * CONV_CO;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float v = t[(tap / 3) * tin + (tap % 3)];
            const float* w = wci + tap * CONV_CO;
#pragma unroll
            for (int co = 0; co < CONV_CO; ++co) acc[co] += v * w[co];
        }
    }
    if (oy >= oh || ox >= ow) return;
    float* o = out + (int64_t)n * CONV_CO * oh * ow + (int64_t)oy * ow + ox;
    const float* mk = mask ? mask + (int64_t)n * CONV_CO * oh * ow + (int64_t)oy * ow + ox : nullptr;
#pragma unroll
    for (int co = 0; co < CONV_CO; ++co) {
        float v = acc[co] + (bias ? bias[co] : 0.f);
        if (relu) v = fmaxf(v, 0.f);
        if (mk) v = mk[(int64_t)co * oh * ow] > 0.f ? v : 0.f;
        o[(int64_t)co * oh * ow] = v;
    }
}

// ---- first layer, forward (stride 2, no padding): the same sums over row-major strips ------------------------------------------------------
// PMC of conv3x3_kernel on this layer at batch 1024 (profiles/r03c_conv_pmc.txt): 182 MB fetched for an 87 MB input (33 x 33 halo tiles, 9 per
// image) and 325 MB written for a 220 MB output (16-pixel row segments of a 41-wide map straddle cache lines), 146 us for 3 GFLOP. Here a
// workgroup takes 256 consecutive output pixels of an image in row-major order: the input rows it needs are one contiguous run per channel
// (read once, + two halo rows), a thread owns one pixel, and the 64 threads of a wave store 256 contiguous bytes per channel. The arithmetic
// per output — channel-major, tap by tap, fp32 FMAs on weights held in SGPRs — is that of conv3x3_kernel, so the results are bit-identical.
// Columns are de-interleaved in LDS (even columns, then odd columns of a row): the stride-2 taps of 64 neighbouring pixels then hit 64
// consecutive words instead of every other one.
constexpr int C1_PASS = 256;
__global__ __launch_bounds__(256) void conv1_strip_kernel(const float* __restrict__ in, const float* __restrict__ Wt, const float* __restrict__ bias,
                                                          float* __restrict__ out, int ci_n, int ih, int iw, int oh, int ow, int in_scale, int relu,
                                                          int max_rows) {
    extern __shared__ float lds[];
    const int npix = oh * ow, n = blockIdx.y;
    const int p0 = blockIdx.x * C1_PASS, p1 = (p0 + C1_PASS < npix ? p0 + C1_PASS : npix) - 1;
    const int y0 = p0 / ow, nrows = 2 * (p1 / ow - y0) + 3;        // input rows 2 y0 .. 2 y1 + 2
    const int he = (iw + 1) >> 1, rowp = 2 * he;                   // even columns [0, he), odd columns [he, 2 he) of a row
    const float* inn = in + ((int64_t)n * ci_n * ih + 2 * y0) * iw;
    for (int i = threadIdx.x; i < ci_n * nrows * iw; i += 256) {
        const int xx = i % iw, rr = (i / iw) % nrows, ci = i / (iw * nrows);
        float v = inn[((int64_t)ci * ih + rr) * iw + xx];
        if (in_scale) v = v / 255.0f - 0.5f;
        lds[(ci * max_rows + rr) * rowp + (xx & 1) * he + (xx >> 1)] = v;
    }
    __syncthreads();
    const int p = p0 + threadIdx.x;
    const int pc = p < npix ? p : npix - 1;                        // tail threads recompute the last pixel and do not store
    const int oy = pc / ow, ox = pc - oy * ow;
    float acc[CONV_CO];
#pragma unroll
    for (int co = 0; co < CONV_CO; ++co) acc[co] = 0.f;
    for (int ci = 0; ci < ci_n; ++ci) {
        const float* t = lds + (ci * max_rows + 2 * (oy - y0)) * rowp + ox;
        const float* wci = Wt + (int64_t)ci * 9 * CONV_CO;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            const float v = t[ky * rowp + (kx & 1) * he + (kx >> 1)];       // column 2 ox + kx
            const float* w = wci + tap * CONV_CO;
#pragma unroll
            for (int co = 0; co < CONV_CO; ++co) acc[co] += v * w[co];
        }
    }
    if (p >= npix) return;
    float* o = out + (int64_t)n * CONV_CO * npix + p;
#pragma unroll
    for (int co = 0; co < CONV_CO; ++co) {
        float v = acc[co] + (bias ? bias[co] : 0.f);
        if (relu) v = fmaxf(v, 0.f);
        o[(int64_t)co * npix] = v;
    }
}

// ---- the 32 -> 32 channel layers on the matrix cores (split-bf16 / bf16 modes) -----------------------------------------------------
// Implicit GEMM per 16 x 16 output tile: M = 32 pixels (two tile rows), N = 32 output channels, K = 9 taps x 32 input channels with
// k = tap * 32 + ci, so the A fragment of a k16 step (8 consecutive ci of one tap for one pixel) is 16 contiguous bytes of a
// channels-last bf16 copy of the input tile in LDS: one ds_read_b128 (pixel stride 80 B: conflict-free for 16 consecutive pixels).
// The NCHW fp32 maps are converted on the way into LDS (hi and, in split mode, lo planes); the layer's weights — 18 k16 steps x 8
// values per lane and plane — are converted once per workgroup and live in registers; a workgroup walks all tiles of one image.
// Split mode forms hi*hi + (hi*lo + lo*hi) as in the H x H GEMMs. The output tile goes back through LDS so that the NCHW stores
// cover 64-byte row segments, with bias / ReLU / the dgrad mask applied there.
typedef __bf16 cbf16x8 __attribute__((ext_vector_type(8)));
typedef float cf32x16 __attribute__((ext_vector_type(16)));
constexpr int CM_PIX = 40;                 // bf16 elements per staged pixel (32 channels + 8 pad = 80 B)

constexpr int CM_THREADS = 512;            // 8 waves x 32 pixels = 256 output pixels per pass
constexpr int CM_PASS = 256;
constexpr int CM_MAXW = 48;                // widest staged row (output width + 2) this kernel takes: ow <= 46
// k-steps unrolled together in the MFMA loop: 3 measured equal to 2 (9.4 vs 9.3 ms per Proto update in bf16x6, 242 vs 240 VGPRs)
#ifndef CM_UNROLL
#define CM_UNROLL 2
#endif
constexpr int CM_IPT = 15;                 // staged (channel pair, y, x) items per thread: 16 * rows * (ow + 2) <= 15 * 512 (host-checked)

// A pass covers 256 consecutive output pixels in row-major order of the map (not a square tile: a 39-wide map would pay for 48 x 48),
// so the staged input is a strip of full-width rows: [rows][ow + 2][CM_PIX] with the tap offsets applied inside it.
// NPL = operand planes: 1 plain bf16, 2 split-bf16 (hi*hi + hi*lo + lo*hi), 3 three-plane split (EXORL_PREC_BF16X6: + hi*l3 + l3*hi + lo*lo,
// products accurate to 3 * 2^-24 — the parity-grade convolution of the pixel agents at ~2x the split-bf16 kernel's MFMA time, where the fp32
// FMA kernel it replaces took 5.8x)
// STAMP (diagnostic build, exorl_gemm_tune bit 8192; no product launch takes it): waves 0 and 7 of every workgroup add up the shader-clock
// cycles of each phase of a pass (convert + LDS write | barrier | fetch issue | MFMA loop with its LDS reads | stores | barrier) over
// the workgroup's passes and leave them in g_conv_stamps[workgroup][2][8] (exorl_debug_conv_stamps reads them back).
__device__ unsigned long long g_conv_stamps[1024 * 2 * 8];
template <int NPL, bool MASK, bool STAMP = false>
__global__ __launch_bounds__(CM_THREADS) void conv3x3_mfma_kernel(const float* __restrict__ in, const float* __restrict__ Wt,
                                                                  const float* __restrict__ bias, const float* __restrict__ mask,
                                                                  float* __restrict__ out, int ih, int iw, int oh, int ow, int pad, int relu,
                                                                  int plane_elems) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cm_lds[];
    constexpr bool X3 = NPL >= 2, X6 = NPL == 3;
    unsigned short* xh = reinterpret_cast<unsigned short*>(cm_lds);                       // [rows][ow + 2][CM_PIX] hi plane
    unsigned short* xl = xh + plane_elems;                                                // second plane (split modes)
    unsigned short* xt = xl + plane_elems;                                                // third plane (three-plane mode)
    cbf16x8* wh = reinterpret_cast<cbf16x8*>(xh + NPL * plane_elems);                     // [18 k-steps][64 lanes] B fragments, 16 B each
    cbf16x8* wl = wh + 18 * 64;                                                           // (CM_FRAG, defined with the shadow kernel)
    cbf16x8* wt = wl + 18 * 64;
    const int n = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kg = lane >> 5, col = lane & 31;
    const float* inn = in + (int64_t)n * CONV_CO * ih * iw;
    // the layer's MFMA B fragments (conv_weight_shadow_kernel made them, hi plane then lo plane): a straight 16-byte copy into LDS
    {
        const uint4* fr = reinterpret_cast<const uint4*>(Wt);
        uint4* dst = reinterpret_cast<uint4*>(wh);
        for (int i = tid; i < NPL * CM_FRAG; i += CM_THREADS) dst[i] = fr[i];
    }
    const int npix = oh * ow, npass = (npix + CM_PASS - 1) / CM_PASS, sw = ow + 2;
    // The input of pass t+1 is fetched into registers while pass t is on the matrix cores; it is converted and written to LDS once
    // every wave is done with pass t. (channel pair, strip row, x) per item; consecutive threads walk x.
    float v0[CM_IPT], v1[CM_IPT];
    auto strip = [&](int pass, int& y0, int& rows) {          // output rows y0 .. y0 + rows - 3 are touched; staged rows = rows
        const int p0 = pass * CM_PASS, p1 = (p0 + CM_PASS < npix ? p0 + CM_PASS : npix) - 1;
        y0 = p0 / ow;
        rows = p1 / ow - y0 + 3;
    };
    // Staging items: a half-wave (32 lanes) owns one channel pair and walks the strip's (row, x) positions in flattened order j = lane + 32 u,
    // so global reads are 128-byte runs of a row, the LDS offset is linear in j, and (row, x) advance by a constant step with one carry —
    // the per-item index arithmetic is a handful of adds (it was two reciprocal divisions per item, and with the conversions made the
    // kernel VALU-bound: ~8 k cycles of staging per SIMD and pass against 3.5 k cycles of MFMA).
    const int cp = tid >> 5, jl = tid & 31;
    const float* in0 = inn + (int64_t)(2 * cp) * ih * iw;
    const float* in1 = in0 + (int64_t)ih * iw;
    const int step_y = 32 / sw, step_x = 32 % sw, yy0 = jl / sw, xx0 = jl % sw;
    typedef float cf32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 cbf16x2 __attribute__((ext_vector_type(2)));
    auto fetch = [&](int pass) {
        int y0, rows;
        strip(pass, y0, rows);
        const int nj = rows * sw;
        int yy = yy0, xx = xx0;
#pragma unroll
        for (int u = 0; u < CM_IPT; ++u) {
            const int gy = y0 - pad + yy, gx = xx - pad;
            const bool ok = jl + 32 * u < nj && (unsigned)gy < (unsigned)ih && (unsigned)gx < (unsigned)iw;
            const int off = ok ? gy * iw + gx : 0;                   // unconditional loads (index clamped), masked afterwards
            const float a0 = in0[off], a1 = in1[off];
            v0[u] = ok ? a0 : 0.f;
            v1[u] = ok ? a1 : 0.f;
            xx += step_x; yy += step_y;
            if (xx >= sw) { xx -= sw; ++yy; }
        }
    };
    fetch(gridDim.y > 1 ? (int)blockIdx.y : 0);
    const float bv = bias ? bias[col] : 0.f;
    // One pass. FULL (every pass but the last): all four pixel quads of a lane lie inside the map and there is a next strip to fetch, so the
    // body has NO run-time branch around a memory instruction — which is what lets the compiler count: the wait in front of the next pass's
    // conversion becomes vmcnt(<this pass's stores>) instead of vmcnt(0), i.e. the stores of pass t drain under pass t + 1 instead of in
    // front of it (gfx9 counts stores in vmcnt too; with the guarded tail path in the loop every pass waited for its predecessor's stores).
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto now = [&]() -> unsigned long long {
        if constexpr (STAMP) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); return __builtin_amdgcn_s_memtime(); }
        return 0ull;
    };
    const unsigned long long t_begin = now();
    auto body = [&](int pass, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        int y0, rows;
        strip(pass, y0, rows);
        const int nj = rows * sw, p0 = pass * CM_PASS;
        const unsigned long long t0 = now();
#pragma unroll
        for (int u = 0; u < CM_IPT; ++u) {
            const int j = jl + 32 * u;
            if (j < nj) {
                const cf32x2 v = {v0[u], v1[u]};
                const cbf16x2 h = __builtin_convertvector(v, cbf16x2);
                const int o = j * CM_PIX + 2 * cp;
                *reinterpret_cast<unsigned int*>(xh + o) = __builtin_bit_cast(unsigned int, h);
                if constexpr (X3) {
                    const cf32x2 r1 = v - __builtin_convertvector(h, cf32x2);
                    const cbf16x2 l = __builtin_convertvector(r1, cbf16x2);
                    *reinterpret_cast<unsigned int*>(xl + o) = __builtin_bit_cast(unsigned int, l);
                    if constexpr (X6) {
                        const cbf16x2 t = __builtin_convertvector(r1 - __builtin_convertvector(l, cf32x2), cbf16x2);
                        *reinterpret_cast<unsigned int*>(xt + o) = __builtin_bit_cast(unsigned int, t);
                    }
                }
            }
        }
        const unsigned long long t1 = now();
        __syncthreads();
        const unsigned long long t2 = now();
        if constexpr (FULL) fetch(pass + 1);
        const unsigned long long t3 = now();
        // C layout of the 32x32 MFMA: reg r of lane l = pixel (r & 3) + 8 (r >> 2) + 4 (l >> 5) of the wave's 32, channel l & 31 — four
        // consecutive pixels of one channel per register quad, i.e. 16 contiguous bytes of the NCHW map: the lane stores them itself
        // (and fetches the dgrad mask the same way, now, behind the MFMA work); no output staging, no second barrier.
        const int pq = p0 + 32 * wave + 4 * kg;                                 // + 8 q + (0..3), q = r >> 2
        float* orow = out + ((int64_t)n * CONV_CO + col) * npix;
        float4 mq[4];
        if constexpr (MASK) {
            const float* mrow = mask + ((int64_t)n * CONV_CO + col) * npix;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int pp = pq + 8 * q;
                if (FULL || pp + 3 < npix) mq[q] = *reinterpret_cast<const float4*>(mrow + pp);        // 4-byte aligned vector load
                else {
                    mq[q].x = pp < npix ? mrow[pp] : 1.f; mq[q].y = pp + 1 < npix ? mrow[pp + 1] : 1.f;
                    mq[q].z = pp + 2 < npix ? mrow[pp + 2] : 1.f; mq[q].w = 1.f;
                }
            }
        }
        cf32x16 acc, accx, accy;               // hi*hi | hi*lo + lo*hi | (three planes) hi*l3 + l3*hi + lo*lo
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accx[r] = 0.f; accy[r] = 0.f; }
        int pa = p0 + 32 * wave + col;                                          // this lane's pixel of the A operand (clamped: tail lanes recompute the last pixel)
        pa = pa < npix ? pa : npix - 1;
        const int abase = ((pa / ow - y0) * sw + pa % ow) * CM_PIX + 8 * kg;
#pragma unroll CM_UNROLL
        for (int s = 0; s < 18; ++s) {                                          // CM_UNROLL k-steps of fragments live at a time (VGPR budget: 256 at 8 waves)
            const int dy = (s >> 1) / 3, dx = (s >> 1) % 3;
            const int o = abase + (dy * sw + dx) * CM_PIX + (s & 1) * 16;
            const cbf16x8 ah = *reinterpret_cast<const cbf16x8*>(xh + o);
            const cbf16x8 bh = wh[s * 64 + lane];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
            if constexpr (X3) {
                const cbf16x8 al = *reinterpret_cast<const cbf16x8*>(xl + o);
                const cbf16x8 bl = wl[s * 64 + lane];
                accx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, accx, 0, 0, 0);
                accx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, accx, 0, 0, 0);
                if constexpr (X6) {
                    const cbf16x8 at = *reinterpret_cast<const cbf16x8*>(xt + o);
                    const cbf16x8 bt = wt[s * 64 + lane];
                    accy = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bt, accy, 0, 0, 0);
                    accy = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at, bh, accy, 0, 0, 0);
                    accy = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bl, accy, 0, 0, 0);
                }
            }
        }
        if constexpr (STAMP) asm volatile("" :: "v"(acc[0]), "v"(accx[0]), "v"(accy[0]));
        const unsigned long long t4 = now();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = (X6 ? (accy[4 * q + e] + accx[4 * q + e]) + acc[4 * q + e] : X3 ? accx[4 * q + e] + acc[4 * q + e] : acc[4 * q + e]) + bv;
                if (relu) v[e] = fmaxf(v[e], 0.f);
            }
            if constexpr (MASK) {
                v[0] = mq[q].x > 0.f ? v[0] : 0.f; v[1] = mq[q].y > 0.f ? v[1] : 0.f;
                v[2] = mq[q].z > 0.f ? v[2] : 0.f; v[3] = mq[q].w > 0.f ? v[3] : 0.f;
            }
            const int pp = pq + 8 * q;
            if (FULL || pp + 3 < npix) *reinterpret_cast<float4*>(orow + pp) = make_float4(v[0], v[1], v[2], v[3]);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (pp + e < npix) orow[pp + e] = v[e];
            }
        }
        const unsigned long long t5 = now();
        __syncthreads();                   // every wave is past its MFMA reads before the next pass rewrites the input planes
        if constexpr (STAMP) {
            const unsigned long long t6 = now();
            st[0] += t1 - t0; st[1] += t2 - t1; st[2] += t3 - t2; st[3] += t4 - t3; st[4] += t5 - t4; st[5] += t6 - t5;
        }
    };
    auto leave = [&]() {
        if constexpr (STAMP) {
            st[6] = now() - t_begin;
            if ((wave == 0 || wave == 7) && lane == 0 && blockIdx.x < 1024 && blockIdx.y == 0)
                for (int i = 0; i < 8; ++i) g_conv_stamps[(blockIdx.x * 2 + (wave ? 1 : 0)) * 8 + i] = st[i];
        }
    };
    if (gridDim.y > 1) {                   // few images (act() on one frame): one pass per workgroup, gridDim.y = npass workgroups per image
        if ((int)blockIdx.y < npass) body((int)blockIdx.y, std::false_type{});
        return;
    }
    for (int pass = 0; pass + 1 < npass; ++pass) body(pass, std::true_type{});
    body(npass - 1, std::false_type{});
    leave();
}

// ---- the same implicit GEMM with the waves specialised (round 3) ------------------------------------------------------------------------
// Stamps of the kernel above (tools/micro/conv_stamp_bench.py, profiles/r03_conv_phase_stamps.txt): inside its MFMA loop the matrix pipe is
// at its issue floor, but the loop is 41-45 % of the kernel — conversion + LDS writes (15 %), issuing the next strip's loads (13-15 %), stores and
// the two barriers run with the pipe idle, because all eight waves walk the phases together and one workgroup owns the CU. Here waves 0-3
// (one per SIMD) only multiply — a pass is 128 pixels, 32 per wave — and waves 4-7 (their SIMD mates) only stage: the input rows live in a
// CIRCULAR buffer of rb full-width rows (slot = row mod rb), so while pass t is on the matrix cores the producers convert the rows pass t + 1
// adds (<= CW_PI * 16 / (ow + 2) rows, fetched into registers one pass earlier) into the slots pass t - 1 has left; one barrier per pass.
// Every input row is staged exactly once (the strip kernel restaged the 3-row halo: 10 rows per 6.6 rows of output). The consumers are alone
// on their SIMD's matrix pipe, so they keep the next k-step's fragments in flight under the current step's MFMAs (explicit double buffer).
constexpr int CW_PASS = 128;               // output pixels per pass
constexpr int CW_PI = 6;                   // staged positions per producer thread and batch: 32 lanes x 6 = 192 positions of one channel quad
// bytes per staged position: the NPL planes of a pixel sit side by side (32 channels x 2 B = 64 B each, 80 B apart), so that a producer's
// stores and a consumer's fragment reads reach all planes from ONE address register with immediate offsets; the stride keeps the 16 lanes of
// a ds_read_b128 group on distinct banks (240 B = 60 dwords and 176 B = 44 dwords: both step through all multiples of 4 mod 64)
__host__ __device__ constexpr int cw_pixb(int npl) { return npl == 3 ? 240 : npl == 2 ? 176 : 80; }
constexpr int CW_DUMMY = 256 * 8 + 176;    // a dummy 8-byte word per producer thread, reached with the same plane offsets (+ 80, + 160) as a real position
// PERSIST: the workgroup walks images blockIdx.x, + gridDim.x, ... (launched with one workgroup per CU): the weight fragments are loaded once, and
// the loads the producers issue past an image's last pass — wasted in the one-image form — fetch the first rows of the workgroup's NEXT image,
// so that between two images only their conversion stands in front of the consumers (the per-image prologue was 15-20 % of the kernel).
template <int NPL, bool MASK, bool STAMP = false, bool PERSIST = false>           // MASK = the dgrad launches: ReLU mask of the layer below AND zero padding (pad = 2)
__global__ __launch_bounds__(CM_THREADS) void conv3x3_ws_kernel(const float* __restrict__ in, const float* __restrict__ Wt,
                                                                const float* __restrict__ bias, const float* __restrict__ mask,
                                                                float* __restrict__ out, int ih, int iw, int oh, int ow, int pad, int relu,
                                                                int rb, int flags, int nimg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cm_lds[];
    constexpr bool X3 = NPL >= 2, X6 = NPL == 3;
    constexpr int PIXB = cw_pixb(NPL);
    const int npix = oh * ow, npass = (npix + CW_PASS - 1) / CW_PASS, sw = ow + 2;
    const int ring_b = rb * sw * PIXB;                                                     // the ring: [rb rows by slot][ow + 2][PIXB]
    unsigned char* ring = cm_lds;                                                          // then CW_DUMMY bytes for the tail items of a batch
    cbf16x8* wh = reinterpret_cast<cbf16x8*>(cm_lds + ((ring_b + CW_DUMMY + 15) & ~15));   // [NPL][18 k-steps][64 lanes] B fragments
    int n = blockIdx.x;                                                                    // the image (PERSIST: the current one)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool producer = (flags & 2) ? wave >= 4 : wave < 4;                              // wave-uniform role; the older half stages (measured)
    const int cw = wave & 3;                                                               // consumer index: pixels 32 cw .. of a pass
    const int kg = lane >> 5, col = lane & 31;
    {
        const uint4* fr = reinterpret_cast<const uint4*>(Wt);
        uint4* dst = reinterpret_cast<uint4*>(wh);
        for (int i = tid; i < NPL * CM_FRAG; i += CM_THREADS) dst[i] = fr[i];
    }
    const int t0 = gridDim.y > 1 ? (int)blockIdx.y : 0, t1 = gridDim.y > 1 ? (t0 + 1 < npass ? t0 + 1 : npass) : npass;
    if (t0 >= npass) return;
    auto end_row = [&](int pass) {             // one past the last strip row pass touches (strip row y = input row y - pad)
        const int p1 = (pass * CW_PASS + CW_PASS < npix ? pass * CW_PASS + CW_PASS : npix) - 1;
        return p1 / ow + 3;
    };
    const int chunk = (32 * CW_PI) / sw;       // rows a batch can hold (host-checked against the most rows a pass adds)
    // ---- producers: (channel quad, 32-lane slice) per thread; positions j = jl + 32 u of rows [ya, yb) in row-major order -------------------
    // Everything about an item that does not depend on the batch is computed once: its byte offset inside the image (row relative to the
    // batch's first row), its byte offset inside the ring (likewise), its row inside the batch (4 bits each) and whether its column is padding.
    // A batch then costs one add per item for its four loads (raw buffer loads through one descriptor per channel of the quad: the range check
    // answers 0, without a fault, for the few addresses in front of or behind the tensor — first rows of image 0 under padding, tail items of
    // the last image) and, at commit, an add, an unsigned wrap (sub + min), a select for tail items (they go to a per-thread dummy behind
    // the ring), the conversions and one 8-byte store per plane — no branch, no division, no 64-bit address arithmetic.
    const int ptid = tid & 255, cq = ptid >> 5, jl = ptid & 31;
    const int plane_b = ih * iw * 4;                                        // bytes of one channel of one image
    const unsigned total_b = (unsigned)nimg * CONV_CO * plane_b;            // host-checked < 2^31
    float* inq = const_cast<float*>(in);
    const auto src0 = __builtin_amdgcn_make_buffer_rsrc(inq, 0, (int)(total_b - 3 * plane_b), 0x00020000);
    const auto src1 = __builtin_amdgcn_make_buffer_rsrc(inq + ih * iw, 0, (int)(total_b - 3 * plane_b), 0x00020000);      // channel + 1: base one plane on
    const auto src2 = __builtin_amdgcn_make_buffer_rsrc(inq + 2 * ih * iw, 0, (int)(total_b - 3 * plane_b), 0x00020000);
    const auto src3 = __builtin_amdgcn_make_buffer_rsrc(inq + 3 * ih * iw, 0, (int)(total_b - 3 * plane_b), 0x00020000);
    const int toff = ring_b + 8 * ptid;
    typedef float cf32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 cbf16x2 __attribute__((ext_vector_type(2)));
    int goff[CW_PI], loff[CW_PI];
    unsigned ypk = 0, xmask = 0;
#pragma unroll
    for (int u = 0; u < CW_PI; ++u) {
        const int j = jl + 32 * u, yy = j / sw, xx = j - yy * sw;
        goff[u] = ((4 * cq * ih + yy) * iw + xx) * 4;
        loff[u] = (yy * sw + xx) * PIXB + 8 * cq;
        ypk |= (unsigned)yy << (4 * u);
        xmask |= ((unsigned)(xx - pad) < (unsigned)iw ? 1u : 0u) << u;
    }
    typedef float Item[CW_PI];                                                // one channel of the quad for the items of a batch
    Item qa0, qa1, qa2, qa3, qb0, qb1, qb2, qb3;                              // two batches in flight: fetched two passes before they are committed
#define QA qa0, qa1, qa2, qa3
#define QB qb0, qb1, qb2, qb3
    // The loaded values are NOT touched in fetch: anything that consumes a load result makes the compiler wait for it on the spot, which
    // put the whole memory latency into the "issue" phase (2-4 k cycles per pass in the stamps, here and in the strip kernel).
    auto fetch_of = [&](int im, int ya, Item& q0, Item& q1, Item& q2, Item& q3) {
        const int boff = (int)((((unsigned)im * CONV_CO * ih + ya - pad) * iw - pad) * 4u);   // image, first row of the batch, padding; may wrap out of range
#pragma unroll
        for (int u = 0; u < CW_PI; ++u) {
            const int vo = goff[u] + boff;
            q0[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(src0, vo, 0, 0));
            q1[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(src1, vo, 0, 0));
            q2[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(src2, vo, 0, 0));
            q3[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(src3, vo, 0, 0));
        }
    };
    auto fetch = [&](int ya, Item& q0, Item& q1, Item& q2, Item& q3) { fetch_of(n, ya, q0, q1, q2, q3); };
    auto commit = [&](int ya, int yb, const Item& q0, const Item& q1, const Item& q2, const Item& q3) {
        const int nrows = yb - ya, nvalid = (nrows * sw - jl + 31) >> 5;      // items u < nvalid lie inside the batch
        const int lbase = (ya % rb) * sw * PIXB;
        unsigned okm = ~0u;                                                  // bit u: the item is a pixel of the map, not padding
        if constexpr (MASK) {                                                // padding columns, and rows outside the map
            unsigned rowin = 0;
            for (int k = 0; k < nrows; ++k) rowin |= ((unsigned)(ya + k - pad) < (unsigned)ih ? 1u : 0u) << k;
            okm = 0;
#pragma unroll
            for (int u = 0; u < CW_PI; ++u) okm |= ((rowin >> ((ypk >> (4 * u)) & 15u)) & 1u) << u;
            okm &= xmask;
        }
#pragma unroll
        for (int u = 0; u < CW_PI; ++u) {
            unsigned o = (unsigned)(loff[u] + lbase);
            const unsigned ow_ = o - (unsigned)ring_b;
            o = o < ow_ ? o : ow_;                                           // slot wrap: o >= ring_b  ->  o - ring_b
            o = u < nvalid ? o : (unsigned)toff;
            cf32x2 v = {q0[u], q1[u]}, w = {q2[u], q3[u]};
            if constexpr (MASK) {
                const bool ok = (okm >> u) & 1u;
                v[0] = ok ? v[0] : 0.f; v[1] = ok ? v[1] : 0.f; w[0] = ok ? w[0] : 0.f; w[1] = ok ? w[1] : 0.f;
            }
            const cbf16x2 hv = __builtin_convertvector(v, cbf16x2), hw_ = __builtin_convertvector(w, cbf16x2);
            *reinterpret_cast<uint2*>(ring + o) = make_uint2(__builtin_bit_cast(unsigned, hv), __builtin_bit_cast(unsigned, hw_));
            if constexpr (X3) {
                const cf32x2 rv = v - __builtin_convertvector(hv, cf32x2), rw = w - __builtin_convertvector(hw_, cf32x2);
                const cbf16x2 lv = __builtin_convertvector(rv, cbf16x2), lw = __builtin_convertvector(rw, cbf16x2);
                *reinterpret_cast<uint2*>(ring + o + 80) = make_uint2(__builtin_bit_cast(unsigned, lv), __builtin_bit_cast(unsigned, lw));
                if constexpr (X6) {
                    const cbf16x2 tv = __builtin_convertvector(rv - __builtin_convertvector(lv, cf32x2), cbf16x2);
                    const cbf16x2 tw = __builtin_convertvector(rw - __builtin_convertvector(lw, cf32x2), cbf16x2);
                    *reinterpret_cast<uint2*>(ring + o + 160) = make_uint2(__builtin_bit_cast(unsigned, tv), __builtin_bit_cast(unsigned, tw));
                }
            }
        }
    };
    // ---- consumers: wave w multiplies pixels p0 + 32 w .. + 31 of the pass ---------------------------------------------------------------
    // A lane's A-operand pixel advances by 128 per pass: its row, column and ring slot are kept incrementally (one division at the start). The
    // barrier of a pass sits right behind its k-loop — the stores, the bias/ReLU/mask arithmetic and the next pass's addresses do not touch the
    // ring, so they run while the producers already refill it.
    const float bv = bias ? bias[col] : 0.f;
    const int adv_r = CW_PASS / ow, adv_x = CW_PASS - adv_r * ow;               // 128 pixels on = adv_r rows and adv_x columns (one carry)
    int cr, cx, cs;                                                            // row, column, slot of the lane's pixel in the current pass
    auto first_pixel = [&]() {
        const int pa = t0 * CW_PASS + 32 * cw + col;
        cr = pa / ow; cx = pa - cr * ow; cs = cr % rb;
    };
    first_pixel();
    const int last_r = (npix - 1) / ow, last_x = npix - 1 - last_r * ow, last_s = last_r % rb;      // tail lanes recompute the last pixel
    unsigned ab[3];                                                            // ring byte offsets of the lane's pixel under tap rows 0..2
    auto prep = [&](int pass) {
        const bool in = pass * CW_PASS + 32 * cw + col < npix;
        const int x = in ? cx : last_x;
        int s0 = in ? cs : last_s;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            ab[d] = (unsigned)((s0 * sw + x) * PIXB + 16 * kg);
            s0 = s0 + 1 >= rb ? s0 + 1 - rb : s0 + 1;
        }
        cx += adv_x; cr += adv_r; cs += adv_r;                                 // the pixel of the pass after
        if (cx >= ow) { cx -= ow; ++cr; ++cs; }
        cs = cs >= rb ? cs - rb : cs;
    };
    // Two accumulator sets: while pass t accumulates into one, the finished sums of pass t - 1 in the other are combined, clamped, masked and
    // stored BETWEEN the k-steps of pass t (the wave's own fillers beside its MFMAs are nearly free; as a block behind the k-loop they were
    // 1.1-1.5 k cycles per pass on a wave that has its SIMD's matrix pipe to itself — 25-40 % on top of the k-loop).
    cf32x16 a0, a1, a2, b0, b1, b2;
    float4 mq[4];
    auto mask_load = [&](int pass) {
        if constexpr (MASK) {
            const int pq = pass * CW_PASS + 32 * cw + 4 * kg;
            const float* mrow = mask + ((int64_t)n * CONV_CO + col) * npix;
            if (pass + 1 < npass) {
#pragma unroll
                for (int q = 0; q < 4; ++q) mq[q] = *reinterpret_cast<const float4*>(mrow + pq + 8 * q);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int pp = pq + 8 * q;
                    mq[q].x = pp < npix ? mrow[pp] : 1.f; mq[q].y = pp + 1 < npix ? mrow[pp + 1] : 1.f;
                    mq[q].z = pp + 2 < npix ? mrow[pp + 2] : 1.f; mq[q].w = pp + 3 < npix ? mrow[pp + 3] : 1.f;
                }
            }
        }
    };
    float ov[4];
    auto out_reg = [&](int i, const cf32x16& p0_, const cf32x16& p1_, const cf32x16& p2_) {          // output register i of a finished pass
        float v = X6 ? (p2_[i] + p1_[i]) + p0_[i] : X3 ? p1_[i] + p0_[i] : p0_[i];
        if (relu) v = fmaxf(v, 0.f);
        if constexpr (MASK) {
            const float m = (i & 3) == 0 ? mq[i >> 2].x : (i & 3) == 1 ? mq[i >> 2].y : (i & 3) == 2 ? mq[i >> 2].z : mq[i >> 2].w;
            v = m > 0.f ? v : 0.f;
        }
        ov[i & 3] = v;
    };
    auto out_store = [&](int pass, int q, bool full) {
        const int pp = pass * CW_PASS + 32 * cw + 4 * kg + 8 * q;
        float* orow = out + ((int64_t)n * CONV_CO + col) * npix;
        if (full || pp + 3 < npix) *reinterpret_cast<float4*>(orow + pp) = make_float4(ov[0], ov[1], ov[2], ov[3]);
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (pp + e < npix) orow[pp + e] = ov[e];
        }
    };
    // k-loop of `pass` into (c0, c1, c2); PREV: the sums of pass - 1 in (p0_, p1_, p2_) leave between the k-steps (its mask was loaded by then)
    auto kloop = [&](int pass, auto prev_tag, cf32x16& c0, cf32x16& c1, cf32x16& c2, const cf32x16& p0_, const cf32x16& p1_, const cf32x16& p2_) {
        constexpr bool PREV = decltype(prev_tag)::value;
#pragma unroll
        for (int i = 0; i < 16; ++i) { c0[i] = bv; c1[i] = 0.f; c2[i] = 0.f; }             // the bias rides in the hi*hi accumulator
        cbf16x8 fa[3][3], fb[3][3];            // fragments of three k-steps: the reads run TWO steps ahead (a step of the two-plane mode is 3 MFMAs =
        auto ld = [&](int s, int b) {          // 96 cycles, less than an LDS round trip under load; one step ahead left a stall per step)
            const unsigned char* o = ring + ab[(s >> 1) / 3] + ((s >> 1) % 3) * PIXB + (s & 1) * 32;
            fa[b][0] = *reinterpret_cast<const cbf16x8*>(o);
            fb[b][0] = wh[s * 64 + lane];
            if constexpr (X3) {
                fa[b][1] = *reinterpret_cast<const cbf16x8*>(o + 80);
                fb[b][1] = wh[(18 + s) * 64 + lane];
            }
            if constexpr (X6) {
                fa[b][2] = *reinterpret_cast<const cbf16x8*>(o + 160);
                fb[b][2] = wh[(36 + s) * 64 + lane];
            }
        };
        ld(0, 0);
        ld(1, 1);
#pragma unroll
        for (int s = 0; s < 18; ++s) {
            const int b = s % 3;
            if (s + 2 < 18) ld(s + 2, (s + 2) % 3);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[b][0], fb[b][0], c0, 0, 0, 0);
            if constexpr (PREV) {
                if (s >= 1 && s <= 16) {
                    out_reg(s - 1, p0_, p1_, p2_);
                    if (((s - 1) & 3) == 3) out_store(pass - 1, (s - 1) >> 2, true);
                }
            }
            if constexpr (X3) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[b][0], fb[b][1], c1, 0, 0, 0);
            if constexpr (X6) c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[b][0], fb[b][2], c2, 0, 0, 0);
            if constexpr (X3) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[b][1], fb[b][0], c1, 0, 0, 0);
            if constexpr (X6) {
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[b][2], fb[b][0], c2, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[b][1], fb[b][1], c2, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);          // keep the reads one step ahead, not eighteen (the scheduler hoisted them all and spilled)
        }
    };
    auto epilogue = [&](int pass, const cf32x16& p0_, const cf32x16& p1_, const cf32x16& p2_) {      // the last pass of the workgroup
        mask_load(pass);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            out_reg(i, p0_, p1_, p2_);
            if ((i & 3) == 3) out_store(pass, i >> 2, pass + 1 < npass);
        }
    };
    // STAMP (diagnostic, tuning bit 8192): consumer wave 0 {pass work, barrier wait, -, -, -, -, whole kernel, prologue} and the first producer wave
    // {even passes: commit, fetch issue, barrier; odd passes: commit + fetch, barrier; -, whole kernel, prologue} into g_conv_stamps, summed over the passes
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto now = [&]() -> unsigned long long {
        if constexpr (STAMP) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); return __builtin_amdgcn_s_memtime(); }
        return 0ull;
    };
    const unsigned long long t_begin = now();
    // The two roles run SEPARATE loops with the same number of barriers (one after the prologue, one per pass). In one shared loop the
    // compiler's wait-count analysis merges the roles at the loop head: the consumers then wait out their own stores (vmcnt(0)) in front of
    // every pass because their registers alias the producers' pending loads, and the producers cannot tell which of their two batches is the
    // older one. The producer loop is unrolled by two for the same reason: batch A is always the older one at its commit, then batch B.
    if constexpr (PERSIST) {
        // launched with gridDim.y == 1 (t0 = 0, t1 = npass) and a first pass whose rows fit two batches (host-checked)
        const int G = gridDim.x, need = end_row(0), ym = chunk < need ? chunk : need;
        const bool odd = npass & 1;
        if (producer) {
            // Past pass npass - 3 the loop's fetches have no rows of this image left to get: row-batch index k = npass and npass + 2 now mean "the
            // next image's first batch", k = npass + 1 its second. With the loop's fixed alternation of the two register sets that leaves the
            // first batch in set A when npass is odd and in set B when it is even — the window below commits them in that order.
            auto fetch_k = [&](int k, Item& q0, Item& q1, Item& q2, Item& q3) {
                const bool nxt = k >= npass;
                fetch_of(nxt ? n + G : n, nxt ? (k == npass + 1 ? ym : 0) : end_row(k - 1), q0, q1, q2, q3);
            };
            if (odd) { fetch(0, QA); fetch(ym, QB); } else { fetch(0, QB); fetch(ym, QA); }
            for (; n < nimg; n += G) {
                const unsigned long long tw = now();
                if (odd) { commit(0, ym, QA); commit(ym, need, QB); } else { commit(0, ym, QB); commit(ym, need, QA); }
                fetch(end_row(0), QA);
                fetch(end_row(1), QB);
                __syncthreads();
                st[7] += now() - tw;
                int t = 0;
                for (; t + 1 < npass; t += 2) {
                    const unsigned long long ta = now();
                    commit(end_row(t), end_row(t + 1), QA);
                    const unsigned long long tb = now();
                    fetch_k(t + 3, QA);
                    const unsigned long long tc = now();
                    __syncthreads();
                    const unsigned long long td = now();
                    commit(end_row(t + 1), end_row(t + 2), QB);
                    fetch_k(t + 4, QB);
                    const unsigned long long te = now();
                    __syncthreads();
                    if constexpr (STAMP) { st[0] += tb - ta; st[1] += tc - tb; st[2] += td - tc; st[3] += te - td; st[4] += now() - te; }
                }
                if (t < npass) __syncthreads();
            }
        } else {
            for (; n < nimg; n += G) {
                const unsigned long long tw = now();
                first_pixel();
                prep(0);
                __syncthreads();
                st[7] += now() - tw;
                unsigned long long ta = now();
                kloop(0, std::false_type{}, a0, a1, a2, a0, a1, a2);
                unsigned long long tc = now();
                __syncthreads();
                prep(1);
                if constexpr (STAMP) { st[0] += tc - ta; st[1] += now() - tc; }
                int t = 1;
                for (; t + 1 < npass; t += 2) {
                    ta = now();
                    mask_load(t - 1);
                    kloop(t, std::true_type{}, b0, b1, b2, a0, a1, a2);
                    tc = now();
                    __syncthreads();
                    prep(t + 1);
                    const unsigned long long td = now();
                    mask_load(t);
                    kloop(t + 1, std::true_type{}, a0, a1, a2, b0, b1, b2);
                    const unsigned long long te = now();
                    __syncthreads();
                    prep(t + 2);
                    if constexpr (STAMP) { st[0] += (tc - ta) + (te - td); st[1] += (td - tc) + (now() - te); }
                }
                if (t < npass) {
                    mask_load(t - 1);
                    kloop(t, std::true_type{}, b0, b1, b2, a0, a1, a2);
                    __syncthreads();
                    epilogue(t, b0, b1, b2);
                } else epilogue(t - 1, a0, a1, a2);
            }
        }
    } else if (producer) {
        const int ya = (t0 * CW_PASS) / ow, need = end_row(t0);
        const int ym = ya + chunk < need ? ya + chunk : need, yn = ym + chunk < need ? ym + chunk : need;
        fetch(ya, QA);
        if (ym < need) fetch(ym, QB);
        commit(ya, ym, QA);
        if (ym < need) commit(ym, yn, QB);
        for (int y = yn; y < need; y += chunk) {                  // maps so narrow that a pass spans more than two batches
            fetch(y, QA);
            commit(y, y + chunk < need ? y + chunk : need, QA);
        }
        // From here on every fetch and commit is unconditional: past the last pass a batch has no rows (end_row stops growing), its loads fall
        // on the next image or out of the descriptor's range (answered with 0) and its stores on the dummy words. With conditions, the
        // paths that skip a fetch made the compiler treat the batch being committed as the youngest one in flight — vmcnt(23..0) instead of
        // vmcnt(47..24): every commit drained the batch fetched just before the barrier.
        fetch(end_row(t0), QA);
        fetch(end_row(t0 + 1), QB);
        __syncthreads();
        st[7] = now() - t_begin;
        int t = t0;
        for (; t + 1 < t1; t += 2) {
            const unsigned long long ta = now();
            if constexpr (STAMP) { asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); st[5] += now() - ta; }      // the wait for batch A alone
            commit(end_row(t), end_row(t + 1), QA);                // the rows pass t + 1 adds, into slots pass t - 1 released at the last barrier
            const unsigned long long tb = now();
            fetch(end_row(t + 2), QA);
            const unsigned long long tc = now();
            __syncthreads();
            const unsigned long long td = now();
            commit(end_row(t + 1), end_row(t + 2), QB);
            fetch(end_row(t + 3), QB);
            const unsigned long long te = now();
            __syncthreads();
            if constexpr (STAMP) { st[0] += tb - ta; st[1] += tc - tb; st[2] += td - tc; st[3] += te - td; st[4] += now() - te; }
        }
        if (t < t1) __syncthreads();                               // the last pass of an odd count: nothing left to stage
    } else {
        prep(t0);
        __syncthreads();
        st[7] = now() - t_begin;
        // pass t0 into set a; then pairs (b with a leaving, a with b leaving); the mask of pass t - 1 is fetched at the head of pass t's k-loop
        unsigned long long ta = now();
        kloop(t0, std::false_type{}, a0, a1, a2, a0, a1, a2);          // no previous pass: the last three arguments are not read
        unsigned long long tc = now();
        __syncthreads();
        prep(t0 + 1);
        if constexpr (STAMP) { st[0] += tc - ta; st[1] += now() - tc; }
        int t = t0 + 1;
        for (; t + 1 < t1; t += 2) {
            ta = now();
            mask_load(t - 1);
            kloop(t, std::true_type{}, b0, b1, b2, a0, a1, a2);
            tc = now();
            __syncthreads();
            prep(t + 1);
            const unsigned long long td = now();
            mask_load(t);
            kloop(t + 1, std::true_type{}, a0, a1, a2, b0, b1, b2);
            const unsigned long long te = now();
            __syncthreads();
            prep(t + 2);
            if constexpr (STAMP) { st[0] += (tc - ta) + (te - td); st[1] += (td - tc) + (now() - te); }
        }
        if (t < t1) {
            mask_load(t - 1);
            kloop(t, std::true_type{}, b0, b1, b2, a0, a1, a2);
            __syncthreads();
            epilogue(t, b0, b1, b2);
        } else epilogue(t - 1, a0, a1, a2);
    }
    if constexpr (STAMP) {
        st[6] = now() - t_begin;
        if ((wave == 0 || wave == 4) && lane == 0 && blockIdx.x < 1024 && blockIdx.y == 0)
            for (int i = 0; i < 8; ++i) g_conv_stamps[(blockIdx.x * 2 + (producer ? 1 : 0)) * 8 + i] = st[i];
    }
}

#undef QA
#undef QB

// wave-specialised kernel: rows of the circular buffer = the rows two consecutive passes span; its budgets
static int conv3x3_ws_rows(int ow) { return (2 * CW_PASS + ow - 2) / ow + 1 + 2; }
static size_t conv3x3_ws_lds(int ow, int npl) {
    const size_t ring = (size_t)conv3x3_ws_rows(ow) * (ow + 2) * cw_pixb(npl);
    return ((ring + CW_DUMMY + 15) & ~(size_t)15) + (size_t)npl * 18 * 64 * 16;
}
static bool conv3x3_ws_fits(int oh, int ow, int npl) {
    const int sw = ow + 2, npix = oh * ow;
    if (sw > CM_MAXW || npix < 1) return false;
    const int chunk = (32 * CW_PI) / sw, npass = (npix + CW_PASS - 1) / CW_PASS;
    auto end_row = [&](int pass) { return ((pass * CW_PASS + CW_PASS < npix ? pass * CW_PASS + CW_PASS : npix) - 1) / ow + 3; };
    for (int t = 0; t + 1 < npass; ++t)
        if (end_row(t + 1) - end_row(t) > chunk) return false;                    // a pass adds more rows than one producer batch holds
    return chunk >= 1 && chunk <= 15 && conv3x3_ws_lds(ow, npl) <= 160 * 1024;
}
// true when the strip of a 256-pixel pass fits the kernel's fixed budgets
static bool conv3x3_mfma_fits(int oh, int ow, int prec = EXORL_PREC_BF16X3) {
    const int rows = (CM_PASS + ow - 2) / ow + 1 + 2;         // most output rows 256 consecutive pixels can touch, + 2
    const int npl = prec == EXORL_PREC_BF16X6 ? 3 : 2;
    const size_t lds = (size_t)npl * round_up((int64_t)rows * (ow + 2) * CM_PIX, 8) * sizeof(unsigned short) + (size_t)npl * 18 * 64 * 16;
    return ow + 2 <= CM_MAXW && 16 * rows * (ow + 2) <= CM_IPT * CM_THREADS && oh * ow >= 1 && lds <= 160 * 1024;
}
static int conv3x3_mfma(const float* in, const float* Wt, const float* bias, const float* mask, float* out, int n, int ih, int iw, int oh, int ow,
                        int pad, int relu, int prec, hipStream_t s) {
    const int npl = prec == EXORL_PREC_BF16X6 ? 3 : (prec == EXORL_PREC_BF16X3 ? 2 : 1);
    const int rows = (CM_PASS + ow - 2) / ow + 1 + 2;
    const int plane = (int)round_up((int64_t)rows * (ow + 2) * CM_PIX, 8);
    const size_t lds = (size_t)npl * plane * sizeof(unsigned short) + (size_t)npl * 18 * 64 * 16;
    EXORL_REQUIRE(lds <= 160 * 1024, "conv3x3_mfma: strip of %d rows x %d columns (%d planes) does not fit LDS", rows, ow + 2, npl);
    static bool attr = false;
    if (!attr) {
        EXORL_CHECK_HIP(hipFuncSetAttribute((const void*)conv3x3_mfma_kernel<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        EXORL_CHECK_HIP(hipFuncSetAttribute((const void*)conv3x3_mfma_kernel<3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        EXORL_CHECK_HIP(hipFuncSetAttribute((const void*)conv3x3_mfma_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        EXORL_CHECK_HIP(hipFuncSetAttribute((const void*)conv3x3_mfma_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        EXORL_CHECK_HIP(hipFuncSetAttribute((const void*)conv3x3_mfma_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        EXORL_CHECK_HIP(hipFuncSetAttribute((const void*)conv3x3_mfma_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    // the wave-specialised kernel: forward launches (no mask, no padding) and dgrad launches (mask and padding); 32-bit buffer offsets.
    // exorl_gemm_tune bit 1073741824 keeps the strip kernel (A/B)
    if (!(tune_variant() & 1073741824) && conv3x3_ws_fits(oh, ow, npl) && (int64_t)n * CONV_CO * ih * iw * 4 < (1ll << 31) && ((mask != nullptr) == (pad != 0))) {
        const int rb = conv3x3_ws_rows(ow);
        const size_t wlds = conv3x3_ws_lds(ow, npl);
        static bool wattr = false;
        static int ncu = 256;
        if (!wattr) {
#define EXORL_WA(NN, MM, SS, PP) EXORL_CHECK_HIP(hipFuncSetAttribute((const void*)conv3x3_ws_kernel<NN, MM, SS, PP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
            EXORL_WA(3, true, false, false); EXORL_WA(3, false, false, false); EXORL_WA(2, true, false, false); EXORL_WA(2, false, false, false);
            EXORL_WA(1, true, false, false); EXORL_WA(1, false, false, false);
            EXORL_WA(3, true, false, true); EXORL_WA(3, false, false, true); EXORL_WA(2, true, false, true); EXORL_WA(2, false, false, true);
            EXORL_WA(1, true, false, true); EXORL_WA(1, false, false, true);
            EXORL_WA(3, false, true, true); EXORL_WA(2, false, true, true);
#undef EXORL_WA
            int dev = 0, v = 0;
            EXORL_CHECK_HIP(hipGetDevice(&dev));
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
            wattr = true;
        }
        const int wpass = (oh * ow + CW_PASS - 1) / CW_PASS, wgy = n <= 8 ? wpass : 1;
        const int wflags = (tune_variant() & 32) ? 2 : 0;          // experiment: the younger half of the workgroup stages
        // persistent form: one workgroup per CU walks the images; needs the first pass's rows in two batches. exorl_gemm_tune bit 8388608: one image per workgroup (A/B)
        const int wchunk = (32 * CW_PI) / (ow + 2), wneed = ((CW_PASS < oh * ow ? CW_PASS : oh * ow) - 1) / ow + 3;
        const bool persist = wgy == 1 && wneed <= 2 * wchunk && !(npl == 3 && mask) && !(tune_variant() & 8388608);      // (three planes + mask: 256 VGPRs and a spill)
        const int gx = persist ? (n < ncu ? n : ncu) : n;
#define EXORL_CW(NN, MM, SS) do { \
            if (persist) hipLaunchKernelGGL((conv3x3_ws_kernel<NN, MM, SS, true>), dim3(gx, 1), dim3(CM_THREADS), wlds, s, in, Wt, bias, mask, out, ih, iw, oh, ow, pad, relu, rb, wflags, n); \
            else hipLaunchKernelGGL((conv3x3_ws_kernel<NN, MM, false, false>), dim3(n, wgy), dim3(CM_THREADS), wlds, s, in, Wt, bias, mask, out, ih, iw, oh, ow, pad, relu, rb, wflags, n); \
        } while (0)
        if ((tune_variant() & 8192) && !mask && npl >= 2 && persist) {          // diagnostic: the stamped build (persistent form)
            if (npl == 3) EXORL_CW(3, false, true); else EXORL_CW(2, false, true);
        } else if (npl == 3) { if (mask) EXORL_CW(3, true, false); else EXORL_CW(3, false, false); }
        else if (npl == 2) { if (mask) EXORL_CW(2, true, false); else EXORL_CW(2, false, false); }
        else               { if (mask) EXORL_CW(1, true, false); else EXORL_CW(1, false, false); }
#undef EXORL_CW
        EXORL_LAUNCH_CHECK();
        return 0;
    }
    // a handful of images (act(): one) cannot fill the chip with one workgroup each: spread an image's 256-pixel passes over workgroups
    const int npass = (oh * ow + CM_PASS - 1) / CM_PASS, gy = n <= 8 ? npass : 1;
#define EXORL_CM(NN, MM) hipLaunchKernelGGL((conv3x3_mfma_kernel<NN, MM>), dim3(n, gy), dim3(CM_THREADS), lds, s, in, Wt, bias, mask, out, ih, iw, oh, ow, pad, relu, plane)
    if ((tune_variant() & 8192) && !mask && npl >= 2) {          // diagnostic: the stamped build of the forward kernel
        static bool sattr = false;
        if (!sattr) {
            EXORL_CHECK_HIP(hipFuncSetAttribute((const void*)conv3x3_mfma_kernel<3, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            EXORL_CHECK_HIP(hipFuncSetAttribute((const void*)conv3x3_mfma_kernel<2, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            sattr = true;
        }
        if (npl == 3) hipLaunchKernelGGL((conv3x3_mfma_kernel<3, false, true>), dim3(n, gy), dim3(CM_THREADS), lds, s, in, Wt, bias, mask, out, ih, iw, oh, ow, pad, relu, plane);
        else hipLaunchKernelGGL((conv3x3_mfma_kernel<2, false, true>), dim3(n, gy), dim3(CM_THREADS), lds, s, in, Wt, bias, mask, out, ih, iw, oh, ow, pad, relu, plane);
        EXORL_LAUNCH_CHECK();
        return 0;
    }
    if (npl == 3)      { if (mask) EXORL_CM(3, true); else EXORL_CM(3, false); }
    else if (npl == 2) { if (mask) EXORL_CM(2, true); else EXORL_CM(2, false); }
    else               { if (mask) EXORL_CM(1, true); else EXORL_CM(1, false); }
#undef EXORL_CM
    EXORL_LAUNCH_CHECK();
    return 0;
}

// prec: EXORL_PREC_F32 -> direct fp32 FMA kernel for every layer; bf16 / split-bf16 -> the 32-channel stride-1 layers on MFMA
static int conv3x3(const float* in, const float* Wt, const float* bias, const float* mask, float* out, int n, int ci_n, int co_n, int ih, int iw,
                   int oh, int ow, int stride, int pad, int in_scale, int relu, hipStream_t s, int prec = EXORL_PREC_F32, const float* frag = nullptr) {
    if (prec != EXORL_PREC_F32 && ci_n == CONV_CO && co_n == CONV_CO && stride == 1 && !in_scale && frag && conv3x3_mfma_fits(oh, ow, prec))
        return conv3x3_mfma(in, frag, bias, mask, out, n, ih, iw, oh, ow, pad, relu, prec, s);
    if (stride == 2 && pad == 0 && !mask && co_n == CONV_CO && ow >= 8 && !(tune_variant() & 16)) {      // first layer forward; exorl_gemm_tune bit 16: the tile kernel (A/B)
        const int max_rows = 2 * ((C1_PASS + ow - 2) / ow) + 3;
        const size_t slds = (size_t)ci_n * max_rows * 2 * ((iw + 1) / 2) * sizeof(float);
        if (slds <= 160 * 1024) {
            static bool sattr = false;
            if (!sattr) {
                EXORL_CHECK_HIP(hipFuncSetAttribute((const void*)conv1_strip_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                sattr = true;
            }
            hipLaunchKernelGGL(conv1_strip_kernel, dim3(cdiv(oh * ow, C1_PASS), n), dim3(256), slds, s, in, Wt, bias, out, ci_n, ih, iw, oh, ow, in_scale,
                               relu, max_rows);
            EXORL_LAUNCH_CHECK();
            return 0;
        }
    }
    const int tin = (CONV_TILE - 1) * stride + 3;
    const size_t lds = (size_t)ci_n * tin * tin * sizeof(float);
    EXORL_REQUIRE(lds <= 160 * 1024 && co_n == CONV_CO, "conv3x3: tile does not fit LDS (ci=%d stride=%d) or co=%d != 32", ci_n, stride, co_n);
    static bool attr = false;
    if (!attr) {
        EXORL_CHECK_HIP(hipFuncSetAttribute((const void*)conv3x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    hipLaunchKernelGGL(conv3x3_kernel, dim3(cdiv(oh, CONV_TILE) * cdiv(ow, CONV_TILE), n), dim3(256), lds, s, in, Wt, bias, mask, out, ci_n,
                       ih, iw, oh, ow, stride, pad, in_scale, relu);
    EXORL_LAUNCH_CHECK();
    return 0;
}

// ---- wgrad: P[n][co][ci][tap] = sum_{y,x} dy[n][co][y][x] * in[n][ci][y*s + ky][x*s + kx];  Pb[n][co] = sum dy -------------------------
constexpr int WG_ROWS = 6;            // output rows per LDS tile
__global__ __launch_bounds__(1024) void conv_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ in, float* __restrict__ P,
                                                          float* __restrict__ Pb, int ci_n, int ih, int iw, int oh, int ow, int stride,
                                                          int in_scale) {
    extern __shared__ float lds[];
    const int n = blockIdx.x;
    // thread = (co, ci, pixel slice): with few input channels (the first layer: 3 or 9) the 32 lanes of an output channel split the
    // pixel columns between them (x = slice, slice + S, ...) instead of idling, and the slices are summed by shuffles at the end
    int ci2 = 1;
    while (ci2 < ci_n) ci2 <<= 1;
    const int S = 32 / ci2;
    const int co = threadIdx.x >> 5, ci = (threadIdx.x & 31) % ci2, slice = (threadIdx.x & 31) / ci2;
    const int in_rows = (WG_ROWS - 1) * stride + 3;
    float* dyt = lds;                                   // [WG_ROWS * ow][32 co]
    float* it = lds + WG_ROWS * ow * CONV_CO;           // [in_rows * iw][ci_n]
    float acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = 0.f;
    float bsum = 0.f;
    const float* dyn = dy + (int64_t)n * CONV_CO * oh * ow;
    const float* inn = in + (int64_t)n * ci_n * ih * iw;
    for (int r0 = 0; r0 < oh; r0 += WG_ROWS) {
        const int nr = oh - r0 < WG_ROWS ? oh - r0 : WG_ROWS;
        __syncthreads();
        for (int i = threadIdx.x; i < nr * ow * CONV_CO; i += 1024) {
            const int p = i % (nr * ow), c2 = i / (nr * ow);                  // coalesced over pixels of one channel
            dyt[p * CONV_CO + c2] = dyn[(int64_t)c2 * oh * ow + (int64_t)r0 * ow + p];
        }
        const int iy0 = r0 * stride, rows_in = (nr - 1) * stride + 3;
        for (int i = threadIdx.x; i < rows_in * iw * ci_n; i += 1024) {
            const int p = i % (rows_in * iw), c2 = i / (rows_in * iw);
            float v = inn[(int64_t)c2 * ih * iw + (int64_t)iy0 * iw + p];
            if (in_scale) v = v / 255.0f - 0.5f;
            it[p * ci_n + c2] = v;
        }
        __syncthreads();
        if (ci < ci_n) {
            for (int y = 0; y < nr; ++y)
                for (int x = slice; x < ow; x += S) {
                    const float d = dyt[(y * ow + x) * CONV_CO + co];
                    const float* ip = it + ((y * stride) * iw + x * stride) * ci_n + ci;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] += d * ip[(ky * iw + kx) * ci_n];
                    if (ci == 0) bsum += d;
                }
        }
    }
    (void)in_rows;
    for (int off = ci2; off < 32; off <<= 1) {           // sum the pixel slices (lanes of one 32-lane group)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] += __shfl_xor(acc[t], off, 64);
        bsum += __shfl_xor(bsum, off, 64);
    }
    if (ci < ci_n && slice == 0) {
        float* Pn = P + ((int64_t)n * CONV_CO + co) * ci_n * 9 + ci * 9;
#pragma unroll
        for (int t = 0; t < 9; ++t) Pn[t] = acc[t];
        if (ci == 0) Pb[(int64_t)n * CONV_CO + co] = bsum;
    }
}

// ---- weight gradients of the 32 -> 32 layers on the matrix cores (bf16 / split-bf16 modes) -------------------------------------
// dW[co][ci][tap] = sum over pixels p of dY[co][p] X[ci][p + tap]: per tap a 32 x 32 GEMM with K = pixels. One workgroup per
// image, nine waves = nine taps (+ the bias gradient on the ninth wave as a product with a ones operand); per 16 x 16 output tile
// the k16 steps are its 16 pixel rows. Both operands are staged channel-major (as they sit in the NCHW maps): dY[co][y][16 x] gives
// the A fragment (8 consecutive x of one channel) as one aligned ds_read_b128; X gets three copies shifted by the tap's dx so that
// the B fragment (8 consecutive x + dx of one input channel) is aligned too. Per-image partials in the layout of the fp32 kernel,
// summed in image order by the same column-sum kernel.
constexpr int WM_THREADS = 512;            // 8 waves (256 VGPRs each): taps 0..7, the last wave also takes tap 8; wave 0 the bias gradient

// NPL operand planes (1 bf16, 2 split-bf16, 3 three-plane split); TH = tile height in output rows: 16, or 8 for three planes — dY and the three
// shifted copies of X take 74 KB of LDS per plane at 16 x 16 (three planes: 221 KB) and 41 KB at 8 x 16 (123 KB)
template <int NPL, int TH>
__global__ __launch_bounds__(WM_THREADS) void conv_wgrad_mfma_kernel(const float* __restrict__ dy, const float* __restrict__ in, float* __restrict__ P,
                                                                     float* __restrict__ Pb, int ih, int iw, int oh, int ow) {
    constexpr bool X3 = NPL >= 2, X6 = NPL == 3;
    constexpr int TW = CONV_TILE;                              // tile width (16 output columns)
    constexpr int DYC = TH * TW + 8;                           // bf16 per dY channel (+16 B: conflict-free across 16 channels)
    constexpr int XR = TH + 2;                                 // X rows of a tile
    constexpr int XC = XR * TW + 8;                            // bf16 per X channel of one shifted copy
    constexpr int DPL = CONV_CO * DYC, XPL = 3 * CONV_CO * XC; // elements per plane
    extern __shared__ __attribute__((aligned(16))) unsigned char wm_lds[];
    unsigned short* dbase = reinterpret_cast<unsigned short*>(wm_lds);            // [plane][32][DYC]
    unsigned short* xbase = dbase + NPL * DPL;                                    // [plane][3 dx][32][XC]
    const int n = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, tap = tid >> 6;                  // wave = tap (wave 7: taps 7 and 8)
    const int kg = lane >> 5, col = lane & 31;
    const int dyy = tap / 3, dxx = tap % 3;
    const float* dyn = dy + (int64_t)n * CONV_CO * oh * ow;
    const float* inn = in + (int64_t)n * CONV_CO * ih * iw;
    cf32x16 acc, accx, accy, accb, acc8, accx8, accy8;          // accb: bias gradient (wave 0); acc8 / accx8 / accy8: tap 8 (wave 7)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accx[r] = 0.f; accy[r] = 0.f; accb[r] = 0.f; acc8[r] = 0.f; accx8[r] = 0.f; accy8[r] = 0.f; }
    cbf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
    auto pack = [](float a, float b, int plane) -> unsigned {     // plane 0: bf16(x); 1: bf16(x - p0); 2: bf16(x - p0 - p1)
        for (int q = 0; q < plane; ++q) { a -= (float)(__bf16)a; b -= (float)(__bf16)b; }
        const __bf16 h0 = (__bf16)a, h1 = (__bf16)b;
        return (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
    };
    const int tiles_x = (ow + TW - 1) / TW, tiles_y = (oh + TH - 1) / TH;
    for (int tile = 0; tile < tiles_x * tiles_y; ++tile) {
        const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
        __syncthreads();
        // dY tile: (co, y, x pair); X tile, three copies: copy dx holds columns tx0 + dx .. tx0 + dx + 15 of rows ty0 .. ty0 + TH + 1:
        // (dx, ci, yy, x pair). Loads are issued in batches before any value is converted and stored, so that their latencies overlap.
        // (Tried: fetching dY and half of X for the NEXT tile ahead of this tile's MFMA loop and the rest in two batches: 256 VGPRs instead of
        // 176, and 609 us per launch instead of 430 — not kept.)
        constexpr int DY_ITEMS = CONV_CO * TH * (TW / 2), DY_PT = DY_ITEMS / WM_THREADS;
        constexpr int X_ITEMS = 3 * CONV_CO * XR * (TW / 2), X_PT = X_ITEMS / WM_THREADS;
        static_assert(DY_ITEMS % WM_THREADS == 0 && X_ITEMS % WM_THREADS == 0 && X_PT % 3 == 0, "staging items divide evenly");
        {
            float va[DY_PT], vb[DY_PT];
#pragma unroll
            for (int u = 0; u < DY_PT; ++u) {
                const int i = tid + WM_THREADS * u;
                const int xp = i % (TW / 2), y = (i / (TW / 2)) % TH, co = i / (TH * TW / 2);
                const int gy = ty0 + y, gx = tx0 + 2 * xp;
                va[u] = (gy < oh && gx < ow) ? dyn[((int64_t)co * oh + gy) * ow + gx] : 0.f;
                vb[u] = (gy < oh && gx + 1 < ow) ? dyn[((int64_t)co * oh + gy) * ow + gx + 1] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < DY_PT; ++u) {
                const int i = tid + WM_THREADS * u;
                const int xp = i % (TW / 2), y = (i / (TW / 2)) % TH, co = i / (TH * TW / 2);
                const int o = co * DYC + y * TW + 2 * xp;
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<unsigned*>(dbase + pl * DPL + o) = pack(va[u], vb[u], pl);
            }
        }
#pragma unroll 1
        for (int part = 0; part < 3; ++part) {
            float va[X_PT / 3], vb[X_PT / 3];
#pragma unroll
            for (int u = 0; u < X_PT / 3; ++u) {
                const int i = tid + WM_THREADS * (part * (X_PT / 3) + u);
                const int xp = i % (TW / 2), yy = (i / (TW / 2)) % XR, ci = (i / (XR * TW / 2)) % CONV_CO, dx = i / (CONV_CO * XR * TW / 2);
                const int gy = ty0 + yy, gx = tx0 + dx + 2 * xp;
                va[u] = (gy < ih && gx < iw) ? inn[((int64_t)ci * ih + gy) * iw + gx] : 0.f;
                vb[u] = (gy < ih && gx + 1 < iw) ? inn[((int64_t)ci * ih + gy) * iw + gx + 1] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < X_PT / 3; ++u) {
                const int i = tid + WM_THREADS * (part * (X_PT / 3) + u);
                const int xp = i % (TW / 2), yy = (i / (TW / 2)) % XR, ci = (i / (XR * TW / 2)) % CONV_CO, dx = i / (CONV_CO * XR * TW / 2);
                const int o = (dx * CONV_CO + ci) * XC + yy * TW + 2 * xp;
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<unsigned*>(xbase + pl * XPL + o) = pack(va[u], vb[u], pl);
            }
        }
        __syncthreads();
#pragma unroll 2
        for (int y = 0; y < TH; ++y) {                           // k16 step = pixel row y of the tile
            const int oa = col * DYC + y * TW + 8 * kg;                                         // A: dY[co = col][y][8 kg ..]
            const int ob = (dxx * CONV_CO + col) * XC + (y + dyy) * TW + 8 * kg;                // B: X[ci = col][y + dy][8 kg + dx ..]
            const cbf16x8 ah = *reinterpret_cast<const cbf16x8*>(dbase + oa);
            const cbf16x8 bh = *reinterpret_cast<const cbf16x8*>(xbase + ob);
            cbf16x8 al = ah, bl = bh, at = ah, bt = bh;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
            if constexpr (X3) {
                al = *reinterpret_cast<const cbf16x8*>(dbase + DPL + oa);
                bl = *reinterpret_cast<const cbf16x8*>(xbase + XPL + ob);
                accx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, accx, 0, 0, 0);
                accx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, accx, 0, 0, 0);
            }
            if constexpr (X6) {
                at = *reinterpret_cast<const cbf16x8*>(dbase + 2 * DPL + oa);
                bt = *reinterpret_cast<const cbf16x8*>(xbase + 2 * XPL + ob);
                accy = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bt, accy, 0, 0, 0);
                accy = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at, bh, accy, 0, 0, 0);
                accy = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bl, accy, 0, 0, 0);
            }
            if (tap == 0) {                                      // wave-uniform: bias gradient = dY . ones (exact in every plane mode: ones is exact)
                accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ones, accb, 0, 0, 0);
                if constexpr (X3) accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, ones, accb, 0, 0, 0);
                if constexpr (X6) accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at, ones, accb, 0, 0, 0);
            }
            if (tap == 7) {                                      // wave-uniform: tap 8 = (dy 2, dx 2)
                const int ob8 = (2 * CONV_CO + col) * XC + (y + 2) * TW + 8 * kg;
                const cbf16x8 bh8 = *reinterpret_cast<const cbf16x8*>(xbase + ob8);
                acc8 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh8, acc8, 0, 0, 0);
                if constexpr (X3) {
                    const cbf16x8 bl8 = *reinterpret_cast<const cbf16x8*>(xbase + XPL + ob8);
                    accx8 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl8, accx8, 0, 0, 0);
                    accx8 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh8, accx8, 0, 0, 0);
                    if constexpr (X6) {
                        const cbf16x8 bt8 = *reinterpret_cast<const cbf16x8*>(xbase + 2 * XPL + ob8);
                        accy8 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bt8, accy8, 0, 0, 0);
                        accy8 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at, bh8, accy8, 0, 0, 0);
                        accy8 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bl8, accy8, 0, 0, 0);
                    }
                }
            }
        }
    }
    // C layout: reg r of lane l = row (r & 3) + 8 (r >> 2) + 4 (l >> 5) (co), column l & 31 (ci)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = (r & 3) + 8 * (r >> 2) + 4 * kg;
        P[(((int64_t)n * CONV_CO + co) * CONV_CO + col) * 9 + tap] = X6 ? (accy[r] + accx[r]) + acc[r] : X3 ? accx[r] + acc[r] : acc[r];
        if (tap == 7) P[(((int64_t)n * CONV_CO + co) * CONV_CO + col) * 9 + 8] = X6 ? (accy8[r] + accx8[r]) + acc8[r] : X3 ? accx8[r] + acc8[r] : acc8[r];
        if (tap == 0 && col == 0) Pb[(int64_t)n * CONV_CO + co] = accb[r];
    }
}

// ---- the same weight gradients on the wave-specialised structure of conv3x3_ws_kernel (round 3) ------------------------------------------
// The tile kernel above stages every 16 x 16 tile three times (one copy per dx), pays for 48 x 48 pixels on a 39 x 39 map and walks its
// phases in lockstep: 430-530 us per launch for the flops the forward pass does in 116-158 us. Here K = the image's pixels in row-major order,
// 16 per MFMA, 128 per pass; per tap dW[ci][co] += X(p + tap)^T dY(p):
//   A (M = ci) comes from the forward kernel's ring of channels-last bf16 planes — staged once, every row once — through the hardware-
//     transposing read (ds_read_b64_tr_b16: 4 pixels x 16 channels per 16-lane group; a position's planes are 64 B apart and positions
//     192 B, so the four pixel rows of a read fall on four disjoint bank quarters);
//   B (N = co) is dY[co][128 pixels] as bf16 planes in a double-buffered LDS image, 272-byte rows (ds_read_b128, conflict-free).
// Waves 0-3 stage (X rows into the ring two batches ahead, dY one pass ahead, and the bias gradient as plain fp32 sums on the way);
// waves 4-7 multiply: consumer w owns taps 2w and 2w + 1 for all eight k16 steps of a pass and tap 8 for steps 2w, 2w + 1 — 18 units of
// NPL-plane products each per pass, the same MFMA count as a forward pass; the four tap-8 partials are added through LDS at the end.
// Split modes keep two accumulators per tap: hi*hi, and every cross term (the cross terms are 2^-8 and 2^-16 of the sum: adding the smaller
// ones into the larger costs 2^-32 of the result).
constexpr int WW_DYP = 272;                 // bytes per dY channel row of the image: 128 pixels x 2 B + 16
__host__ __device__ constexpr int ww_pixb(int npl) { return npl == 1 ? 64 : 192; }
template <int NPL>
__global__ __launch_bounds__(CM_THREADS) void conv_wgrad_ws_kernel(const float* __restrict__ dy, const float* __restrict__ in, float* __restrict__ P,
                                                                   float* __restrict__ Pb, int ih, int iw, int oh, int ow, int rb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cm_lds[];
    constexpr bool X3 = NPL >= 2, X6 = NPL == 3;
    constexpr int PIXB = ww_pixb(NPL);
    const int npix = oh * ow, npass = (npix + CW_PASS - 1) / CW_PASS, sw = ow + 2;
    const int ring_b = rb * sw * PIXB;
    unsigned char* ring = cm_lds;
    unsigned char* dyimg = cm_lds + ((ring_b + CW_DUMMY + 15) & ~15);          // [2 buffers][NPL][32 co][WW_DYP]
    constexpr int DYBUF = NPL * CONV_CO * WW_DYP;
    const int n = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool producer = wave < 4;
    auto end_row = [&](int pass) {
        const int p1 = (pass * CW_PASS + CW_PASS < npix ? pass * CW_PASS + CW_PASS : npix) - 1;
        return p1 / ow + 3;
    };
    typedef float cf32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 cbf16x2 __attribute__((ext_vector_type(2)));
    if (producer) {
        // ---- X rows: as conv3x3_ws_kernel (no padding), with the channel quad as the fast lane index: the 16 lanes of a ds_write_b64 group
        // cover two whole positions (2 x 64 B, 192 B apart) instead of 16 positions 48 dwords apart (8-way on 32 banks)
        const int chunk = (32 * CW_PI) / sw;
        const int ptid = tid, cq = ptid & 7, jl = ptid >> 3;
        const int plane_b = ih * iw * 4;
        const unsigned total_b = (unsigned)gridDim.x * CONV_CO * plane_b;
        float* inq = const_cast<float*>(in);
        const auto src0 = __builtin_amdgcn_make_buffer_rsrc(inq, 0, (int)(total_b - 3 * plane_b), 0x00020000);
        const auto src1 = __builtin_amdgcn_make_buffer_rsrc(inq + ih * iw, 0, (int)(total_b - 3 * plane_b), 0x00020000);
        const auto src2 = __builtin_amdgcn_make_buffer_rsrc(inq + 2 * ih * iw, 0, (int)(total_b - 3 * plane_b), 0x00020000);
        const auto src3 = __builtin_amdgcn_make_buffer_rsrc(inq + 3 * ih * iw, 0, (int)(total_b - 3 * plane_b), 0x00020000);
        const int toff = ring_b + 8 * ptid;
        int goff[CW_PI], loff[CW_PI];
#pragma unroll
        for (int u = 0; u < CW_PI; ++u) {
            const int j = jl + 32 * u, yy = j / sw, xx = j - yy * sw;
            goff[u] = ((4 * cq * ih + yy) * iw + xx) * 4;
            loff[u] = (yy * sw + xx) * PIXB + 8 * cq;
        }
        typedef float Item[CW_PI];
        Item qa0, qa1, qa2, qa3, qb0, qb1, qb2, qb3;
        auto fetch = [&](int ya, Item& q0, Item& q1, Item& q2, Item& q3) {
            const int boff = (n * CONV_CO * ih + ya) * iw * 4;
#pragma unroll
            for (int u = 0; u < CW_PI; ++u) {
                const int vo = goff[u] + boff;
                q0[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(src0, vo, 0, 0));
                q1[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(src1, vo, 0, 0));
                q2[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(src2, vo, 0, 0));
                q3[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(src3, vo, 0, 0));
            }
        };
        auto commit = [&](int ya, int yb, const Item& q0, const Item& q1, const Item& q2, const Item& q3) {
            const int nrows = yb - ya, nvalid = (nrows * sw - jl + 31) >> 5;
            const int lbase = (ya % rb) * sw * PIXB;
#pragma unroll
            for (int u = 0; u < CW_PI; ++u) {
                unsigned o = (unsigned)(loff[u] + lbase);
                const unsigned ow_ = o - (unsigned)ring_b;
                o = o < ow_ ? o : ow_;
                o = u < nvalid ? o : (unsigned)toff;
                const cf32x2 v = {q0[u], q1[u]}, w = {q2[u], q3[u]};
                const cbf16x2 hv = __builtin_convertvector(v, cbf16x2), hw_ = __builtin_convertvector(w, cbf16x2);
                *reinterpret_cast<uint2*>(ring + o) = make_uint2(__builtin_bit_cast(unsigned, hv), __builtin_bit_cast(unsigned, hw_));
                if constexpr (X3) {
                    const cf32x2 rv = v - __builtin_convertvector(hv, cf32x2), rw = w - __builtin_convertvector(hw_, cf32x2);
                    const cbf16x2 lv = __builtin_convertvector(rv, cbf16x2), lw = __builtin_convertvector(rw, cbf16x2);
                    *reinterpret_cast<uint2*>(ring + o + 64) = make_uint2(__builtin_bit_cast(unsigned, lv), __builtin_bit_cast(unsigned, lw));
                    if constexpr (X6) {
                        const cbf16x2 tv = __builtin_convertvector(rv - __builtin_convertvector(lv, cf32x2), cbf16x2);
                        const cbf16x2 tw = __builtin_convertvector(rw - __builtin_convertvector(lw, cf32x2), cbf16x2);
                        *reinterpret_cast<uint2*>(ring + o + 128) = make_uint2(__builtin_bit_cast(unsigned, tv), __builtin_bit_cast(unsigned, tw));
                    }
                }
            }
        };
        // ---- dY: thread = (channel co, pixel quad ql of 8): quads ql + 8 i of a pass; fp32 running sum = the bias gradient
        const int dco = ptid >> 3, dql = ptid & 7;
        const unsigned dy_total = (unsigned)gridDim.x * CONV_CO * npix * 4;
        const auto dsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy), 0, (int)dy_total, 0x00020000);
        const int dgo = ((n * CONV_CO + dco) * npix + 4 * dql) * 4;             // byte offset of the thread's first quad of pass 0
        const int dlo = dco * WW_DYP + 8 * dql;                                  // its byte offset in a plane of the image
        typedef float cf32x4 __attribute__((ext_vector_type(4)));
        cf32x4 da[4], db[4];
        float bsum = 0.f;
        auto dfetch = [&](int pass, cf32x4 (&d)[4]) {
#pragma unroll
            for (int i = 0; i < 4; ++i)                  // dword loads: a quad may straddle the end of the tensor, and only whole dwords are range-checked for sure
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    d[i][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dsrc, dgo + (pass * CW_PASS + 32 * i + e) * 4, 0, 0));
        };
        auto dcommit = [&](int pass, const cf32x4 (&d)[4]) {
            unsigned char* img = dyimg + (pass & 1) * DYBUF + dlo;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                cf32x4 v = d[i];
                const int pp = pass * CW_PASS + 4 * dql + 32 * i;
                if (pp + 3 >= npix) {                                            // the image's last quads: pixels past the map are zeros
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = pp + e < npix ? v[e] : 0.f;
                }
                bsum += (v[0] + v[1]) + (v[2] + v[3]);
                const cf32x2 v01 = {v[0], v[1]}, v23 = {v[2], v[3]};
                const cbf16x2 h01 = __builtin_convertvector(v01, cbf16x2), h23 = __builtin_convertvector(v23, cbf16x2);
                *reinterpret_cast<uint2*>(img + 64 * i) = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
                if constexpr (X3) {
                    const cf32x2 r01 = v01 - __builtin_convertvector(h01, cf32x2), r23 = v23 - __builtin_convertvector(h23, cf32x2);
                    const cbf16x2 l01 = __builtin_convertvector(r01, cbf16x2), l23 = __builtin_convertvector(r23, cbf16x2);
                    *reinterpret_cast<uint2*>(img + CONV_CO * WW_DYP + 64 * i) = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
                    if constexpr (X6) {
                        const cbf16x2 t01 = __builtin_convertvector(r01 - __builtin_convertvector(l01, cf32x2), cbf16x2);
                        const cbf16x2 t23 = __builtin_convertvector(r23 - __builtin_convertvector(l23, cf32x2), cbf16x2);
                        *reinterpret_cast<uint2*>(img + 2 * CONV_CO * WW_DYP + 64 * i) = make_uint2(__builtin_bit_cast(unsigned, t01), __builtin_bit_cast(unsigned, t23));
                    }
                }
            }
        };
        // prologue: rows and dY of pass 0; batches of passes 1 and 2 and dY of passes 1, 2 in flight. Everything unconditional from there on
        // (past the last pass a batch has no rows and dY reads answer 0 or are overwritten by nobody's reads).
        {
            const int need = end_row(0);
            const int ym = chunk < need ? chunk : need, yn = ym + chunk < need ? ym + chunk : need;
            fetch(0, qa0, qa1, qa2, qa3);
            if (ym < need) fetch(ym, qb0, qb1, qb2, qb3);
            dfetch(0, da);
            commit(0, ym, qa0, qa1, qa2, qa3);
            if (ym < need) commit(ym, yn, qb0, qb1, qb2, qb3);
            for (int y = yn; y < need; y += chunk) {
                fetch(y, qa0, qa1, qa2, qa3);
                commit(y, y + chunk < need ? y + chunk : need, qa0, qa1, qa2, qa3);
            }
            dcommit(0, da);
            fetch(end_row(0), qa0, qa1, qa2, qa3);
            fetch(end_row(1), qb0, qb1, qb2, qb3);
            dfetch(1, da);
            dfetch(2, db);
        }
        __syncthreads();
        int t = 0;
        for (; t + 1 < npass; t += 2) {
            commit(end_row(t), end_row(t + 1), qa0, qa1, qa2, qa3);
            dcommit(t + 1, da);
            fetch(end_row(t + 2), qa0, qa1, qa2, qa3);
            dfetch(t + 3, da);
            __syncthreads();
            commit(end_row(t + 1), end_row(t + 2), qb0, qb1, qb2, qb3);
            if (t + 2 < npass) dcommit(t + 2, db);
            fetch(end_row(t + 3), qb0, qb1, qb2, qb3);
            dfetch(t + 4, db);
            __syncthreads();
        }
        if (t < npass) __syncthreads();
        // bias gradient: the 8 lanes of a channel are neighbours
        bsum += __shfl_xor(bsum, 1, 64);
        bsum += __shfl_xor(bsum, 2, 64);
        bsum += __shfl_xor(bsum, 4, 64);
        if (dql == 0) Pb[(int64_t)n * CONV_CO + dco] = bsum;
        __syncthreads();                       // the consumers' tap-8 exchange
        return;
    }
    // ---- consumers ---------------------------------------------------------------------------------------------------------------------------
    const int cw = wave & 3;
    const int kg = lane >> 5, col = lane & 31;
    // transposed A reads: lane 4 q + p of a 16-lane group supplies pixel row q, channels 16 (g & 1) + 4 p .. + 3 of the block; k = 8 (g >> 1) + q (+ 4)
    const int grp = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    const int colb = (16 * (grp & 1) + 4 * tp) * 2;
    const int klo = 8 * (grp >> 1) + tq;
    int tdel[3];                               // ring byte offset of the wave's taps relative to the pixel: (dy * sw + dx) * PIXB
    {
        const int taps[3] = {2 * cw, 2 * cw + 1, 8};
#pragma unroll
        for (int j = 0; j < 3; ++j) tdel[j] = ((taps[j] / 3) * sw + taps[j] % 3) * PIXB;
    }
    // the lane's lo pixel (k = klo of step 0 of pass 0) as (column, ring byte offset of its row); advanced 4 (hi) and 12 (next step) at a time
    int px = klo % ow, prow = (klo / ow) * sw * PIXB, ppix = klo;
    const int rowb = sw * PIXB;
    const int last_x = (npix - 1) % ow, last_row = (((npix - 1) / ow) % rb) * rowb;
    auto advance = [&](int d) {
        px += d; ppix += d;
        if (px >= ow) { px -= ow; prow += rowb; prow = prow >= ring_b ? prow - ring_b : prow; }
    };
    auto pix_base = [&]() -> unsigned {        // ring offset of the lane's current pixel (the last pixel of the map for lanes past it), + its columns
        const bool inm = ppix < npix;
        return (unsigned)((inm ? prow : last_row) + (inm ? px : last_x) * PIXB + colb);
    };
    auto tap_addr = [&](unsigned base, int j) -> const unsigned char* {
        unsigned o = base + (unsigned)tdel[j];
        const unsigned o2 = o - (unsigned)ring_b;
        o = o < o2 ? o : o2;
        return ring + o;
    };
    typedef short v4s16 __attribute__((ext_vector_type(4)));
    typedef short v8s16 __attribute__((ext_vector_type(8)));
    auto tr_frag = [&](const unsigned char* lo, const unsigned char* hi) -> cbf16x8 {
        const v4s16 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16*)(lo));
        const v4s16 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16*)(hi));
        const v8s16 v = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        return __builtin_bit_cast(cbf16x8, v);
    };
    cf32x16 acc[3], accx[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[j][i] = 0.f; accx[j][i] = 0.f; }
    auto unit = [&](int j, const cbf16x8 (&xa)[3], const cbf16x8 (&db)[3]) {            // one tap, one k16 step: NPL-plane products
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[0], db[0], acc[j], 0, 0, 0);
        if constexpr (X3) {
            accx[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[0], db[1], accx[j], 0, 0, 0);
            accx[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[1], db[0], accx[j], 0, 0, 0);
        }
        if constexpr (X6) {
            accx[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[0], db[2], accx[j], 0, 0, 0);
            accx[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[2], db[0], accx[j], 0, 0, 0);
            accx[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[1], db[1], accx[j], 0, 0, 0);
        }
    };
    __syncthreads();
    // fragments of k16 step ks + 1 are read while step ks is on the matrix cores (two register sets, the steps of a pass fully unrolled)
    // Tap 8's fragments are read inside the branch that multiplies them: read one step ahead under the same (wave-uniform) condition and
    // multiplied in the next iteration's branch, the plain-bf16 build produced NaNs in tap 8 on some launches (tools/debug/wgrad_nan_probe.py).
    cbf16x8 db[2][3], x0[2][3], x1[2][3];
    unsigned b8lo[2], b8hi[2];                 // the pixel bases of a step, kept for its tap-8 unit
    for (int t = 0; t < npass; ++t) {
        const unsigned char* dimg = dyimg + (t & 1) * DYBUF + col * WW_DYP + 16 * kg;
        auto ld = [&](int ks, int b) {
            const unsigned blo = pix_base();
            advance(4);
            const unsigned bhi = pix_base();
            advance(12);
            b8lo[b] = blo; b8hi[b] = bhi;
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) db[b][pl] = *reinterpret_cast<const cbf16x8*>(dimg + pl * CONV_CO * WW_DYP + 32 * ks);
            const unsigned char *l0 = tap_addr(blo, 0), *h0 = tap_addr(bhi, 0), *l1 = tap_addr(blo, 1), *h1 = tap_addr(bhi, 1);
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) { x0[b][pl] = tr_frag(l0 + 64 * pl, h0 + 64 * pl); x1[b][pl] = tr_frag(l1 + 64 * pl, h1 + 64 * pl); }
        };
        ld(0, 0);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int b = ks & 1;
            if (ks + 1 < 8) ld(ks + 1, b ^ 1);
            unit(0, x0[b], db[b]);
            unit(1, x1[b], db[b]);
            if ((ks >> 1) == cw) {                         // wave-uniform: this wave's quarter of tap 8
                const unsigned char *l8 = tap_addr(b8lo[b], 2), *h8 = tap_addr(b8hi[b], 2);
                cbf16x8 x8[3];
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) x8[pl] = tr_frag(l8 + 64 * pl, h8 + 64 * pl);
                unit(2, x8, db[b]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    // C layout: reg r of lane l = row (r & 3) + 8 (r >> 2) + 4 (l >> 5) = ci, column l & 31 = co
    float* Pn = P + (int64_t)n * CONV_CO * CONV_CO * 9;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = (r & 3) + 8 * (r >> 2) + 4 * kg;
            Pn[(col * CONV_CO + ci) * 9 + 2 * cw + j] = X3 ? accx[j][r] + acc[j][r] : acc[j][r];
        }
    // tap 8: the four quarters through LDS (the ring is free now), summed in wave order by consumer 0
    float* xch = reinterpret_cast<float*>(ring);
#pragma unroll
    for (int r = 0; r < 16; ++r) xch[(cw * 16 + r) * 64 + lane] = X3 ? accx[2][r] + acc[2][r] : acc[2][r];
    __syncthreads();
    if (cw == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = (r & 3) + 8 * (r >> 2) + 4 * kg;
            Pn[(col * CONV_CO + ci) * 9 + 8] = ((xch[r * 64 + lane] + xch[(16 + r) * 64 + lane]) + xch[(32 + r) * 64 + lane]) + xch[(48 + r) * 64 + lane];
        }
    }
}

static size_t conv_wgrad_ws_lds(int ow, int npl) {
    const size_t ring = (size_t)conv3x3_ws_rows(ow) * (ow + 2) * ww_pixb(npl);
    return ((ring + CW_DUMMY + 15) & ~(size_t)15) + (size_t)2 * npl * CONV_CO * WW_DYP;
}
static bool conv_wgrad_ws_fits(int n, int ih, int iw, int oh, int ow, int npl) {
    const int sw = ow + 2, npix = oh * ow;
    if (ih != oh + 2 || iw != ow + 2 || ow < 16 || sw > CM_MAXW || npix < 3 * CW_PASS) return false;
    const int chunk = (32 * CW_PI) / sw, npass = (npix + CW_PASS - 1) / CW_PASS;
    auto end_row = [&](int pass) { return ((pass * CW_PASS + CW_PASS < npix ? pass * CW_PASS + CW_PASS : npix) - 1) / ow + 3; };
    for (int t = 0; t + 1 < npass; ++t)
        if (end_row(t + 1) - end_row(t) > chunk) return false;
    if (end_row(0) > 2 * chunk + chunk) return false;
    const size_t ring = (size_t)conv3x3_ws_rows(ow) * sw * ww_pixb(npl);
    return chunk >= 1 && conv_wgrad_ws_lds(ow, npl) <= 160 * 1024 && ring >= 4 * 16 * 64 * 4 &&
           (int64_t)n * CONV_CO * ih * iw * 4 < (1ll << 31) && (int64_t)n * CONV_CO * npix * 4 < (1ll << 31);
}

static int conv_wgrad_mfma(const float* dy, const float* in, float* P, float* Pb, int n, int ih, int iw, int oh, int ow, int prec, hipStream_t s) {
    const int npl = prec == EXORL_PREC_BF16X6 ? 3 : (prec == EXORL_PREC_BF16X3 ? 2 : 1);
    if (!(tune_variant() & 64) && conv_wgrad_ws_fits(n, ih, iw, oh, ow, npl)) {          // exorl_gemm_tune bit 64: the tile kernel below (A/B)
        const size_t wlds = conv_wgrad_ws_lds(ow, npl);
        static bool wattr = false;
        if (!wattr) {
            EXORL_CHECK_HIP(hipFuncSetAttribute((const void*)conv_wgrad_ws_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            EXORL_CHECK_HIP(hipFuncSetAttribute((const void*)conv_wgrad_ws_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            EXORL_CHECK_HIP(hipFuncSetAttribute((const void*)conv_wgrad_ws_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            wattr = true;
        }
        const int rb = conv3x3_ws_rows(ow);
        if (npl == 3)      hipLaunchKernelGGL((conv_wgrad_ws_kernel<3>), dim3(n), dim3(CM_THREADS), wlds, s, dy, in, P, Pb, ih, iw, oh, ow, rb);
        else if (npl == 2) hipLaunchKernelGGL((conv_wgrad_ws_kernel<2>), dim3(n), dim3(CM_THREADS), wlds, s, dy, in, P, Pb, ih, iw, oh, ow, rb);
        else               hipLaunchKernelGGL((conv_wgrad_ws_kernel<1>), dim3(n), dim3(CM_THREADS), wlds, s, dy, in, P, Pb, ih, iw, oh, ow, rb);
        EXORL_LAUNCH_CHECK();
        return 0;
    }
    const bool th4 = npl == 3 && (tune_variant() & 2048);            // measured and not adopted: 4 x 16 tiles (74 KB, two workgroups per CU): 9.6 vs 9.3 ms per Proto update
    const int th = npl == 3 ? (th4 ? 4 : 8) : 16;
    const size_t lds = (size_t)npl * (CONV_CO * (th * CONV_TILE + 8) + 3 * CONV_CO * ((th + 2) * CONV_TILE + 8)) * sizeof(unsigned short);
    static bool attr = false;
    if (!attr) {
        EXORL_CHECK_HIP(hipFuncSetAttribute((const void*)conv_wgrad_mfma_kernel<3, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        EXORL_CHECK_HIP(hipFuncSetAttribute((const void*)conv_wgrad_mfma_kernel<3, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        EXORL_CHECK_HIP(hipFuncSetAttribute((const void*)conv_wgrad_mfma_kernel<2, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        EXORL_CHECK_HIP(hipFuncSetAttribute((const void*)conv_wgrad_mfma_kernel<1, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    if (th4)           hipLaunchKernelGGL((conv_wgrad_mfma_kernel<3, 4>), dim3(n), dim3(WM_THREADS), lds, s, dy, in, P, Pb, ih, iw, oh, ow);
    else if (npl == 3) hipLaunchKernelGGL((conv_wgrad_mfma_kernel<3, 8>), dim3(n), dim3(WM_THREADS), lds, s, dy, in, P, Pb, ih, iw, oh, ow);
    else if (npl == 2) hipLaunchKernelGGL((conv_wgrad_mfma_kernel<2, 16>), dim3(n), dim3(WM_THREADS), lds, s, dy, in, P, Pb, ih, iw, oh, ow);
    else               hipLaunchKernelGGL((conv_wgrad_mfma_kernel<1, 16>), dim3(n), dim3(WM_THREADS), lds, s, dy, in, P, Pb, ih, iw, oh, ow);
    EXORL_LAUNCH_CHECK();
    return 0;
}

// ---- first layer (c_in = 3 or 9 input channels, stride 2, pixel scaling): weight gradients on the matrix cores (round 3) -------------------
// The fp32 FMA kernel above spends 525 us per launch at batch 1024 on 3 GFLOP and 307 MB — one LDS read per FMA. As a GEMM per image:
// dW[co][k'] = sum_p dY[co][p] * X[p][k'], k' = ci * 9 + tap (27 or 81 columns), with the bias gradient as one more column of ones. M = 32 output
// channels, N = 32 columns per accumulator tile (NT = 1 for 3 channels, 3 for 9), K = the image's output pixels in row-major order, 16 per MFMA.
// Both operands are built in REGISTERS from fp32 tiles in LDS — dY[co][p] (8 consecutive pixels of one channel: two 16-byte reads), the raw
// input rows (8 stride-2 taps per lane, scaled x / 255 - 0.5 as Encoder.forward does) — and split into NPL bf16 planes there; no plane ever
// exists in memory. A workgroup = one image, 8 waves share the k16 steps of a chunk of RC output rows, accumulate over all chunks, and combine
// their eight accumulators through LDS at the end. Partials in the layout of the fp32 kernel (summed in image order by the same column-sum).
constexpr int W1_RC = 7;                    // output rows per chunk: 7 x 41 = 287 pixels = 18 k16 steps with one pixel of padding
constexpr int W1_THREADS = 512;
template <int NPL, int NT>
__global__ __launch_bounds__(W1_THREADS) void conv1_wgrad_mfma_kernel(const float* __restrict__ dy, const float* __restrict__ in, float* __restrict__ P,
                                                                      float* __restrict__ Pb, int ci_n, int ih, int iw, int oh, int ow) {
    extern __shared__ __attribute__((aligned(16))) unsigned char w1_lds[];
    const int CP = ((W1_RC * ow + 15) / 16) * 16 + 4;                      // dY pitch per channel (floats): whole k16 steps + 4 (bank spread)
    const int XR = 2 * W1_RC + 1;                                          // input rows a chunk touches
    float* dyt = reinterpret_cast<float*>(w1_lds);                         // [32][CP]
    float* xt = dyt + CONV_CO * CP;                                        // [ci_n][XR][iw]
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kg = lane >> 5, col = lane & 31;
    const float* dyn = dy + (int64_t)n * CONV_CO * oh * ow;
    const float* inn = in + (int64_t)n * ci_n * ih * iw;
    const int ncols = ci_n * 9;                                            // real columns; column `ncols` carries the bias gradient
    cf32x16 acc[NT][NPL];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int c = 0; c < NPL; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][c][r] = 0.f;
    // this lane's B columns: k' = 32 t + col -> (ci, ky, kx) offsets into the input tile
    int boff[NT]; bool bok[NT], bone[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int kp = 32 * t + col;
        bok[t] = kp < ncols; bone[t] = kp == ncols;
        const int ci = bok[t] ? kp / 9 : 0, tap = bok[t] ? kp % 9 : 0;
        boff[t] = (ci * XR + tap / 3) * iw + tap % 3;
    }
    auto planes = [](const float (&v)[8], cbf16x8 (&out)[NPL]) {
        float r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = v[j];
#pragma unroll
        for (int c = 0; c < NPL; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) { const __bf16 h = (__bf16)r[j]; out[c][j] = h; r[j] -= (float)h; }
    };
    for (int r0 = 0; r0 < oh; r0 += W1_RC) {
        const int nr = oh - r0 < W1_RC ? oh - r0 : W1_RC, npx = nr * ow, nsteps = (npx + 15) / 16;
        __syncthreads();
        // (channel, pixel) and (channel, row, column) of element i are kept incrementally: the divisions and remainders per element were a
        // large share of this kernel's instructions (same elements, same values)
        {
            int co = tid / CP, pp = tid - co * CP;
            const int dco = W1_THREADS / CP, dpp = W1_THREADS - dco * CP;
            for (int i = tid; i < CONV_CO * CP; i += W1_THREADS) {             // dY chunk, zero beyond the chunk's pixels
                dyt[i] = pp < npx ? dyn[(int64_t)co * oh * ow + (int64_t)r0 * ow + pp] : 0.f;
                co += dco; pp += dpp;
                if (pp >= CP) { pp -= CP; ++co; }
            }
        }
        const int rows_in = 2 * nr + 1, iy0 = 2 * r0;
        {
            const int plane = XR * iw;
            int ci = tid / plane, rem = tid - ci * plane, yy = rem / iw, xx = rem - yy * iw;
            const int srow = W1_THREADS / iw, dxx = W1_THREADS - srow * iw, dci = srow / XR, dyy = srow - dci * XR;
            for (int i = tid; i < ci_n * plane; i += W1_THREADS) {
                xt[i] = (yy < rows_in && iy0 + yy < ih) ? inn[((int64_t)ci * ih + iy0 + yy) * iw + xx] / 255.0f - 0.5f : 0.f;
                xx += dxx; yy += dyy; ci += dci;
                if (xx >= iw) { xx -= iw; ++yy; }
                if (yy >= XR) { yy -= XR; ++ci; }
            }
        }
        __syncthreads();
        for (int st = wave; st < nsteps; st += 8) {
            const int p0 = st * 16 + 8 * kg;
            float av[8];
            {
                const float4 a0 = *reinterpret_cast<const float4*>(dyt + col * CP + p0), a1 = *reinterpret_cast<const float4*>(dyt + col * CP + p0 + 4);
                av[0] = a0.x; av[1] = a0.y; av[2] = a0.z; av[3] = a0.w; av[4] = a1.x; av[5] = a1.y; av[6] = a1.z; av[7] = a1.w;
            }
            cbf16x8 ap[NPL];
            planes(av, ap);
            int py[8], px[8];
            {
                int y = p0 / ow, x = p0 % ow;
#pragma unroll
                for (int j = 0; j < 8; ++j) { py[j] = y; px[j] = x; if (++x == ow) { x = 0; ++y; } }
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float bv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool inside = p0 + j < npx;
                    const float xv = xt[boff[t] + (inside ? (2 * py[j]) * iw + 2 * px[j] : 0)];
                    bv[j] = bok[t] ? (inside ? xv : 0.f) : (bone[t] ? 1.0f : 0.f);
                }
                cbf16x8 bp[NPL];
                planes(bv, bp);
                // plane products by magnitude class: c = 0: p0*p0; c = 1: p0*p1 + p1*p0; c = 2: p0*p2 + p2*p0 + p1*p1
                acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], bp[0], acc[t][0], 0, 0, 0);
                if constexpr (NPL >= 2) {
                    acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], bp[1], acc[t][1], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[1], bp[0], acc[t][1], 0, 0, 0);
                }
                if constexpr (NPL == 3) {
                    acc[t][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], bp[2], acc[t][2], 0, 0, 0);
                    acc[t][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[2], bp[0], acc[t][2], 0, 0, 0);
                    acc[t][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[1], bp[1], acc[t][2], 0, 0, 0);
                }
            }
        }
    }
    // combine the eight waves' accumulators: [wave][t][co][col] floats through LDS (reusing the tiles), then one thread per (co, k')
    __syncthreads();
    float* red = reinterpret_cast<float*>(w1_lds);                                        // [8][NT][32][32]
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = (r & 3) + 8 * (r >> 2) + 4 * kg;                                // C layout: row = co, column = lane & 31
            float v = acc[t][0][r];
            if constexpr (NPL == 2) v = acc[t][1][r] + v;
            if constexpr (NPL == 3) v = (acc[t][2][r] + acc[t][1][r]) + v;
            red[((wave * NT + t) * CONV_CO + co) * 32 + col] = v;
        }
    __syncthreads();
    for (int i = tid; i < NT * CONV_CO * 32; i += W1_THREADS) {
        const int t = i / (CONV_CO * 32), co = (i / 32) % CONV_CO, c = i % 32, kp = 32 * t + c;
        if (kp > ncols) continue;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += red[((w * NT + t) * CONV_CO + co) * 32 + c];
        if (kp < ncols) P[((int64_t)n * CONV_CO + co) * ncols + kp] = v;
        else Pb[(int64_t)n * CONV_CO + co] = v;
    }
}

static bool conv1_wgrad_mfma_fits(int ci_n, int ih, int oh, int ow, int stride) {
    return stride == 2 && (ci_n == 3 || ci_n == 9) && ow <= 48 && oh == ow && ih >= 2 * oh + 1;
}
static int conv1_wgrad_mfma(const float* dy, const float* in, float* P, float* Pb, int n, int ci_n, int ih, int iw, int oh, int ow, int prec, hipStream_t s) {
    const int npl = prec == EXORL_PREC_BF16X6 ? 3 : (prec == EXORL_PREC_BF16X3 ? 2 : 1), nt = ci_n == 3 ? 1 : 3;
    const int cp = ((W1_RC * ow + 15) / 16) * 16 + 4, xr = 2 * W1_RC + 1;
    size_t lds = sizeof(float) * ((size_t)CONV_CO * cp + (size_t)ci_n * xr * iw);
    const size_t red = sizeof(float) * 8 * nt * CONV_CO * 32;
    lds = lds > red ? lds : red;
    EXORL_REQUIRE(lds <= 160 * 1024, "conv1_wgrad_mfma: tiles do not fit LDS");
#define EXORL_W1(NN, TT) do { \
        static bool attr = false; \
        if (!attr) { EXORL_CHECK_HIP(hipFuncSetAttribute((const void*)conv1_wgrad_mfma_kernel<NN, TT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; } \
        hipLaunchKernelGGL((conv1_wgrad_mfma_kernel<NN, TT>), dim3(n), dim3(W1_THREADS), lds, s, dy, in, P, Pb, ci_n, ih, iw, oh, ow); } while (0)
    if (nt == 1) { if (npl == 3) EXORL_W1(3, 1); else if (npl == 2) EXORL_W1(2, 1); else EXORL_W1(1, 1); }
    else         { if (npl == 3) EXORL_W1(3, 3); else if (npl == 2) EXORL_W1(2, 3); else EXORL_W1(1, 3); }
#undef EXORL_W1
    EXORL_LAUNCH_CHECK();
    return 0;
}

// d *= (a > 0): ReLU mask of the top activation against the gradient that arrives from the trunk's Linear
__global__ __launch_bounds__(256) void relu_mask_kernel(float* __restrict__ d, const float* __restrict__ a, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = a[i] > 0.f ? d[i] : 0.f;
}

// ---- encoder geometry -------------------------------------------------------------------------------------------------
struct EncGeom {
    int c_in, hw;                 // input channels, image edge (84 or 64)
    int edge[5];                  // spatial edge of the input and of the 4 activations
    int64_t w_off[4], b_off[4];   // flat parameter offsets (torch order convnet.{0,2,4,6}.{weight,bias}, each padded to 4 floats)
    int64_t total;
};
static EncGeom enc_geom(int c_in, int hw) {
    EncGeom g{};
    g.c_in = c_in; g.hw = hw;
    g.edge[0] = hw;
    g.edge[1] = (hw - 3) / 2 + 1;
    for (int l = 2; l <= 4; ++l) g.edge[l] = g.edge[l - 1] - 2;
    int64_t off = 0;
    for (int l = 0; l < 4; ++l) {
        const int ci = l == 0 ? c_in : CONV_CO;
        g.w_off[l] = off; off += round_up((int64_t)CONV_CO * ci * 9, 4);
        g.b_off[l] = off; off += round_up(CONV_CO, 4);
    }
    g.total = round_up(off, 64);
    return g;
}

}  // namespace exorl

using namespace exorl;

extern "C" {

int exorl_aug_shift(const unsigned char* x_dev, int32_t n, int32_t c, int32_t h, int32_t pad, const int32_t* shifts_dev, uint64_t seed,
                    uint64_t counter, float* out_dev, void* stream) {
    EXORL_REQUIRE(x_dev && out_dev && n > 0 && c > 0 && h > 1 && pad >= 0, "aug_shift: bad arguments");
    if (h <= AUG_MAXH) {
        hipLaunchKernelGGL(aug_shift_rows_kernel, dim3((unsigned)(n * c)), dim3(256), 0, as_stream(stream), x_dev, shifts_dev, seed, counter, out_dev,
                           c, h, pad);
        EXORL_LAUNCH_CHECK();
        return 0;
    }
    const int64_t total = (int64_t)n * c * h * h;
    const int64_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(aug_shift_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, as_stream(stream), x_dev, shifts_dev, seed,
                       counter, out_dev, n, c, h, pad);
    EXORL_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void u8_to_f32_kernel(const unsigned char* __restrict__ x, float* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = (float)x[i];
}
int exorl_debug_conv_stamps(uint64_t* out_host, int32_t n_words) {
    EXORL_REQUIRE(out_host && n_words > 0 && n_words <= 1024 * 2 * 8, "debug_conv_stamps: bad arguments");
    EXORL_CHECK_HIP(hipDeviceSynchronize());
    EXORL_CHECK_HIP(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(exorl::g_conv_stamps), (size_t)n_words * sizeof(uint64_t)));
    return 0;
}
int exorl_u8_to_f32(const unsigned char* x_dev, int64_t n, float* out_dev, void* stream) {
    EXORL_REQUIRE(x_dev && out_dev && n > 0, "u8_to_f32: bad arguments");
    const int64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(u8_to_f32_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, as_stream(stream), x_dev, out_dev, n);
    EXORL_LAUNCH_CHECK();
    return 0;
}

int64_t exorl_encoder_param_floats(int32_t c_in, int32_t hw) { return enc_geom(c_in, hw).total; }
int64_t exorl_encoder_out_dim(int32_t hw) { const EncGeom g = enc_geom(3, hw); return (int64_t)CONV_CO * g.edge[4] * g.edge[4]; }

// floats of scratch for n images: 4 activation maps, 3 gradient maps (ping-pong would do with 2; kept simple), weight shadows, wgrad partials
int64_t exorl_encoder_workspace_floats(int32_t n, int32_t c_in, int32_t hw) {
    const EncGeom g = enc_geom(c_in, hw);
    int64_t f = 0;
    for (int l = 1; l <= 4; ++l) f += round_up((int64_t)n * CONV_CO * g.edge[l] * g.edge[l], 64);          // activations
    for (int l = 1; l <= 3; ++l) f += round_up((int64_t)n * CONV_CO * g.edge[l] * g.edge[l], 64);          // d(activations) 1..3
    f += 2 * 4 * round_up((int64_t)CONV_CO * CONV_CO * 9, 64);                                             // Wf, Wb per layer
    f += 2 * 4 * round_up((int64_t)3 * CM_FRAG * 4, 64);                                                   // MFMA fragment planes (hi, second, third) per layer and direction
    f += round_up((int64_t)n * CONV_CO * CONV_CO * 9, 64) + round_up((int64_t)n * CONV_CO, 64);           // wgrad partials
    return f;
}

struct EncWs { float* act[5]; float* dact[4]; float* wf[4]; float* wb[4]; float* ff[4]; float* fb[4]; float *P, *Pb; };
static EncWs enc_carve(const EncGeom& g, int n, float* ws) {
    EncWs w{};
    int64_t off = 0;
    auto take = [&](int64_t c) { float* p = ws + off; off += round_up(c, 64); return p; };
    for (int l = 1; l <= 4; ++l) w.act[l] = take((int64_t)n * CONV_CO * g.edge[l] * g.edge[l]);
    for (int l = 1; l <= 3; ++l) w.dact[l] = take((int64_t)n * CONV_CO * g.edge[l] * g.edge[l]);
    for (int l = 0; l < 4; ++l) { w.wf[l] = take((int64_t)CONV_CO * CONV_CO * 9); w.wb[l] = take((int64_t)CONV_CO * CONV_CO * 9); }
    for (int l = 0; l < 4; ++l) { w.ff[l] = take((int64_t)3 * CM_FRAG * 4); w.fb[l] = take((int64_t)3 * CM_FRAG * 4); }
    w.P = take((int64_t)n * CONV_CO * CONV_CO * 9);
    w.Pb = take((int64_t)n * CONV_CO);
    return w;
}

// Encoder.forward (ddpg.py:35-39): x (n, c_in, hw, hw) fp32 pixel values (0..255, e.g. the augmentation's output);
// the flattened features (n, 32*e*e) are ws_dev's 4th activation map: *h_out_dev points at them.
int exorl_encoder_forward(const float* params_dev, int32_t c_in, int32_t hw, const float* x_dev, int32_t n, float* ws_dev, float** h_out_dev,
                          void* stream) {
    return exorl_encoder_forward_prec(params_dev, c_in, hw, x_dev, n, ws_dev, h_out_dev, EXORL_PREC_F32, stream);
}
int exorl_encoder_forward_prec(const float* params_dev, int32_t c_in, int32_t hw, const float* x_dev, int32_t n, float* ws_dev, float** h_out_dev,
                               int32_t prec, void* stream) {
    EXORL_REQUIRE(params_dev && x_dev && ws_dev && n > 0 && c_in > 0 && c_in <= 16 && hw >= 16, "encoder_forward: bad arguments");
    EXORL_REQUIRE(prec >= EXORL_PREC_F32 && prec <= EXORL_PREC_BF16X6, "encoder_forward: unknown precision %d", prec);
    hipStream_t s = as_stream(stream);
    const EncGeom g = enc_geom(c_in, hw);
    const EncWs w = enc_carve(g, n, ws_dev);
    const float* in = x_dev;
    const int prec_cfg = prec;                  // the fragment shadows follow the configured mode (the backward pass may still want them)
    if (prec == EXORL_PREC_BF16X3 && (prec_override_mask() & 64)) prec = EXORL_PREC_F32;      // diagnostic (exorl_debug_precision_override)
    for (int l = 0; l < 4; ++l) {
        const int ci = l == 0 ? c_in : CONV_CO;
        const bool frags = l > 0 && prec_cfg != EXORL_PREC_F32;
        hipLaunchKernelGGL(conv_weight_shadow_kernel, dim3(cdiv(CONV_CO * ci * 9, 256)), dim3(256), 0, s, params_dev + g.w_off[l], w.wf[l],
                           l > 0 ? w.wb[l] : nullptr, ci, frags ? reinterpret_cast<uint4*>(w.ff[l]) : nullptr, frags ? reinterpret_cast<uint4*>(w.fb[l]) : nullptr);
        EXORL_LAUNCH_CHECK();
        EXORL_TRY(conv3x3(in, w.wf[l], params_dev + g.b_off[l], nullptr, w.act[l + 1], n, ci, CONV_CO, g.edge[l], g.edge[l], g.edge[l + 1],
                          g.edge[l + 1], l == 0 ? 2 : 1, 0, l == 0 ? 1 : 0, 1, s, prec, frags ? w.ff[l] : nullptr));
        in = w.act[l + 1];
    }
    if (h_out_dev) *h_out_dev = w.act[4];
    return 0;
}

// Backward through the encoder after exorl_encoder_forward on the same x / ws: dh (n, 32*e*e) is overwritten (ReLU mask);
// parameter gradients are written to grads_dev in the parameters' flat layout. No d/d(pixels).
int exorl_encoder_backward(const float* params_dev, int32_t c_in, int32_t hw, const float* x_dev, int32_t n, float* ws_dev, float* dh_dev,
                           float* grads_dev, void* stream) {
    return exorl_encoder_backward_prec(params_dev, c_in, hw, x_dev, n, ws_dev, dh_dev, grads_dev, EXORL_PREC_F32, stream);
}
int exorl_encoder_backward_prec(const float* params_dev, int32_t c_in, int32_t hw, const float* x_dev, int32_t n, float* ws_dev, float* dh_dev,
                                float* grads_dev, int32_t prec, void* stream) {
    EXORL_REQUIRE(params_dev && x_dev && ws_dev && dh_dev && grads_dev && n > 0, "encoder_backward: bad arguments");
    EXORL_REQUIRE(prec >= EXORL_PREC_F32 && prec <= EXORL_PREC_BF16X6, "encoder_backward: unknown precision %d", prec);
    hipStream_t s = as_stream(stream);
    const EncGeom g = enc_geom(c_in, hw);
    const EncWs w = enc_carve(g, n, ws_dev);
    const int64_t top = (int64_t)n * CONV_CO * g.edge[4] * g.edge[4];
    hipLaunchKernelGGL(relu_mask_kernel, dim3((unsigned)((top + 255) / 256 > 4096 ? 4096 : (top + 255) / 256)), dim3(256), 0, s, dh_dev, w.act[4], top);
    EXORL_LAUNCH_CHECK();
    float* d = dh_dev;                                            // d(a_{l+1}), already masked
    for (int l = 3; l >= 0; --l) {
        const int ci = l == 0 ? c_in : CONV_CO, stride = l == 0 ? 2 : 1;
        const float* in = l == 0 ? x_dev : w.act[l];
        const int oh = g.edge[l + 1], ih = g.edge[l];
        const int in_rows = (WG_ROWS - 1) * stride + 3;
        const size_t lds = ((size_t)WG_ROWS * oh * CONV_CO + (size_t)in_rows * ih * ci) * sizeof(float);
        EXORL_REQUIRE(lds <= 160 * 1024, "encoder_backward: wgrad tile does not fit LDS");
        static bool attr = false;
        if (!attr) {
            EXORL_CHECK_HIP(hipFuncSetAttribute((const void*)conv_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr = true;
        }
        const int ov = prec == EXORL_PREC_BF16X3 ? prec_override_mask() : 0;                    // diagnostic (exorl_debug_precision_override)
        const int prec_w = (ov & 256) ? EXORL_PREC_F32 : prec, prec_d = (ov & 128) ? EXORL_PREC_F32 : prec;
        if (prec_w != EXORL_PREC_F32 && l > 0)
            EXORL_TRY(conv_wgrad_mfma(d, in, w.P, w.Pb, n, ih, ih, oh, oh, prec_w, s));
        else if (prec_w != EXORL_PREC_F32 && l == 0 && conv1_wgrad_mfma_fits(ci, ih, oh, oh, stride) && !(tune_variant() & 131072))
            EXORL_TRY(conv1_wgrad_mfma(d, in, w.P, w.Pb, n, ci, ih, ih, oh, oh, prec_w, s));     // exorl_gemm_tune bit 131072: the fp32 FMA kernel (A/B)
        else {
            hipLaunchKernelGGL(conv_wgrad_kernel, dim3(n), dim3(1024), lds, s, d, in, w.P, w.Pb, ci, ih, ih, oh, oh, stride, l == 0 ? 1 : 0);
            EXORL_LAUNCH_CHECK();
        }
        EXORL_TRY(colsum(w.P, grads_dev + g.w_off[l], n, CONV_CO * ci * 9, 1, 0, 0, s));
        EXORL_TRY(colsum(w.Pb, grads_dev + g.b_off[l], n, CONV_CO, 1, 0, 0, s));
        if (l > 0) {              // d(a_l) = full correlation of d(a_{l+1}) with the flipped kernel, masked by a_l > 0
            EXORL_TRY(conv3x3(d, w.wb[l], nullptr, w.act[l], w.dact[l], n, CONV_CO, CONV_CO, oh, oh, ih, ih, 1, 2, 0, 0, s, prec_d, prec_d != EXORL_PREC_F32 ? w.fb[l] : nullptr));
            d = w.dact[l];
        }
    }
    return 0;
}

}  // extern "C"
