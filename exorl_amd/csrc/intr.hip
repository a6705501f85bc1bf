// Intrinsic-reward modules of the reward-free DDPG-backbone agents (states observations): one optimiser step of
// the module on the sampled batch, then the intrinsic reward of that batch under the updated module, written
// where the DDPG critic update reads its reward. Replaces (file:line in the reference repo):
//   RND      agents/unsupervised_learning/rnd.py:13-60 (module), :79-108 (update_rnd, compute_intr_reward)
//   ICM      agents/unsupervised_learning/icm.py:12-45, :64-92
//   ICM-APT  agents/unsupervised_learning/icm_apt.py:13-57, :86-110; utils.PBE / utils.RMS utils/utils.py:257-319
//   Disagreement agents/unsupervised_learning/disagreement.py:11-47, :64-90
//   DIAYN    agents/unsupervised_learning/diayn.py:15-29, :78-127
//   APS      agents/unsupervised_learning/aps.py:63-79, :147-175 (successor-feature net; PBE + task . phi reward)
//   SMM      agents/unsupervised_learning/smm.py:27-112 (VAE, discriminator), :173-246 (updates, reward)
//   Proto    agents/unsupervised_learning/proto.py:14-157 (sinkhorn_knopp, prototypes, candidate queue, kNN reward)
// The modules are plain Linear/ReLU stacks of arbitrary widths (obs_dim, hidden_dim, rep_dim), so the layers run on
// the generic fp32-source grouped GEMM (gemm.hip) with small row/column kernels around it; what the reference's
// autograd graph hides and this file exploits:
//   * RND's target net is frozen and BatchNorm of the same batch is identical in the reward pass, so the target
//     forward runs once per update instead of twice;
//   * the L2-norm losses' gradients are formed in the same kernel that computes the errors (no separate loss pass);
//   * APT's reward is a top-k over pairwise distances (knn.hip) instead of a (B,B,rep) broadcast temporary.
#include <cmath>
#include <vector>

#include "kernels.h"

namespace exorl {

struct ITensor { int64_t off, rows, cols; };
struct RmsState { float M, S; double n; };          // utils.RMS: running mean / variance / count (n starts at 1e-4)

__device__ __forceinline__ float block_sum(float v, float* red) {      // all threads get the total; red: >= 17 floats
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < nw; ++i) s += red[i];
        red[16] = s;
    }
    __syncthreads();
    return red[16];
}

// nn.BatchNorm1d(affine=False) in training mode, then clamp(+-clip) (rnd.py:24-26,49-50): one block per feature.
__global__ __launch_bounds__(256) void bn_clamp_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ out, int rows, int O,
                                                       float clip, float* __restrict__ running /* mean[O] var[O] count */) {
    __shared__ float red[17];
    const int c = blockIdx.x;
    float s = 0.f;
    for (int r = threadIdx.x; r < rows; r += blockDim.x) s += x[(int64_t)r * ldx + c];
    const float mean = block_sum(s, red) / (float)rows;
    float q = 0.f;
    for (int r = threadIdx.x; r < rows; r += blockDim.x) { const float d = x[(int64_t)r * ldx + c] - mean; q += d * d; }
    const float var = block_sum(q, red) / (float)rows;
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    for (int r = threadIdx.x; r < rows; r += blockDim.x) {
        const float v = (x[(int64_t)r * ldx + c] - mean) * rstd;
        out[(int64_t)r * O + c] = fminf(fmaxf(v, -clip), clip);
    }
    if (threadIdx.x == 0) {                          // momentum 0.1; running_var takes the unbiased estimate
        running[c] = 0.9f * running[c] + 0.1f * mean;
        running[O + c] = 0.9f * running[O + c] + 0.1f * var * ((float)rows / (float)(rows > 1 ? rows - 1 : 1));
        if (c == 0) running[2 * O] += 1.0f;
    }
}

// dst[r] = [a[r, 0:ca] | b[r, 0:cb]]
__global__ __launch_bounds__(256) void concat2_kernel(const float* __restrict__ a, int64_t lda, int ca, const float* __restrict__ b, int64_t ldb,
                                                      int cb, float* __restrict__ dst, int rows) {
    const int w = ca + cb;
    const int64_t n = (int64_t)rows * w;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / w;
        const int c = (int)(i - r * w);
        dst[i] = c < ca ? a[r * lda + c] : b[r * ldb + (c - ca)];
    }
}

__global__ __launch_bounds__(256) void relu_bwd_kernel(float* __restrict__ d, const float* __restrict__ a, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        d[i] = a[i] > 0.f ? d[i] : 0.f;
}

// err[b] = mean_j (t - p)^2 (rnd.py:54-56); dpred = d(mean_b err)/dp = -2 (t - p) / R / B. One wave per row.
__global__ __launch_bounds__(256) void rnd_err_kernel(const float* __restrict__ pred, const float* __restrict__ targ, float* __restrict__ err,
                                                      float* __restrict__ dpred, int rows, int R) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float invR = 1.0f / (float)R, invB = 1.0f / (float)rows;
    float s = 0.f;
    for (int j = lane; j < R; j += 64) {
        const float d = targ[(int64_t)row * R + j] - pred[(int64_t)row * R + j];
        s += d * d;
        if (dpred) dpred[(int64_t)row * R + j] = (-2.0f * d * invR) * invB;
    }
    s = wave_sum(s);
    if (lane == 0) err[row] = s * invR;
}

// out[idx] (+)= sum(x) * scale — single block
__global__ __launch_bounds__(1024) void mean_kernel(const float* __restrict__ x, int n, float scale, float* __restrict__ out, int accumulate) {
    __shared__ float red[17];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += x[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) *out = (accumulate ? *out : 0.f) + s * scale;
}

// utils.RMS.__call__ (utils.py:264-276) on n values with batch mean `mean` and unbiased variance `var`; fp32 tensor math
// with the Python-float scalars rounded to fp32 where torch does
__device__ inline void rms_update(RmsState* st, float mean, float var, int bs) {
    const float n = (float)st->n, nb = (float)(st->n + (double)bs), fbs = (float)bs;
    const float delta = mean - st->M;
    const float newM = st->M + delta * fbs / nb;
    const float newS = (st->S * n + var * fbs + delta * delta * n * fbs / nb) / nb;
    st->M = newM; st->S = newS; st->n += (double)bs;
}

// compute_intr_reward (rnd.py:98-103) + the update()'s reward bookkeeping (:131-137): single block
__global__ __launch_bounds__(1024) void rnd_reward_kernel(const float* __restrict__ err, const float* extr, float* reward, int B, float scale,
                                                          RmsState* st, float* __restrict__ metrics) {
    __shared__ float red[17];
    __shared__ float sh_S;
    float s = 0.f, e = 0.f;
    for (int i = threadIdx.x; i < B; i += blockDim.x) { s += err[i]; e += extr ? extr[i] : 0.f; }
    const float mean = block_sum(s, red) / (float)B;
    e = block_sum(e, red);
    float q = 0.f;
    for (int i = threadIdx.x; i < B; i += blockDim.x) { const float d = err[i] - mean; q += d * d; }
    const float var = block_sum(q, red) / (float)(B > 1 ? B - 1 : 1);
    if (threadIdx.x == 0) {
        rms_update(st, mean, var, B);
        sh_S = st->S;
        metrics[EXORL_IM_EXTR_REWARD] = e / (float)B;
        metrics[EXORL_IM_RMS_MEAN] = st->M;
        metrics[EXORL_IM_RMS_STD] = sqrtf(st->S);
    }
    __syncthreads();
    const float denom = sqrtf(sh_S) + 1e-8f;
    float rs = 0.f;
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        const float r = scale * err[i] / denom;
        reward[i] = r;
        rs += r;
    }
    rs = block_sum(rs, red);
    if (threadIdx.x == 0) metrics[EXORL_IM_INTR_REWARD] = rs / (float)B;
}

// ICM errors (icm.py:28-45): fe = ||tgt - pred||_2, be = ||a - tanh(apre)||_2 per row, and the gradients of
// mean(fe) + mean(be) w.r.t. pred / apre. One wave per row. apre == nullptr: forward error only (reward pass).
__global__ __launch_bounds__(256) void icm_err_kernel(const float* __restrict__ pred, int D, const float* __restrict__ tgt, int64_t ldt,
                                                      const float* __restrict__ apre, const float* __restrict__ action, int A,
                                                      float* __restrict__ fe, float* __restrict__ be, float* __restrict__ dpred,
                                                      float* __restrict__ dapre, int rows, float invB) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float s = 0.f;
    for (int j = lane; j < D; j += 64) { const float d = tgt[(int64_t)row * ldt + j] - pred[(int64_t)row * D + j]; s += d * d; }
    const float nf = sqrtf(wave_sum(s));
    if (lane == 0) fe[row] = nf;
    if (dpred)
        for (int j = lane; j < D; j += 64) {
            const float d = tgt[(int64_t)row * ldt + j] - pred[(int64_t)row * D + j];
            dpred[(int64_t)row * D + j] = nf > 0.f ? -(d / nf) * invB : 0.f;
        }
    if (!apre) return;
    float ah = 0.f, d = 0.f;
    if (lane < A) { ah = tanhf(apre[(int64_t)row * A + lane]); d = action[(int64_t)row * A + lane] - ah; }
    const float nb = sqrtf(wave_sum(d * d));
    if (lane == 0) be[row] = nb;
    if (dapre && lane < A) dapre[(int64_t)row * A + lane] = nb > 0.f ? (-(d / nb) * invB) * (1.0f - ah * ah) : 0.f;
}

// The same errors for wide rows (pixel encodings: D = 39200): one WORKGROUP per row and 16-byte accesses. With one wave per row a batch of
// 1024 rows is 1024 waves — four per CU — each walking 157 KB twice in 4-byte steps: 440-610 us per call where the 480 MB it moves cost ~100
// (0.9 ms of an ICM update on pixels, 3.1 ms of a Disagreement update). The second pass re-reads the row from L2.
__global__ __launch_bounds__(256) void icm_err_wide_kernel(const float* __restrict__ pred, int D, const float* __restrict__ tgt, int64_t ldt,
                                                           const float* __restrict__ apre, const float* __restrict__ action, int A,
                                                           float* __restrict__ fe, float* __restrict__ be, float* __restrict__ dpred,
                                                           float* __restrict__ dapre, float invB) {
    __shared__ float red[17];
    const int row = blockIdx.x, lane = threadIdx.x & 63, n4 = D >> 2;
    const float4* p4 = reinterpret_cast<const float4*>(pred + (int64_t)row * D);
    const float4* t4 = reinterpret_cast<const float4*>(tgt + (int64_t)row * ldt);
    float s = 0.f;
    for (int j = threadIdx.x; j < n4; j += 256) {
        const float4 t = t4[j], p = p4[j];
        const float dx = t.x - p.x, dy = t.y - p.y, dz = t.z - p.z, dw = t.w - p.w;
        s += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
    const float nf = sqrtf(block_sum(s, red));
    if (threadIdx.x == 0) fe[row] = nf;
    if (dpred) {
        float4* g4 = reinterpret_cast<float4*>(dpred + (int64_t)row * D);
        for (int j = threadIdx.x; j < n4; j += 256) {
            const float4 t = t4[j], p = p4[j];
            float4 g;
            g.x = nf > 0.f ? -((t.x - p.x) / nf) * invB : 0.f; g.y = nf > 0.f ? -((t.y - p.y) / nf) * invB : 0.f;
            g.z = nf > 0.f ? -((t.z - p.z) / nf) * invB : 0.f; g.w = nf > 0.f ? -((t.w - p.w) / nf) * invB : 0.f;
            g4[j] = g;
        }
    }
    if (!apre || threadIdx.x >= 64) return;
    float ah = 0.f, d = 0.f;
    if (lane < A) { ah = tanhf(apre[(int64_t)row * A + lane]); d = action[(int64_t)row * A + lane] - ah; }
    const float nb = sqrtf(wave_sum(d * d));
    if (lane == 0) be[row] = nb;
    if (dapre && lane < A) dapre[(int64_t)row * A + lane] = nb > 0.f ? (-(d / nb) * invB) * (1.0f - ah * ah) : 0.f;
}
// one launch of the ICM errors: wide rows take the workgroup-per-row kernel (16-byte alignment of every row of pred / tgt / dpred required)
static void launch_icm_err(const float* pred, int D, const float* tgt, int64_t ldt, const float* apre, const float* action, int A, float* fe, float* be,
                           float* dpred, float* dapre, int rows, float invB, hipStream_t s) {
    const bool wide = D >= 2048 && D % 4 == 0 && ldt % 4 == 0 && reinterpret_cast<uintptr_t>(pred) % 16 == 0 && reinterpret_cast<uintptr_t>(tgt) % 16 == 0 &&
                      (!dpred || reinterpret_cast<uintptr_t>(dpred) % 16 == 0);
    if (wide) hipLaunchKernelGGL(icm_err_wide_kernel, dim3(rows), dim3(256), 0, s, pred, D, tgt, ldt, apre, action, A, fe, be, dpred, dapre, invB);
    else hipLaunchKernelGGL(icm_err_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, pred, D, tgt, ldt, apre, action, A, fe, be, dpred, dapre, rows, invB);
}

// reward = log(fe * scale + 1) (icm.py:86-92) + reward bookkeeping; single block
__global__ __launch_bounds__(1024) void icm_reward_kernel(const float* __restrict__ fe, const float* extr, float* reward, int B, float scale,
                                                          float* __restrict__ metrics) {
    __shared__ float red[17];
    float e = 0.f;
    for (int i = threadIdx.x; i < B; i += blockDim.x) e += extr ? extr[i] : 0.f;
    e = block_sum(e, red);
    float rs = 0.f;
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        const float r = logf(fe[i] * scale + 1.0f);
        reward[i] = r;
        rs += r;
    }
    rs = block_sum(rs, red);
    if (threadIdx.x == 0) { metrics[EXORL_IM_EXTR_REWARD] = e / (float)B; metrics[EXORL_IM_INTR_REWARD] = rs / (float)B; }
}

// utils.PBE.__call__ after the top-k (utils.py:301-319): topk (B,k) ascending; single block
__global__ __launch_bounds__(1024) void pbe_reward_kernel(const float* __restrict__ topk, const float* extr, float* reward, int B, int k,
                                                          int avg, int use_rms, float clip, RmsState* st, float* __restrict__ metrics) {
    __shared__ float red[17];
    __shared__ float sh_M;
    const int n = avg ? B * k : B;
    auto val = [&](int i) { return avg ? topk[i] : topk[(int64_t)i * k + (k - 1)]; };
    float e = 0.f;
    for (int i = threadIdx.x; i < B; i += blockDim.x) e += extr ? extr[i] : 0.f;
    e = block_sum(e, red);
    float M = 1.0f;
    if (use_rms) {
        float s = 0.f;
        for (int i = threadIdx.x; i < n; i += blockDim.x) s += val(i);
        const float mean = block_sum(s, red) / (float)n;
        float q = 0.f;
        for (int i = threadIdx.x; i < n; i += blockDim.x) { const float d = val(i) - mean; q += d * d; }
        const float var = block_sum(q, red) / (float)(n > 1 ? n - 1 : 1);
        if (threadIdx.x == 0) { rms_update(st, mean, var, n); sh_M = st->M; }
        __syncthreads();
        M = sh_M;
    }
    float rs = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        float r;
        if (avg) {
            float acc = 0.f;
            for (int j = 0; j < k; ++j) {
                float v = topk[(int64_t)b * k + j];
                if (use_rms) v = v / M;
                if (clip >= 0.f) v = fmaxf(v - clip, 0.f);
                acc += v;
            }
            r = acc / (float)k;
        } else {
            r = topk[(int64_t)b * k + (k - 1)];
            if (use_rms) r = r / M;
            if (clip >= 0.f) r = fmaxf(r - clip, 0.f);
        }
        r = logf(r + 1.0f);
        reward[b] = r;
        rs += r;
    }
    rs = block_sum(rs, red);
    if (threadIdx.x == 0) {
        metrics[EXORL_IM_EXTR_REWARD] = e / (float)B;
        metrics[EXORL_IM_INTR_REWARD] = rs / (float)B;
        metrics[EXORL_IM_RMS_MEAN] = st->M;
        metrics[EXORL_IM_RMS_STD] = sqrtf(st->S);
    }
}

// gradient reaching the trunk output of [obs; next_obs] (icm_apt.py:36-39): rows 0..B-1 from both nets' inputs,
// rows B..2B-1 from the inverse model's input and from being the forward model's regression target (= -dnhat)
__global__ __launch_bounds__(256) void apt_drep_kernel(const float* __restrict__ dxf, int64_t ldf, const float* __restrict__ dxb, int64_t ldb,
                                                       const float* __restrict__ dnhat, float* __restrict__ drep, int B, int R) {
    const int64_t n = (int64_t)B * R;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / R;
        const int c = (int)(i - r * R);
        drep[i] = dxf[r * ldf + c] + dxb[r * ldb + c];
        drep[n + i] = dxb[r * ldb + R + c] - dnhat[i];
    }
}

// d(loss)/d(obs) of plain ICM (icm.py:27-45): obs feeds the forward model's and the inverse model's input; dst is dense (B, O)
__global__ __launch_bounds__(256) void icm_dobs_kernel(const float* __restrict__ dxf, int64_t ldf, const float* __restrict__ dxb, int64_t ldb,
                                                       float* __restrict__ dst, int B, int O) {
    const int64_t n = (int64_t)B * O;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / O;
        const int c = (int)(i - r * O);
        dst[i] = dxf[r * ldf + c] + dxb[r * ldb + c];
    }
}

// SMM: d(loss_vae)/d(obs) = d/d(encoder input) - d/d(decoder output) over the observation columns (the reconstruction target is the
// input itself, F.mse_loss(obs_z, out), smm.py:66)
__global__ __launch_bounds__(256) void smm_dobs_kernel(const float* __restrict__ dx, const float* __restrict__ dout, int64_t ld, float* __restrict__ dst,
                                                       int B, int O) {
    const int64_t n = (int64_t)B * O;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / O;
        const int c = (int)(i - r * O);
        dst[i] = dx[r * ld + c] - dout[r * ld + c];
    }
}

// Disagreement reward (disagreement.py:35-47): unbiased variance over the ensemble's predictions, mean over features
struct PredSet { const float* p[EXORL_MAX_ENSEMBLE]; };
__global__ __launch_bounds__(256) void disagreement_reward_kernel(PredSet ps, int n, float* __restrict__ reward, int rows, int D) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float acc = 0.f;
    for (int j = lane; j < D; j += 64) {
        float mean = 0.f;
        for (int m = 0; m < n; ++m) mean += ps.p[m][(int64_t)row * D + j];
        mean /= (float)n;
        float q = 0.f;
        for (int m = 0; m < n; ++m) { const float d = ps.p[m][(int64_t)row * D + j] - mean; q += d * d; }
        acc += q / (float)(n > 1 ? n - 1 : 1);
    }
    acc = wave_sum(acc);
    if (lane == 0) reward[row] = acc / (float)D;
}

// wide rows (pixel encodings): one workgroup per row, 16-byte accesses, the ensemble loop unrolled over the kernel-argument pointers — the
// wave-per-row kernel above took 2.0 ms on five (1024, 39200) predictions (800 MB: ~160 us of HBM time)
__global__ __launch_bounds__(256) void disagreement_reward_wide_kernel(PredSet ps, int n, float* __restrict__ reward, int D) {
    __shared__ float red[17];
    const int row = blockIdx.x, n4 = D >> 2;
    const float inv_n = (float)n, inv_n1 = (float)(n > 1 ? n - 1 : 1);
    float acc = 0.f;
    for (int j = threadIdx.x; j < n4; j += 256) {
        float4 v[EXORL_MAX_ENSEMBLE];
#pragma unroll
        for (int m = 0; m < EXORL_MAX_ENSEMBLE; ++m)
            if (m < n) v[m] = reinterpret_cast<const float4*>(ps.p[m] + (int64_t)row * D)[j];
        float mx = 0.f, my = 0.f, mz = 0.f, mw = 0.f;
#pragma unroll
        for (int m = 0; m < EXORL_MAX_ENSEMBLE; ++m)
            if (m < n) { mx += v[m].x; my += v[m].y; mz += v[m].z; mw += v[m].w; }
        mx /= inv_n; my /= inv_n; mz /= inv_n; mw /= inv_n;
        float qx = 0.f, qy = 0.f, qz = 0.f, qw = 0.f;
#pragma unroll
        for (int m = 0; m < EXORL_MAX_ENSEMBLE; ++m)
            if (m < n) {
                const float dx = v[m].x - mx, dy = v[m].y - my, dz = v[m].z - mz, dw = v[m].w - mw;
                qx += dx * dx; qy += dy * dy; qz += dz * dz; qw += dw * dw;
            }
        acc += (qx / inv_n1 + qy / inv_n1) + (qz / inv_n1 + qw / inv_n1);
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) reward[row] = acc / (float)D;
}

// DIAYN (diayn.py:94-127): z = argmax(skill), log-softmax of the discriminator logits; nll = -lsm[z], hit = [argmax lsm == z],
// dlogits = (softmax - onehot(z)) / B (CrossEntropyLoss, mean), reward = (lsm[z] - log(1/S)) * scale. One wave per row.
__global__ __launch_bounds__(256) void diayn_kernel(const float* __restrict__ logits, const float* __restrict__ skill, int64_t lds_, int S,
                                                    float* __restrict__ nll, float* __restrict__ hit, float* __restrict__ dlogits,
                                                    float* reward, float scale, int rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float mx = -INFINITY, smx = -INFINITY;
    int amx = 0x7fffffff, asx = 0x7fffffff;
    for (int j = lane; j < S; j += 64) {
        const float v = logits[(int64_t)row * S + j], k = skill[(int64_t)row * lds_ + j];
        if (v > mx) { mx = v; amx = j; }
        if (k > smx) { smx = k; asx = j; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {            // first-occurrence arg-max, as torch.argmax / torch.max
        const float ov = __shfl_xor(mx, off, 64), ok = __shfl_xor(smx, off, 64);
        const int oi = __shfl_xor(amx, off, 64), oj = __shfl_xor(asx, off, 64);
        if (ov > mx || (ov == mx && oi < amx)) { mx = ov; amx = oi; }
        if (ok > smx || (ok == smx && oj < asx)) { smx = ok; asx = oj; }
    }
    float se = 0.f;
    for (int j = lane; j < S; j += 64) se += expf(logits[(int64_t)row * S + j] - mx);
    const float lse = mx + logf(wave_sum(se));
    const float lz = logits[(int64_t)row * S + asx] - lse;
    if (lane == 0) {
        if (nll) { nll[row] = -lz; hit[row] = amx == asx ? 1.f : 0.f; }
        if (reward) reward[row] = (lz - logf(1.0f / (float)S)) * scale;
    }
    if (dlogits) {
        const float invB = 1.0f / (float)rows;
        for (int j = lane; j < S; j += 64)
            dlogits[(int64_t)row * S + j] = (expf(logits[(int64_t)row * S + j] - lse) - (j == asx ? 1.f : 0.f)) * invB;
    }
}

// ---- Proto (proto.py) -------------------------------------------------------------------------------
// F.normalize(x, dim=1): y = x / max(||x||, 1e-12), one wave per row; in place allowed; nrm optional
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* x, float* y, float* __restrict__ nrm, int rows, int D) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float s = 0.f;
    for (int j = lane; j < D; j += 64) { const float v = x[(int64_t)row * D + j]; s += v * v; }
    const float n = fmaxf(sqrtf(wave_sum(s)), 1e-12f);
    for (int j = lane; j < D; j += 64) y[(int64_t)row * D + j] = x[(int64_t)row * D + j] / n;
    if (nrm && lane == 0) nrm[row] = n;
}
// dx = (dy - y (y . dy)) / ||x||
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ nrm,
                                                         float* __restrict__ dx, int rows, int D) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float s = 0.f;
    for (int j = lane; j < D; j += 64) s += y[(int64_t)row * D + j] * dy[(int64_t)row * D + j];
    s = wave_sum(s);
    const float n = nrm[row];
    for (int j = lane; j < D; j += 64) dx[(int64_t)row * D + j] = (dy[(int64_t)row * D + j] - y[(int64_t)row * D + j] * s) / n;
}
// Sinkhorn-Knopp on S = scores / tau in the (B, P) layout of the scores (the reference works on the transpose). The global
// max (Q -= Q.max(), proto.py:16) and the global sum (Q /= Q.sum(), :18) are formed from per-row values, a wave per row:
//   sk_rowmax: rowred[b] = max_p S[b,p]
//   sk_exp   : mx = max_b rowred[b] (every wave re-reduces the B values, fixed order); E = exp(S/tau - mx/tau); rowred[B+b] = sum_p E
//   sk_row(first): total = sum_b rowred[B+b] (again per wave), E /= total
__device__ __forceinline__ float sk_reduce_all(const float* __restrict__ v, int n, int lane, bool is_max) {
    float r = is_max ? -INFINITY : 0.f;
    for (int i = lane; i < n; i += 64) r = is_max ? fmaxf(r, v[i]) : r + v[i];
    return is_max ? wave_max(r) : wave_sum(r);
}
__global__ __launch_bounds__(256) void sk_rowmax_kernel(const float* __restrict__ S, int rows, int P, float* __restrict__ rowred) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float m = -INFINITY;
    for (int j = lane; j < P; j += 64) m = fmaxf(m, S[(int64_t)row * P + j]);
    m = wave_max(m);
    if (lane == 0) rowred[row] = m;
}
__global__ __launch_bounds__(256) void sk_exp_kernel(const float* __restrict__ S, float* __restrict__ E, int rows, int P, float inv_tau,
                                                     float* __restrict__ rowred) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float mx = sk_reduce_all(rowred, rows, lane, true) * inv_tau;
    float s = 0.f;
    for (int j = lane; j < P; j += 64) { const float e = expf(S[(int64_t)row * P + j] * inv_tau - mx); E[(int64_t)row * P + j] = e; s += e; }
    s = wave_sum(s);
    if (lane == 0) rowred[rows + row] = s;
}
// one Sinkhorn half-iteration pair, wave per row b: E[b,p] *= colscale_p, then the row is rescaled to sum `target`
// (1/B inside the loop, 1 for the final normalisation). first != 0: colscale_p = 1/total (the Q /= Q.sum() of proto.py:18).
__global__ __launch_bounds__(256) void sk_row_kernel(float* __restrict__ E, const float* __restrict__ colsum, const float* __restrict__ rowred,
                                                     int rows, int P, int first, float rP, float target) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float total = first ? sk_reduce_all(rowred + rows, rows, lane, false) : 1.0f;
    float s = 0.f;
    for (int j = lane; j < P; j += 64) {
        const float v = E[(int64_t)row * P + j] * (first ? 1.0f / total : rP / colsum[j]);
        E[(int64_t)row * P + j] = v;
        s += v;
    }
    if (first) return;
    s = wave_sum(s);
    for (int j = lane; j < P; j += 64) E[(int64_t)row * P + j] = E[(int64_t)row * P + j] * (target / s);
}
// loss_b = -sum_p q log_softmax(scores/tau); dscores = (softmax * sum_p q - q) / (B tau). One wave per row.
__global__ __launch_bounds__(256) void proto_loss_kernel(const float* __restrict__ scores, const float* __restrict__ q, float* __restrict__ dscores,
                                                         float* __restrict__ loss_row, int rows, int P, float inv_tau) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float mx = -INFINITY, qs = 0.f;
    for (int j = lane; j < P; j += 64) { mx = fmaxf(mx, scores[(int64_t)row * P + j] * inv_tau); qs += q[(int64_t)row * P + j]; }
    mx = wave_max(mx); qs = wave_sum(qs);
    float se = 0.f;
    for (int j = lane; j < P; j += 64) se += expf(scores[(int64_t)row * P + j] * inv_tau - mx);
    const float lse = mx + logf(wave_sum(se));
    float l = 0.f;
    const float sc = inv_tau / (float)rows;
    for (int j = lane; j < P; j += 64) {
        const float lp = scores[(int64_t)row * P + j] * inv_tau - lse, qq = q[(int64_t)row * P + j];
        l -= qq * lp;
        dscores[(int64_t)row * P + j] = (expf(lp) * qs - qq) * sc;
    }
    l = wave_sum(l);
    if (lane == 0) loss_row[row] = l;
}
// Categorical(softmax over the batch of scores[:, p]).sample() by inverse CDF (proto.py:109-112); the chosen z row goes straight
// into the candidate queue (proto.py:115-117). A workgroup owns 16 prototypes: the (B, 16) slab of scores is read as 64-byte row
// segments (a column per workgroup reads B separate cache lines), exp'd into LDS, and 16 lanes run the 16 serial double-precision
// prefix walks side by side (serial in double, as the oracle / torch.multinomial's cumulative table).
template <int PC_P>
__global__ __launch_bounds__(256) void proto_candidates_kernel(const float* __restrict__ scores, const float* __restrict__ z,
                                                               const float* __restrict__ u_in, uint64_t seed, uint64_t counter,
                                                               float* __restrict__ queue, int64_t qrow0, int B, int P, int D,
                                                               int* __restrict__ cand_out) {
    extern __shared__ float pr[];               // [B][PC_P + 1] unnormalised probabilities
    __shared__ float cmax[256 / PC_P][PC_P];
    __shared__ double csum[256 / PC_P][PC_P];
    __shared__ int pick[PC_P];
    const int p0 = blockIdx.x * PC_P;
    constexpr int NBL = 256 / PC_P;
    const int pl = threadIdx.x % PC_P, bl = threadIdx.x / PC_P;      // prototype lane, row lane
    const bool live = p0 + pl < P;
    float mx = -INFINITY;
    for (int b = bl; b < B; b += NBL) {
        const float v = live ? scores[(int64_t)b * P + p0 + pl] : 0.f;
        pr[b * (PC_P + 1) + pl] = v;
        mx = fmaxf(mx, v);
    }
    cmax[bl][pl] = mx;
    __syncthreads();
    mx = cmax[0][pl];
    for (int i = 1; i < NBL; ++i) mx = fmaxf(mx, cmax[i][pl]);
    for (int b = bl; b < B; b += NBL) pr[b * (PC_P + 1) + pl] = expf(pr[b * (PC_P + 1) + pl] - mx);
    __syncthreads();
    // inverse CDF in double, blocked: the NBL row lanes of a prototype each sum a contiguous chunk of rows in row order, one lane
    // chains the chunk totals in chunk order, and the walk to the first row with run > u * total restarts from the prefix of the
    // chunk that contains it. Same sequence of partial sums as one serial walk up to the association of double additions
    // (~1e-16 relative: the pick can differ only if u * total falls that close to a boundary).
    const int chunk = (B + NBL - 1) / NBL;
    {
        double c = 0.0;
        const int b0 = bl * chunk, b1 = b0 + chunk < B ? b0 + chunk : B;
        for (int b = b0; b < b1; ++b) c += (double)pr[b * (PC_P + 1) + pl];
        csum[bl][pl] = c;
    }
    __syncthreads();
    if (threadIdx.x < PC_P && live) {
        const int p = p0 + pl;
        double acc = 0.0;
        for (int i = 0; i < NBL; ++i) acc += csum[i][pl];
        float u;
        if (u_in) u = u_in[p];
        else { uint32_t c[4] = {(uint32_t)p, 7u, (uint32_t)counter, (uint32_t)(counter >> 32)}; Philox::gen(c, seed); u = (float)c[0] * 2.3283064365386963e-10f; }
        const double thr = (double)u * acc;
        double run = 0.0;
        int j = 0;
        while (j < NBL - 1 && !(run + csum[j][pl] > thr)) { run += csum[j][pl]; ++j; }
        const int b1 = (j + 1) * chunk < B ? (j + 1) * chunk : B;
        int k = b1 < B - 1 ? b1 : B - 1;          // reached only if rounding keeps the chunk's own walk at or below thr
        for (int b = j * chunk; b < b1; ++b) { run += (double)pr[b * (PC_P + 1) + pl]; if (run > thr) { k = b; break; } }
        pick[pl] = k;
        if (cand_out) cand_out[p] = k;
    }
    __syncthreads();
    for (int q = 0; q < PC_P && p0 + q < P; ++q) {
        const int k = pick[q];
        for (int j = threadIdx.x; j < D; j += blockDim.x) queue[(qrow0 + p0 + q) * D + j] = z[(int64_t)k * D + j];
    }
}
// reward = topk-th smallest distance (proto.py:119-124) + reward bookkeeping; single block
__global__ __launch_bounds__(1024) void kth_reward_kernel(const float* __restrict__ topk, const float* extr, float* reward, int B, int k,
                                                          float* __restrict__ metrics) {
    __shared__ float red[17];
    float e = 0.f;
    for (int i = threadIdx.x; i < B; i += blockDim.x) e += extr ? extr[i] : 0.f;
    e = block_sum(e, red);
    float rs = 0.f;
    for (int i = threadIdx.x; i < B; i += blockDim.x) { const float r = topk[(int64_t)i * k + (k - 1)]; reward[i] = r; rs += r; }
    rs = block_sum(rs, red);
    if (threadIdx.x == 0) { metrics[EXORL_IM_EXTR_REWARD] = e / (float)B; metrics[EXORL_IM_INTR_REWARD] = rs / (float)B; }
}

// ---- APS (aps.py:147-175) ---------------------------------------------------------------------------
// loss_b = -task . fn with fn = F.normalize(f); df = (dfn - fn (fn . dfn)) / max(||f||, 1e-12), dfn = -task / B. One wave per row.
__global__ __launch_bounds__(256) void aps_loss_kernel(const float* __restrict__ f, const float* __restrict__ task, int64_t ldt,
                                                       float* __restrict__ df, float* __restrict__ loss_row, int rows, int D) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float s = 0.f, tf = 0.f;
    for (int j = lane; j < D; j += 64) { const float v = f[(int64_t)row * D + j]; s += v * v; tf += v * task[(int64_t)row * ldt + j]; }
    const float n = fmaxf(sqrtf(wave_sum(s)), 1e-12f);
    const float tfn = wave_sum(tf) / n;                 // task . fn
    const float invB = 1.0f / (float)rows;
    for (int j = lane; j < D; j += 64) {
        const float fn = f[(int64_t)row * D + j] / n, dfn = -task[(int64_t)row * ldt + j] * invB;
        df[(int64_t)row * D + j] = (dfn - fn * (-tfn * invB)) / n;
    }
    if (lane == 0) loss_row[row] = -tfn;
}
// reward += task . rep / ||rep|| (aps.py:164-168), metrics split into the two parts; single block
__global__ __launch_bounds__(1024) void aps_sf_reward_kernel(const float* __restrict__ rep, const float* __restrict__ task, int64_t ldt,
                                                             float* reward, int B, int D, float* __restrict__ metrics) {
    __shared__ float red[17];
    float ss = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        float n2 = 0.f, tr = 0.f;
        for (int j = 0; j < D; ++j) { const float v = rep[(int64_t)b * D + j]; n2 += v * v; tr += v * task[(int64_t)b * ldt + j]; }
        const float sf = tr / sqrtf(n2);
        reward[b] += sf;
        ss += sf;
    }
    ss = block_sum(ss, red);
    if (threadIdx.x == 0) {
        metrics[EXORL_IM_ENT_REWARD] = metrics[EXORL_IM_INTR_REWARD];
        metrics[EXORL_IM_SF_REWARD] = ss / (float)B;
        metrics[EXORL_IM_INTR_REWARD] += ss / (float)B;
    }
}

// ---- SMM (smm.py) -----------------------------------------------------------------------------------
constexpr int SMM_VAE_HIDDEN = 150, SMM_CODE_DIM = 128;      // smm.py:38-45,98-101: fixed by the reference
__device__ __forceinline__ float philox_normal_i(uint64_t seed, uint64_t counter, uint32_t elem) {
    uint32_t c[4] = {elem, 11u, (uint32_t)counter, (uint32_t)(counter >> 32)};
    Philox::gen(c, seed);
    const float u1 = ((float)c[0] + 1.0f) * 2.3283064365386963e-10f, u2 = (float)c[1] * 2.3283064365386963e-10f;
    return sqrtf(-2.0f * __logf(u1)) * __cosf(6.283185307179586f * u2);
}
// code = eps * exp(0.5 logvar) + mu (smm.py:54-58); eps kept for the backward pass
__global__ __launch_bounds__(256) void vae_code_kernel(const float* __restrict__ mu, const float* __restrict__ lv, const float* __restrict__ eps_in,
                                                       uint64_t seed, uint64_t counter, float* __restrict__ eps, float* __restrict__ code, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float e = eps_in ? eps_in[i] : philox_normal_i(seed, counter, (uint32_t)i);
        eps[i] = e;
        code[i] = e * expf(0.5f * lv[i]) + mu[i];
    }
}
// per row: h_s_z = sum_j (x - out)^2 (smm.py:66-70), d(mean squared error)/d out
__global__ __launch_bounds__(256) void vae_out_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ out, float* __restrict__ dout,
                                                      float* __restrict__ hsz, int rows, int W) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float sc = -2.0f / ((float)rows * (float)W);
    float s = 0.f;
    for (int j = lane; j < W; j += 64) {
        const float d = x[(int64_t)row * ldx + j] - out[(int64_t)row * W + j];
        s += d * d;
        dout[(int64_t)row * W + j] = sc * d;
    }
    s = wave_sum(s);
    if (lane == 0) hsz[row] = s;
}
// wide rows (pixel encodings, W = 39200): one workgroup per row, 16-byte accesses (the wave-per-row kernel took 309 us for 480 MB)
__global__ __launch_bounds__(256) void vae_out_wide_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ out, float* __restrict__ dout,
                                                           float* __restrict__ hsz, int rows, int W) {
    __shared__ float red[17];
    const int row = blockIdx.x, n4 = W >> 2;
    const float sc = -2.0f / ((float)rows * (float)W);
    const float4* x4 = reinterpret_cast<const float4*>(x + (int64_t)row * ldx);
    const float4* o4 = reinterpret_cast<const float4*>(out + (int64_t)row * W);
    float4* g4 = reinterpret_cast<float4*>(dout + (int64_t)row * W);
    float s = 0.f;
    for (int j = threadIdx.x; j < n4; j += 256) {
        const float4 a = x4[j], b = o4[j];
        const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z, dw = a.w - b.w;
        s += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        g4[j] = make_float4(sc * dx, sc * dy, sc * dz, sc * dw);
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) hsz[row] = s;
}
// gradients at (mu, logvar) of beta * KL + reconstruction, given d/d code; kle_row = -0.5 sum (1 + lv - mu^2 - e^lv)
__global__ __launch_bounds__(256) void vae_latent_kernel(const float* __restrict__ dcode, const float* __restrict__ mu, const float* __restrict__ lv,
                                                         const float* __restrict__ eps, float* __restrict__ dmu, float* __restrict__ dlv,
                                                         float* __restrict__ kle_row, int rows, int C, float beta) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float invB = 1.0f / (float)rows;
    float k = 0.f;
    for (int j = lane; j < C; j += 64) {
        const int64_t i = (int64_t)row * C + j;
        const float m = mu[i], l = lv[i], el = expf(l), sd = expf(0.5f * l), dc = dcode[i];
        dmu[i] = dc + beta * m * invB;
        dlv[i] = dc * eps[i] * 0.5f * sd + beta * -0.5f * (1.0f - el) * invB;
        k += 1.0f + l - m * m - el;
    }
    k = wave_sum(k);
    if (lane == 0) kle_row[row] = -0.5f * k;
}
// reward and bookkeeping (smm.py:229-258); single block. The reference's 1-D log_p_star broadcasts its reward to (B,B): per
// sample that is rest_i + mean_j log_p_star_j for the TD gradient, plus var_j(log_p_star_j) in each critic's loss value.
__global__ __launch_bounds__(1024) void smm_reward_kernel(const float* __restrict__ obs, int64_t ld, const float* __restrict__ hsz,
                                                          const float* __restrict__ hzs, const float* extr, float* reward, int B, int Z,
                                                          float sec, float lec, float lcec, float gx, float gy, float* __restrict__ metrics, int encoded) {
    __shared__ float red[17];
    float e = 0.f, sl = 0.f, s1 = 0.f, s2 = 0.f;
    if (encoded) {                 // pixels: p*(s) is ignored (smm.py:232-235), the reward is a plain (B, 1) column
        float rs = 0.f;
        const float hz = lec * logf((float)Z);
        for (int b = threadIdx.x; b < B; b += blockDim.x) {
            e += extr ? extr[b] : 0.f;
            s1 += sec * hsz[b];
            s2 += lcec * hzs[b];
            const float r = sec * hsz[b] + hz + lcec * hzs[b];
            reward[b] = r;
            rs += r;
        }
        e = block_sum(e, red); s1 = block_sum(s1, red); s2 = block_sum(s2, red); rs = block_sum(rs, red);
        if (threadIdx.x == 0) {
            metrics[1] = rs / (float)B; metrics[2] = e / (float)B; metrics[3] = 0.f; metrics[4] = s1 / (float)B; metrics[6] = s2 / (float)B; metrics[7] = 0.f;
        }
        return;
    }
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        e += extr ? extr[b] : 0.f;
        const float dx = obs[(int64_t)b * ld] - gx, dy = obs[(int64_t)b * ld + 1] - gy;
        const float dist = sqrtf(dx * dx + dy * dy);
        sl += logf(dist > 1.0f ? 1.0f / dist : 1.0f);
        s1 += sec * hsz[b];
        s2 += lcec * hzs[b];
    }
    e = block_sum(e, red);
    const float lm = block_sum(sl, red) / (float)B;
    s1 = block_sum(s1, red); s2 = block_sum(s2, red);
    float var = 0.f, rs = 0.f;
    const float hz = lec * logf((float)Z);
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const float dx = obs[(int64_t)b * ld] - gx, dy = obs[(int64_t)b * ld + 1] - gy;
        const float dist = sqrtf(dx * dx + dy * dy);
        const float l = logf(dist > 1.0f ? 1.0f / dist : 1.0f) - lm;
        var += l * l;
        const float r = lm + sec * hsz[b] + hz + lcec * hzs[b];
        reward[b] = r;
        rs += r;
    }
    var = block_sum(var, red); rs = block_sum(rs, red);
    if (threadIdx.x == 0) {
        metrics[1] = rs / (float)B; metrics[2] = e / (float)B; metrics[3] = lm; metrics[4] = s1 / (float)B; metrics[6] = s2 / (float)B;
        metrics[7] = var / (float)B;
    }
}

static int grid_for(int64_t n) { const int64_t b = (n + 255) / 256; return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b)); }

int mlp_forward(const Mlp& m, const float* P, const float* x, int64_t ldx, int rows, int prec, hipStream_t s) {
    const int n = (int)m.L.size();
    for (int l = 0; l < n; ++l) {
        const Lin& L = m.L[l];
        GemmProblem p{l ? m.act[l - 1] : x, P + L.W, m.act[l], P + L.b, rows, L.out, L.in, l ? (int64_t)L.in : ldx, L.in, L.out};
        EXORL_TRY(gemm_grouped(prec, 0, 0, &p, 1, l < n - 1 || m.relu_last, false, s));
    }
    return 0;
}

// n nets of identical shape on the same input (an ensemble): each layer's GEMMs of all nets share one grouped launch
int mlp_forward_many(const Mlp* nets, int n, const float* P, const float* x, int64_t ldx, int rows, int prec, hipStream_t s) {
    EXORL_REQUIRE(n >= 1 && n <= 16, "mlp_forward_many: %d nets", n);
    const int nl = (int)nets[0].L.size();
    for (int l = 0; l < nl; ++l) {
        GemmProblem p[16];
        for (int m = 0; m < n; ++m) {
            const Lin& L = nets[m].L[l];
            p[m] = GemmProblem{l ? nets[m].act[l - 1] : x, P + L.W, nets[m].act[l], P + L.b, rows, L.out, L.in, l ? (int64_t)L.in : ldx, L.in, L.out};
        }
        EXORL_TRY(gemm_grouped(prec, 0, 0, p, n, l < nl - 1 || nets[0].relu_last, false, s));
    }
    return 0;
}
// backward of the same ensemble: masks + bias gradients per net, weight gradients and hidden dgrads grouped; dx (rows, in0) or null
int mlp_backward_many(const Mlp* nets, int n, const float* P, float* G, const float* x, int64_t ldx, int rows, int prec, hipStream_t s, float* dx) {
    EXORL_REQUIRE(n >= 1 && n <= 16, "mlp_backward_many: %d nets", n);
    const int nl = (int)nets[0].L.size();
    for (int l = nl - 1; l >= 0; --l) {
        GemmProblem w[16], g[16];
        for (int m = 0; m < n; ++m) {
            const Lin& L = nets[m].L[l];
            float* d = nets[m].dact[l];
            bool summed = false;
            if (l < nl - 1 || nets[m].relu_last) {
                summed = relu_bwd_colsum(d, nets[m].act[l], G + L.b, rows, L.out, s) == 0;
                if (!summed) {
                    const int64_t cnt = (int64_t)rows * L.out;
                    hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(cnt)), dim3(256), 0, s, d, nets[m].act[l], cnt);
                    EXORL_LAUNCH_CHECK();
                }
            }
            if (!summed) EXORL_TRY(colsum(d, G + L.b, rows, L.out, 1, 0, 0, s));
            w[m] = GemmProblem{d, l ? nets[m].act[l - 1] : x, G + L.W, nullptr, L.out, L.in, rows, L.out, l ? (int64_t)L.in : ldx, L.in};
            if (l) g[m] = GemmProblem{d, P + L.W, nets[m].dact[l - 1], nullptr, rows, L.in, L.out, L.out, L.in, L.in};
        }
        EXORL_TRY(gemm_grouped(prec, 1, 1, w, n, false, false, s));
        if (l) EXORL_TRY(gemm_grouped(prec, 0, 1, g, n, false, false, s));
        else if (dx)                               // d/d(input) summed over the nets in net order (one accumulating launch per net)
            for (int m = 0; m < n; ++m) {
                const Lin& L = nets[m].L[0];
                GemmProblem gx{nets[m].dact[0], P + L.W, dx, nullptr, rows, L.in, L.out, L.out, L.in, L.in};
                EXORL_TRY(gemm_grouped(prec, 0, 1, &gx, 1, false, m > 0, s));
            }
    }
    return 0;
}

// dact[last] holds d(loss)/d(output); writes parameter gradients into G and, if dx, d(loss)/d(input) (rows, in0)
int mlp_backward(const Mlp& m, const float* P, float* G, const float* x, int64_t ldx, int rows, float* dx, int prec, hipStream_t s) {
    const int n = (int)m.L.size();
    for (int l = n - 1; l >= 0; --l) {
        const Lin& L = m.L[l];
        float* d = m.dact[l];
        bool summed = false;
        if (l < n - 1 || m.relu_last) {
            summed = relu_bwd_colsum(d, m.act[l], G + L.b, rows, L.out, s) == 0;      // mask and bias gradient in one pass over dZ
            if (!summed) {
                const int64_t cnt = (int64_t)rows * L.out;
                hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(cnt)), dim3(256), 0, s, d, m.act[l], cnt);
                EXORL_LAUNCH_CHECK();
            }
        }
        if (!summed) EXORL_TRY(colsum(d, G + L.b, rows, L.out, 1, 0, 0, s));
        const float* in = l ? m.act[l - 1] : x;
        GemmProblem w{d, in, G + L.W, nullptr, L.out, L.in, rows, L.out, l ? (int64_t)L.in : ldx, L.in};          // dW[o][i] = sum_r d[r][o] in[r][i]
        EXORL_TRY(gemm_grouped(prec, 1, 1, &w, 1, false, false, s));
        float* dst = l ? m.dact[l - 1] : dx;
        if (dst) {
            GemmProblem g{d, P + L.W, dst, nullptr, rows, L.in, L.out, L.out, L.in, L.in};      // din[r][i] = sum_o d[r][o] W[o][i]
            EXORL_TRY(gemm_grouped(prec, 0, 1, &g, 1, false, false, s));
        }
    }
    return 0;
}

}  // namespace exorl

using namespace exorl;

struct exorl_intr {
    exorl_intr_cfg cfg;
    std::vector<ITensor> tensors;          // module.parameters() order
    int64_t total = 0, trainable = 0;      // flat sizes (floats): all parameters / the prefix the optimiser steps
    float* ws = nullptr;
    bool owns_ws = false;
    float* flat[4] = {nullptr, nullptr, nullptr, nullptr};
    Mlp net[EXORL_MAX_ENSEMBLE];           // RND: predictor, target; ICM(-APT): forward_net, backward_net; Disagreement: the ensemble; DIAYN: [0]
    int n_nets = 0;
    Lin trunk{};                           // APT: Linear(O, R) of the trunk; LayerNorm gain/beta offsets below
    int64_t ln_g = 0, ln_b = 0;
    float *xn = nullptr, *xf = nullptr, *xb = nullptr, *dxf = nullptr, *dxb = nullptr;
    float *x2 = nullptr, *z = nullptr, *rep = nullptr, *xhat = nullptr, *rstd = nullptr, *drep = nullptr, *dz = nullptr, *topk = nullptr, *d2 = nullptr, *skr = nullptr, *splitk = nullptr;    // d2: squared-distance scratch of the kNN (B x n_tgt)
    float *fe = nullptr, *be = nullptr, *metrics = nullptr, *bn = nullptr;
    RmsState* rms = nullptr;
    // Proto: predictor (Linear) in front of net[0] = projector; prototypes C; frozen predictor_target; candidate queue
    Lin pred{}, pred_t{};
    int64_t protos = 0;
    float *z1 = nullptr, *dz1 = nullptr, *sn = nullptr, *nrm = nullptr, *tn = nullptr, *scores_s = nullptr, *scores_t = nullptr,
          *dscores = nullptr, *dsn = nullptr, *colsum_p = nullptr, *scal = nullptr, *queue = nullptr;
    int64_t queue_ptr = 0;
    uint64_t cat_counter = 0;
    // SMM: net[0] = z_pred_net, net[1] = vae.enc (ReLU after both layers), net[2] = vae.dec; the two heads; sub-range of the vae
    Lin enc_mu{}, enc_lv{};
    int64_t vae_off = 0;
    float *mu = nullptr, *lv = nullptr, *eps = nullptr, *code = nullptr, *dcode = nullptr, *dmu = nullptr, *dlv = nullptr, *hsz = nullptr, *hzs = nullptr;
    int64_t t = 0;                         // optimiser steps taken
};

namespace exorl {

struct ICarver {
    float* base; int64_t off = 0;
    explicit ICarver(float* b) : base(b) {}
    float* take(int64_t n) { float* p = base ? base + off : nullptr; off += round_up(n, 64); return p; }
};

static int n_models_of(const exorl_intr_cfg& c) { return c.n_models > 0 ? c.n_models : 5; }     // disagreement.py:12 default

static void describe_intr(exorl_intr* it) {
    const auto& c = it->cfg;
    const int O = c.obs_dim, A = c.act_dim, H = c.hidden_dim, R = c.rep_dim;
    int64_t off = 0;
    auto add = [&](int64_t rows, int64_t cols) { const int64_t o = off; it->tensors.push_back({o, rows, cols}); off += round_up(rows * cols, 4); return o; };
    auto lin = [&](int in, int out) { Lin l{in, out, 0, 0}; l.W = add(out, in); l.b = add(out, 1); return l; };
    it->tensors.clear();
    for (auto& n : it->net) n.L.clear();
    it->n_nets = 2;
    if (c.kind == EXORL_INTR_RND) {
        for (int n = 0; n < 2; ++n) {
            it->net[n].L = {lin(O, H), lin(H, H), lin(H, R)};
            if (n == 0) it->trainable = round_up(off, 64);
            off = round_up(off, 64);
        }
    } else if (c.kind == EXORL_INTR_DISAGREEMENT) {
        it->n_nets = n_models_of(c);
        for (int n = 0; n < it->n_nets; ++n) it->net[n].L = {lin(O + A, H), lin(H, O)};
        it->trainable = round_up(off, 64);
    } else if (c.kind == EXORL_INTR_DIAYN || c.kind == EXORL_INTR_APS) {
        it->n_nets = 1;
        it->net[0].L = {lin(O, H), lin(H, H), lin(H, R)};             // R = skill_dim
        it->trainable = round_up(off, 64);
    } else if (c.kind == EXORL_INTR_SMM) {                            // R = z_dim
        const int W = O + R, V = SMM_VAE_HIDDEN, C = SMM_CODE_DIM;
        it->n_nets = 3;
        it->net[0].L = {lin(O, H), lin(H, H), lin(H, R)};
        off = round_up(off, 64);
        it->vae_off = off;
        it->net[1].L = {lin(W, V), lin(V, V)};
        it->net[1].relu_last = true;
        it->enc_mu = lin(V, C);
        it->enc_lv = lin(V, C);
        it->net[2].L = {lin(C, V), lin(V, V), lin(V, W)};
        it->trainable = round_up(off, 64);
    } else if (c.kind == EXORL_INTR_PROTO) {                          // R = pred_dim, H = proj_dim
        it->n_nets = 1;
        it->pred = lin(O, R);
        it->net[0].L = {lin(R, H), lin(H, R)};
        it->protos = add(c.num_protos, R);
        it->trainable = round_up(off, 64);
        off = it->trainable;
        it->pred_t = lin(O, R);                                       // predictor_target: same padded layout as predictor
    } else {
        int in_f = O + A, in_b = 2 * O, out_f = O;
        if (c.kind == EXORL_INTR_ICM_APT) {
            it->trunk = lin(O, R);
            it->ln_g = add(R, 1); it->ln_b = add(R, 1);
            in_f = R + A; in_b = 2 * R; out_f = R;
        }
        it->net[0].L = {lin(in_f, H), lin(H, out_f)};
        it->net[1].L = {lin(in_b, H), lin(H, A)};
        it->trainable = round_up(off, 64);
    }
    it->total = round_up(off, 64);
}

static void carve_intr(exorl_intr* it, ICarver& c) {
    const auto& g = it->cfg;
    const int64_t B = g.batch, O = g.obs_dim, R = g.rep_dim;
    it->flat[EXORL_T_PARAM] = c.take(it->total);
    for (int w = 1; w < 4; ++w) it->flat[w] = c.take(it->trainable);
    for (int n = 0; n < it->n_nets; ++n) {
        Mlp& m = it->net[n];
        m.act.clear(); m.dact.clear();
        for (const Lin& l : m.L) {
            m.act.push_back(c.take(B * l.out));
            m.dact.push_back((g.kind == EXORL_INTR_RND && n == 1) ? nullptr : c.take(B * l.out));
        }
    }
    it->fe = c.take(B * (g.kind == EXORL_INTR_DISAGREEMENT ? it->n_nets : 1)); it->be = c.take(B);
    it->metrics = c.take(EXORL_N_INTR_METRICS);
    it->rms = reinterpret_cast<RmsState*>(c.take(4));
    if (g.kind == EXORL_INTR_RND) {
        if (!(g.flags & EXORL_INTR_ENCODED)) {     // encoded rows arrive normalised (BatchNorm2d ran on the frames)
            it->xn = c.take(B * O);
            it->bn = c.take(2 * O + 1);
        }
    } else if (g.kind == EXORL_INTR_APS) {
        it->topk = c.take(B * g.knn_k);
        it->d2 = c.take(B * round_up(B, 64));
    } else if (g.kind == EXORL_INTR_SMM) {
        const int64_t C = SMM_CODE_DIM;
        it->mu = c.take(B * C); it->lv = c.take(B * C); it->eps = c.take(B * C); it->code = c.take(B * C); it->dcode = c.take(B * C);
        it->dmu = c.take(B * C); it->dlv = c.take(B * C); it->hsz = c.take(B); it->hzs = c.take(B);
        it->dxf = c.take(B * (O + R));                           // d/d(obs_z) of the VAE encoder (dobs_out)
    } else if (g.kind == EXORL_INTR_PROTO) {
        const int64_t P = g.num_protos;
        it->z1 = c.take(B * R); it->dz1 = c.take(B * R); it->sn = c.take(B * R); it->nrm = c.take(B); it->tn = c.take(B * R);
        it->scores_s = c.take(B * P); it->scores_t = c.take(B * P); it->dscores = c.take(B * P); it->dsn = c.take(B * R);
        it->colsum_p = c.take(P); it->scal = c.take(4); it->skr = c.take(2 * B);
        if (O >= 4096) it->splitk = c.take(16 * B * R);              // pixel features: split-K scratch of the predictor
        it->queue = c.take((int64_t)g.queue_size * R);
        it->topk = c.take(B * g.knn_k);
        it->d2 = c.take(B * round_up(g.queue_size, 64));
    } else if (g.kind == EXORL_INTR_DISAGREEMENT) {
        it->xf = c.take(B * it->net[0].L[0].in);
        it->dxf = c.take(B * it->net[0].L[0].in);
    } else if (g.kind != EXORL_INTR_DIAYN) {
        const int64_t in_f = it->net[0].L[0].in, in_b = it->net[1].L[0].in;
        it->xf = c.take(B * in_f); it->xb = c.take(B * in_b);
        it->dxf = c.take(B * in_f); it->dxb = c.take(B * in_b);
        if (g.kind == EXORL_INTR_ICM_APT) {
            it->x2 = c.take(2 * B * O); it->z = c.take(2 * B * R); it->rep = c.take(2 * B * R); it->xhat = c.take(2 * B * R);
            it->rstd = c.take(2 * B); it->drep = c.take(2 * B * R); it->dz = c.take(2 * B * R);
            it->topk = c.take(B * g.knn_k);
            it->d2 = c.take(B * round_up(B, 64));
        }
    }
}

static int intr_reset_state(exorl_intr* it) {
    const RmsState r0{0.f, 1.f, 1e-4};                    // utils.RMS.__init__ (utils.py:259-262)
    EXORL_CHECK_HIP(hipMemcpy(it->rms, &r0, sizeof(r0), hipMemcpyHostToDevice));
    if (it->bn) {                                         // BatchNorm1d buffers: running_mean 0, running_var 1, num_batches_tracked 0
        std::vector<float> b(2 * it->cfg.obs_dim + 1, 0.f);
        for (int i = 0; i < it->cfg.obs_dim; ++i) b[it->cfg.obs_dim + i] = 1.f;
        EXORL_CHECK_HIP(hipMemcpy(it->bn, b.data(), b.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    return 0;
}

static int launch_mean(const float* x, int n, float scale, float* out, int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(1024), 0, s, x, n, scale, out, accumulate);
    EXORL_LAUNCH_CHECK();
    return 0;
}

static int intr_adam(exorl_intr* it, hipStream_t s) {
    it->t += 1;
    return adam_step(it->flat[EXORL_T_PARAM], it->flat[EXORL_T_GRAD], it->flat[EXORL_T_ADAM_M], it->flat[EXORL_T_ADAM_V], it->trainable,
                     it->cfg.lr, 0.9f, 0.999f, 1e-8f, it->t, nullptr, 0.f, s);
}

int launch_concat(const float* a, int64_t lda, int ca, const float* b, int64_t ldb, int cb, float* dst, int rows, hipStream_t s) {
    hipLaunchKernelGGL(concat2_kernel, dim3(grid_for((int64_t)rows * (ca + cb))), dim3(256), 0, s, a, lda, ca, b, ldb, cb, dst, rows);
    EXORL_LAUNCH_CHECK();
    return 0;
}

// ---- RND -------------------------------------------------------------------------------------------
static int rnd_forward(exorl_intr* it, const exorl_intr_batch& b, bool with_target, float* dpred, hipStream_t s) {
    const auto& c = it->cfg;
    const int B = c.batch, O = c.obs_dim, R = c.rep_dim;
    const float* P = it->flat[EXORL_T_PARAM];
    if (c.flags & EXORL_INTR_ENCODED) {
        // pixels (rnd.py:26-27,35-39,47-53): BatchNorm2d + clamp ran on the frames, in front of the two encoders; obs is the agent's encoder
        // on them (the predictor's first stage), next_obs the frozen encoder copy's output (the target's first stage)
        EXORL_REQUIRE(b.next_obs, "intr_update: RND on encodings reads the frozen encoder's output from next_obs");
        EXORL_TRY(mlp_forward(it->net[0], P, b.obs, b.obs_ld, B, c.precision, s));
        EXORL_TRY(mlp_forward(it->net[1], P, b.next_obs, b.next_obs_ld, B, c.precision, s));
    } else {
        hipLaunchKernelGGL(bn_clamp_kernel, dim3(O), dim3(256), 0, s, b.obs, b.obs_ld, it->xn, B, O, c.clip_val, it->bn);
        EXORL_LAUNCH_CHECK();
        if (with_target) EXORL_TRY(mlp_forward_many(it->net, 2, P, it->xn, O, B, c.precision, s));      // predictor and frozen target: same input, same shapes
        else EXORL_TRY(mlp_forward(it->net[0], P, it->xn, O, B, c.precision, s));
    }
    hipLaunchKernelGGL(rnd_err_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, it->net[0].act[2], it->net[1].act[2], it->fe, dpred, B, R);
    EXORL_LAUNCH_CHECK();
    return 0;
}

// train: 0 reward only, 1 step + reward on the same rows, 2 step only (pixels: the reward pass draws a new augmentation and runs the
// encoder the step has just moved, rnd.py:98-103, so the caller encodes again in between)
static int rnd_update(exorl_intr* it, const exorl_intr_batch& b, int train, hipStream_t s) {
    const auto& c = it->cfg;
    const int B = c.batch;
    const bool enc = (c.flags & EXORL_INTR_ENCODED) != 0;
    EXORL_REQUIRE(enc || train != 2, "intr_update: RND's step-only call belongs to the encoded (pixel) variant");
    if (train) {                                                                                     // rnd.py:79-96
        EXORL_TRY(rnd_forward(it, b, true, it->net[0].dact[2], s));
        EXORL_TRY(launch_mean(it->fe, B, 1.0f / (float)B, it->metrics + EXORL_IM_LOSS, 0, s));
        EXORL_TRY(mlp_backward(it->net[0], it->flat[EXORL_T_PARAM], it->flat[EXORL_T_GRAD], enc ? b.obs : it->xn, enc ? b.obs_ld : (int64_t)c.obs_dim, B,
                               enc ? b.dobs_out : nullptr, c.precision, s));
        EXORL_TRY(intr_adam(it, s));
        if (train == 2) return 0;
    }
    // compute_intr_reward (rnd.py:98-103); states: same batch -> same BatchNorm output and frozen target, only the predictor moved
    EXORL_TRY(rnd_forward(it, b, !train, nullptr, s));
    hipLaunchKernelGGL(rnd_reward_kernel, dim3(1), dim3(1024), 0, s, it->fe, b.extr_reward, b.reward_out, B, c.scale, it->rms, it->metrics);
    EXORL_LAUNCH_CHECK();
    return 0;
}

// ---- ICM / ICM-APT ---------------------------------------------------------------------------------
static int icm_errors(exorl_intr* it, const float* tgt, int64_t ldt, const float* action, int64_t lda, bool grads, bool inverse, hipStream_t s) {
    const auto& c = it->cfg;
    const int D = it->net[0].L[1].out;
    EXORL_REQUIRE(!inverse || lda == c.act_dim, "intr: ICM needs a dense action matrix (ld == action_dim)");
    launch_icm_err(it->net[0].act[1], D, tgt, ldt, inverse ? it->net[1].act[1] : nullptr, action, c.act_dim, it->fe, it->be,
                   grads ? it->net[0].dact[1] : nullptr, grads ? it->net[1].dact[1] : nullptr, c.batch, 1.0f / (float)c.batch, s);
    EXORL_LAUNCH_CHECK();
    return 0;
}

static int icm_update(exorl_intr* it, const exorl_intr_batch& b, bool train, hipStream_t s) {
    const auto& c = it->cfg;
    const int B = c.batch, O = c.obs_dim, A = c.act_dim, prec = c.precision;
    const float* P = it->flat[EXORL_T_PARAM];
    float* G = it->flat[EXORL_T_GRAD];
    EXORL_TRY(launch_concat(b.obs, b.obs_ld, O, b.action, b.action_ld, A, it->xf, B, s));
    if (train) {                                                                                     // icm.py:64-84
        EXORL_TRY(launch_concat(b.obs, b.obs_ld, O, b.next_obs, b.next_obs_ld, O, it->xb, B, s));
        EXORL_TRY(mlp_forward(it->net[0], P, it->xf, O + A, B, prec, s));
        EXORL_TRY(mlp_forward(it->net[1], P, it->xb, 2 * O, B, prec, s));
        EXORL_TRY(icm_errors(it, b.next_obs, b.next_obs_ld, b.action, b.action_ld, true, true, s));
        EXORL_TRY(launch_mean(it->fe, B, 1.0f / (float)B, it->metrics + EXORL_IM_LOSS, 0, s));
        EXORL_TRY(launch_mean(it->be, B, 1.0f / (float)B, it->metrics + EXORL_IM_LOSS, 1, s));
        EXORL_TRY(mlp_backward(it->net[0], P, G, it->xf, O + A, B, b.dobs_out ? it->dxf : nullptr, prec, s));
        EXORL_TRY(mlp_backward(it->net[1], P, G, it->xb, 2 * O, B, b.dobs_out ? it->dxb : nullptr, prec, s));
        if (b.dobs_out) {                          // obs is an encoding: the caller's encoder continues the backward pass (icm.py:64-78)
            hipLaunchKernelGGL(icm_dobs_kernel, dim3(grid_for((int64_t)B * O)), dim3(256), 0, s, it->dxf, (int64_t)(O + A), it->dxb, (int64_t)(2 * O), b.dobs_out, B, O);
            EXORL_LAUNCH_CHECK();
        }
        EXORL_TRY(intr_adam(it, s));
    }
    EXORL_TRY(mlp_forward(it->net[0], P, it->xf, O + A, B, prec, s));                                // icm.py:86-92
    EXORL_TRY(icm_errors(it, b.next_obs, b.next_obs_ld, b.action, b.action_ld, false, false, s));
    hipLaunchKernelGGL(icm_reward_kernel, dim3(1), dim3(1024), 0, s, it->fe, b.extr_reward, b.reward_out, B, c.scale, it->metrics);
    EXORL_LAUNCH_CHECK();
    return 0;
}

static int apt_trunk(exorl_intr* it, const float* x, int64_t ldx, int rows, hipStream_t s) {    // Linear -> LayerNorm -> Tanh (icm_apt.py:21-22)
    const auto& c = it->cfg;
    const float* P = it->flat[EXORL_T_PARAM];
    GemmProblem p{x, P + it->trunk.W, it->z, P + it->trunk.b, rows, c.rep_dim, c.obs_dim, ldx, c.obs_dim, c.rep_dim};
    EXORL_TRY(gemm_grouped(c.precision, 0, 0, &p, 1, false, false, s));
    return ln_tanh_fwd(it->z, P + it->ln_g, P + it->ln_b, it->rep, it->xhat, it->rstd, rows, c.rep_dim, 1, 0, 0, s);
}

static int apt_update(exorl_intr* it, const exorl_intr_batch& b, bool train, hipStream_t s) {
    const auto& c = it->cfg;
    const int B = c.batch, O = c.obs_dim, A = c.act_dim, R = c.rep_dim, prec = c.precision;
    const float* P = it->flat[EXORL_T_PARAM];
    float* G = it->flat[EXORL_T_GRAD];
    if (train) {                                                                                     // icm_apt.py:33-50,86-104
        EXORL_TRY(launch_concat(b.obs, b.obs_ld, O, nullptr, 0, 0, it->x2, B, s));                   // x2 = [obs; next_obs] stacked by rows
        EXORL_TRY(launch_concat(b.next_obs, b.next_obs_ld, O, nullptr, 0, 0, it->x2 + (int64_t)B * O, B, s));
        EXORL_TRY(apt_trunk(it, it->x2, O, 2 * B, s));
        const float* rn = it->rep + (int64_t)B * R;
        EXORL_TRY(launch_concat(it->rep, R, R, b.action, b.action_ld, A, it->xf, B, s));
        EXORL_TRY(launch_concat(it->rep, R, R, rn, R, R, it->xb, B, s));
        EXORL_TRY(mlp_forward(it->net[0], P, it->xf, R + A, B, prec, s));
        EXORL_TRY(mlp_forward(it->net[1], P, it->xb, 2 * R, B, prec, s));
        EXORL_TRY(icm_errors(it, rn, R, b.action, b.action_ld, true, true, s));
        EXORL_TRY(launch_mean(it->fe, B, 1.0f / (float)B, it->metrics + EXORL_IM_LOSS, 0, s));
        EXORL_TRY(launch_mean(it->be, B, 1.0f / (float)B, it->metrics + EXORL_IM_LOSS, 1, s));
        EXORL_TRY(mlp_backward(it->net[0], P, G, it->xf, R + A, B, it->dxf, prec, s));
        EXORL_TRY(mlp_backward(it->net[1], P, G, it->xb, 2 * R, B, it->dxb, prec, s));
        hipLaunchKernelGGL(apt_drep_kernel, dim3(grid_for((int64_t)B * R)), dim3(256), 0, s, it->dxf, (int64_t)(R + A), it->dxb,
                           (int64_t)(2 * R), it->net[0].dact[1], it->drep, B, R);
        EXORL_LAUNCH_CHECK();
        EXORL_TRY(ln_param_grad(it->drep, it->rep, it->xhat, G + it->ln_g, G + it->ln_b, 2 * B, R, 1, 0, 0, s));
        EXORL_TRY(ln_tanh_bwd(it->drep, it->rep, it->xhat, it->rstd, P + it->ln_g, it->dz, 2 * B, R, 1, 0, 0, s));
        EXORL_TRY(colsum(it->dz, G + it->trunk.b, 2 * B, R, 1, 0, 0, s));
        GemmProblem w{it->dz, it->x2, G + it->trunk.W, nullptr, R, O, 2 * B, R, O, O};
        EXORL_TRY(gemm_grouped(prec, 1, 1, &w, 1, false, false, s));
        if (b.dobs_out) {                          // d(loss)/d(obs rows) = dz[:B] W_trunk (next_obs was encoded without a graph, icm_apt.py:116-118)
            GemmProblem gx{it->dz, P + it->trunk.W, b.dobs_out, nullptr, B, O, R, R, O, O};
            EXORL_TRY(gemm_grouped(prec, 0, 1, &gx, 1, false, false, s));
        }
        EXORL_TRY(intr_adam(it, s));
    }
    EXORL_TRY(apt_trunk(it, b.obs, b.obs_ld, B, s));                                                 // icm_apt.py:106-110
    EXORL_TRY(knn_topk(it->rep, B, it->rep, B, R, c.knn_k, it->topk, it->d2, s));
    hipLaunchKernelGGL(pbe_reward_kernel, dim3(1), dim3(1024), 0, s, it->topk, b.extr_reward, b.reward_out, B, c.knn_k, c.knn_avg, c.knn_rms,
                       c.knn_clip, it->rms, it->metrics);
    EXORL_LAUNCH_CHECK();
    return 0;
}

// ---- Disagreement ----------------------------------------------------------------------------------
static int disagreement_update(exorl_intr* it, const exorl_intr_batch& b, bool train, hipStream_t s) {
    const auto& c = it->cfg;
    const int B = c.batch, O = c.obs_dim, A = c.act_dim, prec = c.precision, n = it->n_nets;
    const float* P = it->flat[EXORL_T_PARAM];
    float* G = it->flat[EXORL_T_GRAD];
    EXORL_TRY(launch_concat(b.obs, b.obs_ld, O, b.action, b.action_ld, A, it->xf, B, s));
    if (train) {                                                                                     // disagreement.py:19-33,64-80
        EXORL_TRY(mlp_forward_many(it->net, n, P, it->xf, O + A, B, prec, s));          // the 5 models' layers share launches
        for (int m = 0; m < n; ++m) {
            launch_icm_err(it->net[m].act[1], O, b.next_obs, b.next_obs_ld, nullptr, nullptr, A, it->fe + (int64_t)m * B, nullptr, it->net[m].dact[1],
                           nullptr, B, 1.0f / ((float)B * (float)n), s);
            EXORL_LAUNCH_CHECK();
        }
        EXORL_TRY(mlp_backward_many(it->net, n, P, G, it->xf, O + A, B, prec, s, b.dobs_out ? it->dxf : nullptr));
        if (b.dobs_out) EXORL_TRY(launch_concat(it->dxf, O + A, O, nullptr, 0, 0, b.dobs_out, B, s));
        EXORL_TRY(launch_mean(it->fe, B * n, 1.0f / ((float)B * (float)n), it->metrics + EXORL_IM_LOSS, 0, s));
        EXORL_TRY(intr_adam(it, s));
    }
    PredSet ps{};
    EXORL_TRY(mlp_forward_many(it->net, n, P, it->xf, O + A, B, prec, s));                           // disagreement.py:35-47
    for (int m = 0; m < n; ++m) ps.p[m] = it->net[m].act[1];
    if (b.extr_reward) EXORL_TRY(launch_mean(b.extr_reward, B, 1.0f / (float)B, it->metrics + EXORL_IM_EXTR_REWARD, 0, s));
    bool wide = O >= 2048 && O % 4 == 0;
    for (int m = 0; m < n; ++m) wide = wide && reinterpret_cast<uintptr_t>(ps.p[m]) % 16 == 0;
    if (wide) hipLaunchKernelGGL(disagreement_reward_wide_kernel, dim3(B), dim3(256), 0, s, ps, n, b.reward_out, O);
    else hipLaunchKernelGGL(disagreement_reward_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, ps, n, b.reward_out, B, O);
    EXORL_LAUNCH_CHECK();
    return launch_mean(b.reward_out, B, 1.0f / (float)B, it->metrics + EXORL_IM_INTR_REWARD, 0, s);
}

// ---- DIAYN -----------------------------------------------------------------------------------------
static int diayn_update(exorl_intr* it, const exorl_intr_batch& b, bool train, hipStream_t s) {
    const auto& c = it->cfg;
    const int B = c.batch, O = c.obs_dim, S = c.rep_dim, prec = c.precision;
    const float* P = it->flat[EXORL_T_PARAM];
    if (train) {                                                                                     // diayn.py:78-92,107-127
        EXORL_TRY(mlp_forward(it->net[0], P, b.next_obs, b.next_obs_ld, B, prec, s));
        hipLaunchKernelGGL(diayn_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, it->net[0].act[2], b.skill, b.skill_ld, S, it->fe, it->be,
                           it->net[0].dact[2], (float*)nullptr, c.scale, B);
        EXORL_LAUNCH_CHECK();
        EXORL_TRY(launch_mean(it->fe, B, 1.0f / (float)B, it->metrics + EXORL_IM_LOSS, 0, s));
        EXORL_TRY(launch_mean(it->be, B, 1.0f / (float)B, it->metrics + EXORL_IM_ACC, 0, s));
        EXORL_TRY(mlp_backward(it->net[0], P, it->flat[EXORL_T_GRAD], b.next_obs, b.next_obs_ld, B, b.dobs_out, prec, s));   // dobs_out: d/d(next_obs)
        EXORL_TRY(intr_adam(it, s));
    }
    EXORL_TRY(mlp_forward(it->net[0], P, b.next_obs, b.next_obs_ld, B, prec, s));                    // diayn.py:94-105
    if (b.extr_reward) EXORL_TRY(launch_mean(b.extr_reward, B, 1.0f / (float)B, it->metrics + EXORL_IM_EXTR_REWARD, 0, s));
    hipLaunchKernelGGL(diayn_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, it->net[0].act[2], b.skill, b.skill_ld, S, (float*)nullptr,
                       (float*)nullptr, (float*)nullptr, b.reward_out, c.scale, B);
    EXORL_LAUNCH_CHECK();
    (void)O;
    return launch_mean(b.reward_out, B, 1.0f / (float)B, it->metrics + EXORL_IM_INTR_REWARD, 0, s);
}

// ---- SMM --------------------------------------------------------------------------------------------
static int adam_range(exorl_intr* it, int64_t off, int64_t n, float lr, hipStream_t s) {
    return adam_step(it->flat[EXORL_T_PARAM] + off, it->flat[EXORL_T_GRAD] + off, it->flat[EXORL_T_ADAM_M] + off, it->flat[EXORL_T_ADAM_V] + off, n,
                     lr, 0.9f, 0.999f, 1e-8f, it->t, nullptr, 0.f, s);
}

static int smm_update(exorl_intr* it, const exorl_intr_batch& b, bool train, hipStream_t s) {
    const auto& c = it->cfg;
    const int B = c.batch, O = c.obs_dim, Z = c.rep_dim, W = O + Z, V = SMM_VAE_HIDDEN, C = SMM_CODE_DIM, prec = c.precision;
    EXORL_REQUIRE(b.obs_ld >= W, "intr_update: SMM reads [obs | z] rows (obs_ld >= obs_dim + z_dim)");
    EXORL_REQUIRE(train, "intr_update: SMM's reward is defined by the losses of its own update step (smm.py:226-241); train must be set");
    const float* P = it->flat[EXORL_T_PARAM];
    float* G = it->flat[EXORL_T_GRAD];
    Mlp &zp = it->net[0], &enc = it->net[1], &dec = it->net[2];
    it->t += 1;
    // ---- update_vae (smm.py:173-185, VAE.loss :61-70) on obs_z
    EXORL_TRY(mlp_forward(enc, P, b.obs, b.obs_ld, B, prec, s));
    GemmProblem hd[2] = {{enc.act[1], P + it->enc_mu.W, it->mu, P + it->enc_mu.b, B, C, V, V, V, C},
                         {enc.act[1], P + it->enc_lv.W, it->lv, P + it->enc_lv.b, B, C, V, V, V, C}};
    EXORL_TRY(gemm_grouped(prec, 0, 0, hd, 2, false, false, s));
    hipLaunchKernelGGL(vae_code_kernel, dim3(grid_for((int64_t)B * C)), dim3(256), 0, s, it->mu, it->lv, b.cat_uniform, 0x736d6dull, it->cat_counter++,
                       it->eps, it->code, (int64_t)B * C);
    EXORL_LAUNCH_CHECK();
    EXORL_TRY(mlp_forward(dec, P, it->code, C, B, prec, s));
    if (W >= 2048 && W % 4 == 0 && b.obs_ld % 4 == 0 && reinterpret_cast<uintptr_t>(b.obs) % 16 == 0 && reinterpret_cast<uintptr_t>(dec.act[2]) % 16 == 0 &&
        reinterpret_cast<uintptr_t>(dec.dact[2]) % 16 == 0)
        hipLaunchKernelGGL(vae_out_wide_kernel, dim3(B), dim3(256), 0, s, b.obs, b.obs_ld, dec.act[2], dec.dact[2], it->hsz, B, W);
    else hipLaunchKernelGGL(vae_out_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, b.obs, b.obs_ld, dec.act[2], dec.dact[2], it->hsz, B, W);
    EXORL_LAUNCH_CHECK();
    EXORL_TRY(mlp_backward(dec, P, G, it->code, C, B, it->dcode, prec, s));
    hipLaunchKernelGGL(vae_latent_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, it->dcode, it->mu, it->lv, it->eps, it->dmu, it->dlv, it->fe, B, C, c.vae_beta);
    EXORL_LAUNCH_CHECK();
    EXORL_TRY(launch_mean(it->fe, B, c.vae_beta / (float)B, it->metrics + EXORL_IM_LOSS, 0, s));            // beta * kle
    EXORL_TRY(launch_mean(it->hsz, B, 1.0f / ((float)B * (float)W), it->metrics + EXORL_IM_LOSS, 1, s));    // + mse.mean()
    EXORL_TRY(colsum(it->dmu, G + it->enc_mu.b, B, C, 1, 0, 0, s));
    EXORL_TRY(colsum(it->dlv, G + it->enc_lv.b, B, C, 1, 0, 0, s));
    GemmProblem wg[2] = {{it->dmu, enc.act[1], G + it->enc_mu.W, nullptr, C, V, B, C, V, V}, {it->dlv, enc.act[1], G + it->enc_lv.W, nullptr, C, V, B, C, V, V}};
    EXORL_TRY(gemm_grouped(prec, 1, 1, wg, 2, false, false, s));
    GemmProblem d1{it->dmu, P + it->enc_mu.W, enc.dact[1], nullptr, B, V, C, C, V, V}, d2{it->dlv, P + it->enc_lv.W, enc.dact[1], nullptr, B, V, C, C, V, V};
    EXORL_TRY(gemm_grouped(prec, 0, 1, &d1, 1, false, false, s));
    EXORL_TRY(gemm_grouped(prec, 0, 1, &d2, 1, false, true, s));
    EXORL_TRY(mlp_backward(enc, P, G, b.obs, b.obs_ld, B, b.dobs_out ? it->dxf : nullptr, prec, s));
    if (b.dobs_out) {
        hipLaunchKernelGGL(smm_dobs_kernel, dim3(grid_for((int64_t)B * O)), dim3(256), 0, s, it->dxf, dec.dact[2], (int64_t)W, b.dobs_out, B, O);
        EXORL_LAUNCH_CHECK();
    }
    EXORL_TRY(adam_range(it, it->vae_off, it->trainable - it->vae_off, c.vae_lr, s));
    // ---- update_pred (smm.py:187-200): skill discriminator on the raw observation columns
    EXORL_TRY(mlp_forward(zp, P, b.obs, b.obs_ld, B, prec, s));
    hipLaunchKernelGGL(diayn_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, zp.act[2], b.skill, b.skill_ld, Z, it->hzs, it->be, zp.dact[2], (float*)nullptr, 1.0f, B);
    EXORL_LAUNCH_CHECK();
    EXORL_TRY(launch_mean(it->hzs, B, 1.0f / (float)B, it->metrics + 5, 0, s));
    EXORL_TRY(mlp_backward(zp, P, G, b.obs, b.obs_ld, B, nullptr, prec, s));
    EXORL_TRY(adam_range(it, 0, it->vae_off, c.sp_lr, s));
    // ---- reward (smm.py:229-246)
    hipLaunchKernelGGL(smm_reward_kernel, dim3(1), dim3(1024), 0, s, b.obs, b.obs_ld, it->hsz, it->hzs, b.extr_reward, b.reward_out, B, Z,
                       c.state_ent_coef, c.latent_ent_coef, c.latent_cond_ent_coef, c.goal_x, c.goal_y, it->metrics, (c.flags & EXORL_INTR_ENCODED) ? 1 : 0);
    EXORL_LAUNCH_CHECK();
    return 0;
}

// ---- APS -------------------------------------------------------------------------------------------
static int aps_update(exorl_intr* it, const exorl_intr_batch& b, bool train, hipStream_t s) {
    const auto& c = it->cfg;
    const int B = c.batch, D = c.rep_dim, prec = c.precision;
    const float* P = it->flat[EXORL_T_PARAM];
    if (train) {                                                                                     // aps.py:147-159,170-175
        EXORL_TRY(mlp_forward(it->net[0], P, b.next_obs, b.next_obs_ld, B, prec, s));
        hipLaunchKernelGGL(aps_loss_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, it->net[0].act[2], b.skill, b.skill_ld, it->net[0].dact[2], it->fe, B, D);
        EXORL_LAUNCH_CHECK();
        EXORL_TRY(launch_mean(it->fe, B, 1.0f / (float)B, it->metrics + EXORL_IM_LOSS, 0, s));
        EXORL_TRY(mlp_backward(it->net[0], P, it->flat[EXORL_T_GRAD], b.next_obs, b.next_obs_ld, B, b.dobs_out, prec, s));   // dobs_out: d/d(next_obs)
        EXORL_TRY(intr_adam(it, s));
    }
    EXORL_TRY(mlp_forward(it->net[0], P, b.next_obs, b.next_obs_ld, B, prec, s));                    // aps.py:161-168
    const float* rep = it->net[0].act[2];
    EXORL_TRY(knn_topk(rep, B, rep, B, D, c.knn_k, it->topk, it->d2, s));
    hipLaunchKernelGGL(pbe_reward_kernel, dim3(1), dim3(1024), 0, s, it->topk, b.extr_reward, b.reward_out, B, c.knn_k, c.knn_avg, c.knn_rms,
                       c.knn_clip, it->rms, it->metrics);
    hipLaunchKernelGGL(aps_sf_reward_kernel, dim3(1), dim3(1024), 0, s, rep, b.skill, b.skill_ld, b.reward_out, B, D, it->metrics);
    EXORL_LAUNCH_CHECK();
    return 0;
}

// ---- Proto ------------------------------------------------------------------------------------------
static int launch_l2norm(const float* x, float* y, float* nrm, int rows, int D, hipStream_t s) {
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, y, nrm, rows, D);
    EXORL_LAUNCH_CHECK();
    return 0;
}

// predictor Linear(obs_dim, pred_dim): on pixel features (obs_dim = 39200) a (B, 128) output is 32 tiles of a 39200-long reduction,
// so it runs as a 16-way split-K there
static int proto_predict(exorl_intr* it, const float* x, int64_t ldx, const float* W, const float* bias, float* out, hipStream_t s) {
    const auto& c = it->cfg;
    if (it->splitk) return linear_splitk(c.precision, x, ldx, W, bias, out, c.batch, c.rep_dim, c.obs_dim, it->splitk, 16, s);
    GemmProblem p{x, W, out, bias, c.batch, c.rep_dim, c.obs_dim, ldx, c.obs_dim, c.rep_dim};
    return gemm_grouped(c.precision, 0, 0, &p, 1, false, false, s);
}

static int proto_update(exorl_intr* it, const exorl_intr_batch& b, bool train, hipStream_t s, bool want_reward = true) {
    const auto& c = it->cfg;
    const int B = c.batch, O = c.obs_dim, D = c.rep_dim, P = c.num_protos, prec = c.precision;
    float* Pm = it->flat[EXORL_T_PARAM];
    float* G = it->flat[EXORL_T_GRAD];
    float* C = Pm + it->protos;
    const float inv_tau = 1.0f / c.tau;
    if (train) {                                                                                     // proto.py:126-157
        EXORL_TRY(launch_l2norm(C, C, nullptr, P, D, s));                                            // normalize_protos
        EXORL_TRY(proto_predict(it, b.obs, b.obs_ld, Pm + it->pred.W, Pm + it->pred.b, it->z1, s));
        EXORL_TRY(mlp_forward(it->net[0], Pm, it->z1, D, B, prec, s));
        EXORL_TRY(launch_l2norm(it->net[0].act[1], it->sn, it->nrm, B, D, s));
        GemmProblem ps{it->sn, C, it->scores_s, nullptr, B, P, D, D, D, P};
        EXORL_TRY(gemm_grouped(prec, 0, 0, &ps, 1, false, false, s));
        // target branch (no gradient): predictor_target on next_obs, Sinkhorn assignment
        const float* nt = b.next_obs_target ? b.next_obs_target : b.next_obs;
        const int64_t nt_ld = b.next_obs_target ? b.next_obs_target_ld : b.next_obs_ld;
        EXORL_TRY(proto_predict(it, nt, nt_ld, Pm + it->pred_t.W, Pm + it->pred_t.b, it->tn, s));
        EXORL_TRY(launch_l2norm(it->tn, it->tn, nullptr, B, D, s));
        GemmProblem pq{it->tn, C, it->scores_t, nullptr, B, P, D, D, D, P};
        EXORL_TRY(gemm_grouped(prec, 0, 0, &pq, 1, false, false, s));
        hipLaunchKernelGGL(sk_rowmax_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, it->scores_t, B, P, it->skr);
        hipLaunchKernelGGL(sk_exp_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, it->scores_t, it->scores_t, B, P, inv_tau, it->skr);
        hipLaunchKernelGGL(sk_row_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, it->scores_t, it->colsum_p, it->skr, B, P, 1, 0.f, 0.f);
        EXORL_LAUNCH_CHECK();
        for (int iter = 0; iter < 3; ++iter) {     // u = r / Q.sum(1); Q *= u; Q *= c / Q.sum(0)   (+ the final Q / Q.sum(0) folded into the last pass)
            EXORL_TRY(colsum(it->scores_t, it->colsum_p, B, P, 1, 0, 0, s));
            hipLaunchKernelGGL(sk_row_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, it->scores_t, it->colsum_p, it->skr, B, P, 0, 1.0f / (float)P,
                               iter == 2 ? 1.0f : 1.0f / (float)B);
            EXORL_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(proto_loss_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, it->scores_s, it->scores_t, it->dscores, it->fe, B, P, inv_tau);
        EXORL_LAUNCH_CHECK();
        EXORL_TRY(launch_mean(it->fe, B, 1.0f / (float)B, it->metrics + EXORL_IM_LOSS, 0, s));
        GemmProblem gc{it->dscores, it->sn, G + it->protos, nullptr, P, D, B, P, D, D};              // dC = dscores^T sn
        EXORL_TRY(gemm_grouped(prec, 1, 1, &gc, 1, false, false, s));
        GemmProblem gs{it->dscores, C, it->dsn, nullptr, B, D, P, P, D, D};                          // dsn = dscores C
        EXORL_TRY(gemm_grouped(prec, 0, 1, &gs, 1, false, false, s));
        hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, it->dsn, it->sn, it->nrm, it->net[0].dact[1], B, D);
        EXORL_LAUNCH_CHECK();
        EXORL_TRY(mlp_backward(it->net[0], Pm, G, it->z1, D, B, it->dz1, prec, s));
        EXORL_TRY(colsum(it->dz1, G + it->pred.b, B, D, 1, 0, 0, s));
        GemmProblem gw{it->dz1, b.obs, G + it->pred.W, nullptr, D, O, B, D, b.obs_ld, O};
        EXORL_TRY(gemm_grouped(prec, 1, 1, &gw, 1, false, false, s));
        if (b.dobs_out) {                      // d(loss)/d(obs) = dz1 Wp, before Wp moves: the caller's encoder continues the backward pass
            GemmProblem gx{it->dz1, Pm + it->pred.W, b.dobs_out, nullptr, B, O, D, D, O, O};
            EXORL_TRY(gemm_grouped(prec, 0, 1, &gx, 1, false, false, s));
        }
        EXORL_TRY(intr_adam(it, s));
        // utils.soft_update_params(predictor, predictor_target, encoder_target_tau) (proto.py:202-203): nothing reads the target before the next update
        EXORL_TRY(soft_update(Pm + it->pred.W, Pm + it->pred_t.W, it->pred_t.b + round_up(c.rep_dim, 4) - it->pred_t.W, c.target_tau, s));
    }
    if (!want_reward) return 0;
    // compute_intr_reward(next_obs) (proto.py:103-124)
    EXORL_TRY(launch_l2norm(C, C, nullptr, P, D, s));
    EXORL_TRY(proto_predict(it, b.next_obs, b.next_obs_ld, Pm + it->pred.W, Pm + it->pred.b, it->sn, s));
    EXORL_TRY(launch_l2norm(it->sn, it->sn, nullptr, B, D, s));
    GemmProblem pc{it->sn, C, it->scores_s, nullptr, B, P, D, D, D, P};
    EXORL_TRY(gemm_grouped(prec, 0, 0, &pc, 1, false, false, s));
#define EXORL_PCAND(PP) hipLaunchKernelGGL(proto_candidates_kernel<PP>, dim3(cdiv(P, PP)), dim3(256), (size_t)B * (PP + 1) * sizeof(float), s, it->scores_s, \
                                            it->sn, b.cat_uniform, 0x70726f746full, it->cat_counter++, it->queue, it->queue_ptr, B, P, D, (int*)nullptr)
    if ((size_t)B * 17 * sizeof(float) <= 64 * 1024) EXORL_PCAND(16);
    else if ((size_t)B * 9 * sizeof(float) <= 64 * 1024) EXORL_PCAND(8);
    else {
        EXORL_REQUIRE((size_t)B * 5 * sizeof(float) <= 64 * 1024, "intr: Proto candidate sampling supports batch <= 3276 (got %d)", B);
        EXORL_PCAND(4);
    }
#undef EXORL_PCAND
    EXORL_LAUNCH_CHECK();
    it->queue_ptr = (it->queue_ptr + P) % c.queue_size;
    EXORL_TRY(knn_topk(it->sn, B, it->queue, c.queue_size, D, c.knn_k, it->topk, it->d2, s));
    hipLaunchKernelGGL(kth_reward_kernel, dim3(1), dim3(1024), 0, s, it->topk, b.extr_reward, b.reward_out, B, c.knn_k, it->metrics);
    EXORL_LAUNCH_CHECK();
    return 0;
}

}  // namespace exorl

extern "C" {

static int check_intr_cfg(const exorl_intr_cfg* cfg) {
    EXORL_REQUIRE(cfg, "intr: null cfg");
    EXORL_REQUIRE(cfg->kind >= EXORL_INTR_RND && cfg->kind <= EXORL_INTR_SMM, "intr: unknown kind %d", cfg->kind);
    EXORL_REQUIRE(cfg->kind != EXORL_INTR_SMM || (cfg->rep_dim >= 1 && cfg->rep_dim <= 1024 && cfg->obs_dim >= 2 && cfg->sp_lr > 0.f && cfg->vae_lr > 0.f),
                  "intr: SMM needs z_dim in [1, 1024], obs_dim >= 2 (p* reads obs[:, :2]) and positive sp_lr / vae_lr");
    EXORL_REQUIRE(cfg->kind != EXORL_INTR_APS || (cfg->knn_k >= 1 && cfg->knn_k <= 64 && cfg->knn_k <= cfg->batch && cfg->batch <= 4096),
                  "intr: APS needs 1 <= knn_k <= min(64, batch) and batch <= 4096 (got k=%d B=%d)", cfg->knn_k, cfg->batch);
    EXORL_REQUIRE(cfg->kind != EXORL_INTR_PROTO || (cfg->num_protos >= 1 && cfg->queue_size >= cfg->num_protos && cfg->queue_size % cfg->num_protos == 0 &&
                  cfg->queue_size <= 4096 && cfg->knn_k >= 1 && cfg->knn_k <= 64 && cfg->knn_k <= cfg->queue_size && cfg->tau > 0.f && cfg->batch <= 8192),
                  "intr: Proto needs num_protos >= 1, queue_size a multiple of num_protos and <= 4096, 1 <= topk <= 64, tau > 0 (got %d, %d, %d, %g)",
                  cfg->num_protos, cfg->queue_size, cfg->knn_k, (double)cfg->tau);
    EXORL_REQUIRE(cfg->n_models >= 0 && cfg->n_models <= EXORL_MAX_ENSEMBLE, "intr: n_models=%d out of range (<= %d)", cfg->n_models, EXORL_MAX_ENSEMBLE);
    EXORL_REQUIRE(cfg->obs_dim > 0 && cfg->act_dim > 0 && cfg->act_dim <= 64 && cfg->hidden_dim > 0 && cfg->batch > 0,
                  "intr: unsupported dims O=%d A=%d (<=64) H=%d B=%d", cfg->obs_dim, cfg->act_dim, cfg->hidden_dim, cfg->batch);
    EXORL_REQUIRE(cfg->kind == EXORL_INTR_ICM || cfg->kind == EXORL_INTR_DISAGREEMENT || (cfg->rep_dim > 0 && (cfg->kind != EXORL_INTR_ICM_APT || cfg->rep_dim <= 1024)),
                  "intr: rep_dim=%d out of range (ICM-APT trunk: <= 1024)", cfg->rep_dim);
    EXORL_REQUIRE(cfg->kind != EXORL_INTR_ICM_APT || (cfg->knn_k >= 1 && cfg->knn_k <= 64 && cfg->knn_k <= cfg->batch && cfg->batch <= 4096),
                  "intr: ICM-APT needs 1 <= knn_k <= min(64, batch) and batch <= 4096 (got k=%d B=%d)", cfg->knn_k, cfg->batch);
    EXORL_REQUIRE(cfg->precision >= EXORL_PREC_F32 && cfg->precision <= EXORL_PREC_BF16X6, "intr: unknown precision %d", cfg->precision);
    return 0;
}

size_t exorl_intr_workspace_bytes(const exorl_intr_cfg* cfg) {
    if (check_intr_cfg(cfg) != 0) return 0;
    exorl_intr tmp;
    tmp.cfg = *cfg;
    describe_intr(&tmp);
    ICarver sizing(nullptr);
    carve_intr(&tmp, sizing);
    return (size_t)sizing.off * sizeof(float);
}

int exorl_intr_create(const exorl_intr_cfg* cfg, void* workspace, size_t workspace_bytes, exorl_intr_t** out) {
    EXORL_REQUIRE(out, "intr_create: null argument");
    EXORL_TRY(check_intr_cfg(cfg));
    const size_t bytes = exorl_intr_workspace_bytes(cfg);
    EXORL_REQUIRE(!workspace || (workspace_bytes >= bytes && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0),
                  "intr_create: workspace of %zu bytes (need %zu, 256-byte aligned)", workspace_bytes, bytes);
    auto* it = new exorl_intr();
    it->cfg = *cfg;
    describe_intr(it);
    if (workspace) {
        it->ws = static_cast<float*>(workspace);
    } else {
        if (hipMalloc(&it->ws, bytes) != hipSuccess) {
            set_error("intr_create: hipMalloc(%zu) failed", bytes);
            delete it;
            return 1;
        }
        it->owns_ws = true;
    }
    ICarver c(it->ws);
    carve_intr(it, c);
    int rc = hipMemset(it->ws, 0, bytes) == hipSuccess ? 0 : 1;
    if (rc != 0) set_error("intr_create: hipMemset failed");
    if (rc == 0) rc = intr_reset_state(it);
    if (rc != 0) { if (it->owns_ws) (void)hipFree(it->ws); delete it; return rc; }
    *out = it;
    return 0;
}

int exorl_intr_destroy(exorl_intr_t* it) {
    if (!it) return 0;
    (void)hipDeviceSynchronize();
    if (it->owns_ws) (void)hipFree(it->ws);
    delete it;
    return 0;
}

int exorl_intr_num_tensors(exorl_intr_t* it, int32_t* n) {
    EXORL_REQUIRE(it && n, "intr_num_tensors: null argument");
    *n = (int32_t)it->tensors.size();
    return 0;
}

int exorl_intr_tensor(exorl_intr_t* it, int32_t index, int32_t what, void** ptr, int64_t* rows, int64_t* cols) {
    EXORL_REQUIRE(it && ptr && rows && cols, "intr_tensor: null argument");
    EXORL_REQUIRE(index >= 0 && index < (int32_t)it->tensors.size(), "intr_tensor: index %d out of range", index);
    EXORL_REQUIRE(what >= EXORL_T_PARAM && what <= EXORL_T_ADAM_V, "intr_tensor: unknown buffer %d", what);
    const ITensor& t = it->tensors[index];
    EXORL_REQUIRE(what == EXORL_T_PARAM || t.off < it->trainable, "intr_tensor: tensor %d is frozen (no gradient / optimiser state)", index);
    *ptr = it->flat[what] + t.off; *rows = t.rows; *cols = t.cols;
    return 0;
}

int exorl_intr_flat(exorl_intr_t* it, int32_t what, void** ptr, int64_t* numel) {
    EXORL_REQUIRE(it && ptr && numel, "intr_flat: null argument");
    EXORL_REQUIRE(what >= EXORL_T_PARAM && what <= EXORL_T_ADAM_V, "intr_flat: unknown buffer %d", what);
    *ptr = it->flat[what];
    *numel = what == EXORL_T_PARAM ? it->total : it->trainable;
    return 0;
}

int exorl_intr_state(exorl_intr_t* it, void** rms_dev, void** bn_dev, int64_t* bn_numel) {
    EXORL_REQUIRE(it && rms_dev && bn_dev && bn_numel, "intr_state: null argument");
    *rms_dev = it->rms; *bn_dev = it->bn; *bn_numel = it->bn ? 2 * it->cfg.obs_dim + 1 : 0;
    return 0;
}

int exorl_intr_update(exorl_intr_t* it, const exorl_intr_batch* b, int32_t train, void* stream) {
    EXORL_REQUIRE(it && b && b->obs && b->reward_out, "intr_update: null argument");
    const int k = it->cfg.kind;
    EXORL_REQUIRE(k == EXORL_INTR_RND || k == EXORL_INTR_SMM || b->next_obs, "intr_update: this module needs next_obs");
    EXORL_REQUIRE(k == EXORL_INTR_RND || k == EXORL_INTR_DIAYN || k == EXORL_INTR_PROTO || k == EXORL_INTR_APS || k == EXORL_INTR_SMM || b->action, "intr_update: this module needs action");
    EXORL_REQUIRE((k != EXORL_INTR_DIAYN && k != EXORL_INTR_APS && k != EXORL_INTR_SMM) || b->skill, "intr_update: DIAYN / APS / SMM need the skill / task matrix");
    EXORL_REQUIRE(b->obs_ld >= it->cfg.obs_dim && (!b->next_obs || b->next_obs_ld >= it->cfg.obs_dim) && (!b->action || b->action_ld >= it->cfg.act_dim) &&
                  (!b->skill || (k != EXORL_INTR_DIAYN && k != EXORL_INTR_APS && k != EXORL_INTR_SMM) || b->skill_ld >= it->cfg.rep_dim), "intr_update: a leading dimension is smaller than its row width");
    hipStream_t s = as_stream(stream);
    switch (k) {
        case EXORL_INTR_RND: return rnd_update(it, *b, train, s);
        case EXORL_INTR_ICM: return icm_update(it, *b, train != 0, s);
        case EXORL_INTR_ICM_APT: return apt_update(it, *b, train != 0, s);
        case EXORL_INTR_DISAGREEMENT: return disagreement_update(it, *b, train != 0, s);
        case EXORL_INTR_PROTO: return proto_update(it, *b, train != 0, s, train != 2);
        case EXORL_INTR_APS: return aps_update(it, *b, train != 0, s);
        case EXORL_INTR_SMM: return smm_update(it, *b, train != 0, s);
        default: return diayn_update(it, *b, train != 0, s);
    }
}

int exorl_intr_queue(exorl_intr_t* it, void** queue_dev, int64_t* rows, int64_t* cols, int64_t* ptr_inout, int32_t set) {
    EXORL_REQUIRE(it && queue_dev && rows && cols && ptr_inout && it->queue, "intr_queue: not a Proto module / null argument");
    *queue_dev = it->queue; *rows = it->cfg.queue_size; *cols = it->cfg.rep_dim;
    if (set) {
        EXORL_REQUIRE(*ptr_inout >= 0 && *ptr_inout < it->cfg.queue_size && *ptr_inout % it->cfg.num_protos == 0, "intr_queue: bad write pointer");
        it->queue_ptr = *ptr_inout;
    } else {
        *ptr_inout = it->queue_ptr;
    }
    return 0;
}

int exorl_intr_metrics(exorl_intr_t* it, float* host, void* stream) {
    EXORL_REQUIRE(it && host, "intr_metrics: null argument");
    EXORL_CHECK_HIP(hipMemcpyAsync(host, it->metrics, sizeof(float) * EXORL_N_INTR_METRICS, hipMemcpyDeviceToHost, as_stream(stream)));
    EXORL_CHECK_HIP(hipStreamSynchronize(as_stream(stream)));
    return 0;
}

int exorl_intr_opt_steps(exorl_intr_t* it, int64_t* steps, int32_t set) {
    EXORL_REQUIRE(it && steps, "intr_opt_steps: null argument");
    if (set) { EXORL_REQUIRE(*steps >= 0, "intr_opt_steps: negative step count"); it->t = *steps; }
    else *steps = it->t;
    return 0;
}

int exorl_intr_counter(exorl_intr_t* m, uint64_t* counter_inout, int32_t set) {
    EXORL_REQUIRE(m && counter_inout, "intr_counter: null argument");
    if (set) m->cat_counter = *counter_inout;
    else *counter_inout = m->cat_counter;
    return 0;
}

}  // extern "C"
