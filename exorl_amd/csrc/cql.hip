// CQL-specific kernels: tanh-Gaussian (SquashedNormal) sampling and log-prob, the
// logsumexp conservative penalty and its softmax gradient, the entropy-temperature Adam scalar.
// Reference: /root/reference/agents/offline_learning/cql.py:133-263, /root/reference/utils/utils.py:152-196.
// Everything here is (B,A)- or (10B,1)-sized elementwise/row work around the shared MLP kernels.
#include "kernels.h"

namespace exorl {

__device__ __forceinline__ float philox_normal_k(uint64_t seed, uint64_t counter, uint32_t elem) {
    uint32_t c[4] = {elem, 0u, (uint32_t)counter, (uint32_t)(counter >> 32)};
    Philox::gen(c, seed);
    const float u1 = ((float)c[0] + 1.0f) * 2.3283064365386963e-10f;
    const float u2 = (float)c[1] * 2.3283064365386963e-10f;
    return sqrtf(-2.0f * __logf(u1)) * __cosf(6.283185307179586f * u2);
}
__device__ __forceinline__ float philox_uniform_pm1(uint64_t seed, uint64_t counter, uint32_t elem) {
    uint32_t c[4] = {elem, 1u, (uint32_t)counter, (uint32_t)(counter >> 32)};
    Philox::gen(c, seed);
    return -1.0f + 2.0f * ((float)c[0] * 2.3283064365386963e-10f);
}
__device__ __forceinline__ float draw(const float* buf, const CqlNoise& nz, int k, int64_t e) {
    return buf ? buf[e] : philox_normal_k(nz.seed, (nz.counter_ptr ? *nz.counter_ptr : 0ull) * 8 + k, (uint32_t)e);
}

// mu = tanh(raw[:A]); std = exp(clamp(raw[A:], -10, 2))   (cql.py:24-28)
__device__ __forceinline__ void policy_of(const float* raw_row, int A, int j, float& mu, float& stdv) {
    mu = tanhf(raw_row[j]);
    stdv = expf(fminf(fmaxf(raw_row[A + j], -10.0f), 2.0f));
}

__global__ __launch_bounds__(256) void cql_build_inputs_kernel(const float* __restrict__ obs, const float* __restrict__ action,
                                                               const float* __restrict__ raw2, CqlNoise nz,
                                                               float* __restrict__ xc_next, float* __restrict__ x_all, int B, int O,
                                                               int A, int n) {
    const int W = O + A;
    const int64_t R = (int64_t)(3 * n + 1) * B;
    // rows R .. R+B-1 of the index space are the next_action rows of xc_next (action columns only)
    const int64_t total = (R + B) * W;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / W;
        const int c = (int)(i % W);
        if (row >= R) {                                    // next_action = pi(next_obs).sample()   (cql.py:158-159)
            if (c < O) continue;
            const int b = (int)(row - R), j = c - O;
            float mu, sd;
            policy_of(raw2 + (int64_t)b * 2 * A, A, j, mu, sd);
            xc_next[(int64_t)b * W + c] = tanhf(mu + sd * draw(nz.z_next, nz, 0, (int64_t)b * A + j));
            continue;
        }
        const int blk = (int)(row / ((int64_t)n * B));     // 0 rand, 1 pi(obs), 2 pi(next_obs), 3 data
        const int64_t r = row - (int64_t)blk * n * B;      // = i_sample*B + b   (obs.unsqueeze(0).repeat(n,1,1), cql.py:141-146)
        const int b = (int)(r % B);
        if (c < O) { x_all[i] = obs[(int64_t)b * O + c]; continue; }
        const int j = c - O;
        const int64_t e = r * A + j;                       // element of an (n,B,A) draw
        float v;
        if (blk == 3) {
            v = action[(int64_t)b * A + j];
        } else if (blk == 0) {
            v = nz.u_rand ? nz.u_rand[e] : philox_uniform_pm1(nz.seed, (nz.counter_ptr ? *nz.counter_ptr : 0ull) * 8 + 1, (uint32_t)e);
        } else {
            float mu, sd;
            policy_of(raw2 + (int64_t)(blk == 1 ? B + b : b) * 2 * A, A, j, mu, sd);      // rows B.. = obs half of the stacked forward
            v = tanhf(mu + sd * draw(blk == 1 ? nz.z_cur : nz.z_nxt, nz, blk == 1 ? 2 : 3, e));
        }
        x_all[i] = v;
    }
}

int cql_build_inputs(const float* obs, const float* action, const float* raw2, CqlNoise nz, float* xc_next, float* x_all, int B,
                     int O, int A, int n, hipStream_t s) {
    const int64_t total = ((int64_t)(3 * n + 1) * B + B) * (O + A);
    int blocks = cdiv(total, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(cql_build_inputs_kernel, dim3(blocks), dim3(256), 0, s, obs, action, raw2, nz, xc_next, x_all, B, O, A, n);
    EXORL_LAUNCH_CHECK();
    return 0;
}

template <int NV>
__device__ __forceinline__ void block_sum_c(float (&v)[NV], float (*sm)[16]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float s = wave_sum(v[i]);
        if (lane == 0) sm[i][wave] = s;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float t = 0.f;
        for (int w = 0; w < nw; ++w) t += sm[i][w];
        v[i] = t;
    }
    __syncthreads();
}

// critic_loss = mse(Q1,y) + mse(Q2,y) + alpha * (mean_b lse1 + mean_b lse2 - mean(Q1+Q2))      (cql.py:162-223)
// d/dq[net][row]: alpha * softmax_row / Bg on every row; data rows add 2 (q - y)/Bg - alpha/Bg.
__global__ __launch_bounds__(1024) void cql_critic_dq_kernel(const float* __restrict__ q_all, const float* __restrict__ tq,
                                                             const float* __restrict__ reward, const float* __restrict__ discount,
                                                             float* __restrict__ dq_all, float* __restrict__ metrics, int B, int n,
                                                             float cql_alpha_in, float inv_bg, CqlScalars* lag,
                                                             const AdamConst* __restrict__ cp, float target_penalty, int mode, float* gsum) {
#pragma clang fp contract(off)
    __shared__ float sm[7][16];
    __shared__ float sh_alpha;
    const int K = 3 * n;                     // re-evaluated rows per sample; the data row is the (K+1)-th entry
    const int64_t R = (int64_t)(K + 1) * B;
    // pass 1: the batch scalars (the penalty decides the Lagrange multiplier before any gradient can be formed)
    float v[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};     // r, y, q1, q2, mse, lse, (q1+q2)
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const float r = reward[b];
        const float y = r + discount[b] * fminf(tq[b], tq[B + b]);
        v[0] += r; v[1] += y;
        for (int net = 0; net < 2; ++net) {
            const float* q = q_all + net * R;
            const float qd = q[(int64_t)K * B + b];
            float mx = qd;
            for (int k = 0; k < K; ++k) mx = fmaxf(mx, q[(int64_t)k * B + b]);
            float se = expf(qd - mx);
            for (int k = 0; k < K; ++k) se += expf(q[(int64_t)k * B + b] - mx);
            const float e = qd - y;
            v[2 + net] += qd;
            v[4] += e * e;
            v[5] += logf(se) + mx;
            v[6] += qd;
        }
    }
    block_sum_c<7>(v, sm);
    // Data parallel with the Lagrange multiplier (mode 1, then 2; gsum = two floats of the agent's statistics block): the multiplier's step
    // needs the penalty of the GLOBAL batch before any gradient can be formed, so a first launch only leaves this rank's two sums
    // (sum lse, sum Q1 + Q2), the host sum-all-reduces them, and the second launch steps the multiplier from the global penalty — the same
    // number on every rank, so the replicas' multipliers stay bit-identical — while its metrics stay this rank's partial means (they are
    // all-reduced with the others).
    if (mode == 1) {
        if (threadIdx.x == 0) { gsum[0] = v[5]; gsum[1] = v[6]; }
        return;
    }
    if (threadIdx.x == 0) {
        const float lse = v[5] * inv_bg, pen = lse - v[6] * inv_bg;
        float alpha = cql_alpha_in;
        if (lag) {                           // alpha_loss = -0.5 * clamp(exp(log_alpha), 0, 1e6) * (penalty - target); Adam; re-read
            const AdamConst c = *cp;
            const float ea = expf(lag->log_alpha);
            const float gpen = mode == 2 ? gsum[0] * inv_bg - gsum[1] * inv_bg : pen;
            const float g = (ea >= 0.0f && ea <= 1000000.0f) ? -0.5f * (gpen - target_penalty) * ea : 0.0f;
            float m = lag->m, vv = lag->v, p = lag->log_alpha;
            m = m + c.one_minus_b1 * (g - m);
            vv = vv * c.b2 + (c.one_minus_b2 * g) * g;
            const float denom = sqrtf(vv) / c.bc2_sqrt + c.eps;
            p = p + (c.neg_step_size * m) / denom;
            lag->m = m; lag->v = vv; lag->log_alpha = p;
            alpha = fminf(fmaxf(expf(p), 0.0f), 1000000.0f);
            lag->alpha = alpha;
        }
        sh_alpha = alpha;
        metrics[EXORL_M_BATCH_REWARD] = v[0] * inv_bg;
        metrics[EXORL_M_CRITIC_TARGET_Q] = v[1] * inv_bg;
        metrics[EXORL_M_CRITIC_Q1] = v[2] * inv_bg;
        metrics[EXORL_M_CRITIC_Q2] = v[3] * inv_bg;
        metrics[EXORL_M_CRITIC_CQL_LOGSUM] = lse;
        metrics[EXORL_M_CRITIC_CQL] = pen;
        metrics[EXORL_M_CRITIC_LOSS] = v[4] * inv_bg + alpha * pen;
    }
    __syncthreads();
    const float cql_alpha = sh_alpha;
    // pass 2: per-row gradients
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const float y = reward[b] + discount[b] * fminf(tq[b], tq[B + b]);
        for (int net = 0; net < 2; ++net) {
            const float* q = q_all + net * R;
            float* dq = dq_all + net * R;
            const float qd = q[(int64_t)K * B + b];
            float mx = qd;
            for (int k = 0; k < K; ++k) mx = fmaxf(mx, q[(int64_t)k * B + b]);
            float se = expf(qd - mx);
            for (int k = 0; k < K; ++k) se += expf(q[(int64_t)k * B + b] - mx);
            const float inv = 1.0f / se;
            for (int k = 0; k < K; ++k) dq[(int64_t)k * B + b] = cql_alpha * expf(q[(int64_t)k * B + b] - mx) * inv * inv_bg;
            const float e = qd - y;
            dq[(int64_t)K * B + b] = cql_alpha * expf(qd - mx) * inv * inv_bg + 2.0f * e * inv_bg - cql_alpha * inv_bg;
        }
    }
}

int cql_critic_dq(const float* q_all, const float* tq, const float* reward, const float* discount, float* dq_all, float* metrics,
                  int B, int n, float cql_alpha, float inv_bg, hipStream_t s, CqlScalars* lag, const AdamConst* c_dev, float target_penalty,
                  int mode, float* gsum) {
    EXORL_REQUIRE(mode == 0 || gsum, "cql_critic_dq: the data-parallel modes need the statistics block");
    hipLaunchKernelGGL(cql_critic_dq_kernel, dim3(1), dim3(1024), 0, s, q_all, tq, reward, discount, dq_all, metrics, B, n, cql_alpha, inv_bg,
                       lag, c_dev, target_penalty, mode, gsum);
    EXORL_LAUNCH_CHECK();
    return 0;
}

// log pi of the tanh-Gaussian at y = tanh(x), x = mu + std z (utils.py:176-179: stable log|det J| = 2 (log 2 - x - softplus(-2x)))
__device__ __forceinline__ float squashed_log_prob(float x, float z, float stdv) {
    const float softplus = fmaxf(-2.0f * x, 0.f) + log1pf(expf(-fabsf(2.0f * x)));
    return -0.5f * z * z - logf(stdv) - 0.9189385332046727f - 2.0f * (0.6931471805599453f - x - softplus);
}

__global__ __launch_bounds__(1024) void cql_actor_sample_kernel(const float* __restrict__ raw_obs, CqlNoise nz, float* __restrict__ xc_pi,
                                                                int64_t ld, float* __restrict__ stats, int B, int O, int A) {
    __shared__ float sm[1][16];
    float v[1] = {0.f};
    for (int i = threadIdx.x; i < B * A; i += blockDim.x) {
        const int b = i / A, j = i % A;
        float mu, sd;
        policy_of(raw_obs + (int64_t)b * 2 * A, A, j, mu, sd);
        const float z = draw(nz.z_actor, nz, 4, i);
        const float x = mu + sd * z;
        xc_pi[(int64_t)b * ld + O + j] = tanhf(x);
        v[0] += squashed_log_prob(x, z, sd);
    }
    block_sum_c<1>(v, sm);
    if (threadIdx.x == 0) { stats[0] = v[0]; stats[1] = 0.f; stats[2] = 0.f; stats[3] = 0.f; }
}

int cql_actor_sample(const float* raw_obs, CqlNoise nz, float* xc_pi, int64_t ld, float* stats, int B, int O, int A, hipStream_t s) {
    hipLaunchKernelGGL(cql_actor_sample_kernel, dim3(1), dim3(1024), 0, s, raw_obs, nz, xc_pi, ld, stats, B, O, A);
    EXORL_LAUNCH_CHECK();
    return 0;
}

// alpha_loss = -(log_alpha * mean(log_pi + target_entropy)); Adam step on the scalar; alpha = exp(log_alpha) afterwards.
// Also the actor metrics that need the batch Q: actor_loss = alpha * mean(log_pi) - mean(min Q).
__global__ __launch_bounds__(1024) void cql_alpha_step_kernel(CqlScalars* sc, const float* __restrict__ stats,
                                                              const AdamConst* __restrict__ cp, float* __restrict__ metrics, int B, int A,
                                                              float inv_bg, const float* __restrict__ q) {
#pragma clang fp contract(off)
    __shared__ float sm[1][16];
    float v[1] = {0.f};
    for (int b = threadIdx.x; b < B; b += blockDim.x) v[0] += fminf(q[b], q[B + b]);
    block_sum_c<1>(v, sm);
    if (threadIdx.x != 0) return;
    const AdamConst c = *cp;
    const float mean_lp = stats[0] * inv_bg / (float)A;
    const float target_entropy = -(float)A;
    const float g = -(mean_lp + target_entropy);
    const float alpha_loss = sc->log_alpha * g;
    float m = sc->m, vv = sc->v, p = sc->log_alpha;
    m = m + c.one_minus_b1 * (g - m);
    vv = vv * c.b2 + (c.one_minus_b2 * g) * g;
    const float denom = sqrtf(vv) / c.bc2_sqrt + c.eps;
    p = p + (c.neg_step_size * m) / denom;
    sc->m = m; sc->v = vv; sc->log_alpha = p;
    const float alpha = expf(p);
    sc->alpha = alpha;
    // metrics are per-rank partial contributions: summed over data-parallel ranks they give the global value
    const float wsz = 1.0f / (inv_bg * (float)B);
    metrics[EXORL_M_ACTOR_ALPHA] = alpha / wsz;
    metrics[EXORL_M_ACTOR_ALPHA_LOSS] = alpha_loss / wsz;
    metrics[EXORL_M_ACTOR_ENT] = -mean_lp / wsz;
    metrics[EXORL_M_ACTOR_LOSS] = alpha * mean_lp / wsz - v[0] * inv_bg;
}

int cql_alpha_step(CqlScalars* sc, const float* stats, const AdamConst* c_dev, float* metrics, int B, int A, float inv_bg, const float* q,
                   hipStream_t s) {
    hipLaunchKernelGGL(cql_alpha_step_kernel, dim3(1), dim3(1024), 0, s, sc, stats, c_dev, metrics, B, A, inv_bg, q);
    EXORL_LAUNCH_CHECK();
    return 0;
}

__global__ void cql_act_kernel(const float* __restrict__ raw, const float* __restrict__ noise, uint64_t seed, uint64_t counter,
                               int eval_mode, float* __restrict__ out, int rows, int A) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rows * A; i += gridDim.x * blockDim.x) {
        const int b = i / A, j = i % A;
        float mu, sd;
        policy_of(raw + (int64_t)b * 2 * A, A, j, mu, sd);
        const float z = eval_mode ? 0.f : (noise ? noise[i] : philox_normal_k(seed, counter, (uint32_t)i));
        out[i] = tanhf(mu + sd * z);
    }
}

int cql_act(const float* raw, const float* noise, uint64_t seed, uint64_t counter, int eval_mode, float* out, int rows, int A, hipStream_t s) {
    hipLaunchKernelGGL(cql_act_kernel, dim3(cdiv(rows * A, 256)), dim3(256), 0, s, raw, noise, seed, counter, eval_mode, out, rows, A);
    EXORL_LAUNCH_CHECK();
    return 0;
}

}  // namespace exorl
