// Loss epilogues, TD target, TruncatedNormal sampling and input staging for the DDPG-family update
// (SURVEY K5-K8):
//   TD target + 2x MSE                 td3_bc.py:122-131, td3.py:120-129, ddpg.py:243-252
//   TD3+BC / TD3 / DDPG actor losses    td3_bc.py:151-155, td3.py:149-152, ddpg.py:274-280
//   BC negative log-likelihood          bc.py:83-84
//   TruncatedNormal.sample              utils/utils.py:140-149 (straight-through clamp, :135-138)
// (B,1)/(B,A)-sized work: one 1024-thread workgroup each, deterministic block reductions. Every "mean" is
// taken over batch*world_size (inv_bg) so that a sum all-reduce over data-parallel ranks reproduces the
// single-process large-batch update.
#include "kernels.h"

namespace exorl {

template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float (*sm)[16]) {
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

// xa = [next_obs ; obs] (2B x O); xc_cur = [obs | action]; xc_next[:, :O] = next_obs; xc_pi[:, :O] = obs
__global__ void prepare_inputs_kernel(const float* __restrict__ obs, const float* __restrict__ action,
                                      const float* __restrict__ next_obs, float* __restrict__ xa,
                                      float* __restrict__ xc_cur, float* __restrict__ xc_next,
                                      float* __restrict__ xc_pi, int B, int O, int A, int has_critic, StepState* st,
                                      int advance_replay) {
    if (st && blockIdx.x == 0 && threadIdx.x == 0) step_begin_device(st, advance_replay);   // counters += ; Adam scalars
    const int W = O + A;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * W; i += gridDim.x * blockDim.x) {
        const int m = i / W, c = i % W;
        if (c < O) {
            const float o = obs[m * O + c], no = next_obs[m * O + c];
            xa[m * O + c] = no;
            xa[(B + m) * O + c] = o;
            if (has_critic) {
                xc_cur[m * W + c] = o;
                xc_next[m * W + c] = no;
                xc_pi[m * W + c] = o;
            }
        } else if (has_critic) {
            xc_cur[m * W + c] = action[m * A + (c - O)];
        }
    }
}

int prepare_inputs(const float* obs, const float* action, const float* next_obs, float* xa, float* xc_cur,
                   float* xc_next, float* xc_pi, int B, int O, int A, int has_critic, StepState* st, int advance_replay,
                   hipStream_t s) {
    const int n = B * (O + A);
    hipLaunchKernelGGL(prepare_inputs_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, obs, action, next_obs, xa, xc_cur,
                       xc_next, xc_pi, B, O, A, has_critic, st, advance_replay);
    EXORL_LAUNCH_CHECK();
    return 0;
}

__device__ __forceinline__ float philox_normal(uint64_t seed, uint64_t counter, uint32_t elem) {
    uint32_t c[4] = {elem, 0u, (uint32_t)counter, (uint32_t)(counter >> 32)};
    Philox::gen(c, seed);
    const float u1 = ((float)c[0] + 1.0f) * 2.3283064365386963e-10f;     // (0, 1]
    const float u2 = (float)c[1] * 2.3283064365386963e-10f;
    return sqrtf(-2.0f * __logf(u1)) * __cosf(6.283185307179586f * u2);
}

__global__ __launch_bounds__(256) void reduce_pairs_kernel(const float* __restrict__ parts, int chunks, float* __restrict__ out) {
    __shared__ float sm[2][16];
    float v[2] = {0.f, 0.f};
    for (int i = threadIdx.x; i < chunks; i += blockDim.x) { v[0] += parts[2 * i]; v[1] += parts[2 * i + 1]; }
    block_sum<2>(v, sm);
    if (threadIdx.x == 0) { out[0] = v[0]; out[1] = v[1]; }
}

int reduce_pairs(const float* parts, int chunks, float* out, hipStream_t s) {
    hipLaunchKernelGGL(reduce_pairs_kernel, dim3(1), dim3(256), 0, s, parts, chunks, out);
    EXORL_LAUNCH_CHECK();
    return 0;
}

__global__ void debug_philox_normal_kernel(uint64_t seed, uint64_t counter, int64_t n, float* __restrict__ out) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
        out[e] = philox_normal(seed, counter, (uint32_t)e);
}

// dst[m][j] = clamp(mu + clamp(noise*std, +-clip), +-(1-1e-6));  optional sum of log N(dst; mu, std)
__global__ __launch_bounds__(1024) void sample_action_kernel(const float* __restrict__ mu, NoiseSpec noise, float stddev_val,
                                                             float clip, int use_clip, float* __restrict__ dst,
                                                             int64_t dst_ld, int B, int A, float* logprob_out,
                                                             float logprob_scale, const float* stddev_ptr) {
    const float stddev = stddev_ptr ? *stddev_ptr : stddev_val;
    __shared__ float sm[1][16];
    float lp[1] = {0.f};
    const float log_norm = -__logf(stddev) - 0.9189385332046727f;   // -log(std) - log(sqrt(2 pi))
    for (int i = threadIdx.x; i < B * A; i += blockDim.x) {
        const int m = i / A, j = i % A;
        const float z = noise.buf ? noise.buf[i]
                                  : philox_normal(noise.seed, noise.counter + (noise.counter_ptr ? *noise.counter_ptr : 0ull), (uint32_t)i);
        float eps = z * stddev;
        if (use_clip) eps = fminf(fmaxf(eps, -clip), clip);
        const float mv = mu[i];
        float x = mv + eps;
        x = fminf(fmaxf(x, -1.0f + 1e-6f), 1.0f - 1e-6f);
        dst[(int64_t)m * dst_ld + j] = x;
        const float d = x - mv;
        lp[0] += -(d * d) / (2.0f * stddev * stddev) + log_norm;
    }
    if (logprob_out) {
        block_sum<1>(lp, sm);
        if (threadIdx.x == 0) *logprob_out = lp[0] * logprob_scale;
    }
}

__global__ __launch_bounds__(256) void sample_actions2_kernel(const float* __restrict__ mu2, const float* __restrict__ noise_c,
                                                              const float* __restrict__ noise_a, uint64_t seed,
                                                              const uint64_t* __restrict__ counter_ptr, float stddev_val, float clip,
                                                              float* __restrict__ dst_next, float* __restrict__ dst_pi,
                                                              int64_t dst_ld, int B, int A, const float* stddev_ptr) {
    const float stddev = stddev_ptr ? *stddev_ptr : stddev_val;
    const int n = B * A;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 2 * n; i += gridDim.x * blockDim.x) {
        const int half = i >= n, e = half ? i - n : i;
        const float* nb = half ? noise_a : noise_c;
        const float z = nb ? nb[e] : philox_normal(seed, (uint64_t)half + (counter_ptr ? *counter_ptr : 0ull), (uint32_t)e);
        const float eps = fminf(fmaxf(z * stddev, -clip), clip);
        const float x = fminf(fmaxf(mu2[i] + eps, -1.0f + 1e-6f), 1.0f - 1e-6f);
        (half ? dst_pi : dst_next)[(int64_t)(e / A) * dst_ld + e % A] = x;
    }
}

int sample_actions2(const float* mu2, const float* noise_c, const float* noise_a, uint64_t seed, const uint64_t* counter_ptr,
                    float stddev, float clip, float* dst_next, float* dst_pi, int64_t dst_ld, int B, int A, hipStream_t s,
                    const float* stddev_ptr) {
    hipLaunchKernelGGL(sample_actions2_kernel, dim3(cdiv(2 * B * A, 256)), dim3(256), 0, s, mu2, noise_c, noise_a, seed, counter_ptr,
                       stddev, clip, dst_next, dst_pi, dst_ld, B, A, stddev_ptr);
    EXORL_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void repeat_sample_kernel(const float* __restrict__ obs, const float* __restrict__ mu,
                                                            const float* __restrict__ noise, uint64_t seed,
                                                            const uint64_t* __restrict__ counter_ptr, uint64_t counter,
                                                            float stddev_val, float clip, float* __restrict__ xc, int B, int O,
                                                            int A, int n, const float* stddev_ptr) {
    const float stddev = stddev_ptr ? *stddev_ptr : stddev_val;
    const int W = O + A;
    const int64_t total = (int64_t)B * n * W;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / W), c = (int)(i % W);
        const int b = row / n;                              // einops 'b x -> (b n) x': row = b*n + sample
        if (c < O) {
            xc[i] = obs[b * O + c];
        } else {
            const int j = c - O, e = row * A + j;
            const float z = noise ? noise[e] : philox_normal(seed, counter + (counter_ptr ? *counter_ptr : 0ull), (uint32_t)e);
            const float eps = fminf(fmaxf(z * stddev, -clip), clip);
            xc[i] = fminf(fmaxf(mu[b * A + j] + eps, -1.0f + 1e-6f), 1.0f - 1e-6f);
        }
    }
}

int repeat_sample(const float* obs, const float* mu, const float* noise, uint64_t seed, const uint64_t* counter_ptr, uint64_t counter,
                  float stddev, float clip, float* xc_rep, int B, int O, int A, int n, hipStream_t s, const float* stddev_ptr) {
    const int64_t total = (int64_t)B * n * (O + A);
    int blocks = cdiv(total, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(repeat_sample_kernel, dim3(blocks), dim3(256), 0, s, obs, mu, noise, seed, counter_ptr, counter, stddev, clip,
                       xc_rep, B, O, A, n, stddev_ptr);
    EXORL_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void crr_weights_kernel(const float* __restrict__ q_rep, const float* __restrict__ q_data,
                                                          float* __restrict__ w, int B, int n, int weight_func) {
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) {
        float vsum = 0.f;
        for (int i = 0; i < n; ++i) vsum += fminf(q_rep[b * n + i], q_rep[(int64_t)B * n + b * n + i]);
        const float adv = fminf(q_data[b], q_data[B + b]) - vsum / (float)n;
        float wv;
        if (weight_func == EXORL_CRR_IDENTITY) wv = adv;
        else if (weight_func == EXORL_CRR_INDICATOR) wv = adv > 0.f ? 1.0f : 0.0f;       // sign(relu(A))
        else wv = fminf(fmaxf(expf(adv), 0.0f), 20.0f);
        w[b] = wv;
    }
}

int crr_weights(const float* q_rep, const float* q_data, float* w, int B, int n, int weight_func, hipStream_t s) {
    hipLaunchKernelGGL(crr_weights_kernel, dim3(cdiv(B, 256)), dim3(256), 0, s, q_rep, q_data, w, B, n, weight_func);
    EXORL_LAUNCH_CHECK();
    return 0;
}

int sample_action(const float* mu, NoiseSpec noise, float stddev, float clip, int use_clip, float* dst, int64_t dst_ld,
                  int B, int A, float* logprob_sum, hipStream_t s, const float* stddev_ptr, int world_size) {
    // the log-prob metric is a partial mean over the GLOBAL batch, like every other metric (agents._metrics sum-all-reduces them)
    hipLaunchKernelGGL(sample_action_kernel, dim3(1), dim3(1024), 0, s, mu, noise, stddev, clip, use_clip, dst, dst_ld,
                       B, A, logprob_sum, 1.0f / ((float)B * (float)world_size), stddev_ptr);
    EXORL_LAUNCH_CHECK();
    return 0;
}

// y = r + D*min(tq1,tq2); dq_i = 2 (q_i - y) * inv_bg; metrics (partial means over the global batch)
__global__ __launch_bounds__(1024) void critic_loss_kernel(const float* __restrict__ q, const float* __restrict__ tq,
                                                           const float* __restrict__ reward,
                                                           const float* __restrict__ discount, float* __restrict__ dq,
                                                           float* __restrict__ metrics, int B, float inv_bg) {
    __shared__ float sm[5][16];
    float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int m = threadIdx.x; m < B; m += blockDim.x) {
        const float r = reward[m];
        const float y = r + discount[m] * fminf(tq[m], tq[B + m]);
        const float q1 = q[m], q2 = q[B + m];
        const float e1 = q1 - y, e2 = q2 - y;
        dq[m] = 2.0f * e1 * inv_bg;
        dq[B + m] = 2.0f * e2 * inv_bg;
        v[0] += r; v[1] += y; v[2] += q1; v[3] += q2; v[4] += e1 * e1 + e2 * e2;
    }
    block_sum<5>(v, sm);
    if (threadIdx.x == 0) {
        metrics[EXORL_M_BATCH_REWARD] = v[0] * inv_bg;
        metrics[EXORL_M_CRITIC_TARGET_Q] = v[1] * inv_bg;
        metrics[EXORL_M_CRITIC_Q1] = v[2] * inv_bg;
        metrics[EXORL_M_CRITIC_Q2] = v[3] * inv_bg;
        metrics[EXORL_M_CRITIC_LOSS] = v[4] * inv_bg;
    }
}

int critic_loss(const float* q, const float* tq, const float* reward, const float* discount, float* dq,
                float* metrics, int B, float inv_bg, hipStream_t s) {
    hipLaunchKernelGGL(critic_loss_kernel, dim3(1), dim3(1024), 0, s, q, tq, reward, discount, dq, metrics, B, inv_bg);
    EXORL_LAUNCH_CHECK();
    return 0;
}

// stats[0] = sum |min(q1,q2)|, stats[1] = sum min(q1,q2)   (local sums; all-reduced by the caller under DP)
__global__ __launch_bounds__(1024) void actor_stats_kernel(const float* __restrict__ q, float* __restrict__ stats,
                                                           float* __restrict__ metrics, int B) {
    __shared__ float sm[2][16];
    float v[2] = {0.f, 0.f};
    for (int m = threadIdx.x; m < B; m += blockDim.x) {
        const float qq = fminf(q[m], q[B + m]);
        v[0] += fabsf(qq);
        v[1] += qq;
    }
    block_sum<2>(v, sm);
    if (threadIdx.x == 0) {
        stats[0] = v[0]; stats[1] = v[1]; stats[2] = 0.f; stats[3] = 0.f;
        metrics[EXORL_M_Q_ABS_SUM] = v[0];
        metrics[EXORL_M_Q_SUM] = v[1];
    }
}

__global__ __launch_bounds__(256) void sf_q_kernel(const float* __restrict__ feat, const float* __restrict__ task, int64_t task_ld,
                                                   float* __restrict__ q, int B, int sf, int nets) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nets * B; i += gridDim.x * blockDim.x) {
        const int m = i % B;
        float acc = 0.f;
        for (int j = 0; j < sf; ++j) acc += task[(int64_t)m * task_ld + j] * feat[(int64_t)i * sf + j];
        q[i] = acc;
    }
}

int sf_q(const float* feat, const float* task, int64_t task_ld, float* q, int B, int sf, int nets, hipStream_t s) {
    hipLaunchKernelGGL(sf_q_kernel, dim3(cdiv(nets * B, 256)), dim3(256), 0, s, feat, task, task_ld, q, B, sf, nets);
    EXORL_LAUNCH_CHECK();
    return 0;
}

int actor_stats(const float* q, float* stats, int B, hipStream_t s) {
    // metrics block directly follows the 4-float stats block (agent.hip lays them out that way)
    hipLaunchKernelGGL(actor_stats_kernel, dim3(1), dim3(1024), 0, s, q, stats, stats + 4, B);
    EXORL_LAUNCH_CHECK();
    return 0;
}

// dq_i[m] = -lambda * inv_bg * w_i[m]; torch.min routes the gradient to the smaller Q, 0.5/0.5 on ties
__global__ void actor_dq_kernel(const float* __restrict__ q, const float* __restrict__ stats, float* __restrict__ dq,
                                int B, float inv_bg, float alpha, int use_lambda) {
    const float lambda = use_lambda ? alpha / (stats[0] * inv_bg) : 1.0f;
    const float g = -lambda * inv_bg;
    for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < B; m += gridDim.x * blockDim.x) {
        const float q1 = q[m], q2 = q[B + m];
        const float w1 = q1 < q2 ? 1.0f : (q1 == q2 ? 0.5f : 0.0f);
        dq[m] = g * w1;
        dq[B + m] = g * (1.0f - w1);
    }
}

int actor_dq(const float* q, const float* stats, float* dq, int B, float inv_bg, float alpha, int use_lambda,
             hipStream_t s) {
    hipLaunchKernelGGL(actor_dq_kernel, dim3(cdiv(B, 256)), dim3(256), 0, s, q, stats, dq, B, inv_bg, alpha, use_lambda);
    EXORL_LAUNCH_CHECK();
    return 0;
}

// Gradient at the actor's pre-tanh output.
//  TD3+BC: dmu = da + 2 (mu - a)/(Bg*A);  TD3/DDPG: dmu = da;  BC: dmu = -(a - mu)/std^2 / Bg
//  dpre = dmu * (1 - mu^2).  Also writes the actor_loss metric (partial over the global batch).
__global__ __launch_bounds__(1024) void actor_dmu_kernel(const float* __restrict__ da, int64_t da_ld, int da_nets,
                                                         int64_t da_net_stride,
                                                         const float* __restrict__ mu, const float* __restrict__ a_data,
                                                         const float* __restrict__ reward, const float* __restrict__ w,
                                                         float* __restrict__ dpre, const float* __restrict__ stats,
                                                         float* __restrict__ metrics, int B, int A, float inv_bg,
                                                         float alpha, int kind, float stddev_val, const float* stddev_ptr) {
    const float stddev = stddev_ptr ? *stddev_ptr : stddev_val;
    __shared__ float sm[2][16];
    float v[2] = {0.f, 0.f};
    if (reward)
        for (int m = threadIdx.x; m < B; m += blockDim.x) v[1] += reward[m];
    const float bc_coef = 2.0f * inv_bg / (float)A;
    const float inv_var = 1.0f / (stddev * stddev);
    const float log_norm = -__logf(stddev) - 0.9189385332046727f;
    for (int i = threadIdx.x; i < B * A; i += blockDim.x) {
        const int m = i / A, j = i % A;
        const float mv = mu[i];
        float dmu;
        if (kind == EXORL_AGENT_TD3_BC) {
            const float d = mv - a_data[i];
            float dav = 0.f;
            for (int t = 0; t < da_nets; ++t) dav += da[t * da_net_stride + (int64_t)m * da_ld + j];
            dmu = dav + bc_coef * d;
            v[0] += d * d;
        } else if (kind == EXORL_AGENT_BC) {
            const float d = a_data[i] - mv;
            dmu = -d * inv_var * inv_bg;
            v[0] += d * d * 0.5f * inv_var - log_norm;     // -log N(a; mu, std)
        } else if (kind == EXORL_AGENT_CRR) {
            const float d = a_data[i] - mv;
            dmu = -w[m] * d * inv_var * inv_bg;
            v[0] += w[m] * (d * d * 0.5f * inv_var - log_norm);   // -(log_prob * w), crr.py:186
        } else {
            dmu = 0.f;
            for (int t = 0; t < da_nets; ++t) dmu += da[t * da_net_stride + (int64_t)m * da_ld + j];
        }
        dpre[i] = dmu * (1.0f - mv * mv);
    }
    block_sum<2>(v, sm);
    if (threadIdx.x == 0) {
        if (reward) metrics[EXORL_M_BATCH_REWARD] = v[1] * inv_bg;
        float loss;
        if (kind == EXORL_AGENT_TD3_BC) {
            const float lambda = alpha / (stats[0] * inv_bg);
            loss = -lambda * metrics[EXORL_M_Q_SUM] * inv_bg + v[0] * inv_bg / (float)A;
            metrics[EXORL_M_BC_SUM] = v[0];
        } else if (kind == EXORL_AGENT_BC || kind == EXORL_AGENT_CRR) {
            loss = v[0] * inv_bg;
        } else {
            loss = -metrics[EXORL_M_Q_SUM] * inv_bg;
        }
        metrics[EXORL_M_ACTOR_LOSS] = loss;
    }
}

int actor_dmu(const float* da, int64_t da_ld, int da_nets, int64_t da_net_stride, const float* mu, const float* a_data, const float* reward, const float* w, float* dpre, float* stats,
              float* metrics, int B, int A, float inv_bg, float alpha, int kind, float stddev, hipStream_t s, const float* stddev_ptr) {
    hipLaunchKernelGGL(actor_dmu_kernel, dim3(1), dim3(1024), 0, s, da, da_ld, da_nets, da_net_stride, mu, a_data, reward, w, dpre, stats, metrics, B, A,
                       inv_bg, alpha, kind, stddev, stddev_ptr);
    EXORL_LAUNCH_CHECK();
    return 0;
}

}  // namespace exorl

extern "C" int exorl_debug_philox_normal(uint64_t seed, uint64_t counter, int64_t n, float* out_dev, void* stream) {
    using namespace exorl;
    EXORL_REQUIRE(out_dev && n > 0 && n <= (int64_t)1 << 31, "debug_philox_normal: bad arguments");
    hipLaunchKernelGGL(debug_philox_normal_kernel, dim3(cdiv(n, 256) > 4096 ? 4096 : cdiv(n, 256)), dim3(256), 0, as_stream(stream), seed, counter, n, out_dev);
    EXORL_LAUNCH_CHECK();
    return 0;
}
