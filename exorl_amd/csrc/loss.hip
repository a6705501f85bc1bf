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

// ---- act() in ONE launch (td3_bc.py:107-117, ddpg.py:221-238): the online loop calls the policy once per environment step on ONE
// observation, so the cost is latency, not FLOPs. Every workgroup recomputes the trunk for the (<= ACT_FAST_ROWS) rows — Linear(in, H) through
// the transposed fp32 shadow (coalesced), LayerNorm, tanh: 24 k MACs and 100 KB of L2 reads for the walker actor — takes FOUR neurons of the
// Linear(H, H) + ReLU layer (one wave each: 4 KB of W1 per wave; the 4 MB of fp32 master weights are spread over H / 4 workgroups, no bf16
// shadow and no GEMM needed at one row), forms its share of the head dots and leaves it in `part`. The last workgroup to arrive (one atomic
// ticket per workgroup) sums the shares in workgroup order — deterministic — applies bias + tanh and the TruncatedNormal draw (clip = None)
// and stores the action rows to `out`, which may be pinned host memory: no separate head, sampling or copy launch, no host round trip
// before the result. fp32 FMA throughout (the parity mode of every precision setting: act() is never the bottleneck on FLOPs).
struct ActFastArgs {
    const float* x; int64_t ldx;            // observation rows (device) — or, when x == nullptr, the rows embedded below (kernel arguments)
    const float *w0t, *P;                   // W0T shadow [in][H]; flat fp32 parameters
    int64_t b0, g, beta, W1, b1, W2, b2;    // offsets in P
    float* part;                            // [H / 4][rows][nout] head shares
    unsigned int* ticket;
    const float* noise;                     // (rows, nout) standard normals (device), or null -> Philox(seed, counter) unless kn != 0
    uint64_t seed, counter;
    float* out;                             // (rows, nout)
    float stddev;
    int rows, in_dim, H, nout, eval_mode, kn;
    float kx[ACT_FAST_ROWS * 256];          // embedded observation rows (x == nullptr)
    float knoise[ACT_FAST_ROWS * 16];       // embedded noise rows (kn != 0)
};

template <int FIRST>
__global__ __launch_bounds__(256) void act_fast_kernel(const ActFastArgs a) {
    extern __shared__ float act_sm[];                   // h1[rows][H], then scratch
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, H = a.H, rows = a.rows, I = a.in_dim, A = a.nout;
    float* h1 = act_sm;
    __shared__ float red[ACT_FAST_ROWS][2][4];
    __shared__ float h2s[ACT_FAST_ROWS][4];
    __shared__ unsigned int last;
    // ---- trunk: z = W0 x + b0 (every workgroup, all H columns)
    for (int r = 0; r < rows; ++r) {
        const float* x = a.x ? a.x + (int64_t)r * a.ldx : a.kx + r * 256;
        float s1 = 0.f;
        for (int n = tid; n < H; n += 256) {
            float z = a.P[a.b0 + n];
            if constexpr (FIRST == 0) {
                for (int i = 0; i < I; ++i) z = fmaf(x[i], a.w0t[(int64_t)i * H + n], z);
            } else {                                    // pixel policy: Linear(feature_dim, H) + ReLU on the trunk's output, torch layout W[n][i]
                const float* wrow = a.P + a.g + (int64_t)n * I;
                for (int i = 0; i < I; ++i) z = fmaf(x[i], wrow[i], z);
                z = fmaxf(z, 0.f);
            }
            h1[r * H + n] = z;
            s1 += z;
        }
        s1 = wave_sum(s1);
        if (lane == 0) red[r][0][wave] = s1;
    }
    __syncthreads();
    if constexpr (FIRST == 0) {
    for (int r = 0; r < rows; ++r) {                    // LayerNorm (biased variance about the mean, eps 1e-5) + tanh
        const float mean = (red[r][0][0] + red[r][0][1] + red[r][0][2] + red[r][0][3]) / (float)H;
        float s2 = 0.f;
        for (int n = tid; n < H; n += 256) { const float d = h1[r * H + n] - mean; s2 += d * d; }
        s2 = wave_sum(s2);
        if (lane == 0) red[r][1][wave] = s2;
    }
    __syncthreads();
    for (int r = 0; r < rows; ++r) {
        const float mean = (red[r][0][0] + red[r][0][1] + red[r][0][2] + red[r][0][3]) / (float)H;
        const float var = (red[r][1][0] + red[r][1][1] + red[r][1][2] + red[r][1][3]) / (float)H;
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        for (int n = tid; n < H; n += 256) h1[r * H + n] = tanhf((h1[r * H + n] - mean) * rstd * a.P[a.g + n] + a.P[a.beta + n]);
    }
    __syncthreads();
    }
    // ---- Linear(H, H) + ReLU: wave w -> neuron 4 blockIdx.x + w, all rows
    const int nn = 4 * blockIdx.x + wave;
    if (nn < H) {
        const float4* w1 = reinterpret_cast<const float4*>(a.P + a.W1 + (int64_t)nn * H);
        float acc[ACT_FAST_ROWS];
#pragma unroll
        for (int r = 0; r < ACT_FAST_ROWS; ++r) acc[r] = 0.f;
        for (int k4 = lane; k4 < H / 4; k4 += 64) {
            const float4 w = w1[k4];
#pragma unroll
            for (int r = 0; r < ACT_FAST_ROWS; ++r)
                if (r < rows) {
                    const float4 h = *reinterpret_cast<const float4*>(h1 + r * H + 4 * k4);
                    acc[r] = fmaf(w.x, h.x, fmaf(w.y, h.y, fmaf(w.z, h.z, fmaf(w.w, h.w, acc[r]))));
                }
        }
#pragma unroll
        for (int r = 0; r < ACT_FAST_ROWS; ++r)
            if (r < rows) {
                const float v = wave_sum(acc[r]);
                if (lane == 0) h2s[r][wave] = fmaxf(v + a.P[a.b1 + nn], 0.f);
            }
    } else if (lane == 0) {
        for (int r = 0; r < rows; ++r) h2s[r][wave] = 0.f;
    }
    __syncthreads();
    // ---- this workgroup's share of the head dots
    if (tid < rows * A) {
        const int r = tid / A, j = tid % A;
        float v = 0.f;
        for (int w = 0; w < 4; ++w) {
            const int n2 = 4 * blockIdx.x + w;
            if (n2 < H) v = fmaf(a.P[a.W2 + (int64_t)j * H + n2], h2s[r][w], v);
        }
        a.part[((int64_t)blockIdx.x * rows + r) * A + j] = v;
    }
    // publish the shares, draw a ticket (cdna_hip_programming.md, in-launch split-K reduction): every wave drains its stores, ONE lane makes
    // the agent-scope release and the relaxed ticket add — a __threadfence() in all 256 threads of all H / 4 workgroups writes the XCD's L2
    // back once per caller, which is what the first version of this kernel spent most of its 30 us on
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        last = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (!last) return;
    // ---- last workgroup: fixed-order sum over the workgroups (thread t holds workgroup t, t + 256, ...; wave tree, then the four waves)
    float* fin = act_sm;                                // h1 is dead
    const int G = gridDim.x;
    for (int e = 0; e < rows * A; ++e) {
        float v = 0.f;
        for (int gidx = tid; gidx < G; gidx += 256) v += a.part[(int64_t)gidx * rows * A + e];
        v = wave_sum(v);
        if (lane == 0) fin[e * 4 + wave] = v;
    }
    __syncthreads();
    if (tid < rows * A) {
        const int j = tid % A;
        const float pre = ((fin[tid * 4] + fin[tid * 4 + 1]) + (fin[tid * 4 + 2] + fin[tid * 4 + 3])) + a.P[a.b2 + j];
        const float mu = tanhf(pre);
        float o = mu;
        if (!a.eval_mode) {                             // TruncatedNormal(mu, std).sample(clip=None): clamp to +-(1 - 1e-6)
            const float z = a.kn ? a.knoise[tid] : (a.noise ? a.noise[tid] : philox_normal(a.seed, a.counter, (uint32_t)tid));
            o = fminf(fmaxf(mu + z * a.stddev, -1.0f + 1e-6f), 1.0f - 1e-6f);
        }
        a.out[tid] = o;
    }
    if (tid == 0) *a.ticket = 0u;                       // armed for the next call (launches on one stream are ordered)
}

bool act_fast_supported(int rows, int in_dim, int H, int nout) {
    return rows >= 1 && rows <= ACT_FAST_ROWS && in_dim <= 256 && nout <= 16 && H % 4 == 0 && H >= 16 && rows * nout <= 256;
}

int act_fast(const ActFast& f, hipStream_t s) {
    ActFastArgs a{};
    a.x = f.x_dev; a.ldx = f.in_dim;
    a.w0t = f.w0t; a.P = f.P; a.b0 = f.b0; a.g = f.g; a.beta = f.beta; a.W1 = f.W1; a.b1 = f.b1; a.W2 = f.W2; a.b2 = f.b2;
    a.part = f.part; a.ticket = f.ticket; a.noise = f.noise_dev; a.seed = f.seed; a.counter = f.counter; a.out = f.out; a.stddev = f.stddev;
    a.rows = f.rows; a.in_dim = f.in_dim; a.H = f.H; a.nout = f.nout; a.eval_mode = f.eval_mode; a.kn = 0;
    if (f.first_relu) a.g = f.W0;                        // row-major first-layer weight rides in the (unused) gain slot
    if (f.x_host) {
        a.x = nullptr;
        for (int r = 0; r < f.rows; ++r) memcpy(a.kx + r * 256, f.x_host + (int64_t)r * f.in_dim, sizeof(float) * f.in_dim);
    }
    if (f.noise_host && !f.eval_mode) { a.kn = 1; memcpy(a.knoise, f.noise_host, sizeof(float) * f.rows * f.nout); }
    const size_t lds = sizeof(float) * ((size_t)f.rows * f.H > 1024 ? (size_t)f.rows * f.H : 1024);
    if (f.first_relu) hipLaunchKernelGGL(act_fast_kernel<1>, dim3(cdiv(f.H, 4)), dim3(256), lds, s, a);
    else hipLaunchKernelGGL(act_fast_kernel<0>, dim3(cdiv(f.H, 4)), dim3(256), lds, s, a);
    EXORL_LAUNCH_CHECK();
    return 0;
}

// ---- pixel act(): the trunk Linear(39200 (+ meta), F) + LayerNorm + tanh of ONE encoding in one launch. Workgroup g takes a slab of the 39200
// columns for all F outputs (the 7.8 MB of trunk weights are read exactly once, spread over the chip), leaves F partial dots; the last
// workgroup to arrive sums them in workgroup order, adds the meta columns and the bias, normalises and writes tanh(LN(z)) (F floats).
struct TrunkOneArgs {
    const float *x, *meta, *W, *b, *gain, *beta;        // x: R floats; W: [F][D] with D = R + M
    float *part, *out;                                  // part: [gridDim.x][F]
    unsigned int* ticket;
    int R, M, F, slab;
};
__global__ __launch_bounds__(256) void trunk_one_kernel(const TrunkOneArgs a) {
    __shared__ float z[1024];
    __shared__ float red[2][4];
    __shared__ unsigned int last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, D = a.R + a.M;
    const int k0 = blockIdx.x * a.slab, k1 = k0 + a.slab < a.R ? k0 + a.slab : a.R;
    for (int f = wave; f < a.F; f += 4) {
        const float* w = a.W + (int64_t)f * D;
        float acc = 0.f;
        for (int k = k0 + lane; k < k1; k += 64) acc = fmaf(a.x[k], w[k], acc);
        acc = wave_sum(acc);
        if (lane == 0) a.part[(int64_t)blockIdx.x * a.F + f] = acc;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        last = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (!last) return;
    const int G = gridDim.x;
    for (int f = wave; f < a.F; f += 4) {               // fixed order: lane l holds workgroups l, l + 64, ...; wave tree
        float v = 0.f;
        for (int g = lane; g < G; g += 64) v += a.part[(int64_t)g * a.F + f];
        v = wave_sum(v);
        if (lane == 0) {
            const float* w = a.W + (int64_t)f * D + a.R;
            for (int j = 0; j < a.M; ++j) v = fmaf(a.meta[j], w[j], v);
            z[f] = v + a.b[f];
        }
    }
    __syncthreads();
    float s1 = 0.f;
    for (int f = tid; f < a.F; f += 256) s1 += z[f];
    s1 = wave_sum(s1);
    if (lane == 0) red[0][wave] = s1;
    __syncthreads();
    const float mean = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / (float)a.F;
    float s2 = 0.f;
    for (int f = tid; f < a.F; f += 256) { const float d = z[f] - mean; s2 += d * d; }
    s2 = wave_sum(s2);
    if (lane == 0) red[1][wave] = s2;
    __syncthreads();
    const float rstd = 1.0f / sqrtf((red[1][0] + red[1][1] + red[1][2] + red[1][3]) / (float)a.F + 1e-5f);
    for (int f = tid; f < a.F; f += 256) a.out[f] = tanhf((z[f] - mean) * rstd * a.gain[f] + a.beta[f]);
    if (tid == 0) *a.ticket = 0u;
}

int trunk_one(const float* x, const float* meta, const float* W, const float* b, const float* gain, const float* beta, float* part, unsigned int* ticket,
              float* out, int R, int M, int F, hipStream_t s) {
    EXORL_REQUIRE(F >= 1 && F <= 1024 && R >= 1 && M >= 0, "trunk_one: unsupported dims");
    const int grid = 256;
    TrunkOneArgs a{x, meta, W, b, gain, beta, part, out, ticket, R, M, F, (int)round_up(cdiv(R, grid), 4)};
    hipLaunchKernelGGL(trunk_one_kernel, dim3(grid), dim3(256), 0, s, a);
    EXORL_LAUNCH_CHECK();
    return 0;
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
