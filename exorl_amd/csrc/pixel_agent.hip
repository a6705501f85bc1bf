// DDPG on pixel observations: augmentation, conv encoder, the pixel Actor / Critic and their update.
// Replaces (file:line in the reference repo, agents/unsupervised_learning/ddpg.py):
//   Actor (obs_type == 'pixels')   :42-76   trunk Linear(repr, feature_dim)+LN+Tanh, policy Linear-ReLU-Linear-ReLU-Linear, tanh
//   Critic (obs_type == 'pixels')  :79-123  trunk on the encoding, action concatenated AFTER the trunk, Q1/Q2 three-layer heads
//   aug_and_encode :213-215, update_critic :240-268 (encoder_opt steps with critic_opt), update_actor :270-292, update :298-328, act :221-238
// Built from the blocks of the other translation units: pixels.hip (RandomShiftsAug, conv encoder forward/backward), the generic
// grouped GEMM + Linear/ReLU stack helpers (intr.hip), LayerNorm/tanh row kernels (rowops.hip), loss/sampling kernels (loss.hip),
// Adam/Polyak (optim.hip). The one shape the generic GEMM handles badly is the trunk's Linear(39200, 50): 16 row tiles of a
// K = 39200 reduction would occupy 16 CUs, so it runs as a 16-way split-K (one grouped launch, 256 workgroups) + a fixed-order reduce.
#include <vector>

#include "kernels.h"

namespace exorl {

struct PTensor { int64_t off, rows, cols; };
struct PNet {                       // trunk Linear(D, F) + LayerNorm(F) + Tanh, then n_heads three-layer heads on F (+A)
    int D = 0, F = 0, H = 0, head_in = 0, out = 0, n_heads = 0;
    Lin trunk{};
    int64_t g = 0, beta = 0;
    Mlp head[2];
    int64_t total = 0;
    std::vector<PTensor> tensors;
};

static PNet make_pnet(int D, int F, int H, int head_in, int out, int n_heads) {
    PNet n;
    n.D = D; n.F = F; n.H = H; n.head_in = head_in; n.out = out; n.n_heads = n_heads;
    int64_t off = 0;
    auto add = [&](int64_t r, int64_t c) { const int64_t o = off; n.tensors.push_back({o, r, c}); off += round_up(r * c, 4); return o; };
    auto lin = [&](int in, int o2) { Lin l{in, o2, 0, 0}; l.W = add(o2, in); l.b = add(o2, 1); return l; };
    n.trunk = lin(D, F);
    n.g = add(F, 1); n.beta = add(F, 1);
    for (int i = 0; i < n_heads; ++i) n.head[i].L = {lin(head_in, H), lin(H, H), lin(H, out)};
    n.total = round_up(off, 64);
    return n;
}

constexpr int SPLITK = 16;

// z[m][f] = bias[f] + sum_s P[s][m][f]
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ P, const float* __restrict__ bias, float* __restrict__ z,
                                                            int64_t n, int F, int splits) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float acc = bias[i % F];
        for (int s = 0; s < splits; ++s) acc += P[(int64_t)s * n + i];
        z[i] = acc;
    }
}
__global__ __launch_bounds__(256) void tanh_kernel(float* __restrict__ x, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] = tanhf(x[i]);
}
// dst[m][j] = a[m*lda + c0 + j] + b[m*ldb + c0 + j]   (b == nullptr: a alone)
__global__ __launch_bounds__(256) void add_cols_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ b, int64_t ldb, int c0,
                                                       int nc, float* __restrict__ dst, int rows) {
    const int64_t n = (int64_t)rows * nc;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / nc;
        const int j = (int)(i - m * nc);
        dst[i] = b ? a[m * lda + c0 + j] + b[m * ldb + c0 + j] : a[m * lda + c0 + j];
    }
}
// straight-through sample (utils.py:135-138) then tanh: dpre = da * (1 - mu^2)
__global__ __launch_bounds__(256) void dpre_kernel(const float* __restrict__ da, const float* __restrict__ mu, float* __restrict__ dpre, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dpre[i] = da[i] * (1.0f - mu[i] * mu[i]);
}

// CriticSF (aps.py:17-60): each head emits sf_dim successor features, Q_n = task . features_n; backward: d/d(features_n) = dQ_n * task
__global__ __launch_bounds__(256) void sf_dot_kernel(const float* __restrict__ f0, const float* __restrict__ f1, const float* __restrict__ task,
                                                     int64_t ldt, float* __restrict__ q, int B, int S) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * B) return;
    const int n = i / B, m = i - n * B;
    const float* f = (n ? f1 : f0) + (int64_t)m * S;
    float acc = 0.f;
    for (int j = 0; j < S; ++j) acc += task[(int64_t)m * ldt + j] * f[j];
    q[i] = acc;
}
__global__ __launch_bounds__(256) void sf_dout_kernel(const float* __restrict__ dq, const float* __restrict__ task, int64_t ldt, float* __restrict__ d0,
                                                      float* __restrict__ d1, int B, int S) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * B * S) return;
    const int n = i / (B * S), r = i - n * B * S, m = r / S, j = r - m * S;
    (n ? d1 : d0)[r] = dq[n * B + m] * task[(int64_t)m * ldt + j];
}

// nn.BatchNorm2d(affine=False) in training mode on (n, c, h, w) images, then clamp(+-clip) (rnd.py:26-27,47-50). Statistics per channel over
// n * h * w values in double (torch's CPU kernel accumulates float inputs in double): partial sums per (channel, chunk), a fixed-order
// finish, the centred second pass the same way, then one elementwise pass. running_var takes the unbiased estimate, momentum 0.1.
constexpr int BN2_CHUNKS = 128;
__global__ __launch_bounds__(256) void bn2d_partial_kernel(const float* __restrict__ x, int n, int c, int64_t hw, const double* __restrict__ mean,
                                                           double* __restrict__ part) {
    __shared__ double red[256];
    const int ch = blockIdx.x, k = blockIdx.y;
    const int64_t per = (int64_t)n * hw, lo = per * k / BN2_CHUNKS, hi = per * (k + 1) / BN2_CHUNKS;
    const double m = mean ? mean[ch] : 0.0;
    double acc = 0.0;
    // (image, pixel) of element i kept incrementally: a 64-bit division per element was most of this kernel's 92 us on 87 MB
    int64_t i = lo + threadIdx.x, img = i / hw, p = i - img * hw;
    const int64_t step_img = (int64_t)blockDim.x / hw, step_p = (int64_t)blockDim.x - step_img * hw;
    for (; i < hi; i += blockDim.x) {
        const double v = (double)x[(img * c + ch) * hw + p] - m;
        acc += mean ? v * v : v;
        img += step_img; p += step_p;
        if (p >= hw) { p -= hw; ++img; }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[ch * BN2_CHUNKS + k] = red[0];
}
// stage 0: mean[ch] = sum(part) / count; stage 1: var, running statistics (stats: mean[c] var[c] count)
__global__ void bn2d_finish_kernel(const double* __restrict__ part, double* __restrict__ mean, double* __restrict__ var, float* __restrict__ running,
                                   int c, double count, int stage) {
    const int ch = threadIdx.x;
    if (ch >= c) return;
    double acc = 0.0;
    for (int k = 0; k < BN2_CHUNKS; ++k) acc += part[ch * BN2_CHUNKS + k];
    if (stage == 0) { mean[ch] = acc / count; return; }
    var[ch] = acc / count;
    running[ch] = 0.9f * running[ch] + 0.1f * (float)mean[ch];
    running[c + ch] = 0.9f * running[c + ch] + 0.1f * (float)(acc / (count > 1.0 ? count - 1.0 : 1.0));
    if (ch == 0) running[2 * c] += 1.0f;
}
__global__ __launch_bounds__(256) void bn2d_apply_kernel(float* __restrict__ x, int c, int64_t hw, int64_t total, const double* __restrict__ mean,
                                                         const double* __restrict__ var, float clip) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)((i / hw) % c);
        const double inv = 1.0 / sqrt(var[ch] + 1e-5);               // torch's CPU kernel: out = x * alpha + beta with alpha = invstd, beta = -mean * invstd
        const float v = x[i] * (float)inv + (float)(-mean[ch] * inv);
        x[i] = fminf(fmaxf(v, -clip), clip);
    }
}

static int grid1(int64_t n) { const int64_t b = (n + 255) / 256; return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }

}  // namespace exorl

using namespace exorl;

struct TrunkAct { float *z, *h, *xhat, *rstd; };

struct exorl_pixel_agent {
    exorl_pixel_cfg cfg;
    int R = 0;                                   // repr_dim
    PNet actor, critic;
    int64_t enc_total = 0;
    float* ws = nullptr;
    // flat[net][what]; nets: 0 encoder, 1 actor, 2 critic, 3 critic_target (params only)
    float* flat[4][4] = {{nullptr}};
    unsigned char *obs = nullptr, *next_obs = nullptr;
    float *action = nullptr, *reward = nullptr, *discount = nullptr, *meta = nullptr;      // meta: (batch, meta_dim) skill / task rows
    float *aug_o = nullptr, *aug_n = nullptr, *enc_ws_o = nullptr, *enc_ws_n = nullptr, *feat_o = nullptr, *feat_n = nullptr;
    float* splitk = nullptr; float* splitk2 = nullptr;     // split-K partials of a trunk forward (two: paired trunks share a launch)
    TrunkAct ta_n{}, ta_o{}, tt{}, tc{};         // actor on next_obs / obs, target critic, critic
    float *xq_t = nullptr, *xq_c = nullptr, *dxq[2] = {nullptr, nullptr};
    float *q = nullptr, *tq = nullptr, *dq = nullptr, *mu_n = nullptr, *mu_o = nullptr, *dmu = nullptr, *dh = nullptr, *dz = nullptr, *dfeat = nullptr;
    float *stats = nullptr, *metrics = nullptr;
    int32_t* shifts = nullptr;
    float* act_ws = nullptr;                     // B = 1 inference scratch
    float* act_part = nullptr; unsigned int* act_ticket = nullptr;     // fused act(): trunk shares, head shares, trunk output; two tickets
    int64_t t = 0, t_enc = 0;                    // Adam step counts: critic_opt / actor_opt, encoder_opt (equal for plain DDPG)
    uint64_t noise_counter = 0, aug_counter = 0, act_counter = 0, rnd_aug_counter = 0;
    // Proto on pixels (proto.py:46-85): encoder_target (Polyak copy) and the encoder's second Adam state (proto_opt's)
    float *enc_target = nullptr, *enc_m2 = nullptr, *enc_v2 = nullptr;
    float* bn2d = nullptr;                       // RND on pixels: BatchNorm2d running_mean[c] running_var[c] num_batches_tracked
    double* bn_scratch = nullptr;                // mean[16] var[16] partials[16 * BN2_CHUNKS]
    int64_t t2 = 0;
    bool augmented = false;
    bool have_feat_o = false, have_feat_n = false;     // feat_o / feat_n hold the online encoder's pass of exorl_pixel_agent_encode
    bool train_encoder = true;      // false: update_critic received obs.detach() (proto.py:190-193): encoder_opt.step() finds no gradients
};

namespace exorl {

struct PCarver {
    float* base; int64_t off = 0;
    explicit PCarver(float* b) : base(b) {}
    float* take(int64_t n) { float* p = base ? base + off : nullptr; off += round_up(n, 64); return p; }
};

static void take_mlp(Mlp& m, PCarver& c, int64_t rows, float* last_act, float* last_dact) {
    m.act.clear(); m.dact.clear();
    for (size_t l = 0; l < m.L.size(); ++l) {
        const bool last = l + 1 == m.L.size();
        m.act.push_back(last && last_act ? last_act : c.take(rows * m.L[l].out));
        m.dact.push_back(last && last_dact ? last_dact : c.take(rows * m.L[l].out));
    }
}

static void pcarve(exorl_pixel_agent* a, PCarver& c) {
    const auto& g = a->cfg;
    const int64_t B = g.batch, A = g.act_dim, F = g.feature_dim, R = a->R, img = (int64_t)g.c_in * g.hw * g.hw;
    a->flat[0][0] = c.take(a->enc_total);
    for (int w = 1; w < 4; ++w) a->flat[0][w] = c.take(a->enc_total);
    for (int w = 0; w < 4; ++w) a->flat[1][w] = c.take(a->actor.total);
    for (int w = 0; w < 4; ++w) a->flat[2][w] = c.take(a->critic.total);
    a->flat[3][0] = c.take(a->critic.total);
    a->enc_target = c.take(a->enc_total); a->enc_m2 = c.take(a->enc_total); a->enc_v2 = c.take(a->enc_total);
    a->bn2d = c.take(2 * 16 + 1);
    a->bn_scratch = reinterpret_cast<double*>(c.take(2 * (32 + 16 * BN2_CHUNKS)));
    a->obs = reinterpret_cast<unsigned char*>(c.take((B * img + 3) / 4));
    a->next_obs = reinterpret_cast<unsigned char*>(c.take((B * img + 3) / 4));
    a->action = c.take(B * A); a->reward = c.take(B); a->discount = c.take(B);
    a->meta = c.take(B * (g.meta_dim > 0 ? g.meta_dim : 1));
    a->aug_o = c.take(B * img); a->aug_n = c.take(B * img);
    const int64_t ews = exorl_encoder_workspace_floats((int32_t)B, g.c_in, g.hw);
    a->enc_ws_o = c.take(ews); a->enc_ws_n = c.take(ews);
    a->splitk = c.take((int64_t)(SPLITK + 1) * B * F);
    a->splitk2 = c.take((int64_t)(SPLITK + 1) * B * F);
    for (TrunkAct* t : {&a->ta_n, &a->ta_o, &a->tt, &a->tc}) { t->z = c.take(B * F); t->h = c.take(B * F); t->xhat = c.take(B * F); t->rstd = c.take(B); }
    a->xq_t = c.take(B * (F + A)); a->xq_c = c.take(B * (F + A));
    a->dxq[0] = c.take(B * (F + A)); a->dxq[1] = c.take(B * (F + A));
    a->q = c.take(2 * B); a->tq = c.take(2 * B); a->dq = c.take(2 * B);
    a->mu_n = c.take(B * A); a->mu_o = c.take(B * A); a->dmu = c.take(B * A);
    a->dh = c.take(B * F); a->dz = c.take(B * F); a->dfeat = c.take(B * R);
    a->stats = c.take(4 + EXORL_N_METRICS);
    a->metrics = a->stats ? a->stats + 4 : nullptr;
    a->shifts = reinterpret_cast<int32_t*>(c.take(4 * B));
    // actor policy: outputs land in mu_* (policy on next_obs shares the Mlp buffers: it is consumed before the obs pass runs)
    take_mlp(a->actor.head[0], c, B, nullptr, nullptr);
    // critic heads: outputs are the halves of q (B each); gradients at the outputs are the halves of dq
    // (CriticSF: the heads emit sf_dim features into their own buffers, q / dq hold the task-weighted scalars)
    const bool sf = g.sf_dim > 0;
    for (int i = 0; i < 2; ++i) take_mlp(a->critic.head[i], c, B, (a->q && !sf) ? a->q + i * B : nullptr, (a->dq && !sf) ? a->dq + i * B : nullptr);
    a->act_ws = c.take(exorl_encoder_workspace_floats(1, g.c_in, g.hw) + img + 4 * F + 2 * 1024 + 64);
    a->act_part = c.take(256 * F + (int64_t)cdiv(g.hidden_dim, 4) * ACT_FAST_ROWS * 16 + F);   // trunk shares [256][F] | head shares | trunk output h[F]
    a->act_ticket = reinterpret_cast<unsigned int*>(c.take(8));
}

// z = [x | meta] W0^T + b0 (split-K over the encoding's columns, one more slab for the meta columns), then LayerNorm + tanh.
// The trunk's input is cat([encoding, skill]) for the meta-conditioned agents (diayn.py:163-165, ddpg.py:305-312): the two parts stay
// where they are (x: rows x R inside the encoder workspace, meta: rows x M) and enter as separate GEMM problems.
static int trunk_forward(exorl_pixel_agent* a, const PNet& n, const float* P, const float* x, const float* meta, int rows, const TrunkAct& t, int prec,
                         hipStream_t s) {
    const int D = n.D, F = n.F, R = a->R, M = D - R;
    int kc = (int)round_up(cdiv(R, SPLITK), 4);
    GemmProblem p[SPLITK + 1];                 // all slabs in one launch: SPLITK x rows/64 workgroups
    int cnt = 0;
    for (int i = 0; i < SPLITK; ++i) {
        const int k0 = i * kc;
        if (k0 >= R) break;
        const int k = R - k0 < kc ? R - k0 : kc;
        p[cnt] = GemmProblem{x + k0, P + n.trunk.W + k0, a->splitk + (int64_t)cnt * rows * F, nullptr, rows, F, k, R, D, F};
        ++cnt;
    }
    if (M > 0) {
        EXORL_REQUIRE(meta, "pixel_agent: the trunk takes %d meta columns but no meta rows were given", M);
        p[cnt] = GemmProblem{meta, P + n.trunk.W + R, a->splitk + (int64_t)cnt * rows * F, nullptr, rows, F, M, M, D, F};
        ++cnt;
    }
    EXORL_TRY(gemm_grouped(prec, 0, 0, p, cnt, false, false, s));
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid1((int64_t)rows * F)), dim3(256), 0, s, a->splitk, P + n.trunk.b, t.z, (int64_t)rows * F, F, cnt);
    EXORL_LAUNCH_CHECK();
    return ln_tanh_fwd(t.z, P + n.g, P + n.beta, t.h, t.xhat, t.rstd, rows, F, 1, 0, 0, s);
}

// Two trunks on the SAME encoding (actor + target critic on next_obs, actor + critic on obs) in one launch with their k-slabs interleaved:
// slab i of net A and of net B are neighbouring problems, their workgroups land on the same XCDs (linear id mod 8) one dispatch round apart,
// and the second reads the 160 MB of encodings from L2 instead of HBM. Without meta columns only (2 x 16 problems = the launch limit).
static int trunk_forward_pair(exorl_pixel_agent* a, const PNet& na, const float* Pa, const TrunkAct& ta, const PNet& nb, const float* Pb, const TrunkAct& tb,
                              const float* x, int rows, int prec, hipStream_t s) {
    const int F = na.F, R = a->R, D = na.D;
    EXORL_REQUIRE(D == R && nb.D == R && nb.F == F, "trunk_forward_pair: needs two meta-free trunks of one width");
    const int kc = (int)round_up(cdiv(R, SPLITK), 4);
    GemmProblem p[2 * SPLITK];
    int cnt = 0;
    for (int i = 0; i < SPLITK; ++i) {
        const int k0 = i * kc;
        if (k0 >= R) break;
        const int k = R - k0 < kc ? R - k0 : kc;
        p[2 * cnt] = GemmProblem{x + k0, Pa + na.trunk.W + k0, a->splitk + (int64_t)cnt * rows * F, nullptr, rows, F, k, R, D, F};
        p[2 * cnt + 1] = GemmProblem{x + k0, Pb + nb.trunk.W + k0, a->splitk2 + (int64_t)cnt * rows * F, nullptr, rows, F, k, R, D, F};
        ++cnt;
    }
    EXORL_TRY(gemm_grouped(prec, 0, 0, p, 2 * cnt, false, false, s));
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid1((int64_t)rows * F)), dim3(256), 0, s, a->splitk, Pa + na.trunk.b, ta.z, (int64_t)rows * F, F, cnt);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid1((int64_t)rows * F)), dim3(256), 0, s, a->splitk2, Pb + nb.trunk.b, tb.z, (int64_t)rows * F, F, cnt);
    EXORL_LAUNCH_CHECK();
    EXORL_TRY(ln_tanh_fwd(ta.z, Pa + na.g, Pa + na.beta, ta.h, ta.xhat, ta.rstd, rows, F, 1, 0, 0, s));
    return ln_tanh_fwd(tb.z, Pb + nb.g, Pb + nb.beta, tb.h, tb.xhat, tb.rstd, rows, F, 1, 0, 0, s);
}

// dh (rows, F) at the trunk output -> parameter grads (W0, b0, gain, beta) and optionally d/d(encoding) (rows, R)
static int trunk_backward(exorl_pixel_agent* a, const PNet& n, const float* P, float* G, const float* x, const float* meta, int rows, const TrunkAct& t,
                          const float* dh, float* dx, int prec, hipStream_t s) {
    const int D = n.D, F = n.F, R = a->R, M = D - R;
    EXORL_TRY(ln_param_grad(dh, t.h, t.xhat, G + n.g, G + n.beta, rows, F, 1, 0, 0, s));
    EXORL_TRY(ln_tanh_bwd(dh, t.h, t.xhat, t.rstd, P + n.g, a->dz, rows, F, 1, 0, 0, s));
    EXORL_TRY(colsum(a->dz, G + n.trunk.b, rows, F, 1, 0, 0, s));
    GemmProblem w{a->dz, x, G + n.trunk.W, nullptr, F, R, rows, F, R, D};
    EXORL_TRY(gemm_grouped(prec, 1, 1, &w, 1, false, false, s));
    if (M > 0) {
        GemmProblem wm{a->dz, meta, G + n.trunk.W + R, nullptr, F, M, rows, F, M, D};
        EXORL_TRY(gemm_grouped(prec, 1, 1, &wm, 1, false, false, s));
    }
    if (dx) {
        GemmProblem g{a->dz, P + n.trunk.W, dx, nullptr, rows, R, F, F, D, R};
        EXORL_TRY(gemm_grouped(prec, 0, 1, &g, 1, false, false, s));
    }
    return 0;
}

static int padam(exorl_pixel_agent* a, int net, int64_t n, float* target, hipStream_t s) {
    return adam_step(a->flat[net][0], a->flat[net][1], a->flat[net][2], a->flat[net][3], n, a->cfg.lr, 0.9f, 0.999f, 1e-8f, net == 0 ? a->t_enc : a->t,
                     target, target ? a->cfg.tau : 0.f, s);
}

}  // namespace exorl

extern "C" {

static int check_pcfg(const exorl_pixel_cfg* c) {
    EXORL_REQUIRE(c, "pixel_agent: null cfg");
    EXORL_REQUIRE(c->c_in >= 1 && c->c_in <= 16 && (c->hw == 84 || c->hw == 64), "pixel_agent: obs_shape (%d, %d, %d) unsupported (84x84 or 64x64, <= 16 channels)",
                  c->c_in, c->hw, c->hw);
    EXORL_REQUIRE(c->act_dim >= 1 && c->act_dim <= 64 && c->feature_dim >= 1 && c->feature_dim <= 1024 && c->hidden_dim >= 1 && c->batch >= 1,
                  "pixel_agent: unsupported dims A=%d feature_dim=%d H=%d B=%d", c->act_dim, c->feature_dim, c->hidden_dim, c->batch);
    EXORL_REQUIRE(c->precision >= EXORL_PREC_F32 && c->precision <= EXORL_PREC_BF16X6, "pixel_agent: unknown precision %d", c->precision);
    EXORL_REQUIRE(c->meta_dim >= 0 && c->meta_dim <= 256, "pixel_agent: meta_dim=%d unsupported (0..256)", c->meta_dim);
    EXORL_REQUIRE(c->sf_dim == 0 || (c->sf_dim >= 1 && c->sf_dim == c->meta_dim), "pixel_agent: sf_dim=%d must equal meta_dim=%d (the task vector is the meta row, aps.py:236-238)",
                  c->sf_dim, c->meta_dim);
    return 0;
}

static void pdescribe(exorl_pixel_agent* a) {
    const auto& c = a->cfg;
    a->R = (int)exorl_encoder_out_dim(c.hw);
    a->enc_total = exorl_encoder_param_floats(c.c_in, c.hw);
    a->actor = make_pnet(a->R + c.meta_dim, c.feature_dim, c.hidden_dim, c.feature_dim, c.act_dim, 1);
    a->critic = make_pnet(a->R + c.meta_dim, c.feature_dim, c.hidden_dim, c.feature_dim + c.act_dim, c.sf_dim > 0 ? c.sf_dim : 1, 2);
}

size_t exorl_pixel_agent_workspace_bytes(const exorl_pixel_cfg* cfg) {
    if (check_pcfg(cfg) != 0) return 0;
    exorl_pixel_agent tmp;
    tmp.cfg = *cfg;
    pdescribe(&tmp);
    PCarver c(nullptr);
    pcarve(&tmp, c);
    return (size_t)c.off * sizeof(float);
}

int exorl_pixel_agent_create(const exorl_pixel_cfg* cfg, void* workspace, size_t workspace_bytes, exorl_pixel_agent_t** out) {
    EXORL_REQUIRE(out && workspace, "pixel_agent_create: null argument (the workspace is caller-owned)");
    EXORL_TRY(check_pcfg(cfg));
    const size_t bytes = exorl_pixel_agent_workspace_bytes(cfg);
    EXORL_REQUIRE(workspace_bytes >= bytes && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "pixel_agent_create: workspace of %zu bytes (need %zu, 256-byte aligned)",
                  workspace_bytes, bytes);
    auto* a = new exorl_pixel_agent();
    a->cfg = *cfg;
    pdescribe(a);
    a->ws = static_cast<float*>(workspace);
    PCarver c(a->ws);
    pcarve(a, c);
    if (hipMemset(a->ws, 0, bytes) != hipSuccess) { set_error("pixel_agent_create: hipMemset failed"); delete a; return 1; }
    {                                            // BatchNorm2d buffers: running_mean 0, running_var 1, num_batches_tracked 0
        float ones[16];
        for (float& o : ones) o = 1.0f;
        if (hipMemcpy(a->bn2d + cfg->c_in, ones, sizeof(float) * cfg->c_in, hipMemcpyHostToDevice) != hipSuccess) { set_error("pixel_agent_create: hipMemcpy failed"); delete a; return 1; }
    }
    *out = a;
    return 0;
}

int exorl_pixel_agent_destroy(exorl_pixel_agent_t* a) {
    if (!a) return 0;
    (void)hipDeviceSynchronize();
    delete a;
    return 0;
}

// nets: 0 encoder (convnet.{0,2,4,6}.{weight,bias}; weights reported as (32, ci*9)), 1 actor, 2 critic, 3 critic_target
int exorl_pixel_agent_num_tensors(exorl_pixel_agent_t* a, int32_t net, int32_t* n) {
    EXORL_REQUIRE(a && n && net >= 0 && net <= 3, "pixel_agent_num_tensors: bad arguments");
    *n = net == 0 ? 8 : (int32_t)(net == 1 ? a->actor.tensors.size() : a->critic.tensors.size());
    return 0;
}

int exorl_pixel_agent_tensor(exorl_pixel_agent_t* a, int32_t net, int32_t index, int32_t what, void** ptr, int64_t* rows, int64_t* cols) {
    EXORL_REQUIRE(a && ptr && rows && cols && net >= 0 && net <= 3 && what >= 0 && what <= 3 && (net != 3 || what == 0), "pixel_agent_tensor: bad arguments");
    if (net == 0) {
        EXORL_REQUIRE(index >= 0 && index < 8, "pixel_agent_tensor: encoder tensor %d out of range", index);
        int64_t off = 0;
        for (int l = 0; l < 4; ++l) {
            const int64_t ci = l == 0 ? a->cfg.c_in : 32;
            if (index == 2 * l) { *ptr = a->flat[0][what] + off; *rows = 32; *cols = ci * 9; return 0; }
            off += round_up(32 * ci * 9, 4);
            if (index == 2 * l + 1) { *ptr = a->flat[0][what] + off; *rows = 32; *cols = 1; return 0; }
            off += 32;
        }
    }
    const PNet& n = net == 1 ? a->actor : a->critic;
    EXORL_REQUIRE(index >= 0 && index < (int32_t)n.tensors.size(), "pixel_agent_tensor: index %d out of range", index);
    *ptr = a->flat[net][what] + n.tensors[index].off; *rows = n.tensors[index].rows; *cols = n.tensors[index].cols;
    return 0;
}

int exorl_pixel_agent_sync_target(exorl_pixel_agent_t* a, void* stream) {
    EXORL_REQUIRE(a, "pixel_agent_sync_target: null handle");
    EXORL_CHECK_HIP(hipMemcpyAsync(a->flat[3][0], a->flat[2][0], a->critic.total * sizeof(float), hipMemcpyDeviceToDevice, as_stream(stream)));
    return 0;
}

int exorl_pixel_agent_set_batch(exorl_pixel_agent_t* a, const unsigned char* obs, const float* action, const float* reward, const float* discount,
                                const unsigned char* next_obs, void* stream) {
    EXORL_REQUIRE(a && obs && action && reward && discount && next_obs, "pixel_agent_set_batch: null argument");
    hipStream_t s = as_stream(stream);
    const size_t B = a->cfg.batch, img = (size_t)a->cfg.c_in * a->cfg.hw * a->cfg.hw;
    EXORL_CHECK_HIP(hipMemcpyAsync(a->obs, obs, B * img, hipMemcpyDeviceToDevice, s));
    EXORL_CHECK_HIP(hipMemcpyAsync(a->next_obs, next_obs, B * img, hipMemcpyDeviceToDevice, s));
    EXORL_CHECK_HIP(hipMemcpyAsync(a->action, action, B * a->cfg.act_dim * 4, hipMemcpyDeviceToDevice, s));
    EXORL_CHECK_HIP(hipMemcpyAsync(a->reward, reward, B * 4, hipMemcpyDeviceToDevice, s));
    EXORL_CHECK_HIP(hipMemcpyAsync(a->discount, discount, B * 4, hipMemcpyDeviceToDevice, s));
    return 0;
}

int exorl_pixel_agent_batch_slots(exorl_pixel_agent_t* a, exorl_batch_out* out) {
    EXORL_REQUIRE(a && out, "pixel_agent_batch_slots: null argument");
    const int64_t img = (int64_t)a->cfg.c_in * a->cfg.hw * a->cfg.hw;
    out->obs = a->obs; out->obs_stride = img;
    out->action = a->action; out->action_stride = a->cfg.act_dim;
    out->reward = a->reward; out->discount = a->discount;
    out->next_obs = a->next_obs; out->next_obs_stride = img;
    out->meta = a->cfg.meta_dim > 0 ? a->meta : nullptr; out->meta_stride = a->cfg.meta_dim;
    return 0;
}

// One DDPG update on the batch in the slots. shifts_*: (B,2) int32 augmentation shifts (utils.py:244-248) or null -> Philox;
// noise_*: (B,A) standard normals for the two TruncatedNormal draws (critic target first, actor second) or null -> Philox.
// RandomShiftsAug of obs and next_obs (ddpg.py:213-215; proto.py:168-170 augments once and encodes several times)
int exorl_pixel_agent_augment(exorl_pixel_agent_t* a, const int32_t* shifts_obs, const int32_t* shifts_next, void* stream) {
    EXORL_REQUIRE(a, "pixel_agent_augment: null handle");
    hipStream_t s = as_stream(stream);
    const auto& c = a->cfg;
    EXORL_TRY(exorl_aug_shift(a->obs, c.batch, c.c_in, c.hw, 4, shifts_obs, c.seed, 2 * a->aug_counter, a->aug_o, s));
    EXORL_TRY(exorl_aug_shift(a->next_obs, c.batch, c.c_in, c.hw, 4, shifts_next, c.seed, 2 * a->aug_counter + 1, a->aug_n, s));
    a->aug_counter += 1;
    a->augmented = true;
    return 0;
}

// encoder (target != 0: encoder_target) on the augmented obs (which == 0) or next_obs (which == 1); *feat_out_dev -> (batch, repr_dim)
int exorl_pixel_agent_encode(exorl_pixel_agent_t* a, int32_t which, int32_t target, float** feat_out_dev, void* stream) {
    EXORL_REQUIRE(a && feat_out_dev && a->augmented && (which == 0 || which == 1), "pixel_agent_encode: bad arguments (augment first)");
    const auto& c = a->cfg;
    EXORL_TRY(exorl_encoder_forward_prec(target ? a->enc_target : a->flat[0][0], c.c_in, c.hw, which ? a->aug_n : a->aug_o, c.batch,
                                         which ? a->enc_ws_n : a->enc_ws_o, feat_out_dev, c.precision, stream));
    if (!target) { (which ? a->feat_n : a->feat_o) = *feat_out_dev; (which ? a->have_feat_n : a->have_feat_o) = true; }
    else (which ? a->have_feat_n : a->have_feat_o) = false;           // the workspace now holds the target encoder's pass
    return 0;
}

// Backward through the encoder pass last run by exorl_pixel_agent_encode(which, 0) from dfeat_dev (batch, repr_dim; overwritten), then
// one Adam step of the encoder with optimiser state `opt` (0: encoder_opt, ddpg.py:188-190; 1: the encoder's slots in proto_opt, proto.py:75-78)
int exorl_pixel_agent_encoder_step(exorl_pixel_agent_t* a, int32_t which, float* dfeat_dev, int32_t opt, void* stream) {
    EXORL_REQUIRE(a && dfeat_dev && (which == 0 || which == 1) && opt >= 0 && opt <= 2, "pixel_agent_encoder_step: bad arguments");
    hipStream_t s = as_stream(stream);
    const auto& c = a->cfg;
    EXORL_TRY(exorl_encoder_backward_prec(a->flat[0][0], c.c_in, c.hw, which ? a->aug_n : a->aug_o, c.batch, which ? a->enc_ws_n : a->enc_ws_o, dfeat_dev,
                                     a->flat[0][1], a->cfg.precision, s));
    if (opt == 0) { a->t_enc += 1; return padam(a, 0, a->enc_total, nullptr, s); }
    a->t2 += 1;
    EXORL_TRY(adam_step(a->flat[0][0], a->flat[0][1], a->enc_m2, a->enc_v2, a->enc_total, c.lr, 0.9f, 0.999f, 1e-8f, a->t2, nullptr, 0.f, s));
    if (opt == 2) { a->t_enc += 1; return padam(a, 0, a->enc_total, nullptr, s); }      // rnd.py:86-89: rnd_opt.step() then encoder_opt.step(), same gradients
    return 0;
}

// RND on pixels (rnd.py:47-53): x = clamp(BatchNorm2d(RandomShiftsAug(obs))) -> the agent's encoder (*feat_pred_dev, kept for
// exorl_pixel_agent_encoder_step(0, ...)) and the frozen copy held in the encoder_target slot (*feat_target_dev). shifts_dev: (batch, 2) or null.
int exorl_pixel_agent_rnd_features(exorl_pixel_agent_t* a, const int32_t* shifts_dev, float clip_val, float** feat_pred_dev, float** feat_target_dev,
                                   void* stream) {
    EXORL_REQUIRE(a && feat_pred_dev && feat_target_dev && clip_val > 0.f, "pixel_agent_rnd_features: bad arguments");
    hipStream_t s = as_stream(stream);
    const auto& c = a->cfg;
    const int64_t hw = (int64_t)c.hw * c.hw, total = (int64_t)c.batch * c.c_in * hw;
    EXORL_TRY(exorl_aug_shift(a->obs, c.batch, c.c_in, c.hw, 4, shifts_dev, c.seed, (1ull << 62) | a->rnd_aug_counter, a->aug_o, s));
    a->rnd_aug_counter += 1;
    double *mean = a->bn_scratch, *var = mean + 16, *part = mean + 32;
    const double count = (double)c.batch * (double)hw;
    hipLaunchKernelGGL(bn2d_partial_kernel, dim3(c.c_in, BN2_CHUNKS), dim3(256), 0, s, a->aug_o, c.batch, c.c_in, hw, (const double*)nullptr, part);
    hipLaunchKernelGGL(bn2d_finish_kernel, dim3(1), dim3(16), 0, s, part, mean, var, a->bn2d, c.c_in, count, 0);
    hipLaunchKernelGGL(bn2d_partial_kernel, dim3(c.c_in, BN2_CHUNKS), dim3(256), 0, s, a->aug_o, c.batch, c.c_in, hw, (const double*)mean, part);
    hipLaunchKernelGGL(bn2d_finish_kernel, dim3(1), dim3(16), 0, s, part, mean, var, a->bn2d, c.c_in, count, 1);
    hipLaunchKernelGGL(bn2d_apply_kernel, dim3(grid1(total)), dim3(256), 0, s, a->aug_o, c.c_in, hw, total, (const double*)mean, (const double*)var, clip_val);
    EXORL_LAUNCH_CHECK();
    a->augmented = false;                        // aug_o no longer holds a plain augmentation; aug_n is stale
    a->have_feat_o = a->have_feat_n = false;
    EXORL_TRY(exorl_encoder_forward_prec(a->flat[0][0], c.c_in, c.hw, a->aug_o, c.batch, a->enc_ws_o, feat_pred_dev, c.precision, stream));
    return exorl_encoder_forward_prec(a->enc_target, c.c_in, c.hw, a->aug_o, c.batch, a->enc_ws_n, feat_target_dev, c.precision, stream);
}

// BatchNorm2d buffers of RND's normalize_obs: running_mean[c_in], running_var[c_in], num_batches_tracked (as float)
int exorl_pixel_agent_bn_state(exorl_pixel_agent_t* a, void** ptr_dev, int64_t* n_floats) {
    EXORL_REQUIRE(a && ptr_dev && n_floats, "pixel_agent_bn_state: null argument");
    *ptr_dev = a->bn2d; *n_floats = 2 * a->cfg.c_in + 1;
    return 0;
}

// encoder_target: init != 0 copies the encoder (deepcopy at construction), else Polyak with rate tau (proto.py:200-201)
int exorl_pixel_agent_encoder_target(exorl_pixel_agent_t* a, float tau, int32_t init, void* stream) {
    EXORL_REQUIRE(a, "pixel_agent_encoder_target: null handle");
    if (init) {
        EXORL_CHECK_HIP(hipMemcpyAsync(a->enc_target, a->flat[0][0], a->enc_total * sizeof(float), hipMemcpyDeviceToDevice, as_stream(stream)));
        return 0;
    }
    return soft_update(a->flat[0][0], a->enc_target, a->enc_total, tau, as_stream(stream));
}

int exorl_pixel_agent_set_train_encoder(exorl_pixel_agent_t* a, int32_t enable) {
    EXORL_REQUIRE(a, "pixel_agent_set_train_encoder: null handle");
    a->train_encoder = enable != 0;
    return 0;
}

int exorl_pixel_agent_encoder_target_ptr(exorl_pixel_agent_t* a, void** ptr_dev) {
    EXORL_REQUIRE(a && ptr_dev, "pixel_agent_encoder_target_ptr: null argument");
    *ptr_dev = a->enc_target;
    return 0;
}

// training state that is not a tensor view: Adam step counts (critic/actor/encoder share one, proto_opt's encoder state has its
// own) and the Philox counters of the noise / augmentation / act() streams — what a pickled agent needs to continue bit-identically
int exorl_pixel_agent_state(exorl_pixel_agent_t* a, int64_t* steps3, uint64_t* counters4) {
    EXORL_REQUIRE(a && steps3 && counters4, "pixel_agent_state: null argument");
    steps3[0] = a->t; steps3[1] = a->t2; steps3[2] = a->t_enc;
    counters4[0] = a->noise_counter; counters4[1] = a->aug_counter; counters4[2] = a->act_counter; counters4[3] = a->rnd_aug_counter;
    return 0;
}
int exorl_pixel_agent_set_state(exorl_pixel_agent_t* a, const int64_t* steps3, const uint64_t* counters4) {
    EXORL_REQUIRE(a && steps3 && counters4 && steps3[0] >= 0 && steps3[1] >= 0 && steps3[2] >= 0, "pixel_agent_set_state: bad arguments");
    a->t = steps3[0]; a->t2 = steps3[1]; a->t_enc = steps3[2];
    a->noise_counter = counters4[0]; a->aug_counter = counters4[1]; a->act_counter = counters4[2]; a->rnd_aug_counter = counters4[3];
    a->augmented = false;
    a->have_feat_o = a->have_feat_n = false;
    return 0;
}
int exorl_pixel_agent_encoder_opt2(exorl_pixel_agent_t* a, void** m_dev, void** v_dev, int64_t* n) {
    EXORL_REQUIRE(a && m_dev && v_dev && n, "pixel_agent_encoder_opt2: null argument");
    *m_dev = a->enc_m2; *v_dev = a->enc_v2; *n = a->enc_total;
    return 0;
}

int exorl_pixel_agent_update(exorl_pixel_agent_t* a, float stddev, const int32_t* shifts_obs, const int32_t* shifts_next, const float* noise_c,
                             const float* noise_a, void* stream) {
    EXORL_REQUIRE(a && stddev > 0.f, "pixel_agent_update: bad arguments");
    hipStream_t s = as_stream(stream);
    const auto& c = a->cfg;
    const int B = c.batch, A = c.act_dim, F = c.feature_dim, FA = F + A, prec = c.precision;
    const float inv_b = 1.0f / (float)B;
    float *Pe = a->flat[0][0], *Pa = a->flat[1][0], *Pc = a->flat[2][0], *Pt = a->flat[3][0];
    float *Ge = a->flat[0][1], *Ga = a->flat[1][1], *Gc = a->flat[2][1];
    a->t += 1;
    const float* mt = c.meta_dim > 0 ? a->meta : nullptr;
    const int S = c.sf_dim;
    auto sf_q = [&]() -> int {                 // Q_n = task . features_n (aps.py:55-58) -> a->q
        if (S == 0) return 0;
        hipLaunchKernelGGL(sf_dot_kernel, dim3(cdiv(2 * B, 256)), dim3(256), 0, s, a->critic.head[0].act[2], a->critic.head[1].act[2], mt, (int64_t)c.meta_dim, a->q, B, S);
        EXORL_LAUNCH_CHECK();
        return 0;
    };
    auto sf_dout = [&]() -> int {              // dQ_n -> gradient at the heads' feature outputs
        if (S == 0) return 0;
        hipLaunchKernelGGL(sf_dout_kernel, dim3(cdiv(2 * B * S, 256)), dim3(256), 0, s, a->dq, mt, (int64_t)c.meta_dim, a->critic.head[0].dact[2], a->critic.head[1].dact[2], B, S);
        EXORL_LAUNCH_CHECK();
        return 0;
    };
    // ---- aug_and_encode (ddpg.py:213-215, 312-315); shifts_obs == (const int32_t*)-1: keep the images exorl_pixel_agent_augment made;
    // (const int32_t*)-2: keep the encodings exorl_pixel_agent_encode(0 / 1, online) made as well — the agents that step the encoder
    // through their own module pass the critic the encodings computed BEFORE that step (icm.py:97-131, diayn.py:137-170), detached
    const bool keep_feat = shifts_obs == reinterpret_cast<const int32_t*>(-2);
    if (keep_feat) {
        EXORL_REQUIRE(a->have_feat_o && a->have_feat_n && !a->train_encoder, "pixel_agent_update: kept encodings need exorl_pixel_agent_encode(0) and (1) "
                      "with the online encoder first, and set_train_encoder(0) (they are detached)");
    } else {
        if (shifts_obs != reinterpret_cast<const int32_t*>(-1)) EXORL_TRY(exorl_pixel_agent_augment(a, shifts_obs, shifts_next, stream));
        EXORL_REQUIRE(a->augmented, "pixel_agent_update: no augmented batch");
        EXORL_TRY(exorl_encoder_forward_prec(Pe, c.c_in, c.hw, a->aug_o, B, a->enc_ws_o, &a->feat_o, a->cfg.precision, s));
        EXORL_TRY(exorl_encoder_forward_prec(Pe, c.c_in, c.hw, a->aug_n, B, a->enc_ws_n, &a->feat_n, a->cfg.precision, s));
    }
    a->have_feat_o = a->have_feat_n = false;
    // ---- update_critic (ddpg.py:240-268)
    const bool pair = c.meta_dim == 0 && !(tune_variant() & 4096);      // exorl_gemm_tune bit 4096: separate trunk launches (A/B)
    if (pair) EXORL_TRY(trunk_forward_pair(a, a->actor, Pa, a->ta_n, a->critic, Pt, a->tt, a->feat_n, B, prec, s));
    else EXORL_TRY(trunk_forward(a, a->actor, Pa, a->feat_n, mt, B, a->ta_n, prec, s));
    Mlp& pol = a->actor.head[0];
    EXORL_TRY(mlp_forward(pol, Pa, a->ta_n.h, F, B, prec, s));
    EXORL_CHECK_HIP(hipMemcpyAsync(a->mu_n, pol.act[2], sizeof(float) * B * A, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(tanh_kernel, dim3(grid1((int64_t)B * A)), dim3(256), 0, s, a->mu_n, (int64_t)B * A);
    EXORL_LAUNCH_CHECK();
    if (!pair) EXORL_TRY(trunk_forward(a, a->critic, Pt, a->feat_n, mt, B, a->tt, prec, s));
    EXORL_TRY(launch_concat(a->tt.h, F, F, a->mu_n, A, A, a->xq_t, B, s));           // action columns overwritten by the sample below
    NoiseSpec nc{noise_c, c.seed, 2 * a->noise_counter, nullptr};
    EXORL_TRY(sample_action(a->mu_n, nc, stddev, c.stddev_clip, 1, a->xq_t + F, FA, B, A, nullptr, s));
    // the target's Q heads reuse the critic's Mlp buffers (outputs to q), then are copied to tq
    EXORL_TRY(mlp_forward_many(a->critic.head, 2, Pt, a->xq_t, FA, B, prec, s));          // Q1 and Q2 layer by layer in shared launches
    EXORL_TRY(sf_q());
    EXORL_CHECK_HIP(hipMemcpyAsync(a->tq, a->q, sizeof(float) * 2 * B, hipMemcpyDeviceToDevice, s));
    EXORL_TRY(trunk_forward(a, a->critic, Pc, a->feat_o, mt, B, a->tc, prec, s));
    EXORL_TRY(launch_concat(a->tc.h, F, F, a->action, A, A, a->xq_c, B, s));
    EXORL_TRY(mlp_forward_many(a->critic.head, 2, Pc, a->xq_c, FA, B, prec, s));
    EXORL_TRY(sf_q());
    EXORL_TRY(critic_loss(a->q, a->tq, a->reward, a->discount, a->dq, a->metrics, B, inv_b, s));
    EXORL_TRY(sf_dout());
    // both heads' backward passes layer by layer in shared launches; d/d(input) of the two is summed by the second head's accumulating dgrad
    // (dx0 + dx1, the same fp32 addition add_cols_kernel used to make of two buffers)
    EXORL_TRY(mlp_backward_many(a->critic.head, 2, Pc, Gc, a->xq_c, FA, B, prec, s, a->dxq[0]));
    hipLaunchKernelGGL(add_cols_kernel, dim3(grid1((int64_t)B * F)), dim3(256), 0, s, a->dxq[0], (int64_t)FA, (const float*)nullptr, (int64_t)FA, 0, F, a->dh, B);
    EXORL_LAUNCH_CHECK();
    EXORL_TRY(trunk_backward(a, a->critic, Pc, Gc, a->feat_o, mt, B, a->tc, a->dh, a->train_encoder ? a->dfeat : nullptr, prec, s));
    if (a->train_encoder) EXORL_TRY(exorl_encoder_backward_prec(Pe, c.c_in, c.hw, a->aug_o, B, a->enc_ws_o, a->dfeat, Ge, a->cfg.precision, s));
    EXORL_TRY(padam(a, 2, a->critic.total, nullptr, s));
    if (a->train_encoder) { a->t_enc += 1; EXORL_TRY(padam(a, 0, a->enc_total, nullptr, s)); }
    // ---- update_actor (ddpg.py:270-292) on obs.detach(): the encoding computed above, the critic just updated
    if (pair) EXORL_TRY(trunk_forward_pair(a, a->actor, Pa, a->ta_o, a->critic, Pc, a->tc, a->feat_o, B, prec, s));
    else EXORL_TRY(trunk_forward(a, a->actor, Pa, a->feat_o, mt, B, a->ta_o, prec, s));
    EXORL_TRY(mlp_forward(pol, Pa, a->ta_o.h, F, B, prec, s));
    EXORL_CHECK_HIP(hipMemcpyAsync(a->mu_o, pol.act[2], sizeof(float) * B * A, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(tanh_kernel, dim3(grid1((int64_t)B * A)), dim3(256), 0, s, a->mu_o, (int64_t)B * A);
    EXORL_LAUNCH_CHECK();
    if (!pair) EXORL_TRY(trunk_forward(a, a->critic, Pc, a->feat_o, mt, B, a->tc, prec, s));
    EXORL_TRY(launch_concat(a->tc.h, F, F, a->mu_o, A, A, a->xq_c, B, s));
    NoiseSpec na{noise_a, c.seed, 2 * a->noise_counter + 1, nullptr};
    EXORL_TRY(sample_action(a->mu_o, na, stddev, c.stddev_clip, 1, a->xq_c + F, FA, B, A, a->metrics + EXORL_M_ACTOR_LOGPROB, s));
    a->noise_counter += 1;
    EXORL_TRY(mlp_forward_many(a->critic.head, 2, Pc, a->xq_c, FA, B, prec, s));
    EXORL_TRY(sf_q());
    EXORL_TRY(actor_stats(a->q, a->stats, B, s));
    EXORL_TRY(actor_dq(a->q, a->stats, a->dq, B, inv_b, 0.f, 0, s));
    EXORL_TRY(sf_dout());
    EXORL_TRY(mlp_backward_many(a->critic.head, 2, Pc, Gc, a->xq_c, FA, B, prec, s, a->dxq[0]));     // Gc: scratch now
    hipLaunchKernelGGL(add_cols_kernel, dim3(grid1((int64_t)B * A)), dim3(256), 0, s, a->dxq[0], (int64_t)FA, (const float*)nullptr, (int64_t)FA, F, A, a->dmu, B);
    hipLaunchKernelGGL(dpre_kernel, dim3(grid1((int64_t)B * A)), dim3(256), 0, s, a->dmu, a->mu_o, pol.dact[2], (int64_t)B * A);
    EXORL_LAUNCH_CHECK();
    EXORL_TRY(mlp_backward(pol, Pa, Ga, a->ta_o.h, F, B, a->dh, prec, s));
    EXORL_TRY(trunk_backward(a, a->actor, Pa, Ga, a->feat_o, mt, B, a->ta_o, a->dh, nullptr, prec, s));
    EXORL_TRY(padam(a, 1, a->actor.total, nullptr, s));
    // ---- soft update (ddpg.py:326-327)
    return soft_update(Pc, Pt, a->critic.total, c.tau, s);
}

// actor_loss = -mean(min Q) is stats[1] / B (actor_stats); the rest comes from critic_loss / sample_action
int exorl_pixel_agent_metrics(exorl_pixel_agent_t* a, float* host, void* stream) {
    EXORL_REQUIRE(a && host, "pixel_agent_metrics: null argument");
    float buf[4 + EXORL_N_METRICS];
    EXORL_CHECK_HIP(hipMemcpyAsync(buf, a->stats, sizeof(buf), hipMemcpyDeviceToHost, as_stream(stream)));
    EXORL_CHECK_HIP(hipStreamSynchronize(as_stream(stream)));
    for (int i = 0; i < EXORL_N_METRICS; ++i) host[i] = buf[4 + i];
    host[EXORL_M_ACTOR_LOSS] = -buf[1] / (float)a->cfg.batch;
    return 0;
}

// act (ddpg.py:221-238) for one uint8 image (no augmentation): mean action (eval) or TruncatedNormal sample without clip
int exorl_pixel_agent_act(exorl_pixel_agent_t* a, const unsigned char* obs_dev, const float* meta_dev, float stddev, int32_t eval_mode,
                          const float* noise_dev, float* action_out_dev, void* stream) {
    EXORL_REQUIRE(a && obs_dev && action_out_dev && (a->cfg.meta_dim == 0 || meta_dev), "pixel_agent_act: null argument (meta_dim=%d needs meta_dev)",
                  a ? a->cfg.meta_dim : 0);
    hipStream_t s = as_stream(stream);
    const auto& c = a->cfg;
    const int A = c.act_dim, F = c.feature_dim, prec = c.precision;
    const int64_t img = (int64_t)c.c_in * c.hw * c.hw;
    float* ws = a->act_ws;
    float* x = ws; ws += round_up(img, 64);
    float* ews = ws;
    EXORL_TRY(exorl_u8_to_f32(obs_dev, img, x, s));           // act() sees the raw frame: no augmentation (ddpg.py:223-224)
    float* feat = nullptr;
    EXORL_TRY(exorl_encoder_forward_prec(a->flat[0][0], c.c_in, c.hw, x, 1, ews, &feat, a->cfg.precision, s));
    Mlp& pol0 = a->actor.head[0];
    if (act_fast_supported(1, F, c.hidden_dim, A) && F <= 256 && !(tune_variant() & 256)) {
        // two launches behind the encoder: the 39200-wide trunk with an in-launch combine + LayerNorm + tanh, then the policy
        // (Linear + ReLU recomputed per workgroup, four Linear(H, H) neurons per workgroup, head + tanh + TruncatedNormal draw in the last one)
        const float* Pa = a->flat[1][0];
        const PNet& n = a->actor;
        float* tpart = a->act_part;
        float* hpart = tpart + 256 * F;
        float* hvec = hpart + (int64_t)cdiv(c.hidden_dim, 4) * ACT_FAST_ROWS * 16;
        EXORL_TRY(trunk_one(feat, meta_dev, Pa + n.trunk.W, Pa + n.trunk.b, Pa + n.g, Pa + n.beta, tpart, a->act_ticket, hvec, a->R, c.meta_dim, F, s));
        EXORL_REQUIRE(eval_mode || stddev > 0.f, "pixel_agent_act: stddev must be > 0 in sampling mode");
        ActFast f{};
        f.x_dev = hvec; f.P = Pa; f.first_relu = 1; f.W0 = pol0.L[0].W; f.b0 = pol0.L[0].b;
        f.W1 = pol0.L[1].W; f.b1 = pol0.L[1].b; f.W2 = pol0.L[2].W; f.b2 = pol0.L[2].b;
        f.part = hpart; f.ticket = a->act_ticket + 1; f.noise_dev = noise_dev; f.seed = c.seed;
        f.counter = (eval_mode || noise_dev) ? 0ull : ((1ull << 63) | a->act_counter++);
        f.out = action_out_dev; f.stddev = stddev; f.rows = 1; f.in_dim = F; f.H = c.hidden_dim; f.nout = A; f.eval_mode = eval_mode;
        return act_fast(f, s);
    }
    // B = 1 reuses the batch-sized trunk / policy buffers (act() is never called inside update())
    EXORL_TRY(trunk_forward(a, a->actor, a->flat[1][0], feat, meta_dev, 1, a->ta_n, prec, s));
    Mlp& pol = a->actor.head[0];
    EXORL_TRY(mlp_forward(pol, a->flat[1][0], a->ta_n.h, F, 1, prec, s));
    EXORL_CHECK_HIP(hipMemcpyAsync(a->mu_n, pol.act[2], sizeof(float) * A, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(tanh_kernel, dim3(1), dim3(256), 0, s, a->mu_n, (int64_t)A);
    EXORL_LAUNCH_CHECK();
    if (eval_mode) {
        EXORL_CHECK_HIP(hipMemcpyAsync(action_out_dev, a->mu_n, sizeof(float) * A, hipMemcpyDeviceToDevice, s));
        return 0;
    }
    NoiseSpec nz{noise_dev, c.seed, (1ull << 63) | a->act_counter++, nullptr};
    return sample_action(a->mu_n, nz, stddev, 0.f, 0, action_out_dev, A, 1, A, nullptr, s);
}

}  // extern "C"
