// Internal C++ interface between the kernel translation units of libexorl_hip.so.
#pragma once
#include <vector>

#include "common.h"

namespace exorl {

struct GemmProblem {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    int M, N, K;
    int64_t lda, ldb, ldc;
};
int gemm_grouped(int precision, int a_layout, int b_layout, const GemmProblem* probs, int count, bool relu,
                 bool accumulate, hipStream_t s);

struct Gemm16Problem {
    const unsigned short* A;
    const unsigned short* B;
    float* C;
    const float* bias;
    int M, N, K;
    int64_t lda, ldb, ldc;
    // split-bf16 ("bf16x3") operands: x = hi + lo with hi = bf16(x), lo = bf16(x - hi); the product is formed as
    // hi*hi + hi*lo + lo*hi in fp32 accumulators (relative error ~2^-16: per-step losses stay within 1e-4 of the fp32 reference)
    const unsigned short* A_lo = nullptr;
    const unsigned short* B_lo = nullptr;
    // Scalar head folded into the epilogue (the 128 x TN "p" kernels only; forward launches with relu): instead of storing C the workgroup leaves,
    // per output row, the dot of relu(acc + bias) with head_w over each wave's column block: head_part[row][column block] (N / (TN / 2) floats per
    // row, gemm16_head_slots() says how many). For a net whose hidden activations nobody reads again — the Polyak target critic of a TD step.
    const float* head_w = nullptr;
    float* head_part = nullptr;
    // > 0 (the 128 x TN "p" kernels only; a multiple of 4): only columns [0, n_store) of C exist — N is the padded width of the operand planes.
    // The planes adapter uses it for outputs whose width is not a multiple of the tile (39200-wide module layers) instead of a padded C + copy.
    int n_store = 0;
};
// slots per row the "p" kernels would write for these problems' head_part (N / wave column block), or 0 when the launch would not take them
int gemm16_head_slots(const Gemm16Problem* probs, int count);
// operands stored as bf16 in memory (fast mode); same layout conventions as gemm_grouped
int gemm16_grouped(int a_layout, int b_layout, const Gemm16Problem* probs, int count, bool relu, bool accumulate, hipStream_t s);
// wgrad (a_layout 1) and dgrad (a_layout 0) problems of one backward pass in a single launch; b_layout 1, plain store
int gemm16_grouped_mixed(const int* a_layouts, const Gemm16Problem* probs, int count, hipStream_t s);

// ---- row-wise / column-wise layer kernels (rowops.hip). `nets` independent nets are processed by one
// launch (blockIdx.y); a* strides are in floats between nets for activations, p* for parameters.
int ln_tanh_fwd(const float* z, const float* gain, const float* beta, float* h, float* xhat, float* rstd,
                int rows, int H, int nets, int64_t astride, int64_t pstride, hipStream_t s);
int ln_tanh_bwd(const float* dh, const float* h, const float* xhat, const float* rstd, const float* gain, float* dz,
                int rows, int H, int nets, int64_t astride, int64_t pstride, hipStream_t s);
int ln_param_grad(const float* dh, const float* h, const float* xhat, float* dgain, float* dbeta,
                  int rows, int H, int nets, int64_t astride, int64_t pstride, hipStream_t s);
int colsum(const float* x, float* out, int rows, int cols, int nets, int64_t astride, int64_t pstride, hipStream_t s);
// fused ReLU backward + column sum (0 = done; 1 = shapes need relu_bwd + colsum separately)
int relu_bwd_colsum(float* x, const float* act, float* out, int rows, int cols, hipStream_t s);
// out = x W^T + b with the reduction cut into `splits` slabs in one grouped launch (scratch: splits x rows x F floats), summed in slab order
int linear_splitk(int prec, const float* x, int64_t ldx, const float* W, const float* b, float* out, int rows, int F, int K, float* scratch,
                  int splits, hipStream_t s);
int head_fwd(const float* a, const float* W, const float* b, float* out, int rows, int H, int nout, int tanh_out,
             int nets, int64_t astride, int64_t pstride, int64_t ostride, hipStream_t s);
int head_bwd_dx(const float* dout, const float* W, const float* a, float* dz, int rows, int H, int nout,
                int nets, int64_t astride, int64_t pstride, int64_t dstride, hipStream_t s);
int head_bwd_params(const float* dout, const float* a, const float* dz, float* dW, float* db_hidden, float* db_out,
                    int rows, int H, int nout, int nets, int64_t astride, int64_t pstride, int64_t dstride, hipStream_t s);

// ---- fused layer kernels (fused.hip)
int trunk_fwd(const float* x, int64_t ldx, const float* W0T, const float* b0, const float* gain, const float* beta,
              float* h, float* xhat, float* rstd, unsigned short* h_bf16, unsigned short* xhat_bf16, int rows, int in_dim,
              int H, int nets, int64_t astride, int64_t pstride, int64_t tstride, hipStream_t s);
// per-net operands of one trunk / scalar head, so that nets whose parameters live in different buffers (critic and its Polyak
// target) share a launch
// W0l / hl / xhl: lo planes of the split-bf16 mode (x = hi + lo); null in plain bf16 mode
struct TrunkItem { const float* x; const unsigned short* W0b; const float *b0, *gain, *beta; float* rstd; unsigned short *hb, *xhb;
                   const unsigned short* W0l; unsigned short *hl, *xhl; };
struct TrunkBatch { TrunkItem it[4]; };
int trunk_fwd16_batch(const TrunkBatch& tb, int count, int64_t ldx, int rows, int in_dim, int H, hipStream_t s);
struct HeadItem { const float* a; const float* W; const float* b; float* out; };
struct HeadBatch { HeadItem it[4]; };
int head_fwd1_batch(const HeadBatch& hb, int count, int rows, int H, hipStream_t s);
// MFMA variant for the bf16 fast mode (H % 128 == 0): W0b = bf16 shadow [nets][H][round_up(in_dim, 32)], zero padded
bool trunk_fwd16_supported(int H);
int trunk_fwd16(const float* x, int64_t ldx, const unsigned short* W0b, const float* b0, const float* gain, const float* beta, float* rstd,
                unsigned short* h_bf16, unsigned short* xhat_bf16, int rows, int in_dim, int H, int nets, int64_t astride, int64_t pstride,
                hipStream_t s, const unsigned short* W0l = nullptr, unsigned short* h_lo = nullptr, unsigned short* xhat_lo = nullptr);
int ln_bwd(float* dh, const float* h, const float* xhat, const unsigned short* h_bf16, const unsigned short* xhat_bf16,
           const float* rstd, const float* gain, float* P, int rows, int H, int nets, int64_t astride, int64_t pstride,
           int want_params, hipStream_t s, const float* w0t = nullptr, int64_t tstride = 0, float* dx = nullptr, int dx_cols = 0,
           const unsigned short* h_lo = nullptr, const unsigned short* xhat_lo = nullptr, const float* beta = nullptr);
int trunk_chunks(int rows);
// k smallest L2 distances of every src row to the tgt rows, ascending (knn.hip); d2 = scratch, n_src x round_up(n_tgt, 64) floats
int knn_topk(const float* src, int n_src, const float* tgt, int n_tgt, int dim, int k, float* out, float* d2, hipStream_t s);
int outer_reduce(const float* u, int64_t ldu, int J, const float* v, float* P, int rows, int H, int nets, int64_t vstride,
                 hipStream_t s);
int outer_chunks(int rows);
// Optional epilogue of the actor head on the stacked [next_obs; obs] batch: the two TruncatedNormal draws of a DDPG-family
// step (rows < B -> next_action into dst_next, rows >= B -> pi(obs) sample into dst_pi), td3_bc.py:125,151 — saves the
// separate sampling kernel. noise_* null -> Philox(seed, *counter_ptr + half).
struct SampleSpec {
    const float* stddev_ptr;   // device-resident exploration std (StepState::stddev); null -> the `stddev` value below
    const float *noise_c, *noise_a;
    uint64_t seed;
    const uint64_t* counter_ptr;
    float stddev, clip;
    float *dst_next, *dst_pi;
    int64_t dst_ld;
    int B;
};
int head_fwd4(const float* a, const float* W, const float* b, float* out, int rows, int H, int nout, int tanh_out,
              int nets, int64_t astride, int64_t pstride, int64_t ostride, hipStream_t s, const SampleSpec* sample = nullptr);
// Where head_bwd takes d(loss)/d(head output) from: a buffer, or computed on the fly from the loss definition so that
// the (B,1)/(B,A)-sized loss kernels need not sit between the forward and the backward pass.
#define EXORL_DOUT_BUFFER   0
#define EXORL_DOUT_TD       1   /* 2 (q_net - (r + D min(tq1,tq2))) * inv_bg                    td3_bc.py:127-131 */
#define EXORL_DOUT_ACTOR_Q  2   /* -lambda * inv_bg * [q_net is the min] (0.5 on ties)            td3_bc.py:152-155 */
#define EXORL_DOUT_CQL_ACTOR 4  /* tanh-Gaussian actor, 2A outputs: reparameterised (alpha*log_pi - Q).mean()   cql.py:236-255 */
#define EXORL_DOUT_ACTOR_MU 3   /* (sum_t da_t + bc term) * (1 - mu^2)  /  BC: -(a-mu)/std^2*inv_bg td3_bc.py:155, bc.py:83 */
struct DoutSpec {
    int mode;
    const float* buf;        // BUFFER: (nets, rows, nout)
    const float *q, *tq, *reward, *discount;   // TD / ACTOR_Q: q, tq are (2, rows)
    const float* stats;      // ACTOR_Q: stats[0] = sum |Q| over the global batch
    const float *da, *mu, *a_data;             // ACTOR_MU: da (da_nets, rows, nout), mu / a_data (rows, nout)
    const float* w;          // ACTOR_MU, CRR: per-sample advantage weight (rows)
    const float *raw, *z;    // CQL_ACTOR: actor head outputs (rows, 2A) [mu_raw | log_std_raw]; rsample noise (rows, A) or null
    const float* alpha_ptr;  // CQL_ACTOR: exp(log_actor_alpha) after its optimiser step (device scalar)
    uint64_t seed, counter; const uint64_t* counter_ptr;   // Philox identity of z when z == nullptr
    const float* task;       // TD / ACTOR_Q with a successor-feature critic (aps.py:50-58): dQ/d(feature j) = task[m][j]
    int64_t task_ld;
    const float* lam_parts;  // ACTOR_MU after qhead(ACTOR): per-chunk [sum |min Q|, sum min Q]; lambda = alpha / mean|Q| is applied here
    int lam_chunks;          //   (the critic backward is linear in its output gradient, so it ran with lambda = 1)
    int da_nets;
    int kind;                // EXORL_AGENT_*
    int use_lambda;
    float inv_bg, alpha, stddev;
    const float* stddev_ptr;   // BC / CRR: device-resident std (StepState::stddev); null -> `stddev`
};
int head_bwd(const DoutSpec& dspec, const float* W, const float* a, float* dz, unsigned short* dz_bf16, float* P, int rows,
             int H, int nout, int nets, int64_t astride, int64_t pstride, int want_params, hipStream_t s, unsigned short* dz_lo = nullptr);
int head_chunks(int rows);
int tune_variant();      // exorl_gemm_tune's experiment bits (0 = defaults)
int prec_override_mask();    // exorl_debug_precision_override's bits (0 = none; diagnostic)
// Scalar critic heads, forward and backward in one kernel (single-GPU whole-step path, no metrics): Q1,Q2 (and the target's
// Q1',Q2') row dots, the loss gradient at the head output, dz2 = dQ * W2 * [h2 > 0] and the per-chunk parameter partials.
//   mode 0 (critic step, td3_bc.py:126-131): dQ_n = 2 (Q_n - (r + D min(Q1',Q2'))) * inv_bg
//   mode 1 (actor step, td3_bc.py:152-155):  dQ_n = -inv_bg * [Q_n is the min] (0.5 on ties); per-chunk sum|min Q|, sum min Q -> abs_part
struct QHeadArgs {
    const float* a[4];       // h2 rows of critic net 0, 1 and (mode 0) target net 0, 1
    const float* W[4];
    const float* b[4];
    float *q, *tq;           // (2, rows) outputs
    const float *reward, *discount;
    float* dz; unsigned short* dzb; unsigned short* dzl; int64_t act;     // dzl: lo plane (split-bf16) or null
    float* P;                // head partials [net][chunk][(1+1)H + 16] or null
    float* abs_part;         // [chunk][2]
    const float* tpart[2];   // mode 0, folded target heads: per row `tslots` partial dots of target net 0 / 1 (gemm16 head_part) instead of a[2], a[3]
    int tslots;
    int rows, H, mode;
    float inv_bg;
};
int qhead(const QHeadArgs& q, hipStream_t s);
int qhead_chunks(int rows);     // partial rows qhead writes per net (finer than head_chunks)
struct FinalizeArgs {
    const float* Ph; int head_chunks; int n_heads; int64_t head_stride;    // head partials; stride between heads in G
    int64_t gW2, gb1, gb2;                                                 // offsets of head 0's tensors in G
    const float* Pt; int trunk_chunks; int n_trunks; int64_t trunk_stride; // LN/bias column-sum partials
    const float* Pw; int w_chunks;                                         // first-layer weight partials [k][c]
    int64_t gW0, gb0, gg, gbeta;
    int H, nout, in_dim;
    float* G;
};
int finalize_grads(const FinalizeArgs& f, hipStream_t s);

// Products and sums are rounded separately (no FMA contraction): the op order of torch's CPU kernels.
__device__ __forceinline__ float polyak(float p, float t, float tau, float one_minus_tau) {
#pragma clang fp contract(off)
    const float a = tau * p;
    const float b = one_minus_tau * t;
    return a + b;
}
struct AdamConst;
// One launch for "sum the per-workgroup gradient partials" and the optimiser step (single-GPU steps: nothing has to be
// exchanged between the two): the first blocks reduce the partials of every tensor except the H x H weights and step those
// elements in place (Adam, Polyak target, derived copies), the remaining blocks stream the H x H weights' Adam update.
struct FusedAdamArgs {
    float *p, *g, *m, *v, *target;
    const AdamConst* c;                // device
    int n_heads; int64_t w1_off[2];    // flat offsets of the H x H weights
    unsigned long long* bump;
};
struct ShadowSpec;
int finalize_adam(const FinalizeArgs& f, const FusedAdamArgs& a, const ShadowSpec& sh, hipStream_t s, int part = 0);

// Derived copies of a net's weights that the kernels read: W0T[in][H] per trunk (coalesced first-layer reads)
// and, in bf16 mode, W1 as bf16 per head (MFMA operand). Kept current by the Adam kernel itself.
struct ShadowSpec {
    int n_trunks, n_heads, in_dim, H;
    int64_t w0_off[2], w1_off[2];      // flat offsets of W0 (per trunk) and W1 (per head)
    float* w0t;                        // [n_trunks][in][H]
    unsigned short* w1b;               // [n_heads][H][H] bf16, or nullptr
    unsigned short* w0b;               // [n_trunks][H][round_up(in, 32)] bf16 (zero padded), or nullptr
    float* t_w0t;                      // same for the Polyak target (nullptr if none)
    unsigned short* t_w1b;
    unsigned short* t_w0b;
    unsigned short *w1l, *w0l, *t_w1l, *t_w0l;     // lo planes of the bf16 copies (split-bf16 mode), same shapes; or nullptr
};
int refresh_shadows(const float* p, int64_t n, const ShadowSpec& sh, bool target, hipStream_t s);

// ---- loss / sampling kernels (loss.hip)
struct NoiseSpec {           // where TruncatedNormal noise comes from
    const float* buf;        // (B,A) standard normal draws, or nullptr -> Philox(seed, counter)
    uint64_t seed;
    uint64_t counter;              // draw id = counter + (counter_ptr ? *counter_ptr : 0)
    const uint64_t* counter_ptr;   // device-resident base (advanced by step_begin) so a captured graph replays fresh draws
};

struct AdamConst {           // fp32 scalars of one torch.optim.Adam step (bias corrections folded in)
    float one_minus_b1, b2, one_minus_b2, bc2_sqrt, eps, neg_step_size, tau, one_minus_tau;
};

__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, const AdamConst& c) {
#pragma clang fp contract(off)
    // exp_avg.lerp_(g, 1-b1); exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2); p.addcdiv_(m, sqrt(v)/bc2_sqrt + eps, -lr/bc1)
    m = m + c.one_minus_b1 * (g - m);
    v = v * c.b2 + (c.one_minus_b2 * g) * g;
    const float denom = sqrtf(v) / c.bc2_sqrt + c.eps;
    p = p + (c.neg_step_size * m) / denom;
}

// Device-resident per-agent step state: everything that changes from one update() to the next, so the whole
// step can be captured once in a hipGraph and replayed with no host-side argument patching.
struct StepState {
    uint64_t replay_counter;     // Philox batch counter of the sampler bound to the graph
    uint64_t noise_counter;      // Philox draw counter for action noise (2 draws per step)
    long long t_actor, t_critic; // Adam step counts
    double b1t, b2t_actor_unused, b1t_c, b2t_c;   // running beta^t products: [actor b1^t, (pad), critic b1^t, critic b2^t]
    double b2t;                  // actor b2^t
    AdamConst actor, critic;
    double lr, b1, b2, eps, tau;  // the Python-float hyper-parameters torch.optim.Adam holds (dec7() of the fp32 ABI values)
    int has_critic;
    float stddev;                // exploration std of the step (utils.schedule value): read through a pointer, so a schedule
                                 // that moves every step needs no re-capture of the hipGraph
};
int step_begin(StepState* st, int advance_replay, hipStream_t s);
// act() for up to ACT_FAST_ROWS observation rows in one launch (loss.hip, act_fast_kernel): trunk + LayerNorm + tanh + Linear(H,H) + ReLU +
// head + tanh + TruncatedNormal draw. x_host / noise_host (host pointers) travel as kernel arguments; `out` may be pinned host memory.
constexpr int ACT_FAST_ROWS = 2;            // x 256 floats of embedded observation: the kernel-argument block stays under 4 KB
struct ActFast {
    const float *x_dev, *x_host;            // one of the two
    const float *w0t, *P;
    int64_t b0, g, beta, W1, b1, W2, b2;
    float* part; unsigned int* ticket;      // scratch: cdiv(H, 4) * rows * nout floats; one zero-initialised word
    const float *noise_dev, *noise_host;    // at most one; both null -> Philox(seed, counter)
    uint64_t seed, counter;
    float* out;
    float stddev;
    int rows, in_dim, H, nout, eval_mode;
    int first_relu = 0; int64_t W0 = 0;     // pixel policy: the first layer is Linear(in_dim, H) + ReLU with torch-layout weights at offset W0 of P
                                            // (no W0T shadow, no LayerNorm); b0 is its bias
};
bool act_fast_supported(int rows, int in_dim, int H, int nout);
int act_fast(const ActFast& f, hipStream_t s);
// pixel act(): tanh(LayerNorm([x | meta] W^T + b)) for one encoding, split over 256 workgroups with an in-launch combine (loss.hip)
int trunk_one(const float* x, const float* meta, const float* W, const float* b, const float* gain, const float* beta, float* part, unsigned int* ticket,
              float* out, int R, int M, int F, hipStream_t s);
// out[0] = sum_i parts[2i], out[1] = sum_i parts[2i+1] in a fixed order (qhead's per-chunk [sum |min Q|, sum min Q] -> the 4-float
// statistics buffer that is all-reduced under data parallelism)
int reduce_pairs(const float* parts, int chunks, float* out, hipStream_t s);
int set_device_float(float* dst, float value, hipStream_t s);

__host__ __device__ inline void fill_adam_const(AdamConst& c, double b1t, double b2t, double lr, double b1, double b2, double eps, double tau) {
    // double-precision scalar math, as torch's _single_tensor_adam does in Python floats; beta^t comes from a running
    // product kept in the step state (one multiply per step instead of two software pow() calls on the critical path)
    const double bc1 = 1.0 - b1t;
    const double bc2 = 1.0 - b2t;
    c.one_minus_b1 = (float)(1.0 - b1);
    c.b2 = (float)b2;
    c.one_minus_b2 = (float)(1.0 - b2);
    c.bc2_sqrt = (float)sqrt(bc2);
    c.eps = (float)eps;
    c.neg_step_size = (float)(-(lr / bc1));
    c.tau = (float)tau;
    c.one_minus_tau = (float)(1.0 - tau);
}

// One thread: advance the counters and pre-compute both optimisers' scalars for this step.
__device__ inline void step_begin_device(StepState* st, int advance_replay) {
    if (advance_replay) st->replay_counter += 1;
    st->noise_counter += 2;
    st->t_actor += 1;
    st->b1t *= st->b1;
    st->b2t *= st->b2;
    fill_adam_const(st->actor, st->b1t, st->b2t, st->lr, st->b1, st->b2, st->eps, 0.0);
    if (st->has_critic) {
        st->t_critic += 1;
        st->b1t_c *= st->b1;
        st->b2t_c *= st->b2;
        fill_adam_const(st->critic, st->b1t_c, st->b2t_c, st->lr, st->b1, st->b2, st->eps, st->tau);
    }
}


int prepare_inputs(const float* obs, const float* action, const float* next_obs, float* xa, float* xc_cur,
                   float* xc_next, float* xc_pi, int B, int O, int A, int has_critic, StepState* st, int advance_replay,
                   hipStream_t s);
// both TruncatedNormal draws of a DDPG-family step in one launch: rows 0..B-1 of mu2 -> next_action (draw 0),
// rows B..2B-1 -> pi(obs) sample (draw 1)
int sample_actions2(const float* mu2, const float* noise_c, const float* noise_a, uint64_t seed, const uint64_t* counter_ptr,
                    float stddev, float clip, float* dst_next, float* dst_pi, int64_t dst_ld, int B, int A, hipStream_t s,
                    const float* stddev_ptr = nullptr);
int sample_action(const float* mu, NoiseSpec noise, float stddev, float clip, int use_clip, float* dst, int64_t dst_ld,
                  int B, int A, float* logprob_sum, hipStream_t s, const float* stddev_ptr = nullptr, int world_size = 1);
// ---- CQL (cql.py:152-263)
struct CqlNoise {            // the five draws of one CQL update; null buffers -> Philox(seed, *counter_ptr + k)
    const float *z_next, *u_rand, *z_cur, *z_nxt, *z_actor;
    uint64_t seed;
    const uint64_t* counter_ptr;
};
struct CqlScalars {          // device-resident entropy-temperature state (cql.py:96-101,241-247)
    float log_alpha, m, v, alpha;
};
// raw2: actor head outputs on [next_obs; obs] (2B, 2A). Writes next_action into xc_next[:, O:] and the (3n+1)B critic
// rows x_all = [obs | rand] (nB) ; [obs | pi(obs) samples] (nB) ; [obs | pi(next_obs) samples] (nB) ; [obs | a_data] (B)
int cql_build_inputs(const float* obs, const float* action, const float* raw2, CqlNoise nz, float* xc_next, float* x_all, int B,
                     int O, int A, int n, hipStream_t s);
// per-row d(critic_loss)/dQ over the (3n+1)B rows of both nets + metrics (TD loss, logsumexp penalty)
// lag != nullptr: the penalty weight is exp(log_critic_alpha), stepped by Adam inside the kernel before it is used (cql.py:201-213)
int cql_critic_dq(const float* q_all, const float* tq, const float* reward, const float* discount, float* dq_all, float* metrics,
                  int B, int n, float cql_alpha, float inv_bg, hipStream_t s, CqlScalars* lag = nullptr, const AdamConst* c_dev = nullptr,
                  float target_penalty = 0.f,
                  int mode = 0, float* gsum = nullptr);
// rsample of pi(obs): y = tanh(mu + std z) -> xc_pi[:, O:]; stats[0] = sum log_pi (per element, cql.py:239)
int cql_actor_sample(const float* raw_obs, CqlNoise nz, float* xc_pi, int64_t ld, float* stats, int B, int O, int A, hipStream_t s);
// scalar Adam step on log_actor_alpha from the (all-reduced) sum of log_pi; writes alpha = exp(log_alpha) and metrics
int cql_alpha_step(CqlScalars* sc, const float* stats, const AdamConst* c_dev, float* metrics, int B, int A, float inv_bg, const float* q,
                   hipStream_t s);
// policy output for act(): tanh(mu) (eval) or tanh(mu + std z)
int cql_act(const float* raw, const float* noise, uint64_t seed, uint64_t counter, int eval_mode, float* out, int rows, int A, hipStream_t s);
int head_bwd_wide(const DoutSpec& dspec, const float* W, const float* a, float* dz, unsigned short* dz_bf16, float* P, int rows, int H,
                  int nout, int64_t astride, int64_t pstride, int want_params, hipStream_t s, unsigned short* dz_lo = nullptr);

// CRR (crr.py:121-142): xc_rep[(b*n+i)] = [obs_b | TruncatedNormal(mu_b).sample(clip)] for i < n
int repeat_sample(const float* obs, const float* mu, const float* noise, uint64_t seed, const uint64_t* counter_ptr, uint64_t counter,
                  float stddev, float clip, float* xc_rep, int B, int O, int A, int n, hipStream_t s, const float* stddev_ptr = nullptr);
// w_b = f(min(Q1,Q2)(s_b, a_b) - mean_i min(Q1,Q2)(s_b, a_bi))
int crr_weights(const float* q_rep, const float* q_data, float* w, int B, int n, int weight_func, hipStream_t s);
int critic_loss(const float* q, const float* tq, const float* reward, const float* discount, float* dq,
                float* metrics, int B, float inv_bg, hipStream_t s);
int actor_stats(const float* q, float* stats, int B, hipStream_t s);
// successor-feature critic: q[net][m] = sum_j task[m][j] * feat[net][m][j]   (aps.py:55-58)
int sf_q(const float* feat, const float* task, int64_t task_ld, float* q, int B, int sf, int nets, hipStream_t s);
int actor_dq(const float* q, const float* stats, float* dq, int B, float inv_bg, float alpha, int use_lambda,
             hipStream_t s);
int actor_dmu(const float* da, int64_t da_ld, int da_nets, int64_t da_net_stride, const float* mu, const float* a_data, const float* reward, const float* w, float* dpre, float* stats,
              float* metrics, int B, int A, float inv_bg, float alpha, int kind, float stddev, hipStream_t s, const float* stddev_ptr = nullptr);

// ---- optimiser (optim.hip)
int adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
              int64_t t, float* target, float tau, hipStream_t s);
int soft_update(const float* p, float* target, int64_t n, float tau, hipStream_t s);
// Adam with the scalars read from device memory (StepState), for graph-replayable steps.
// bump != nullptr: block 0 also increments *bump (the replay counter of a captured step: nothing later in the step reads it)
int adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const AdamConst* c_dev, float* target,
                  const ShadowSpec* shadows, hipStream_t s, uint64_t* bump = nullptr);

// ---- plain Linear/ReLU stacks on the generic grouped GEMM (intr.hip); shared by the intrinsic modules and the pixel agent
struct Lin { int in, out; int64_t W, b; };           // offsets into a flat parameter buffer, torch layout W[out][in]
struct Mlp {                                         // Linear-ReLU-...-Linear; the last layer's output is raw unless relu_last
    std::vector<Lin> L;
    bool relu_last = false;
    std::vector<float*> act, dact;                   // per layer: (rows, out) activations and their gradients
};
int mlp_forward(const Mlp& m, const float* P, const float* x, int64_t ldx, int rows, int prec, hipStream_t s);
// dact[last] holds d(loss)/d(output); writes parameter gradients into G and, if dx, d(loss)/d(input) (rows, in0)
int mlp_backward(const Mlp& m, const float* P, float* G, const float* x, int64_t ldx, int rows, float* dx, int prec, hipStream_t s);
// n nets of identical shapes (twin Q heads, Disagreement's ensemble) layer by layer in shared launches; dx (rows, in0) receives the SUM of the nets' d/d(input)
int mlp_forward_many(const Mlp* nets, int n, const float* P, const float* x, int64_t ldx, int rows, int prec, hipStream_t s);
int mlp_backward_many(const Mlp* nets, int n, const float* P, float* G, const float* x, int64_t ldx, int rows, int prec, hipStream_t s, float* dx);
int launch_concat(const float* a, int64_t lda, int ca, const float* b, int64_t ldb, int cb, float* dst, int rows, hipStream_t s);

// ---- replay (replay.hip) internal entry points used by the agent's graph capture
// Optional fan-out of the sampled rows into the agent's staged network inputs (what prepare_inputs would do in a second
// kernel): xa = [next_obs ; obs], xc_cur = [obs | action], xc_next[:, :O] = next_obs, xc_pi[:, :O] = obs. fp32 state
// observations only. st != nullptr: thread 0 also runs step_begin_device(st, 0).
struct StageOut {
    float *xa, *xc_cur, *xc_next, *xc_pi;
    int O, A, B, has_critic;
    StepState* st;
};
int replay_sample_impl(exorl_replay* r, int32_t batch, int32_t nstep, float gamma, int32_t sampler,
                       const int32_t* pairs_host, const exorl_batch_out* out, int32_t* pairs_out_host, hipStream_t s,
                       const uint64_t* dev_counter, const StageOut* stage = nullptr);
int replay_obs_bytes(exorl_replay* r);
// comm.cpp (exorl_comm is the C ABI's opaque struct, declared at global scope in exorl_hip.h)
int comm_allreduce_sum(exorl_comm* c, float* buf, int64_t n, hipStream_t s);
int comm_nranks(const exorl_comm* c);

uint64_t replay_philox_counter(exorl_replay* r);
// what a sample call does before its launch, without drawing: episode table upload, pair buffer, nstep vs the shortest episode
int replay_prepare(exorl_replay* r, int32_t batch, int32_t nstep, hipStream_t s);
void replay_advance_philox(exorl_replay* r, uint64_t n);

}  // namespace exorl
