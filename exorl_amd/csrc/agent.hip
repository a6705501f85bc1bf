// Agent update orchestration: TD3+BC, TD3, BC and the DDPG (states) backbone as one stream-ordered chain of
// the kernels in gemm.hip / rowops.hip / loss.hip / optim.hip.
// Replaces (file:line in the reference repo):
//   td3_bc.py:119-189 update_critic/update_actor/update    td3.py:117-186    bc.py:78-110
//   unsupervised_learning/ddpg.py:240-328
// Structure exploited (what the reference's autograd graph hides):
//   * the actor is evaluated on next_obs (critic target, td3_bc.py:124) and on obs (actor loss, :149) with the
//     SAME weights -> one stacked 2B-row forward;
//   * the twin critics / twin heads are independent nets -> every layer is a 2-problem grouped GEMM;
//   * the actor step needs only the critic's dgrad, and of dX only the action columns (the reference also
//     computes and discards the critic wgrad there — SURVEY 8d);
//   * Polyak averaging of the target is fused into the critic's Adam pass.
// Data parallel: the step is split in 4 phases at the three points where a global batch quantity is needed
// (critic grads, sum|Q| for lambda (td3_bc.py:154), actor grads); the caller all-reduces between phases.
#include <cmath>
#include <vector>

#include "kernels.h"

namespace exorl {

struct TensorDesc { int64_t off, rows, cols; };

// A net = n_trunks x [Linear(in,H) LN Tanh] feeding n_heads x [Linear(H,H) ReLU Linear(H,out)].
// twin critic: 2 trunks, 2 heads (head i on trunk i); shared critic: 1 trunk, 2 heads; actor: 1, 1.
struct NetDesc {
    int in_dim = 0, out_dim = 0, H = 0, n_trunks = 0, n_heads = 0;
    int64_t W0 = 0, b0 = 0, g = 0, beta = 0, trunk_stride = 0;   // offsets of trunk 0's tensors; stride to trunk 1
    int64_t W1 = 0, b1 = 0, W2 = 0, b2 = 0, head_stride = 0;     // offsets of head 0's tensors; stride to head 1
    int64_t total = 0;                                           // padded flat size (floats)
    std::vector<TensorDesc> tensors;                             // reference parameters() order
};

static int64_t pad4(int64_t n) { return round_up(n, 4); }

static NetDesc make_net(int in_dim, int out_dim, int H, int n_trunks, int n_heads) {
    NetDesc d;
    d.in_dim = in_dim; d.out_dim = out_dim; d.H = H; d.n_trunks = n_trunks; d.n_heads = n_heads;
    int64_t off = 0;
    auto add = [&](int64_t rows, int64_t cols) { const int64_t o = off; d.tensors.push_back({o, rows, cols}); off += pad4(rows * cols); return o; };
    auto add_trunk = [&](bool first) {
        const int64_t w = add(H, in_dim), b = add(H, 1), gg = add(H, 1), be = add(H, 1);
        if (first) { d.W0 = w; d.b0 = b; d.g = gg; d.beta = be; }
        return w;
    };
    auto add_head = [&](bool first) {
        const int64_t w1 = add(H, H), bb1 = add(H, 1), w2 = add(out_dim, H), bb2 = add(out_dim, 1);
        if (first) { d.W1 = w1; d.b1 = bb1; d.W2 = w2; d.b2 = bb2; }
        return w1;
    };
    if (n_trunks == n_heads) {             // [trunk0 head0][trunk1 head1]
        int64_t t0 = 0, h0 = 0;
        for (int i = 0; i < n_trunks; ++i) {
            const int64_t t = add_trunk(i == 0);
            const int64_t h = add_head(i == 0);
            if (i == 0) { t0 = t; h0 = h; }
            if (i == 1) { d.trunk_stride = t - t0; d.head_stride = h - h0; }
        }
    } else {                               // [trunk][head0][head1]
        add_trunk(true);
        int64_t h0 = 0;
        for (int i = 0; i < n_heads; ++i) {
            const int64_t h = add_head(i == 0);
            if (i == 0) h0 = h;
            if (i == 1) d.head_stride = h - h0;
        }
    }
    d.total = round_up(off, 64);
    return d;
}

// Second stream for independent branches of the step (target forward || critic forward, wgrad || dgrad chain).
// Used while capturing: the hipGraph then has parallel branches, so kernels that fill only part of the 256 CUs
// (64x64-tile GEMMs, row kernels) overlap instead of queueing behind each other.
struct Fork {
    hipStream_t aux = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool on = false;
    int fork(hipStream_t s) const {            // aux continues after everything enqueued on s so far
        if (!on) return 0;
        EXORL_CHECK_HIP(hipEventRecord(ev_fork, s));
        EXORL_CHECK_HIP(hipStreamWaitEvent(aux, ev_fork, 0));
        return 0;
    }
    int join(hipStream_t s) const {            // s waits for everything enqueued on aux
        if (!on) return 0;
        EXORL_CHECK_HIP(hipEventRecord(ev_join, aux));
        EXORL_CHECK_HIP(hipStreamWaitEvent(s, ev_join, 0));
        return 0;
    }
    hipStream_t side(hipStream_t s) const { return on ? aux : s; }
};

// *l: lo planes of the split-bf16 mode (same shapes as the bf16 buffers; x = hi + lo), null otherwise
struct FwdBufs { float *h1, *xhat, *rstd, *h2, *out; unsigned short *h1b, *xhatb, *h1l, *xhatl; };
struct BwdBufs { float *dz2, *dh1; unsigned short *dz2b, *dz2l; };
struct NetShadow { float* w0t; unsigned short* w1b; unsigned short* w0b; unsigned short *w1l, *w0l; };   // W0 transposed per trunk; W1 as bf16 per head; W0 as K-padded bf16
// bf16 activation pipeline: plain bf16 mode, or split-bf16 with hi/lo planes (H, batch multiples of 64: see planes_ok)
static bool fast16(int prec, const NetShadow& sh) { return prec == EXORL_PREC_BF16 || (prec == EXORL_PREC_BF16X3 && sh.w1l); }
static Gemm16Problem g16(const unsigned short* A, const unsigned short* Al, const unsigned short* B, const unsigned short* Bl, int64_t aoff,
                         int64_t boff, float* C, const float* bias, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc) {
    Gemm16Problem p{A + aoff, B + boff, C, bias, M, N, K, lda, ldb, ldc};
    if (Al && Bl) { p.A_lo = Al + aoff; p.B_lo = Bl + boff; }
    return p;
}
struct Partials { float *Ph, *Pt, *Pw; };                  // per-chunk partial gradients (fused.hip)
// Adam on the H x H weights launched on a side stream as soon as their gradient exists (right behind the wgrad GEMM): it is the HBM-bound
// 80 % of the optimiser pass and depends on nothing the LayerNorm-backward -> first-layer-wgrad chain produces, so the two run side by side
// (graph: parallel branches) and the optimiser launch that follows the chain only handles the small tensors.
struct EarlyW1 { const Fork* fo; FusedAdamArgs fa; ShadowSpec sh; bool* launched; };

static ShadowSpec shadow_spec(const NetDesc& d, const NetShadow& sh, const NetShadow* target) {
    ShadowSpec s{};
    s.n_trunks = d.n_trunks; s.n_heads = d.n_heads; s.in_dim = d.in_dim; s.H = d.H;
    for (int t = 0; t < d.n_trunks; ++t) s.w0_off[t] = d.W0 + t * d.trunk_stride;
    for (int i = 0; i < d.n_heads; ++i) s.w1_off[i] = d.W1 + i * d.head_stride;
    s.w0t = sh.w0t; s.w1b = sh.w1b; s.w0b = sh.w0b; s.w1l = sh.w1l; s.w0l = sh.w0l;
    s.t_w0t = target ? target->w0t : nullptr;
    s.t_w1b = target ? target->w1b : nullptr;
    s.t_w0b = target ? target->w0b : nullptr;
    s.t_w1l = target ? target->w1l : nullptr;
    s.t_w0l = target ? target->w0l : nullptr;
    return s;
}

static int net_forward(const NetDesc& d, const float* P, const NetShadow& sh, const float* x, int64_t ldx, int rows,
                       const FwdBufs& f, bool save, bool tanh_out, int prec, hipStream_t s, const SampleSpec* sample = nullptr,
                       bool no_head = false) {
    const int H = d.H;
    const int64_t act = (int64_t)rows * H;
    const bool bf = fast16(prec, sh);
    // fast mode keeps the trunk activations as bf16 only (MFMA operand + LN backward input): 4 B/elem written instead of 10
    if (bf && sh.w0b && trunk_fwd16_supported(H))
        EXORL_TRY(trunk_fwd16(x, ldx, sh.w0b, P + d.b0, P + d.g, P + d.beta, save ? f.rstd : nullptr, f.h1b, save ? f.xhatb : nullptr, rows,
                              d.in_dim, H, d.n_trunks, act, d.trunk_stride, s, sh.w0l, f.h1l, f.xhatl));
    else
        EXORL_TRY(trunk_fwd(x, ldx, sh.w0t, P + d.b0, P + d.g, P + d.beta, bf ? nullptr : f.h1, (save && !bf) ? f.xhat : nullptr,
                            save ? f.rstd : nullptr, bf ? f.h1b : nullptr, (bf && save) ? f.xhatb : nullptr, rows, d.in_dim, H,
                            d.n_trunks, act, d.trunk_stride, (int64_t)d.in_dim * H, s));
    if (bf) {
        Gemm16Problem q[2];
        for (int i = 0; i < d.n_heads; ++i)
            q[i] = g16(f.h1b, f.h1l, sh.w1b, sh.w1l, (d.n_trunks == d.n_heads ? i : 0) * act, (int64_t)i * H * H, f.h2 + i * act,
                       P + d.b1 + i * d.head_stride, rows, H, H, H, H, H);
        EXORL_TRY(gemm16_grouped(0, 0, q, d.n_heads, true, false, s));
    } else {
        GemmProblem p[2];
        for (int i = 0; i < d.n_heads; ++i)
            p[i] = GemmProblem{f.h1 + (d.n_trunks == d.n_heads ? i : 0) * act, P + d.W1 + i * d.head_stride, f.h2 + i * act,
                               P + d.b1 + i * d.head_stride, rows, H, H, H, H, H};
        EXORL_TRY(gemm_grouped(prec, 0, 0, p, d.n_heads, true, false, s));
    }
    if (no_head) return 0;                       // the caller runs the fused head forward+backward (qhead)
    if (H % 4 == 0)
        EXORL_TRY(head_fwd4(f.h2, P + d.W2, P + d.b2, f.out, rows, H, d.out_dim, tanh_out ? 1 : 0, d.n_heads, act, d.head_stride,
                            (int64_t)rows * d.out_dim, s, sample));
    else
        EXORL_TRY(head_fwd(f.h2, P + d.W2, P + d.b2, f.out, rows, H, d.out_dim, tanh_out ? 1 : 0, d.n_heads, act, d.head_stride,
                           (int64_t)rows * d.out_dim, s));
    return 0;
}

// The critic on (obs, a_data) and its Polyak target on (next_obs, next_action) in shared launches (bf16 mode): the two forward
// chains are independent, same shapes, different weights -> 3 launches instead of 6, each filling the chip twice as deep.
static bool forward2_supported(const NetDesc& d, int prec, const NetShadow& sa, const NetShadow& sb) {
    return fast16(prec, sa) && fast16(prec, sb) && sa.w0b && sb.w0b && trunk_fwd16_supported(d.H) && d.out_dim == 1 && d.n_heads == 2 && d.H % 4 == 0;
}
// fold_a (with no_head): net A's scalar heads are folded into the GEMM epilogue (Gemm16Problem::head_part) when the launch takes the 128 x TN
// kernels — A's h2 buffer then holds, per head, rows x *slots_a partial dots instead of hidden activations (qhead sums them)
static int net_forward2(const NetDesc& d, const float* Pa, const NetShadow& sa, const float* xa, const FwdBufs& fa, bool save_a,
                        const float* Pb, const NetShadow& sb, const float* xb, const FwdBufs& fb, bool save_b, int64_t ldx, int rows,
                        hipStream_t s, bool no_head = false, bool fold_a = false, int* slots_a = nullptr) {
    const int H = d.H;
    const int64_t act = (int64_t)rows * H, wst = (int64_t)H * round_up(d.in_dim, 32);
    const float* P[2] = {Pa, Pb};
    const NetShadow* sh[2] = {&sa, &sb};
    const float* x[2] = {xa, xb};
    const FwdBufs* f[2] = {&fa, &fb};
    const bool save[2] = {save_a, save_b};
    TrunkBatch tb{};
    int nt = 0;
    for (int k = 0; k < 2; ++k)
        for (int t = 0; t < d.n_trunks; ++t)
            tb.it[nt++] = TrunkItem{x[k], sh[k]->w0b + t * wst, P[k] + d.b0 + t * d.trunk_stride, P[k] + d.g + t * d.trunk_stride,
                                    P[k] + d.beta + t * d.trunk_stride, save[k] ? f[k]->rstd + (int64_t)t * rows : nullptr,
                                    f[k]->h1b + t * act, save[k] ? f[k]->xhatb + t * act : nullptr, sh[k]->w0l ? sh[k]->w0l + t * wst : nullptr,
                                    f[k]->h1l ? f[k]->h1l + t * act : nullptr, (save[k] && f[k]->xhatl) ? f[k]->xhatl + t * act : nullptr};
    EXORL_TRY(trunk_fwd16_batch(tb, nt, ldx, rows, d.in_dim, H, s));
    Gemm16Problem q[4];
    HeadBatch hb{};
    int nq = 0;
    for (int k = 0; k < 2; ++k)
        for (int i = 0; i < d.n_heads; ++i) {
            q[nq] = g16(f[k]->h1b, f[k]->h1l, sh[k]->w1b, sh[k]->w1l, (d.n_trunks == d.n_heads ? i : 0) * act, (int64_t)i * H * H, f[k]->h2 + i * act,
                        P[k] + d.b1 + i * d.head_stride, rows, H, H, H, H, H);
            hb.it[nq] = HeadItem{f[k]->h2 + i * act, P[k] + d.W2 + i * d.head_stride, P[k] + d.b2 + i * d.head_stride, f[k]->out + (int64_t)i * rows};
            ++nq;
        }
    if (slots_a) *slots_a = 0;
    if (no_head && fold_a && slots_a && !(tune_variant() & 1024)) {     // exorl_gemm_tune bit 1024: keep the target's hidden activations (A/B)
        const int slots = gemm16_head_slots(q, nq);
        if (slots > 0 && (int64_t)slots <= H) {
            for (int i = 0; i < d.n_heads; ++i) {
                q[i].head_w = P[0] + d.W2 + i * d.head_stride;
                q[i].head_part = f[0]->h2 + i * act;
            }
            *slots_a = slots;
        }
    }
    EXORL_TRY(gemm16_grouped(0, 0, q, nq, true, false, s));
    if (no_head) return 0;
    return head_fwd1_batch(hb, nq, rows, H, s);
}

// G == nullptr: dgrad only (no parameter gradients). dx (n_trunks x rows x dx_cols, one slab per trunk — the
// consumer adds them) receives d/dx[:, col0:col0+dx_cols].
static int net_backward(const NetDesc& d, const float* P, const NetShadow& sh, float* G, const Partials& pt, const float* x,
                        int64_t ldx, int rows, const FwdBufs& f, const DoutSpec& dout, const BwdBufs& b, float* dx, int dx_col0,
                        int dx_cols, int prec, hipStream_t s, const Fork& fk, FinalizeArgs* defer = nullptr, bool have_dz2 = false,
                        const EarlyW1* early = nullptr) {
    const int H = d.H;
    const int64_t act = (int64_t)rows * H;
    const bool paired = d.n_trunks == d.n_heads;
    const bool bf = fast16(prec, sh);
    if (have_dz2) {
        // dz2 and the head partials were produced by qhead
    } else if (d.out_dim > 16) {
        EXORL_REQUIRE(d.n_heads == 1, "net_backward: wide heads are single-net only");
        EXORL_TRY(head_bwd_wide(dout, P + d.W2, f.h2, bf ? nullptr : b.dz2, bf ? b.dz2b : nullptr, G ? pt.Ph : nullptr, rows, H, d.out_dim,
                                act, d.head_stride, G ? 1 : 0, s, bf ? b.dz2l : nullptr));
    } else {
        EXORL_TRY(head_bwd(dout, P + d.W2, f.h2, bf ? nullptr : b.dz2, bf ? b.dz2b : nullptr, G ? pt.Ph : nullptr, rows, H, d.out_dim,
                           d.n_heads, act, d.head_stride, G ? 1 : 0, s, bf ? b.dz2l : nullptr));
    }
    if (bf) {
        Gemm16Problem q[4];
        int at[4];
        int nq = 0;
        if (G && !fk.on) {                           // wgrad + dgrad in one launch (independent readers of dz2)
            for (int i = 0; i < d.n_heads; ++i) {    // dW1_i[n][k] = sum_m dz2_i[m][n] h1[m][k]
                at[nq] = 1;
                q[nq++] = g16(b.dz2b, b.dz2l, f.h1b, f.h1l, i * act, (paired ? i : 0) * act, G + d.W1 + i * d.head_stride, nullptr,
                              H, H, rows, H, H, H);
            }
            const int nd = paired ? d.n_heads : 1;   // shared trunk: the heads' dgrads add into one dh1 -> only the first joins
            for (int i = 0; i < nd; ++i) {           // dh1[m][k] = sum_n dz2_i[m][n] W1_i[n][k]
                at[nq] = 0;
                q[nq++] = g16(b.dz2b, b.dz2l, sh.w1b, sh.w1l, i * act, (int64_t)i * H * H, b.dh1 + (paired ? i : 0) * act, nullptr,
                              rows, H, H, H, H, H);
            }
            EXORL_TRY(gemm16_grouped_mixed(at, q, nq, s));
            if (early && early->fo->on && defer) {
                EXORL_TRY(early->fo->fork(s));
                EXORL_TRY(finalize_adam(FinalizeArgs{}, early->fa, early->sh, early->fo->aux, 2));
                *early->launched = true;
            }
            for (int i = nd; i < d.n_heads; ++i) {
                Gemm16Problem r = g16(b.dz2b, b.dz2l, sh.w1b, sh.w1l, i * act, (int64_t)i * H * H, b.dh1, nullptr, rows, H, H, H, H, H);
                EXORL_TRY(gemm16_grouped(0, 1, &r, 1, false, true, s));
            }
        } else {
            if (G) {
                for (int i = 0; i < d.n_heads; ++i)
                    q[i] = g16(b.dz2b, b.dz2l, f.h1b, f.h1l, i * act, (paired ? i : 0) * act, G + d.W1 + i * d.head_stride, nullptr,
                               H, H, rows, H, H, H);
                EXORL_TRY(fk.fork(s));                 // wgrad only feeds the optimiser: off the dgrad -> LN-backward chain
                EXORL_TRY(gemm16_grouped(1, 1, q, d.n_heads, false, false, fk.side(s)));
            }
            for (int i = 0; i < d.n_heads; ++i)
                q[i] = g16(b.dz2b, b.dz2l, sh.w1b, sh.w1l, i * act, (int64_t)i * H * H, b.dh1 + (paired ? i : 0) * act, nullptr,
                           rows, H, H, H, H, H);
            if (paired) {
                EXORL_TRY(gemm16_grouped(0, 1, q, d.n_heads, false, false, s));
            } else {
                for (int i = 0; i < d.n_heads; ++i) EXORL_TRY(gemm16_grouped(0, 1, q + i, 1, false, i > 0, s));
            }
        }
    } else {
        GemmProblem p[2];
        if (G) {
            for (int i = 0; i < d.n_heads; ++i)
                p[i] = GemmProblem{b.dz2 + i * act, f.h1 + (paired ? i : 0) * act, G + d.W1 + i * d.head_stride, nullptr,
                                   H, H, rows, H, H, H};
            EXORL_TRY(fk.fork(s));
            EXORL_TRY(gemm_grouped(prec, 1, 1, p, d.n_heads, false, false, fk.side(s)));
        }
        for (int i = 0; i < d.n_heads; ++i)
            p[i] = GemmProblem{b.dz2 + i * act, P + d.W1 + i * d.head_stride, b.dh1 + (paired ? i : 0) * act, nullptr,
                               rows, H, H, H, H, H};
        if (paired) {
            EXORL_TRY(gemm_grouped(prec, 0, 1, p, d.n_heads, false, false, s));
        } else {
            for (int i = 0; i < d.n_heads; ++i) EXORL_TRY(gemm_grouped(prec, 0, 1, p + i, 1, false, i > 0, s));
        }
    }
    const bool dx_fused = dx && !G;              // dgrad-only pass: d/d(input columns) in the LayerNorm-backward kernel itself
    EXORL_TRY(ln_bwd(b.dh1, f.h1, f.xhat, bf ? f.h1b : nullptr, bf ? f.xhatb : nullptr, f.rstd, P + d.g, pt.Pt, rows, H, d.n_trunks,
                     act, d.trunk_stride, G ? 1 : 0, s, dx_fused ? sh.w0t + (int64_t)dx_col0 * H : nullptr, (int64_t)d.in_dim * H,
                     dx_fused ? dx : nullptr, dx_cols, bf ? f.h1l : nullptr, bf ? f.xhatl : nullptr, (bf && trunk_fwd16_supported(H)) ? P + d.beta : nullptr));
    if (dx && !dx_fused)       // dx[m][j] = sum_c dz0[m][c] W0[c][col0+j]: a row-dot against rows col0.. of the transposed shadow
        EXORL_TRY(head_fwd4(b.dh1, sh.w0t + (int64_t)dx_col0 * H, nullptr, dx, rows, H, dx_cols, 0, d.n_trunks, act,
                            (int64_t)d.in_dim * H, (int64_t)rows * dx_cols, s));
    if (G) {
        EXORL_TRY(outer_reduce(x, ldx, d.in_dim, b.dh1, pt.Pw, rows, H, d.n_trunks, act, s));
        EXORL_TRY(fk.join(s));                     // wgrad branch done before the gradients are finalised
        FinalizeArgs fa{};
        fa.Ph = pt.Ph; fa.head_chunks = have_dz2 ? qhead_chunks(rows) : head_chunks(rows); fa.n_heads = d.n_heads; fa.head_stride = d.head_stride;
        fa.gW2 = d.W2; fa.gb1 = d.b1; fa.gb2 = d.b2;
        fa.Pt = pt.Pt; fa.trunk_chunks = trunk_chunks(rows); fa.n_trunks = d.n_trunks; fa.trunk_stride = d.trunk_stride;
        fa.Pw = pt.Pw; fa.w_chunks = outer_chunks(rows);
        fa.gW0 = d.W0; fa.gb0 = d.b0; fa.gg = d.g; fa.gbeta = d.beta;
        fa.H = H; fa.nout = d.out_dim; fa.in_dim = d.in_dim; fa.G = G;
        if (defer) *defer = fa;                    // the optimiser launch sums the partials itself (finalize_adam)
        else EXORL_TRY(finalize_grads(fa, s));
    }
    return 0;
}

struct Carver {           // lays sub-buffers out in one workspace; base == nullptr -> sizing pass
    float* base;
    int64_t off = 0;
    explicit Carver(float* b) : base(b) {}
    float* take(int64_t n) {
        float* p = base ? base + off : nullptr;
        off += round_up(n, 64);
        return p;
    }
};

constexpr int ACT_ROWS = 64;

}  // namespace exorl

using namespace exorl;

struct exorl_agent {
    exorl_agent_cfg cfg;
    NetDesc actor, critic;
    bool has_critic = false;
    bool owns_ws = false;
    float* ws = nullptr;
    size_t ws_bytes = 0;
    // flat parameter state: [net][what]
    float* flat[3][4] = {{nullptr}};
    // batch slots
    float *obs = nullptr, *action = nullptr, *reward = nullptr, *discount = nullptr, *next_obs = nullptr;
    // staged inputs
    float *xa = nullptr, *xc_cur = nullptr, *xc_next = nullptr, *xc_pi = nullptr;
    FwdBufs fa{}, ft{}, fc{};    // actor (2B rows), target critic, critic
    BwdBufs bc{}, ba{};
    float *dq = nullptr, *da = nullptr, *dpre = nullptr, *abs_part = nullptr;
    float *sfq = nullptr, *sftq = nullptr;           // APS: scalar Q = task . successor features of critic / target (2, B)
    float *x_all = nullptr, *dq_all = nullptr;     // CQL: (3n+1)B critic rows and their per-row loss gradients
    CqlScalars* cql = nullptr;                      // CQL: log_actor_alpha + Adam moments + alpha (device)
    float *xc_rep = nullptr, *crr_w = nullptr;     // CRR: repeated (obs, sampled action) inputs; advantage weights
    FwdBufs fr{};                                   // CRR: critic forward on B*num_value_samples rows (no grad)
    NetShadow sh_actor{}, sh_critic{}, sh_target{};
    ShadowSpec spec_actor{}, spec_critic{};
    Partials pa{}, pc{};
    float *stats = nullptr, *metrics = nullptr;      // contiguous: stats[4] then metrics[EXORL_N_METRICS]
    float *act_x = nullptr, *act_noise = nullptr;
    float* act_part = nullptr; unsigned int* act_ticket = nullptr;      // one-launch act(): head shares per workgroup, arrival ticket
    FwdBufs fact{};
    StepState* state = nullptr;  // device-resident counters + Adam scalars (graph-replayable)
    const float *noise_c = nullptr, *noise_a = nullptr;   // caller-supplied noise of the step in flight (parity tests)
    int64_t actor_t = 0, critic_t = 0;       // host mirrors of state->t_*
    uint64_t act_noise_counter = 0;
    float inv_bg = 0.f;
    float dev_stddev = -1.f;     // host mirror of state->stddev (the last value enqueued for it)
    // captured step (sample + update) — exorl_agent_enable_graph
    hipGraphExec_t graph_exec = nullptr;
    hipGraph_t graph = nullptr;
    hipStream_t capture_stream = nullptr;
    exorl_replay* graph_replay = nullptr;
    bool capturing = false;
    Fork fk{};                   // parallel-branch plumbing (active while capturing)
    bool parallel_branches = false;  // measured slower than one chain on MI355X (2987 vs 3259 steps/s): opt-in
    bool fuse_opt = false;           // whole-step call on one GPU: partial-gradient reduction happens inside the optimiser launch
    bool whole_step = false;         // exorl_agent_update / the captured graph drive all four phases: the fused scalar-head kernels may run
    // data parallel: RCCL communicator attached by exorl_agent_set_comm; the library enqueues the all-reduces between the phases
    exorl_comm* comm = nullptr;
    hipStream_t comm_stream = nullptr;           // the 16-byte statistic travels here while the critic's backward pass runs
    hipEvent_t ev_stats_ready = nullptr, ev_stats_done = nullptr;
    FinalizeArgs pend_c{}, pend_a{};
    Fork fo;                         // side stream of the early H x H optimiser pass (EarlyW1)
    bool w1_early[2] = {false, false};   // [0] critic, [1] actor: the H x H part of this step's optimiser pass is already in flight
    bool staged_by_sampler = false;  // captured step: the sampler's gather kernel writes the staged inputs and runs step_begin
    bool want_metrics = true;    // the (B,1)-sized metric reductions are skipped when the caller never reads them (use_tb=False)
    int tq_slots = 0;            // > 0: this step's target critic left per-row partial head dots in ft.h2 (folded into the forward GEMM)
};

namespace exorl {

// split-bf16 mode runs the bf16 activation pipeline on hi/lo planes when every H x H GEMM tiles by 64 (the LDS-DMA kernel has no
// edge handling); other shapes take the fp32 activation pipeline with the operands split inside the GEMM (gemm_kernel<BF16X3>)
static bool planes_ok(const exorl_agent_cfg& cfg) {
    return cfg.precision == EXORL_PREC_BF16X3 && cfg.hidden_dim % 64 == 0 && cfg.batch % 64 == 0 && trunk_fwd16_supported(cfg.hidden_dim);
}

static void carve(exorl_agent* a, Carver& c) {
    const auto& cfg = a->cfg;
    const int64_t B = cfg.batch, O = cfg.obs_dim, A = cfg.act_dim, H = cfg.hidden_dim, W = O + A;
    const int64_t AO = a->actor.out_dim;                                     // actor head width (2A for CQL)
    const int64_t RC = cfg.kind == EXORL_AGENT_CQL ? (3 * cfg.n_samples + 1) * B : B;   // rows of the critic's gradient pass
    for (int w = 0; w < 4; ++w) a->flat[EXORL_NET_ACTOR][w] = c.take(a->actor.total);
    if (a->has_critic) {
        for (int w = 0; w < 4; ++w) a->flat[EXORL_NET_CRITIC][w] = c.take(a->critic.total);
        a->flat[EXORL_NET_CRITIC_TARGET][EXORL_T_PARAM] = c.take(a->critic.total);
    }
    a->obs = c.take(B * O); a->action = c.take(B * A); a->reward = c.take(B); a->discount = c.take(B); a->next_obs = c.take(B * O);
    a->xa = c.take(2 * B * O);
    auto take_u16 = [&](int64_t n) { return reinterpret_cast<unsigned short*>(c.take((n + 1) / 2)); };
    const bool x3 = planes_ok(cfg);
    const bool bf = cfg.precision == EXORL_PREC_BF16 || x3;
    auto take_lo = [&](int64_t n) { return x3 ? take_u16(n) : nullptr; };
    a->fa = FwdBufs{c.take(2 * B * H), c.take(2 * B * H), c.take(2 * B), c.take(2 * B * H), c.take(2 * B * AO), bf ? take_u16(2 * B * H) : nullptr,
                    bf ? take_u16(2 * B * H) : nullptr, take_lo(2 * B * H), take_lo(2 * B * H)};
    a->ba = BwdBufs{c.take(B * H), c.take(B * H), bf ? take_u16(B * H) : nullptr, take_lo(B * H)};
    a->sh_actor = NetShadow{c.take(O * H), bf ? take_u16(H * H) : nullptr, bf ? take_u16(H * round_up(O, 32)) : nullptr, take_lo(H * H),
                            take_lo(H * round_up(O, 32))};
    a->pa = Partials{c.take((int64_t)head_chunks(B) * ((AO + 1) * H + 32)), c.take((int64_t)trunk_chunks(B) * 3 * H),
                     c.take((int64_t)outer_chunks(B) * O * H)};
    a->dpre = c.take(B * AO);
    a->stats = c.take(4 + EXORL_N_METRICS);
    a->metrics = a->stats ? a->stats + 4 : nullptr;
    a->state = reinterpret_cast<StepState*>(c.take((sizeof(StepState) + 3) / 4));
    a->act_x = c.take(ACT_ROWS * O);
    a->act_noise = c.take(ACT_ROWS * A);
    a->act_part = c.take((int64_t)cdiv(H, 4) * ACT_FAST_ROWS * 16);
    a->act_ticket = reinterpret_cast<unsigned int*>(c.take(4));
    a->fact = FwdBufs{c.take(ACT_ROWS * H), nullptr, nullptr, c.take(ACT_ROWS * H), c.take(ACT_ROWS * AO), bf ? take_u16(ACT_ROWS * H) : nullptr, nullptr,
                      nullptr, nullptr};
    if (a->has_critic) {
        const int64_t nt = a->critic.n_trunks;
        a->xc_cur = c.take(B * W); a->xc_next = c.take(B * W); a->xc_pi = c.take(B * W);
        const int64_t od = a->critic.out_dim;
        a->ft = FwdBufs{c.take(nt * B * H), nullptr, nullptr, c.take(2 * B * H), c.take(2 * B * od), bf ? take_u16(nt * B * H) : nullptr, nullptr,
                        take_lo(nt * B * H), nullptr};
        a->fc = FwdBufs{c.take(nt * RC * H), c.take(nt * RC * H), c.take(nt * RC), c.take(2 * RC * H), c.take(2 * RC * od), bf ? take_u16(nt * RC * H) : nullptr,
                        bf ? take_u16(nt * RC * H) : nullptr, take_lo(nt * RC * H), take_lo(nt * RC * H)};
        a->bc = BwdBufs{c.take(2 * RC * H), c.take(nt * RC * H), bf ? take_u16(2 * RC * H) : nullptr, take_lo(2 * RC * H)};
        if (cfg.kind == EXORL_AGENT_CQL) {
            a->x_all = c.take(RC * W);
            a->dq_all = c.take(2 * RC);
            a->cql = reinterpret_cast<CqlScalars*>(c.take(16));
        }
        a->dq = c.take(2 * B);
        if (cfg.kind == EXORL_AGENT_APS) { a->sfq = c.take(2 * B); a->sftq = c.take(2 * B); }
        a->abs_part = c.take(2 * (int64_t)qhead_chunks(B));
        a->da = c.take(nt * B * A);
        if (cfg.kind == EXORL_AGENT_CRR) {
            const int64_t R = B * cfg.num_value_samples;
            a->xc_rep = c.take(R * W);
            a->crr_w = c.take(B);
            a->fr = FwdBufs{bf ? nullptr : c.take(nt * R * H), nullptr, nullptr, c.take(2 * R * H), c.take(2 * R), bf ? take_u16(nt * R * H) : nullptr, nullptr,
                            take_lo(nt * R * H), nullptr};
        }
        a->sh_critic = NetShadow{c.take(nt * W * H), bf ? take_u16(2 * H * H) : nullptr, bf ? take_u16(nt * H * round_up(W, 32)) : nullptr,
                                 take_lo(2 * H * H), take_lo(nt * H * round_up(W, 32))};
        a->sh_target = NetShadow{c.take(nt * W * H), bf ? take_u16(2 * H * H) : nullptr, bf ? take_u16(nt * H * round_up(W, 32)) : nullptr,
                                 take_lo(2 * H * H), take_lo(nt * H * round_up(W, 32))};
        a->pc = Partials{c.take(2 * (int64_t)qhead_chunks(RC) * ((od + 1) * H + 32)), c.take(nt * (int64_t)trunk_chunks(RC) * 3 * H),
                         c.take(nt * (int64_t)outer_chunks(RC) * W * H)};
    }
}

static int describe(exorl_agent* a, const exorl_agent_cfg* cfg) {
    EXORL_REQUIRE(cfg, "agent: null cfg");
    EXORL_REQUIRE(cfg->kind >= EXORL_AGENT_TD3_BC && cfg->kind <= EXORL_AGENT_APS, "agent: unknown kind %d", cfg->kind);
    EXORL_REQUIRE(cfg->kind != EXORL_AGENT_APS || (cfg->sf_dim >= 1 && cfg->sf_dim <= 16 && cfg->sf_dim < cfg->obs_dim),
                  "agent: APS needs 1 <= sf_dim <= 16 and obs_dim = observation + sf_dim (got sf_dim=%d obs_dim=%d)", cfg->sf_dim, cfg->obs_dim);
    EXORL_REQUIRE(cfg->kind != EXORL_AGENT_CQL || (cfg->n_samples >= 1 && cfg->n_samples <= 16 && cfg->act_dim <= 16),
                  "agent: CQL needs 1 <= n_samples <= 16 and action_dim <= 16 (got %d, %d)", cfg->n_samples, cfg->act_dim);
    EXORL_REQUIRE(!cfg->use_critic_lagrange || cfg->kind == EXORL_AGENT_CQL, "agent: use_critic_lagrange is a CQL option");
    EXORL_REQUIRE(cfg->kind != EXORL_AGENT_CRR || (cfg->num_value_samples >= 1 && cfg->num_value_samples <= 64 &&
                  cfg->weight_func >= EXORL_CRR_IDENTITY && cfg->weight_func <= EXORL_CRR_EXP),
                  "agent: CRR needs 1 <= num_value_samples <= 64 and a valid weight_func (got %d, %d)", cfg->num_value_samples, cfg->weight_func);
    EXORL_REQUIRE(cfg->obs_dim > 0 && cfg->act_dim > 0 && cfg->act_dim <= 16 && cfg->hidden_dim >= 4 && cfg->hidden_dim <= 1024 && cfg->hidden_dim % 4 == 0 &&
                  cfg->obs_dim + cfg->act_dim <= 256 &&
                  cfg->batch > 0, "agent: unsupported dims O=%d A=%d (<=16) H=%d (multiple of 4, <=1024) B=%d; O+A <= 256", cfg->obs_dim, cfg->act_dim,
                  cfg->hidden_dim, cfg->batch);
    EXORL_REQUIRE(cfg->precision == EXORL_PREC_F32 || cfg->precision == EXORL_PREC_BF16 || cfg->precision == EXORL_PREC_BF16X3, "agent: unknown precision %d", cfg->precision);
    EXORL_REQUIRE(cfg->world_size >= 1, "agent: world_size must be >= 1");
    EXORL_REQUIRE(cfg->precision != EXORL_PREC_BF16 || (cfg->hidden_dim % 8 == 0 && cfg->batch % 8 == 0),
                  "agent: bf16 precision needs hidden_dim and batch to be multiples of 8 (got H=%d B=%d)", cfg->hidden_dim, cfg->batch);
    a->cfg = *cfg;
    a->has_critic = cfg->kind != EXORL_AGENT_BC;
    a->actor = make_net(cfg->obs_dim, cfg->kind == EXORL_AGENT_CQL ? 2 * cfg->act_dim : cfg->act_dim, cfg->hidden_dim, 1, 1);
    if (a->has_critic)
        a->critic = make_net(cfg->obs_dim + cfg->act_dim, cfg->kind == EXORL_AGENT_APS ? cfg->sf_dim : 1, cfg->hidden_dim,
                             (cfg->kind == EXORL_AGENT_DDPG || cfg->kind == EXORL_AGENT_APS) ? 1 : 2, 2);
    a->inv_bg = 1.0f / ((float)cfg->batch * (float)cfg->world_size);
    return 0;
}

// update(): draw `which` (0 = critic target, 1 = actor) of this step, counter base lives in StepState
static NoiseSpec noise_spec(exorl_agent* a, const float* buf, int which) {
    return NoiseSpec{buf, a->cfg.seed, (uint64_t)which, buf ? nullptr : &a->state->noise_counter};
}
// act(): separate counter space (top bit set) so exploration draws never collide with update draws
static NoiseSpec act_noise_spec(exorl_agent* a, const float* buf) {
    NoiseSpec n{buf, a->cfg.seed, 0, nullptr};
    if (!buf) n.counter = (1ull << 63) | a->act_noise_counter++;
    return n;
}

static int push_opt_steps(exorl_agent* a) {
    long long t[2] = {a->actor_t, a->critic_t};
    EXORL_CHECK_HIP(hipMemcpy(&a->state->t_actor, t, sizeof(t), hipMemcpyHostToDevice));
    // beta^t as the device forms it (a running product, one multiply per step) so that a restored agent continues bit-identically
    auto prod = [](double b, long long t) { double r = 1.0; for (long long i = 0; i < t; ++i) r *= b; return r; };
    const double p[4] = {prod(0.9, a->actor_t), 0.0, prod(0.9, a->critic_t), prod(0.999, a->critic_t)};
    const double p2 = prod(0.999, a->actor_t);
    EXORL_CHECK_HIP(hipMemcpy(&a->state->b1t, p, sizeof(p), hipMemcpyHostToDevice));
    EXORL_CHECK_HIP(hipMemcpy(&a->state->b2t, &p2, sizeof(p2), hipMemcpyHostToDevice));
    return 0;
}

static int ensure_opt_fork(exorl_agent* a) {
    if (a->fo.aux) return 0;
    EXORL_CHECK_HIP(hipStreamCreateWithFlags(&a->fo.aux, hipStreamNonBlocking));
    EXORL_CHECK_HIP(hipEventCreateWithFlags(&a->fo.ev_fork, hipEventDisableTiming));
    EXORL_CHECK_HIP(hipEventCreateWithFlags(&a->fo.ev_join, hipEventDisableTiming));
    return 0;
}
// Off by default, exorl_gemm_tune bit 16777216 turns it on. Measured (TD3+BC, H = B = 1024, bf16x3): the two branches do overlap, but every
// cross-stream edge costs ~6 us on this runtime (event record -> wait, in a captured graph as well) and the co-running kernels slow each
// other (outer_reduce 8.6 -> 17.8 us): the critic's backward tail went from 41 to 54 us, the step from 0.290 to 0.340 ms.
static bool opt_overlap_enabled() { return (tune_variant() & 16777216) != 0; }
static FusedAdamArgs fused_adam_args(exorl_agent* a, int net, const NetDesc& d, const AdamConst* c, float* target, uint64_t* bump) {
    float** f = a->flat[net];
    return FusedAdamArgs{f[EXORL_T_PARAM], f[EXORL_T_GRAD], f[EXORL_T_ADAM_M], f[EXORL_T_ADAM_V], target, c, d.n_heads,
                         {d.W1, d.W1 + d.head_stride}, reinterpret_cast<unsigned long long*>(bump)};
}
static EarlyW1 early_w1(exorl_agent* a, int net) {
    const bool critic = net == EXORL_NET_CRITIC;
    return EarlyW1{&a->fo, fused_adam_args(a, net, critic ? a->critic : a->actor, critic ? &a->state->critic : &a->state->actor,
                                           critic ? a->flat[EXORL_NET_CRITIC_TARGET][EXORL_T_PARAM] : nullptr, nullptr),
                   critic ? a->spec_critic : a->spec_actor, &a->w1_early[critic ? 0 : 1]};
}
static int opt_step(exorl_agent* a, int net, const NetDesc& d, const FinalizeArgs& pend, const AdamConst* c, float* target, const ShadowSpec& spec,
                    uint64_t* bump, hipStream_t s) {
    float** f = a->flat[net];
    if (a->fuse_opt) {
        bool& early = a->w1_early[net == EXORL_NET_CRITIC ? 0 : 1];
        EXORL_TRY(finalize_adam(pend, fused_adam_args(a, net, d, c, target, bump), spec, s, early ? 1 : 0));
        if (early) EXORL_TRY(a->fo.join(s));
        early = false;
        return 0;
    }
    return adam_step_dev(f[EXORL_T_PARAM], f[EXORL_T_GRAD], f[EXORL_T_ADAM_M], f[EXORL_T_ADAM_V], d.total, c, target, &spec, s, bump);
}
static int opt_step_critic(exorl_agent* a, hipStream_t s) {
    return opt_step(a, EXORL_NET_CRITIC, a->critic, a->pend_c, &a->state->critic, a->flat[EXORL_NET_CRITIC_TARGET][EXORL_T_PARAM], a->spec_critic,
                    nullptr, s);
}

// fused scalar-head path: whole-step call on one GPU, nobody reads the metrics, twin scalar heads
static bool qfuse(const exorl_agent* a) {
    return a->whole_step && !a->fk.on && !a->want_metrics && a->has_critic && a->critic.n_heads == 2 && a->critic.out_dim == 1 && a->cfg.hidden_dim % 4 == 0 &&
           (a->cfg.kind == EXORL_AGENT_TD3_BC || a->cfg.kind == EXORL_AGENT_TD3 || a->cfg.kind == EXORL_AGENT_DDPG);
}
static int run_qhead(exorl_agent* a, int mode, hipStream_t s) {
    const NetDesc& d = a->critic;
    const int B = a->cfg.batch, H = d.H;
    const int64_t act = (int64_t)B * H;
    const float* Pc = a->flat[EXORL_NET_CRITIC][EXORL_T_PARAM];
    const float* Pt = a->flat[EXORL_NET_CRITIC_TARGET][EXORL_T_PARAM];
    const bool bf = a->bc.dz2b != nullptr;
    QHeadArgs q{};
    for (int i = 0; i < 2; ++i) {
        q.a[i] = a->fc.h2 + i * act; q.W[i] = Pc + d.W2 + i * d.head_stride; q.b[i] = Pc + d.b2 + i * d.head_stride;
        q.a[2 + i] = a->ft.h2 + i * act; q.W[2 + i] = Pt + d.W2 + i * d.head_stride; q.b[2 + i] = Pt + d.b2 + i * d.head_stride;
    }
    q.q = a->fc.out; q.tq = a->ft.out; q.reward = a->reward; q.discount = a->discount;
    q.tpart[0] = q.tpart[1] = nullptr; q.tslots = 0;
    if (mode == 0 && a->tq_slots > 0) {          // the target's heads were folded into its forward GEMM (net_forward2): ft.h2 holds partial dots
        q.tpart[0] = a->ft.h2; q.tpart[1] = a->ft.h2 + act; q.tslots = a->tq_slots;
    }
    q.dz = bf ? nullptr : a->bc.dz2; q.dzb = bf ? a->bc.dz2b : nullptr; q.dzl = bf ? a->bc.dz2l : nullptr; q.act = act;
    q.P = mode == 0 ? a->pc.Ph : nullptr;
    q.abs_part = a->abs_part;
    q.rows = B; q.H = H; q.mode = mode; q.inv_bg = a->inv_bg;
    return qhead(q, s);
}

// -- phase 0: everything up to the critic gradients -------------------------------------------------
static int phase0(exorl_agent* a, float stddev, const float* noise_c, hipStream_t s) {
    const auto& cfg = a->cfg;
    const int B = cfg.batch, O = cfg.obs_dim, A = cfg.act_dim, W = O + A, prec = cfg.precision;
    // stages the inputs and (thread 0) advances the device-side step state: counters, Adam scalars of this step
    if (!a->staged_by_sampler)
        EXORL_TRY(prepare_inputs(a->obs, a->action, a->next_obs, a->xa, a->xc_cur, a->xc_next, a->xc_pi, B, O, A, a->has_critic, a->state,
                                 a->capturing ? 1 : 0, s));
    a->actor_t += 1;
    if (a->has_critic) a->critic_t += 1;
    if (!a->has_critic) return 0;
    const float* Pa = a->flat[EXORL_NET_ACTOR][EXORL_T_PARAM];
    const float* Pc = a->flat[EXORL_NET_CRITIC][EXORL_T_PARAM];
    const float* Pt = a->flat[EXORL_NET_CRITIC_TARGET][EXORL_T_PARAM];
    // actor on [next_obs; obs] in one pass (td3_bc.py:124 and :149 use the same weights)
    // next_action = dist.sample(clip) (td3_bc.py:125) and the actor-step sample pi(obs) (:151, it does not depend on the
    // critic update) straight into the two critic input buffers, as the epilogue of the actor head
    a->noise_c = noise_c;
    const bool fused_sample = cfg.hidden_dim % 4 == 0 && A > 1;
    SampleSpec sp{&a->state->stddev, noise_c, cfg.kind == EXORL_AGENT_CRR ? nullptr : a->noise_a, cfg.seed, &a->state->noise_counter, stddev, cfg.stddev_clip,
                  a->xc_next + O, a->xc_pi + O, W, B};
    EXORL_TRY(net_forward(a->actor, Pa, a->sh_actor, a->xa, O, 2 * B, a->fa, true, true, prec, s, fused_sample ? &sp : nullptr));
    if (!fused_sample)
        EXORL_TRY(sample_actions2(a->fa.out, noise_c, cfg.kind == EXORL_AGENT_CRR ? nullptr : a->noise_a, cfg.seed,
                                  &a->state->noise_counter, stddev, cfg.stddev_clip, a->xc_next + O, a->xc_pi + O, W, B, A, s, &a->state->stddev));
    const bool qf = qfuse(a);
    a->tq_slots = 0;
    if (!a->fk.on && forward2_supported(a->critic, prec, a->sh_target, a->sh_critic)) {
        EXORL_TRY(net_forward2(a->critic, Pt, a->sh_target, a->xc_next, a->ft, false, Pc, a->sh_critic, a->xc_cur, a->fc, true, W, B, s, qf, qf,
                               &a->tq_slots));
    } else {
        EXORL_TRY(a->fk.fork(s));               // target critic (td3_bc.py:126) and critic (td3_bc.py:130) forwards are independent
        EXORL_TRY(net_forward(a->critic, Pt, a->sh_target, a->xc_next, W, B, a->ft, false, false, prec, a->fk.side(s), nullptr, qf));
        EXORL_TRY(net_forward(a->critic, Pc, a->sh_critic, a->xc_cur, W, B, a->fc, true, false, prec, s, nullptr, qf));
        EXORL_TRY(a->fk.join(s));
    }
    if (qf) EXORL_TRY(run_qhead(a, 0, s));       // Q, Q', TD gradient, dz2 and head partials in one kernel
    const bool aps = cfg.kind == EXORL_AGENT_APS;
    const float* task = aps ? a->obs + (O - cfg.sf_dim) : nullptr;          // obs rows are [observation | task] (aps.py:236-238)
    const float *qv = a->fc.out, *tqv = a->ft.out;
    if (aps) {                                  // Q = task . successor features (aps.py:55-58)
        EXORL_TRY(sf_q(a->ft.out, task, O, a->sftq, B, cfg.sf_dim, 2, s));
        EXORL_TRY(sf_q(a->fc.out, task, O, a->sfq, B, cfg.sf_dim, 2, s));
        qv = a->sfq; tqv = a->sftq;
    }
    if (a->want_metrics)
        EXORL_TRY(critic_loss(qv, tqv, a->reward, a->discount, a->dq, a->metrics, B, a->inv_bg, s));   // :133-137
    const EarlyW1 ew = early_w1(a, EXORL_NET_CRITIC);
    DoutSpec td{};                              // d(2 x MSE)/dQ computed where it is consumed (:127-131)
    td.mode = EXORL_DOUT_TD; td.q = qv; td.tq = tqv; td.reward = a->reward; td.discount = a->discount; td.inv_bg = a->inv_bg;
    td.task = task; td.task_ld = O;
    EXORL_TRY(net_backward(a->critic, Pc, a->sh_critic, a->flat[EXORL_NET_CRITIC][EXORL_T_GRAD], a->pc, a->xc_cur, W, B, a->fc, td,
                           a->bc, nullptr, 0, 0, prec, s, a->fk, a->fuse_opt ? &a->pend_c : nullptr, qf, &ew));                  // :141
    return 0;
}

// -- phase 1: critic optimiser step (+ fused soft update), critic re-evaluated at pi(obs) -------------
static int phase1(exorl_agent* a, float stddev, const float* noise_a, hipStream_t s) {
    if (!a->has_critic) return 0;
    const auto& cfg = a->cfg;
    const int B = cfg.batch, O = cfg.obs_dim, A = cfg.act_dim, W = O + A, prec = cfg.precision;
    EXORL_TRY(opt_step_critic(a, s));
    if (cfg.kind == EXORL_AGENT_CRR) {          // crr.py:170-179 with the updated critic, no gradients
        const int n = cfg.num_value_samples;
        const float* Pc = a->flat[EXORL_NET_CRITIC][EXORL_T_PARAM];
        EXORL_TRY(repeat_sample(a->obs, a->fa.out + (int64_t)B * A, noise_a, cfg.seed, noise_a ? nullptr : &a->state->noise_counter, 1,
                                stddev, cfg.stddev_clip, a->xc_rep, B, O, A, n, s, &a->state->stddev));
        EXORL_TRY(net_forward(a->critic, Pc, a->sh_critic, a->xc_rep, W, B * n, a->fr, false, false, prec, s));      // compute_value
        EXORL_TRY(net_forward(a->critic, Pc, a->sh_critic, a->xc_cur, W, B, a->fc, false, false, prec, s));          // Q(s, a_data)
        EXORL_TRY(crr_weights(a->fr.out, a->fc.out, a->crr_w, B, n, cfg.weight_func, s));
        return 0;
    }
    // pi(obs) sample already sits in xc_pi (phase 0); DDPG logs its log-prob (ddpg.py:276,289)
    if ((cfg.kind == EXORL_AGENT_DDPG || cfg.kind == EXORL_AGENT_APS) && a->want_metrics)
        EXORL_TRY(sample_action(a->fa.out + (int64_t)B * A, noise_spec(a, noise_a, 1), stddev, cfg.stddev_clip, 1, a->xc_pi + O, W, B, A,
                                a->metrics + EXORL_M_ACTOR_LOGPROB, s, &a->state->stddev, cfg.world_size));
    const bool qf = qfuse(a);
    EXORL_TRY(net_forward(a->critic, a->flat[EXORL_NET_CRITIC][EXORL_T_PARAM], a->sh_critic, a->xc_pi, W, B, a->fc, true, false, prec, s,
                          nullptr, qf));
    if (cfg.kind == EXORL_AGENT_APS) {
        EXORL_TRY(sf_q(a->fc.out, a->obs + (O - cfg.sf_dim), O, a->sfq, B, cfg.sf_dim, 2, s));
        EXORL_TRY(actor_stats(a->sfq, a->stats, B, s));
    } else if (!qf) {
        EXORL_TRY(actor_stats(a->fc.out, a->stats, B, s));
    }
    return 0;
}

// -- phase 2: actor gradients -----------------------------------------------------------------------
static int phase2(exorl_agent* a, float stddev, hipStream_t s) {
    const auto& cfg = a->cfg;
    const int B = cfg.batch, O = cfg.obs_dim, A = cfg.act_dim, W = O + A, H = cfg.hidden_dim, prec = cfg.precision;
    const float* Pa = a->flat[EXORL_NET_ACTOR][EXORL_T_PARAM];
    if (a->has_critic && cfg.kind != EXORL_AGENT_CRR) {
        DoutSpec dq{};                          // -lambda/Bg routed to the smaller Q (td3_bc.py:152-155)
        dq.mode = EXORL_DOUT_ACTOR_Q; dq.q = a->fc.out; dq.stats = a->stats; dq.inv_bg = a->inv_bg; dq.alpha = cfg.alpha;
        if (cfg.kind == EXORL_AGENT_APS) { dq.q = a->sfq; dq.task = a->obs + (O - cfg.sf_dim); dq.task_ld = O; }
        dq.use_lambda = cfg.kind == EXORL_AGENT_TD3_BC;
        const bool qf = qfuse(a);
        if (qf) EXORL_TRY(run_qhead(a, 1, s));   // Q(s, pi(s)), its min-routing gradient (lambda applied later), dz2, sum|Q| partials
        if (qf && a->comm && dq.use_lambda) {    // lambda needs sum|Q| over the GLOBAL batch (td3_bc.py:154): reduce it while the critic's
            EXORL_TRY(reduce_pairs(a->abs_part, qhead_chunks(B), a->stats, s));          // backward pass runs (it does not depend on lambda)
            EXORL_CHECK_HIP(hipEventRecord(a->ev_stats_ready, s));
            EXORL_CHECK_HIP(hipStreamWaitEvent(a->comm_stream, a->ev_stats_ready, 0));
            EXORL_TRY(comm_allreduce_sum(a->comm, a->stats, 4, a->comm_stream));
            EXORL_CHECK_HIP(hipEventRecord(a->ev_stats_done, a->comm_stream));
        }
        EXORL_TRY(net_backward(a->critic, a->flat[EXORL_NET_CRITIC][EXORL_T_PARAM], a->sh_critic, nullptr, a->pc, a->xc_pi, W, B, a->fc,
                               dq, a->bc, a->da, O, A, prec, s, a->fk, nullptr, qf));
    }
    // the obs half (rows B..2B) of the stacked actor forward
    FwdBufs f{a->fa.h1 + (int64_t)B * H, a->fa.xhat + (int64_t)B * H, a->fa.rstd + B, a->fa.h2 + (int64_t)B * H,
              a->fa.out + (int64_t)B * A, a->fa.h1b ? a->fa.h1b + (int64_t)B * H : nullptr,
              a->fa.xhatb ? a->fa.xhatb + (int64_t)B * H : nullptr, a->fa.h1l ? a->fa.h1l + (int64_t)B * H : nullptr,
              a->fa.xhatl ? a->fa.xhatl + (int64_t)B * H : nullptr};
    if (!a->has_critic)       // BC (bc.py:82): the only forward of the step
        EXORL_TRY(net_forward(a->actor, Pa, a->sh_actor, a->xa + (int64_t)B * O, O, B, f, true, true, prec, s));
    if (a->want_metrics)                        // actor_loss / batch_reward(BC) metrics only (the gradient is formed in head_bwd)
        EXORL_TRY(actor_dmu(a->da, A, a->has_critic ? a->critic.n_trunks : 0, (int64_t)B * A, f.out, a->action,
                            a->has_critic ? nullptr : a->reward, a->crr_w, a->dpre, a->stats, a->metrics, B, A, a->inv_bg, cfg.alpha,
                            cfg.kind, stddev, s, &a->state->stddev));
    DoutSpec dm{};
    dm.mode = EXORL_DOUT_ACTOR_MU; dm.da = a->da; dm.da_nets = a->has_critic ? a->critic.n_trunks : 0; dm.mu = f.out; dm.a_data = a->action;
    dm.kind = cfg.kind; dm.inv_bg = a->inv_bg; dm.stddev = stddev; dm.stddev_ptr = &a->state->stddev; dm.w = a->crr_w;
    if (qfuse(a)) {                             // lambda = alpha / mean|Q| from the per-chunk sums qhead left behind
        dm.lam_parts = a->abs_part; dm.lam_chunks = qhead_chunks(B); dm.use_lambda = cfg.kind == EXORL_AGENT_TD3_BC; dm.alpha = cfg.alpha;
        if (a->comm && dm.use_lambda) {          // the all-reduced statistic (one 'chunk': stats[0] = global sum |Q|)
            EXORL_CHECK_HIP(hipStreamWaitEvent(s, a->ev_stats_done, 0));
            dm.lam_parts = a->stats; dm.lam_chunks = 1;
        }
    }
    const EarlyW1 ew = early_w1(a, EXORL_NET_ACTOR);
    EXORL_TRY(net_backward(a->actor, Pa, a->sh_actor, a->flat[EXORL_NET_ACTOR][EXORL_T_GRAD], a->pa, a->xa + (int64_t)B * O, O, B, f,
                           dm, a->ba, nullptr, 0, 0, prec, s, a->fk, a->fuse_opt ? &a->pend_a : nullptr, false, &ew));
    return 0;
}

static int phase3(exorl_agent* a, hipStream_t s) {
    return opt_step(a, EXORL_NET_ACTOR, a->actor, a->pend_a, &a->state->actor, nullptr, a->spec_actor,
                    a->staged_by_sampler ? &a->state->replay_counter : nullptr, s);
}

// ---- CQL (cql.py:152-263) ---------------------------------------------------------------------------
static CqlNoise cql_noise(exorl_agent* a) {
    const int64_t BA = (int64_t)a->cfg.batch * a->cfg.act_dim, n = a->cfg.n_samples;
    const float* nc = a->noise_c;
    CqlNoise z{};
    z.z_next = nc; z.u_rand = nc ? nc + BA : nullptr; z.z_cur = nc ? nc + (1 + n) * BA : nullptr;
    z.z_nxt = nc ? nc + (1 + 2 * n) * BA : nullptr; z.z_actor = a->noise_a;
    z.seed = a->cfg.seed; z.counter_ptr = &a->state->noise_counter;
    return z;
}

// Phase 0 in two halves. One GPU (and data parallel without the Lagrange multiplier) runs them back to back. With use_critic_lagrange under
// torch.distributed the host drives them as phases 4 and 5 and sum-all-reduces the statistics block in between (cql.hip, cql_critic_dq_kernel).
static int cql_phase0a(exorl_agent* a, hipStream_t s, bool split) {
    const auto& cfg = a->cfg;
    const int B = cfg.batch, O = cfg.obs_dim, A = cfg.act_dim, W = O + A, n = cfg.n_samples, prec = cfg.precision;
    const int R = (3 * n + 1) * B;
    if (!a->staged_by_sampler)
        EXORL_TRY(prepare_inputs(a->obs, a->action, a->next_obs, a->xa, a->xc_cur, a->xc_next, a->xc_pi, B, O, A, 1, a->state,
                                 a->capturing ? 1 : 0, s));
    a->actor_t += 1;
    a->critic_t += 1;
    const float* Pa = a->flat[EXORL_NET_ACTOR][EXORL_T_PARAM];
    const float* Pc = a->flat[EXORL_NET_CRITIC][EXORL_T_PARAM];
    EXORL_TRY(net_forward(a->actor, Pa, a->sh_actor, a->xa, O, 2 * B, a->fa, true, false, prec, s));      // raw (mu | log_std) on [next_obs; obs]
    EXORL_TRY(cql_build_inputs(a->obs, a->action, a->fa.out, cql_noise(a), a->xc_next, a->x_all, B, O, A, n, s));
    EXORL_TRY(net_forward(a->critic, a->flat[EXORL_NET_CRITIC_TARGET][EXORL_T_PARAM], a->sh_target, a->xc_next, W, B, a->ft, false, false,
                          prec, s));                                                                     // cql.py:160
    EXORL_TRY(net_forward(a->critic, Pc, a->sh_critic, a->x_all, W, R, a->fc, true, false, prec, s));     // cql.py:166,179-184 in one pass
    if (split)      // this rank's penalty sums into stats[2], stats[3]
        EXORL_TRY(cql_critic_dq(a->fc.out, a->ft.out, a->reward, a->discount, a->dq_all, a->metrics, B, n, cfg.alpha, a->inv_bg, s,
                                a->cql + 1, &a->state->critic, cfg.target_cql_penalty, 1, a->stats + 2));
    return 0;
}

static int cql_phase0b(exorl_agent* a, hipStream_t s, bool split) {
    const auto& cfg = a->cfg;
    const int B = cfg.batch, O = cfg.obs_dim, A = cfg.act_dim, W = O + A, n = cfg.n_samples, prec = cfg.precision;
    const int R = (3 * n + 1) * B;
    const float* Pc = a->flat[EXORL_NET_CRITIC][EXORL_T_PARAM];
    EXORL_TRY(cql_critic_dq(a->fc.out, a->ft.out, a->reward, a->discount, a->dq_all, a->metrics, B, n, cfg.alpha, a->inv_bg, s,
                            cfg.use_critic_lagrange ? a->cql + 1 : nullptr, &a->state->critic, cfg.target_cql_penalty, split ? 2 : 0, a->stats + 2));
    DoutSpec d{};
    d.mode = EXORL_DOUT_BUFFER; d.buf = a->dq_all;
    EXORL_TRY(net_backward(a->critic, Pc, a->sh_critic, a->flat[EXORL_NET_CRITIC][EXORL_T_GRAD], a->pc, a->x_all, W, R, a->fc, d, a->bc,
                           nullptr, 0, 0, prec, s, a->fk, a->fuse_opt ? &a->pend_c : nullptr));
    return 0;
}

static int cql_phase0(exorl_agent* a, hipStream_t s) {
    EXORL_REQUIRE(!(a->cfg.use_critic_lagrange && a->cfg.world_size > 1), "agent_update_phase: CQL with use_critic_lagrange under data parallelism "
                  "runs phase 0 as phases 4 and 5 with a sum-all-reduce of exorl_agent_stats in between");
    EXORL_TRY(cql_phase0a(a, s, false));
    return cql_phase0b(a, s, false);
}

static int cql_phase1(exorl_agent* a, hipStream_t s) {
    const auto& cfg = a->cfg;
    const int B = cfg.batch, O = cfg.obs_dim, A = cfg.act_dim, W = O + A, prec = cfg.precision;
    EXORL_TRY(opt_step_critic(a, s));
    EXORL_TRY(cql_actor_sample(a->fa.out + (int64_t)B * 2 * A, cql_noise(a), a->xc_pi, W, a->stats, B, O, A, s));    // cql.py:237-239
    EXORL_TRY(net_forward(a->critic, a->flat[EXORL_NET_CRITIC][EXORL_T_PARAM], a->sh_critic, a->xc_pi, W, B, a->fc, true, false, prec, s));
    return 0;
}

static int cql_phase2(exorl_agent* a, hipStream_t s) {
    const auto& cfg = a->cfg;
    const int B = cfg.batch, O = cfg.obs_dim, A = cfg.act_dim, W = O + A, H = cfg.hidden_dim, prec = cfg.precision;
    EXORL_TRY(cql_alpha_step(a->cql, a->stats, &a->state->actor, a->metrics, B, A, a->inv_bg, a->fc.out, s));    // cql.py:241-247
    DoutSpec dq{};
    dq.mode = EXORL_DOUT_ACTOR_Q; dq.q = a->fc.out; dq.stats = a->stats; dq.inv_bg = a->inv_bg; dq.use_lambda = 0;
    EXORL_TRY(net_backward(a->critic, a->flat[EXORL_NET_CRITIC][EXORL_T_PARAM], a->sh_critic, nullptr, a->pc, a->xc_pi, W, B, a->fc, dq,
                           a->bc, a->da, O, A, prec, s, a->fk));
    FwdBufs f{a->fa.h1 + (int64_t)B * H, a->fa.xhat + (int64_t)B * H, a->fa.rstd + B, a->fa.h2 + (int64_t)B * H,
              a->fa.out + (int64_t)B * 2 * A, a->fa.h1b ? a->fa.h1b + (int64_t)B * H : nullptr,
              a->fa.xhatb ? a->fa.xhatb + (int64_t)B * H : nullptr, a->fa.h1l ? a->fa.h1l + (int64_t)B * H : nullptr,
              a->fa.xhatl ? a->fa.xhatl + (int64_t)B * H : nullptr};
    DoutSpec dm{};
    dm.mode = EXORL_DOUT_CQL_ACTOR; dm.da = a->da; dm.da_nets = a->critic.n_trunks; dm.raw = f.out; dm.z = a->noise_a;
    dm.alpha_ptr = &a->cql->alpha; dm.inv_bg = a->inv_bg; dm.seed = cfg.seed; dm.counter = 4; dm.counter_ptr = &a->state->noise_counter;
    EXORL_TRY(net_backward(a->actor, a->flat[EXORL_NET_ACTOR][EXORL_T_PARAM], a->sh_actor, a->flat[EXORL_NET_ACTOR][EXORL_T_GRAD], a->pa,
                           a->xa + (int64_t)B * O, O, B, f, dm, a->ba, nullptr, 0, 0, prec, s, a->fk, a->fuse_opt ? &a->pend_a : nullptr));
    return 0;
}

static int release_graph(exorl_agent* a) {
    if (a->graph_exec) { EXORL_CHECK_HIP(hipGraphExecDestroy(a->graph_exec)); a->graph_exec = nullptr; }
    if (a->graph) { EXORL_CHECK_HIP(hipGraphDestroy(a->graph)); a->graph = nullptr; }
    a->graph_replay = nullptr;
    return 0;
}

}  // namespace exorl

extern "C" {

size_t exorl_agent_workspace_bytes(const exorl_agent_cfg* cfg) {
    exorl_agent tmp;
    if (describe(&tmp, cfg) != 0) return 0;
    Carver c(nullptr);
    carve(&tmp, c);
    return (size_t)c.off * sizeof(float);
}

int exorl_agent_create(const exorl_agent_cfg* cfg, void* workspace, size_t workspace_bytes, exorl_agent_t** out) {
    EXORL_REQUIRE(out, "agent_create: null out");
    auto* a = new exorl_agent();
    if (int rc = describe(a, cfg)) { delete a; return rc; }
    Carver sizing(nullptr);
    carve(a, sizing);
    const size_t need = (size_t)sizing.off * sizeof(float);
    if (workspace) {
        if (workspace_bytes < need || (uintptr_t)workspace % 256 != 0) {
            set_error("agent_create: workspace %zu B (need %zu B, 256-byte aligned)", workspace_bytes, need);
            delete a;
            return 2;
        }
        a->ws = static_cast<float*>(workspace);
    } else {
        hipError_t e = hipMalloc((void**)&a->ws, need);
        if (e != hipSuccess) { set_error("agent_create: hipMalloc(%zu) -> %s", need, hipGetErrorString(e)); delete a; return 1; }
        a->owns_ws = true;
    }
    a->ws_bytes = need;
    hipError_t e = hipMemset(a->ws, 0, need);
    if (e != hipSuccess) { set_error("agent_create: hipMemset -> %s", hipGetErrorString(e)); if (a->owns_ws) (void)hipFree(a->ws); delete a; return 1; }
    Carver c(a->ws);
    carve(a, c);
    a->spec_actor = shadow_spec(a->actor, a->sh_actor, nullptr);
    if (a->has_critic) a->spec_critic = shadow_spec(a->critic, a->sh_critic, &a->sh_target);
    StepState st{};
    st.lr = dec7(cfg->lr); st.b1 = 0.9; st.b2 = 0.999; st.eps = 1e-8; st.tau = dec7(cfg->tau);      // torch.optim.Adam defaults
    st.has_critic = a->has_critic ? 1 : 0;
    st.b1t = st.b2t = st.b1t_c = st.b2t_c = 1.0;
    if (a->cql) { CqlScalars sc{0.f, 0.f, 0.f, 1.0f}; (void)hipMemcpy(a->cql, &sc, sizeof(sc), hipMemcpyHostToDevice); }
    e = hipMemcpy(a->state, &st, sizeof(st), hipMemcpyHostToDevice);
    if (e != hipSuccess) { set_error("agent_create: state upload -> %s", hipGetErrorString(e)); if (a->owns_ws) (void)hipFree(a->ws); delete a; return 1; }
    *out = a;
    return 0;
}

int exorl_agent_destroy(exorl_agent_t* a) {
    if (!a) return 0;
    (void)release_graph(a);
    if (a->capture_stream) (void)hipStreamDestroy(a->capture_stream);
    if (a->comm_stream) { (void)hipStreamDestroy(a->comm_stream); (void)hipEventDestroy(a->ev_stats_ready); (void)hipEventDestroy(a->ev_stats_done); }
    if (a->fk.aux) { (void)hipStreamDestroy(a->fk.aux); (void)hipEventDestroy(a->fk.ev_fork); (void)hipEventDestroy(a->fk.ev_join); }
    if (a->fo.aux) { (void)hipStreamDestroy(a->fo.aux); (void)hipEventDestroy(a->fo.ev_fork); (void)hipEventDestroy(a->fo.ev_join); }
    if (a->owns_ws) (void)hipFree(a->ws);
    delete a;
    return 0;
}

static const NetDesc* net_of(exorl_agent_t* a, int32_t net) {
    if (net == EXORL_NET_ACTOR) return &a->actor;
    if ((net == EXORL_NET_CRITIC || net == EXORL_NET_CRITIC_TARGET) && a->has_critic) return &a->critic;
    return nullptr;
}

int exorl_agent_num_tensors(exorl_agent_t* a, int32_t net, int32_t* n) {
    EXORL_REQUIRE(a && n, "agent_num_tensors: null argument");
    const NetDesc* d = net_of(a, net);
    EXORL_REQUIRE(d, "agent_num_tensors: agent has no net %d", net);
    *n = (int32_t)d->tensors.size();
    return 0;
}

int exorl_agent_tensor(exorl_agent_t* a, int32_t net, int32_t index, int32_t what, void** ptr, int64_t* rows, int64_t* cols) {
    EXORL_REQUIRE(a && ptr && rows && cols, "agent_tensor: null argument");
    const NetDesc* d = net_of(a, net);
    EXORL_REQUIRE(d, "agent_tensor: agent has no net %d", net);
    EXORL_REQUIRE(index >= 0 && index < (int)d->tensors.size(), "agent_tensor: index %d out of range", index);
    EXORL_REQUIRE(what >= 0 && what < 4 && a->flat[net][what], "agent_tensor: net %d has no buffer kind %d", net, what);
    *ptr = a->flat[net][what] + d->tensors[index].off;
    *rows = d->tensors[index].rows;
    *cols = d->tensors[index].cols;
    return 0;
}

int exorl_agent_flat(exorl_agent_t* a, int32_t net, int32_t what, void** ptr, int64_t* numel) {
    EXORL_REQUIRE(a && ptr && numel, "agent_flat: null argument");
    const NetDesc* d = net_of(a, net);
    EXORL_REQUIRE(d && what >= 0 && what < 4 && a->flat[net][what], "agent_flat: no buffer net=%d what=%d", net, what);
    *ptr = a->flat[net][what];
    *numel = d->total;
    return 0;
}

int exorl_agent_params_changed(exorl_agent_t* a, int32_t sync_target, void* stream) {
    EXORL_REQUIRE(a, "agent_params_changed: null handle");
    hipStream_t s = as_stream(stream);
    if (sync_target && a->has_critic)
        EXORL_CHECK_HIP(hipMemcpyAsync(a->flat[EXORL_NET_CRITIC_TARGET][EXORL_T_PARAM], a->flat[EXORL_NET_CRITIC][EXORL_T_PARAM],
                                       a->critic.total * sizeof(float), hipMemcpyDeviceToDevice, s));
    EXORL_TRY(refresh_shadows(a->flat[EXORL_NET_ACTOR][EXORL_T_PARAM], a->actor.total, a->spec_actor, false, s));
    if (a->has_critic) {
        EXORL_TRY(refresh_shadows(a->flat[EXORL_NET_CRITIC][EXORL_T_PARAM], a->critic.total, a->spec_critic, false, s));
        EXORL_TRY(refresh_shadows(a->flat[EXORL_NET_CRITIC_TARGET][EXORL_T_PARAM], a->critic.total, a->spec_critic, true, s));
    }
    return 0;
}

int exorl_agent_batch_slots(exorl_agent_t* a, exorl_batch_out* out) {
    EXORL_REQUIRE(a && out, "agent_batch_slots: null argument");
    const int64_t O = a->cfg.obs_dim, A = a->cfg.act_dim;
    out->obs = a->obs; out->obs_stride = O * 4;
    out->action = a->action; out->action_stride = A;
    out->reward = a->reward; out->discount = a->discount;
    out->next_obs = a->next_obs; out->next_obs_stride = O * 4;
    out->meta = nullptr; out->meta_stride = 0;
    return 0;
}

int exorl_agent_set_batch(exorl_agent_t* a, const float* obs, const float* action, const float* reward, const float* discount,
                          const float* next_obs, void* stream) {
    EXORL_REQUIRE(a && obs && action && reward && discount && next_obs, "agent_set_batch: null argument");
    hipStream_t s = as_stream(stream);
    const size_t B = a->cfg.batch, O = a->cfg.obs_dim, A = a->cfg.act_dim;
    EXORL_CHECK_HIP(hipMemcpyAsync(a->obs, obs, B * O * 4, hipMemcpyDeviceToDevice, s));
    EXORL_CHECK_HIP(hipMemcpyAsync(a->action, action, B * A * 4, hipMemcpyDeviceToDevice, s));
    EXORL_CHECK_HIP(hipMemcpyAsync(a->reward, reward, B * 4, hipMemcpyDeviceToDevice, s));
    EXORL_CHECK_HIP(hipMemcpyAsync(a->discount, discount, B * 4, hipMemcpyDeviceToDevice, s));
    EXORL_CHECK_HIP(hipMemcpyAsync(a->next_obs, next_obs, B * O * 4, hipMemcpyDeviceToDevice, s));
    return 0;
}

int exorl_agent_update_phase(exorl_agent_t* a, int32_t phase, float stddev, const float* noise_c, const float* noise_a, void* stream) {
    EXORL_REQUIRE(a, "agent_update_phase: null handle");
    EXORL_REQUIRE(stddev > 0.f || a->cfg.kind == EXORL_AGENT_CQL, "agent_update_phase: stddev must be > 0");
    hipStream_t s = as_stream(stream);
    a->noise_a = noise_a;
    // the kernels read the exploration std from the device step state; enqueue a new value only when the schedule moved
    // (never while capturing: a captured graph is std-agnostic, exorl_agent_step_graph writes it before the launch)
    if (!a->capturing && a->cfg.kind != EXORL_AGENT_CQL && stddev != a->dev_stddev) {
        EXORL_TRY(set_device_float(&a->state->stddev, stddev, s));
        a->dev_stddev = stddev;
    }
    if (a->cfg.kind == EXORL_AGENT_CQL) {
        a->noise_c = noise_c;
        switch (phase) {
            case 0: return cql_phase0(a, s);
            case 1: return cql_phase1(a, s);
            case 2: return cql_phase2(a, s);
            case 3: return phase3(a, s);
            case 4: return cql_phase0a(a, s, true);        // use_critic_lagrange under data parallelism: phase 0 up to the penalty sums ...
            case 5: return cql_phase0b(a, s, true);        // ... and from the global penalty on
        }
    }
    switch (phase) {
        case 0: return phase0(a, stddev, noise_c, s);
        case 1: return phase1(a, stddev, noise_a, s);
        case 2: return phase2(a, stddev, s);
        case 3: return phase3(a, s);
    }
    set_error("agent_update_phase: phase %d out of range", phase);
    return 2;
}

// The four phases of one update with the data-parallel exchanges between them (a->comm != nullptr); shared by the eager whole-step call
// and by the graph capture.
// one GPU: nothing is exchanged between the phases, the optimiser launches reduce the gradient partials themselves.
// data parallel: gradients are finalised into the flat buffers, sum-all-reduced over RCCL on this stream, then stepped.
static int whole_step_body(exorl_agent* a, float stddev, const float* noise_c, const float* noise_a, hipStream_t s) {
    const bool dp = a->comm != nullptr;
    const int kind = a->cfg.kind;
    // CQL's Lagrange multiplier steps on the penalty of the GLOBAL batch inside phase 0: with a communicator phase 0 runs as 4 | all-reduce | 5
    const bool lag_dp = dp && kind == EXORL_AGENT_CQL && a->cfg.use_critic_lagrange;
    int rc = exorl_agent_update_phase(a, lag_dp ? 4 : 0, stddev, noise_c, noise_a, s);
    if (rc == 0 && lag_dp) rc = comm_allreduce_sum(a->comm, a->stats, 4, s);
    if (rc == 0 && lag_dp) rc = exorl_agent_update_phase(a, 5, stddev, noise_c, noise_a, s);
    if (rc == 0 && dp && a->has_critic) rc = comm_allreduce_sum(a->comm, a->flat[EXORL_NET_CRITIC][EXORL_T_GRAD], a->critic.total, s);
    if (rc == 0) rc = exorl_agent_update_phase(a, 1, stddev, noise_c, noise_a, s);
    // TD3+BC's sum |Q| (lambda) / CQL's sum log pi (entropy temperature); the fused scalar-head path moves it inside phase 2
    if (rc == 0 && dp && ((kind == EXORL_AGENT_TD3_BC && !qfuse(a)) || kind == EXORL_AGENT_CQL)) rc = comm_allreduce_sum(a->comm, a->stats, 4, s);
    if (rc == 0) rc = exorl_agent_update_phase(a, 2, stddev, noise_c, noise_a, s);
    if (rc == 0 && dp) rc = comm_allreduce_sum(a->comm, a->flat[EXORL_NET_ACTOR][EXORL_T_GRAD], a->actor.total, s);
    if (rc == 0) rc = exorl_agent_update_phase(a, 3, stddev, noise_c, noise_a, s);
    return rc;
}

int exorl_agent_update(exorl_agent_t* a, float stddev, const float* noise_c, const float* noise_a, void* stream) {
    EXORL_REQUIRE(a, "agent_update: null handle");
    const bool dp = a->comm != nullptr;
    EXORL_REQUIRE(a->cfg.world_size == 1 || dp, "agent_update: world_size=%d needs a communicator (exorl_agent_set_comm), or drive "
                  "exorl_agent_update_phase and all-reduce between the phases yourself", a->cfg.world_size);
    hipStream_t s = as_stream(stream);
    a->whole_step = true;
    a->fuse_opt = !dp && !a->fk.on;
    if (a->fuse_opt && opt_overlap_enabled()) { EXORL_TRY(ensure_opt_fork(a)); a->fo.on = true; }
    const int rc = whole_step_body(a, stddev, noise_c, noise_a, s);
    a->fuse_opt = false;
    a->whole_step = false;
    a->fo.on = false;
    a->w1_early[0] = a->w1_early[1] = false;
    return rc;
}

int exorl_agent_set_comm(exorl_agent_t* a, exorl_comm_t* c) {
    EXORL_REQUIRE(a, "agent_set_comm: null handle");
    EXORL_REQUIRE(!a->graph_exec, "agent_set_comm: disable the captured graph first");
    EXORL_REQUIRE(!c || comm_nranks(c) == a->cfg.world_size, "agent_set_comm: communicator has %d ranks, the agent was built for world_size=%d "
                  "(its means are over batch * world_size)", comm_nranks(c), a->cfg.world_size);
    if (c && !a->comm_stream) {
        EXORL_CHECK_HIP(hipStreamCreateWithFlags(&a->comm_stream, hipStreamNonBlocking));
        EXORL_CHECK_HIP(hipEventCreateWithFlags(&a->ev_stats_ready, hipEventDisableTiming));
        EXORL_CHECK_HIP(hipEventCreateWithFlags(&a->ev_stats_done, hipEventDisableTiming));
    }
    a->comm = c;
    return 0;
}

int exorl_agent_stats_buffer(exorl_agent_t* a, void** ptr, int64_t* numel) {
    EXORL_REQUIRE(a && ptr && numel, "agent_stats_buffer: null argument");
    *ptr = a->stats;
    *numel = 4;
    return 0;
}

// one launch for <= ACT_FAST_ROWS rows of a tanh-mean policy (every agent kind but CQL's tanh-Gaussian); obs / noise on the device or on the host
static bool act_fast_ok(const exorl_agent* a, int n) {
    return a->cfg.kind != EXORL_AGENT_CQL && act_fast_supported(n, a->cfg.obs_dim, a->cfg.hidden_dim, a->actor.out_dim) && !(tune_variant() & 256);
}
static int act_fast_launch(exorl_agent* a, const float* obs_dev, const float* obs_host, int n, float stddev, int eval_mode, const float* noise_dev,
                           const float* noise_host, float* out, hipStream_t s) {
    EXORL_REQUIRE(eval_mode || stddev > 0.f, "agent_act: stddev must be > 0 in sampling mode");
    const NetDesc& d = a->actor;
    ActFast f{};
    f.x_dev = obs_dev; f.x_host = obs_host; f.w0t = a->sh_actor.w0t; f.P = a->flat[EXORL_NET_ACTOR][EXORL_T_PARAM];
    f.b0 = d.b0; f.g = d.g; f.beta = d.beta; f.W1 = d.W1; f.b1 = d.b1; f.W2 = d.W2; f.b2 = d.b2;
    f.part = a->act_part; f.ticket = a->act_ticket; f.noise_dev = noise_dev; f.noise_host = noise_host;
    f.seed = a->cfg.seed;
    f.counter = (eval_mode || noise_dev || noise_host) ? 0ull : ((1ull << 63) | a->act_noise_counter++);      // act_noise_spec's counter space
    f.out = out; f.stddev = stddev; f.rows = n; f.in_dim = a->cfg.obs_dim; f.H = a->cfg.hidden_dim; f.nout = d.out_dim; f.eval_mode = eval_mode;
    return act_fast(f, s);
}

int exorl_agent_act_host(exorl_agent_t* a, const float* obs_host, int32_t n, float stddev, int32_t eval_mode, const float* noise_host,
                         float* action_out, void* stream) {
    EXORL_REQUIRE(a && obs_host && action_out && n > 0, "agent_act_host: bad arguments");
    EXORL_REQUIRE(act_fast_ok(a, n), "agent_act_host: needs <= %d rows, a tanh-mean policy (not CQL), obs_dim <= 256 and hidden_dim %% 4 == 0; use exorl_agent_act",
                  ACT_FAST_ROWS);
    return act_fast_launch(a, nullptr, obs_host, n, stddev, eval_mode, nullptr, noise_host, action_out, as_stream(stream));
}

int exorl_agent_act(exorl_agent_t* a, const float* obs, int32_t n, float stddev, int32_t eval_mode, const float* noise,
                    float* out, void* stream) {
    EXORL_REQUIRE(a && obs && out && n > 0, "agent_act: bad arguments");
    hipStream_t s = as_stream(stream);
    const int O = a->cfg.obs_dim, A = a->cfg.act_dim;
    if (act_fast_ok(a, n)) return act_fast_launch(a, obs, nullptr, n, stddev, eval_mode, noise, nullptr, out, s);
    for (int r0 = 0; r0 < n; r0 += ACT_ROWS) {
        const int rows = n - r0 < ACT_ROWS ? n - r0 : ACT_ROWS;
        NetShadow sh = a->sh_actor;                 // act() rows do not tile by 64: split-bf16 takes the in-GEMM split here
        sh.w1l = sh.w0l = nullptr;
        EXORL_TRY(net_forward(a->actor, a->flat[EXORL_NET_ACTOR][EXORL_T_PARAM], sh, obs + (int64_t)r0 * O, O, rows, a->fact, false,
                              a->cfg.kind != EXORL_AGENT_CQL,
                              a->cfg.precision, s));
        if (a->cfg.kind == EXORL_AGENT_CQL) {       // SquashedNormal: mean = tanh(loc), sample = tanh(loc + std z)  (cql.py:122-131)
            EXORL_TRY(cql_act(a->fact.out, noise ? noise + (int64_t)r0 * A : nullptr, a->cfg.seed, (1ull << 63) | a->act_noise_counter++,
                              eval_mode, out + (int64_t)r0 * A, rows, A, s));
        } else if (eval_mode) {
            EXORL_CHECK_HIP(hipMemcpyAsync(out + (int64_t)r0 * A, a->fact.out, (size_t)rows * A * 4, hipMemcpyDeviceToDevice, s));
        } else {
            EXORL_REQUIRE(stddev > 0.f, "agent_act: stddev must be > 0 in sampling mode");
            EXORL_TRY(sample_action(a->fact.out, act_noise_spec(a, noise ? noise + (int64_t)r0 * A : nullptr), stddev, 0.f, 0,
                                    out + (int64_t)r0 * A, A, rows, A, nullptr, s));
        }
    }
    return 0;
}

int exorl_agent_metrics(exorl_agent_t* a, float* host, void* stream) {
    EXORL_REQUIRE(a && host, "agent_metrics: null argument");
    hipStream_t s = as_stream(stream);
    EXORL_CHECK_HIP(hipMemcpyAsync(host, a->metrics, EXORL_N_METRICS * sizeof(float), hipMemcpyDeviceToHost, s));
    EXORL_CHECK_HIP(hipStreamSynchronize(s));
    return 0;
}

int exorl_agent_opt_steps(exorl_agent_t* a, int64_t* actor_steps, int64_t* critic_steps) {
    EXORL_REQUIRE(a, "agent_opt_steps: null handle");
    if (actor_steps) *actor_steps = a->actor_t;
    if (critic_steps) *critic_steps = a->critic_t;
    return 0;
}

int exorl_agent_set_opt_steps(exorl_agent_t* a, int64_t actor_steps, int64_t critic_steps) {
    EXORL_REQUIRE(a && actor_steps >= 0 && critic_steps >= 0, "agent_set_opt_steps: bad arguments");
    a->actor_t = actor_steps;
    a->critic_t = critic_steps;
    return push_opt_steps(a);
}

int exorl_agent_enable_graph(exorl_agent_t* a, exorl_replay_t* r, int32_t nstep, float gamma, float stddev, void* stream) {
    EXORL_REQUIRE(a && r, "agent_enable_graph: null argument");
    // a previous step (eager or a graph launch) may still be running on the caller's stream and reads the buffers the new graph is
    // built over; releasing a graph exec that is executing is not allowed either. Setup call: a full stream sync is fine here.
    EXORL_CHECK_HIP(hipStreamSynchronize(as_stream(stream)));
    EXORL_REQUIRE(a->cfg.world_size == 1 || a->comm, "agent_enable_graph: a data-parallel step needs the library's communicator (exorl_agent_set_comm) to be captured; "
                  "steps whose collectives run in the caller's code are enqueued eagerly");
    EXORL_REQUIRE(stddev > 0.f && nstep >= 1, "agent_enable_graph: bad stddev/nstep");
    EXORL_TRY(release_graph(a));
    if (!a->capture_stream) EXORL_CHECK_HIP(hipStreamCreateWithFlags(&a->capture_stream, hipStreamNonBlocking));
    exorl_batch_out slots;
    EXORL_TRY(exorl_agent_batch_slots(a, &slots));
    // what a sample call would do on the host side (episode table upload, pair buffer, nstep vs episode lengths) — no draw is spent
    EXORL_TRY(replay_prepare(r, a->cfg.batch, nstep, a->capture_stream));
    const uint64_t ctr = replay_philox_counter(r);
    EXORL_CHECK_HIP(hipMemcpyAsync(&a->state->replay_counter, &ctr, sizeof(ctr), hipMemcpyHostToDevice, a->capture_stream));
    EXORL_TRY(set_device_float(&a->state->stddev, stddev, a->capture_stream));
    a->dev_stddev = stddev;
    EXORL_CHECK_HIP(hipStreamSynchronize(a->capture_stream));
    const int64_t t_a = a->actor_t, t_c = a->critic_t;
    if (!a->fk.aux) {
        EXORL_CHECK_HIP(hipStreamCreateWithFlags(&a->fk.aux, hipStreamNonBlocking));
        EXORL_CHECK_HIP(hipEventCreateWithFlags(&a->fk.ev_fork, hipEventDisableTiming));
        EXORL_CHECK_HIP(hipEventCreateWithFlags(&a->fk.ev_join, hipEventDisableTiming));
    }
    EXORL_TRY(ensure_opt_fork(a));
    EXORL_CHECK_HIP(hipStreamBeginCapture(a->capture_stream, hipStreamCaptureModeThreadLocal));
    a->capturing = true;
    a->fk.on = a->parallel_branches;
    // fp32 state observations: the gather kernel writes the networks' staged inputs itself and runs step_begin (one kernel
    // boundary less per step); the replay counter is then advanced by the step's last kernel (the actor's optimiser pass)
    StageOut stage{a->xa, a->xc_cur, a->xc_next, a->xc_pi, a->cfg.obs_dim, a->cfg.act_dim, a->cfg.batch, a->has_critic ? 1 : 0, a->state};
    a->staged_by_sampler = replay_obs_bytes(r) == a->cfg.obs_dim * 4;
    int rc = replay_sample_impl(r, a->cfg.batch, nstep, gamma, EXORL_SAMPLER_PHILOX, nullptr, &slots, nullptr, a->capture_stream,
                                &a->state->replay_counter, a->staged_by_sampler ? &stage : nullptr);
    a->fuse_opt = !a->comm && !a->fk.on;
    a->fo.on = a->fuse_opt && opt_overlap_enabled();
    a->whole_step = true;
    if (rc == 0) rc = whole_step_body(a, stddev, nullptr, nullptr, a->capture_stream);      // RCCL's all-reduces are captured with the kernels around them
    a->fuse_opt = false;
    a->fo.on = false;
    a->w1_early[0] = a->w1_early[1] = false;
    a->whole_step = false;
    a->staged_by_sampler = false;
    a->capturing = false;
    a->fk.on = false;
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(a->capture_stream, &g);
    a->actor_t = t_a; a->critic_t = t_c;             // capture enqueued nothing: undo the host-side bookkeeping
    replay_advance_philox(r, (uint64_t)-1);
    if (rc != 0) { if (g) (void)hipGraphDestroy(g); return rc; }
    if (e != hipSuccess) { set_error("agent_enable_graph: hipStreamEndCapture -> %s", hipGetErrorString(e)); return 1; }
    a->graph = g;
    EXORL_CHECK_HIP(hipGraphInstantiate(&a->graph_exec, g, nullptr, nullptr, 0));
    a->graph_replay = r;
    return 0;
}

int exorl_agent_noise_counter(exorl_agent_t* a, uint64_t* counter_out, void* stream) {
    EXORL_REQUIRE(a && counter_out, "agent_noise_counter: null argument");
    hipStream_t s = as_stream(stream);
    EXORL_CHECK_HIP(hipMemcpyAsync(counter_out, &a->state->noise_counter, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    EXORL_CHECK_HIP(hipStreamSynchronize(s));
    return 0;
}

int exorl_agent_cql_alpha(exorl_agent_t* a, float* host, int32_t set) {
    EXORL_REQUIRE(a && host && a->cql, "agent_cql_alpha: not a CQL agent / null argument");
    const int nsc = a->cfg.use_critic_lagrange ? 2 : 1;      // [1] = log_critic_alpha (cql.py:103-105)
    for (int i = 0; i < nsc; ++i) {
        float* h = host + 3 * i;
        if (set) {
            CqlScalars sc{h[0], h[1], h[2], expf(h[0])};
            EXORL_CHECK_HIP(hipMemcpy(a->cql + i, &sc, sizeof(sc), hipMemcpyHostToDevice));
        } else {
            CqlScalars sc;
            EXORL_CHECK_HIP(hipMemcpy(&sc, a->cql + i, sizeof(sc), hipMemcpyDeviceToHost));
            h[0] = sc.log_alpha; h[1] = sc.m; h[2] = sc.v;
        }
    }
    return 0;
}

int exorl_agent_set_parallel_branches(exorl_agent_t* a, int32_t enable) {
    EXORL_REQUIRE(a, "agent_set_parallel_branches: null handle");
    EXORL_REQUIRE(!a->graph_exec, "agent_set_parallel_branches: disable the captured graph first");
    a->parallel_branches = enable != 0;
    return 0;
}

int exorl_agent_set_metrics(exorl_agent_t* a, int32_t enable) {
    EXORL_REQUIRE(a, "agent_set_metrics: null handle");
    EXORL_REQUIRE(!a->graph_exec, "agent_set_metrics: disable the captured graph first");
    a->want_metrics = enable != 0;
    return 0;
}

int exorl_agent_disable_graph(exorl_agent_t* a) {
    EXORL_REQUIRE(a, "agent_disable_graph: null handle");
    return release_graph(a);
}

// One captured step: replay sample (Philox) + update phases 0..3, one hipGraphLaunch.
int exorl_agent_step_graph(exorl_agent_t* a, float stddev, void* stream) {
    EXORL_REQUIRE(a && a->graph_exec, "agent_step_graph: no captured graph (call exorl_agent_enable_graph)");
    EXORL_REQUIRE(stddev > 0.f || a->cfg.kind == EXORL_AGENT_CQL, "agent_step_graph: stddev must be > 0");
    if (a->cfg.kind != EXORL_AGENT_CQL && stddev != a->dev_stddev) {       // schedule moved: one scalar write ahead of the launch, same stream
        EXORL_TRY(set_device_float(&a->state->stddev, stddev, as_stream(stream)));
        a->dev_stddev = stddev;
    }
    EXORL_CHECK_HIP(hipGraphLaunch(a->graph_exec, as_stream(stream)));
    a->actor_t += 1;
    if (a->has_critic) a->critic_t += 1;
    replay_advance_philox(a->graph_replay, 1);
    return 0;
}

}  // extern "C"
