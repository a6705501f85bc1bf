/*
 * exorl_hip.h — C ABI of libexorl_hip.so, the MI355X (gfx950) backend for the exorl RL-update hot path.
 *
 * Drop-in boundary. Each entry point names the reference interface it replaces
 * (paths relative to the reference repo AOS55/exorl):
 *
 *   exorl_replay_*          utils/replay_buffer.py:153-239  ReplayBuffer (_store_episode, eviction,
 *                           _sample: episode pick + start index + gather + n-step return) and
 *                           :241-277 make_replay_loader / DataLoader collate (batched output)
 *   exorl_agent_create      agents/offline_learning/td3_bc.py:59-100 (and td3.py, bc.py,
 *                           agents/unsupervised_learning/ddpg.py:126-197) constructors
 *   exorl_agent_update*     td3_bc.py:119-189, td3.py:117-186, bc.py:78-110, ddpg.py:240-328
 *                           (update_critic / update_actor / soft_update_params / update)
 *   exorl_agent_act         td3_bc.py:107-117, ddpg.py:221-238
 *   exorl_adam_step         torch.optim.Adam.step as used at td3_bc.py:96-97,142,160
 *   exorl_soft_update       utils/utils.py:44-47
 *   exorl_knn_*             utils/utils.py:279-319 (PBE) and unsupervised_learning/proto.py:114-119
 *
 * Conventions: plain pointers and sizes only (no torch types). Every function returns 0 on success,
 * non-zero on error; exorl_last_error() returns the message of the calling thread's last error.
 * All device work is enqueued on the `stream` argument (a hipStream_t passed as void*; NULL = the
 * default stream) and is asynchronous unless stated. Pointers named *_dev are device pointers,
 * *_host host pointers. One host thread per GPU; handles are not thread-safe.
 */
#ifndef EXORL_HIP_H
#define EXORL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EXORL_ABI_VERSION 8      /* 8 (round 3): EXORL_PREC_BF16X6, exorl_agent_act_host, exorl_debug_precision_override */

const char* exorl_last_error(void);
int exorl_abi_version(void);
/* device name / CU count of the current HIP device (fails if none). */
int exorl_device_info(char* name_out, int name_len, int* num_cus, int64_t* hbm_bytes);

/* ------------------------------------------------------------------------------------------------
 * Replay: episodic buffer resident in HBM.
 * Arena layout (SoA, row = one time-step; episode = len+1 consecutive rows, row 0 the dummy reset step):
 *   obs[rows][obs_bytes]  action[rows][act_dim] f32  reward[rows] f32  discount[rows] f32  meta[rows][meta_dim] f32
 * ---------------------------------------------------------------------------------------------- */
typedef struct exorl_replay exorl_replay_t;

typedef struct {
    int32_t obs_bytes;      /* bytes per observation row: O*4 (f32 states) or C*H*W (u8 pixels); multiple of 4 */
    int32_t act_dim;
    int32_t meta_dim;       /* total width of the concatenated meta columns (0 = none) */
    int32_t max_episodes;
    int64_t capacity_rows;  /* arena rows = transitions + one dummy row per episode */
} exorl_replay_cfg;

typedef struct {            /* where one sampled minibatch is written (device memory, caller-owned) */
    void*   obs;       int64_t obs_stride;       /* bytes between consecutive samples */
    float*  action;    int64_t action_stride;    /* floats between consecutive samples */
    float*  reward;                              /* (B) contiguous */
    float*  discount;                            /* (B) contiguous */
    void*   next_obs;  int64_t next_obs_stride;  /* bytes */
    float*  meta;      int64_t meta_stride;      /* floats; may be NULL when meta_dim == 0 */
} exorl_batch_out;

#define EXORL_SAMPLER_MT19937 0   /* reference-exact index stream (CPython random + NumPy legacy RandomState) */
#define EXORL_SAMPLER_PHILOX  1   /* device-side counter-based stream, same distribution */
#define EXORL_SAMPLER_GIVEN   2   /* (episode position, start idx) pairs supplied by the caller */

int exorl_replay_create(const exorl_replay_cfg* cfg, exorl_replay_t** out);
int exorl_replay_destroy(exorl_replay_t* r);
/* Copies one episode (rows = len+1 host rows per array) into the arena; returns its slot id. */
int exorl_replay_append_episode(exorl_replay_t* r, const void* obs_host, const float* act_host,
                                const float* rew_host, const float* disc_host, const float* meta_host,
                                int32_t rows, int32_t* slot_out);
int exorl_replay_evict(exorl_replay_t* r, int32_t slot);
/* Sampling order of resident episodes = the reference's sorted `_episode_fns` list (replay_buffer.py:184). */
int exorl_replay_set_order(exorl_replay_t* r, const int32_t* slots_host, int32_t n);
int exorl_replay_num_rows(exorl_replay_t* r, int64_t* live_rows, int64_t* used_rows);
/* MT19937 states as exposed by random.getstate()[1] / np.random.get_state()[1:3]. */
int exorl_replay_seed_mt(exorl_replay_t* r, const uint32_t* py_key624, int32_t py_pos,
                         const uint32_t* np_key624, int32_t np_pos);
/* Convenience for callers without a Python interpreter: random.seed(py_seed); np.random.seed(np_seed). */
int exorl_replay_seed_mt_ints(exorl_replay_t* r, uint64_t py_seed, uint32_t np_seed);
int exorl_replay_get_mt(exorl_replay_t* r, uint32_t* py_key624, int32_t* py_pos,
                        uint32_t* np_key624, int32_t* np_pos);
int exorl_replay_seed_philox(exorl_replay_t* r, uint64_t seed);
/* One minibatch: B x (obs[idx-1], action[idx], n-step reward, n-step discount, obs[idx+n-1], meta[idx-1]).
 * pairs_host: for EXORL_SAMPLER_GIVEN, B x {position in order, start idx >= 1};
 * pairs_out_host: if non-NULL receives the pairs drawn (MT19937 mode only; host-generated). */
int exorl_replay_sample(exorl_replay_t* r, int32_t batch, int32_t nstep, float gamma, int32_t sampler,
                        const int32_t* pairs_host, const exorl_batch_out* out, int32_t* pairs_out_host,
                        void* stream);

/* Synchronous: the (position, start idx) pairs of the last sample call, read back from the device. */
int exorl_replay_last_pairs(exorl_replay_t* r, int32_t batch, int32_t* pairs_host, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Data-parallel communicator: RCCL over xGMI, one per process / GPU. The reference has no distributed path; these
 * are the exchanges its single-process update implies when the batch is sharded: critic gradients before critic_opt.step()
 * (td3_bc.py:140-142), the batch-global sum |Q| behind lambda (:154), actor gradients before actor_opt.step() (:158-160).
 * ---------------------------------------------------------------------------------------------- */
typedef struct exorl_comm exorl_comm_t;
#define EXORL_COMM_ID_BYTES 128
/* Rank 0 makes the id and hands the bytes to the other ranks by any channel (torch.distributed broadcast, MPI, a file). */
int exorl_comm_unique_id(void* id_out_host /* EXORL_COMM_ID_BYTES */);
/* Collective over all ranks, on the caller's current HIP device. */
int exorl_comm_init(int32_t rank, int32_t nranks, const void* id_host, exorl_comm_t** out);
int exorl_comm_destroy(exorl_comm_t* c);
/* In-place float32 sum all-reduce, enqueued on `stream`. */
int exorl_comm_allreduce(exorl_comm_t* c, float* buf_dev, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Agents
 * ---------------------------------------------------------------------------------------------- */
typedef struct exorl_agent exorl_agent_t;

#define EXORL_AGENT_TD3_BC 0
#define EXORL_AGENT_TD3    1
#define EXORL_AGENT_BC     2
#define EXORL_AGENT_DDPG   3   /* states; shared-trunk critic (ddpg.py:79-123) */
#define EXORL_AGENT_CRR    4   /* agents/offline_learning/crr.py:59-219 */
#define EXORL_AGENT_CQL    5   /* agents/offline_learning/cql.py:59-286 */
#define EXORL_AGENT_APS    6   /* agents/unsupervised_learning/aps.py:12-60,268-320: DDPG with the successor-feature critic
                                  (heads emit sf_dim features, Q = task . features; the task is the last sf_dim columns of obs) */

#define EXORL_CRR_IDENTITY  0   /* crr.py:132-142 adv_transform */
#define EXORL_CRR_INDICATOR 1
#define EXORL_CRR_EXP       2

#define EXORL_PREC_F32  0      /* v_mfma_f32_32x32x2_f32: exact fp32 products, parity mode */
#define EXORL_PREC_BF16 1      /* v_mfma_f32_32x32x16_bf16: bf16 operands, fp32 accumulate, fp32 master weights */
#define EXORL_PREC_BF16X3 2    /* split-bf16: every GEMM operand x = hi + lo (two bf16), product = hi*hi + hi*lo + lo*hi on the bf16 MFMA,
                                  fp32 accumulate; everything else as EXORL_PREC_F32. ~2^-16 relative product error: per-step losses stay
                                  within the 1e-4 parity bar of the fp32 reference (tests/test_gpu_agent.py) at about twice fp32 mode's rate */
#define EXORL_PREC_BF16X6 3    /* three-plane split-bf16 (pixel agents and intrinsic modules only): x = hi + mid + lo, three bf16 planes = the 24
                                  significand bits of fp32 held exactly; product = hi*hi + (hi*mid + mid*hi) + (hi*lo + lo*hi + mid*mid) on the bf16
                                  MFMA, fp32 accumulate, the three dropped terms <= 3 * 2^-24 per product — fp32-grade products at 6 / 16 of the fp32
                                  MFMA's cycles. Generic GEMMs and the 32-channel convolutions (forward, dgrad); the convolution weight gradients
                                  and the first layer run as in EXORL_PREC_F32. */

#define EXORL_NET_ACTOR         0
#define EXORL_NET_CRITIC        1
#define EXORL_NET_CRITIC_TARGET 2

#define EXORL_T_PARAM 0
#define EXORL_T_GRAD  1
#define EXORL_T_ADAM_M 2
#define EXORL_T_ADAM_V 3

/* metrics[] layout written by exorl_agent_metrics (same keys as the reference's metrics dict) */
#define EXORL_M_BATCH_REWARD    0
#define EXORL_M_CRITIC_TARGET_Q 1
#define EXORL_M_CRITIC_Q1       2
#define EXORL_M_CRITIC_Q2       3
#define EXORL_M_CRITIC_LOSS     4
#define EXORL_M_ACTOR_LOSS      5
#define EXORL_M_ACTOR_LOGPROB   6
#define EXORL_M_Q_ABS_SUM       7   /* local sum |Q| (DP scalar all-reduce operand) */
#define EXORL_M_Q_SUM           8   /* local sum  Q  */
#define EXORL_M_BC_SUM          9   /* local sum (mu-a)^2 */
#define EXORL_M_CRITIC_CQL        10
#define EXORL_M_CRITIC_CQL_LOGSUM 11
#define EXORL_M_ACTOR_ALPHA       12
#define EXORL_M_ACTOR_ALPHA_LOSS  13
#define EXORL_M_ACTOR_ENT         14
#define EXORL_N_METRICS        16

typedef struct {
    int32_t kind;
    int32_t obs_dim, act_dim, hidden_dim;
    int32_t batch;            /* rows per update on THIS rank */
    int32_t precision;        /* EXORL_PREC_* */
    int32_t world_size;       /* data-parallel ranks; losses are means over batch*world_size */
    int32_t sf_dim;           /* APS: successor-feature width (aps.yaml: 10); obs_dim includes it */
    float   lr, tau, alpha, stddev_clip;
    uint64_t seed;            /* Philox stream for action noise when no noise buffer is given */
    int32_t num_value_samples; /* CRR: actions sampled per state for V(s) (crr.yaml: 10) */
    int32_t weight_func;       /* CRR: EXORL_CRR_* */
    int32_t n_samples;         /* CQL: action samples per source (cql.yaml: 3); `alpha` is then the CQL penalty weight */
    int32_t use_critic_lagrange; /* CQL: learn the penalty weight (cql.py:201-213); data parallel through exorl_agent_update_phase 4 / 5 (exorl_agent_update does that itself when a communicator is set) */
    float   target_cql_penalty;  /* CQL Lagrange target (cql.yaml: 5.0) */
    int32_t reserved3;
} exorl_agent_cfg;

size_t exorl_agent_workspace_bytes(const exorl_agent_cfg* cfg);
/* workspace_dev: caller-allocated device memory of at least exorl_agent_workspace_bytes (e.g. a torch
 * tensor's data_ptr) or NULL to let the library hipMalloc it. Contents are zero-initialised. */
int exorl_agent_create(const exorl_agent_cfg* cfg, void* workspace_dev, size_t workspace_bytes,
                       exorl_agent_t** out);
int exorl_agent_destroy(exorl_agent_t* a);
/* Number of parameter tensors of a net, in the reference's nn.Module.parameters() order. */
int exorl_agent_num_tensors(exorl_agent_t* a, int32_t net, int32_t* n);
/* Device pointer + logical shape (rows x cols; cols = 1 for vectors) of one tensor. */
int exorl_agent_tensor(exorl_agent_t* a, int32_t net, int32_t index, int32_t what,
                       void** ptr_dev, int64_t* rows, int64_t* cols);
/* The flat (padded) buffer that holds all tensors of a net back to back: all-reduce operand. */
int exorl_agent_flat(exorl_agent_t* a, int32_t net, int32_t what, void** ptr_dev, int64_t* numel);
/* Must be called after parameters were written from outside (load_state_dict, init): refreshes derived
 * copies (bf16 shadows) and, if sync_target != 0, copies critic -> critic_target (td3_bc.py:93). */
int exorl_agent_params_changed(exorl_agent_t* a, int32_t sync_target, void* stream);
/* Where the sampler should write this agent's minibatch (zero-copy hand-off replay -> update). */
int exorl_agent_batch_slots(exorl_agent_t* a, exorl_batch_out* out);
/* Copies a caller-provided device batch (contiguous f32 (B,O) (B,A) (B) (B) (B,O)) into the batch slots. */
int exorl_agent_set_batch(exorl_agent_t* a, const float* obs_dev, const float* action_dev,
                          const float* reward_dev, const float* discount_dev, const float* next_obs_dev,
                          void* stream);
/* One gradient step on the batch currently in the batch slots.
 * noise_critic_dev / noise_actor_dev: (B,A) standard-normal draws (reference order; for CRR the second
 * draw is (B*num_value_samples, A), crr.py:125) or NULL for device Philox.
 * CQL (draw order cql.py:159,170-176,238): noise_critic_dev = [z_next (B,A) | u_rand (n,B,A) uniform(-1,1) | z_cur (n,B,A) |
 * z_nxt (n,B,A)] back to back, noise_actor_dev = z_actor (B,A). Runs phases 0..3 back to back (world_size 1). */
int exorl_agent_update(exorl_agent_t* a, float stddev, const float* noise_critic_dev,
                       const float* noise_actor_dev, void* stream);
/* Data-parallel form: the caller all-reduces (sum) between phases:
 *   phase 0 -> critic grads ready       (all-reduce EXORL_NET_CRITIC / EXORL_T_GRAD flat buffer)
 *   phase 1 -> critic Adam + soft update, actor-side Q statistics ready (all-reduce the 4-float stats buffer)
 *   phase 2 -> actor grads ready        (all-reduce EXORL_NET_ACTOR / EXORL_T_GRAD flat buffer)
 *   phase 3 -> actor Adam
 * BC has phases 2 and 3 only.
 * CQL with use_critic_lagrange under data parallelism (cql.py:199-213: the multiplier's own step needs the penalty of the GLOBAL batch before
 * any critic gradient exists): phase 0 is driven as
 *   phase 4 -> forwards done, this rank's penalty sums in the stats buffer (all-reduce the 4-float stats buffer)
 *   phase 5 -> multiplier stepped from the global penalty, critic grads ready (then as after phase 0). */
int exorl_agent_update_phase(exorl_agent_t* a, int32_t phase, float stddev, const float* noise_critic_dev,
                             const float* noise_actor_dev, void* stream);
int exorl_agent_stats_buffer(exorl_agent_t* a, void** ptr_dev, int64_t* numel);
/* Attach a communicator of cfg.world_size ranks (NULL detaches): exorl_agent_update then runs the four phases AND the all-reduces
 * between them on `stream` — no host round trip inside a step. The 16-byte statistic is reduced on a second stream while the critic's
 * backward pass runs (lambda enters only at the actor head: the critic backward is linear in its output gradient). */
int exorl_agent_set_comm(exorl_agent_t* a, exorl_comm_t* c);
/* CQL: entropy temperature state (log_actor_alpha and its Adam moments), host <-> device: host[0..2]; with
 * use_critic_lagrange also host[3..5] = log_critic_alpha and its Adam moments (pass a 6-float array). */
int exorl_agent_cql_alpha(exorl_agent_t* a, float* log_alpha_host, int32_t set);
/* Policy inference for n rows: out = mean (eval) or TruncatedNormal sample (clip=None). */
int exorl_agent_act(exorl_agent_t* a, const float* obs_dev, int32_t n, float stddev, int32_t eval_mode,
                    const float* noise_dev, float* action_out_dev, void* stream);
/* The same for the online loop's one observation per environment step (pretrain.py:271-283 -> ddpg.py:221-238 / td3_bc.py:107-117): obs_host
 * (n <= 2 rows) and noise_host (or NULL -> device Philox) are HOST pointers copied into the kernel arguments, `action_out` is any
 * device-accessible address — pinned host memory makes the result visible after a stream synchronise with no copy. One kernel launch, no other
 * device work. Not for CQL's tanh-Gaussian policy (exorl_agent_act handles it). */
int exorl_agent_act_host(exorl_agent_t* a, const float* obs_host, int32_t n, float stddev, int32_t eval_mode,
                         const float* noise_host, float* action_out, void* stream);
/* Synchronous: copies the metric block of the last update to host. */
int exorl_agent_metrics(exorl_agent_t* a, float* metrics_host, void* stream);
/* The reference computes its metrics dict only under use_tb (td3_bc.py:133,162,175); enable = 0 skips the metric
 * reductions of the step (default: enabled). */
int exorl_agent_set_metrics(exorl_agent_t* a, int32_t enable);
/* Captured graphs run independent parts of the step (target || critic forward, wgrad || dgrad chain) as parallel
 * branches on a second stream (default: off — one chain measured faster on MI355X). */
int exorl_agent_set_parallel_branches(exorl_agent_t* a, int32_t enable);
int exorl_agent_opt_steps(exorl_agent_t* a, int64_t* actor_steps, int64_t* critic_steps);
int exorl_agent_set_opt_steps(exorl_agent_t* a, int64_t actor_steps, int64_t critic_steps);
/* Captures exorl_replay_sample(PHILOX) into the agent's batch slots + exorl_agent_update into one hipGraph;
 * exorl_agent_step_graph replays it (all per-step counters, Adam scalars and the exploration std live in device memory, so a
 * stddev schedule — utils.schedule, td3_bc.py:169 — that moves every step costs one scalar write, not a re-capture).
 * enable_graph synchronises `stream` (the caller's stream, where earlier steps may still run) before touching anything and does not
 * consume a Philox batch. step_graph launches on `stream`; stddev is the schedule's value for this step. */
int exorl_agent_enable_graph(exorl_agent_t* a, exorl_replay_t* r, int32_t nstep, float gamma, float stddev, void* stream);
int exorl_agent_step_graph(exorl_agent_t* a, float stddev, void* stream);
int exorl_agent_disable_graph(exorl_agent_t* a);
/* Test hooks for the device-side noise stream (synchronous): the Philox draw counter after the steps enqueued so far, and the
 * standard-normal block the update kernels generate for (seed, counter) — out[e], e = row * act_dim + column. With these a test
 * can replay a captured-graph trajectory (device sampler + device noise) through the CPU oracle. */
int exorl_agent_noise_counter(exorl_agent_t* a, uint64_t* counter_out, void* stream);
int exorl_debug_philox_normal(uint64_t seed, uint64_t counter, int64_t n, float* out_dev, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Stand-alone operators (used by the agents above; exported for tests and for callers' own nets)
 * ---------------------------------------------------------------------------------------------- */
/* C[M,N] = op(A) * op(B) (+bias[N]) (ReLU) (C +=), fp32 in memory.
 * a_layout: 0 = A stored [M][K] (lda >= K), 1 = A stored [K][M] (lda >= M)
 * b_layout: 0 = B stored [N][K] (ldb >= K)  (nn.Linear weight), 1 = B stored [K][N] (ldb >= N) */
int exorl_gemm(int32_t precision, int32_t a_layout, int32_t b_layout, int32_t M, int32_t N, int32_t K,
               const float* A_dev, int64_t lda, const float* B_dev, int64_t ldb, float* C_dev, int64_t ldc,
               const float* bias_dev, int32_t relu, int32_t accumulate, void* stream);
/* Same with A and B stored as bf16 in memory (the fast mode's operand format); fp32 accumulate and output. */
int exorl_gemm_bf16(int32_t a_layout, int32_t b_layout, int32_t M, int32_t N, int32_t K, const uint16_t* A_dev, int64_t lda,
                    const uint16_t* B_dev, int64_t ldb, float* C_dev, int64_t ldc, const float* bias_dev, int32_t relu,
                    int32_t accumulate, void* stream);
/* Grouped form on bf16 planes, as the agents launch it: up to 4 problems of one shape in ONE launch, each operand given as a hi plane and
 * (split-bf16) a lo plane with x = hi + lo, or lo == NULL for plain bf16. a_layouts[i] selects problem i's A layout (a Linear's wgrad
 * and dgrad share a launch with different A layouts); b_layout is common. No bias. Pointer arrays are host arrays of device pointers. */
int exorl_gemm_planes(int32_t count, const int32_t* a_layouts, int32_t b_layout, int32_t M, int32_t N, int32_t K,
                      const uint16_t* const* A_hi_dev, const uint16_t* const* A_lo_dev, int64_t lda,
                      const uint16_t* const* B_hi_dev, const uint16_t* const* B_lo_dev, int64_t ldb,
                      float* const* C_dev, int64_t ldc, int32_t relu, void* stream);
/* Tuning switch for the bf16-operand GEMM (tools/micro/gemm_bench.py): -1 = default heuristics. */
int exorl_gemm_tune(int32_t variant);
/* Diagnostic only (tools/debug/config4_ablation.py): in EXORL_PREC_BF16X3 mode, run the selected product families with exact fp32 products.
 * bits 1/2 forward GEMM narrow/wide, 4/8 wgrad, 16/32 dgrad ("wide": a dimension >= 8192), 64/128/256 convolution forward/dgrad/wgrad.
 * No reference counterpart; the product never calls it. */
int exorl_debug_precision_override(int32_t mask);
/* Diagnostic (tools/micro/stamp_bench.py; tuning bit 33554432 selects the stamped build of the forward H x H GEMM): per workgroup
 * {s_memtime x 4, s_memrealtime x 4} at entry / first k-step / last k-step / stores drained; 8 words per workgroup, <= 1024 workgroups. */
int exorl_debug_gemm_stamps(uint64_t* out_host, int32_t n_words);
/* Diagnostic (tools/micro/conv_stamp_bench.py; tuning bit 8192 selects the stamped build of the 32 -> 32 forward convolution): per
 * workgroup (= image) and for waves 0 and 7, shader-clock cycles summed over the image's passes: {convert + LDS write, barrier, fetch
 * issue, MFMA loop incl. its LDS reads, stores, barrier, whole kernel, 0}; 16 words per workgroup, <= 1024 workgroups. */
int exorl_debug_conv_stamps(uint64_t* out_host, int32_t n_words);
/* Measurement hook (bench.py roofline leg): time every GEMM launch with HIP events on its own stream. */
int exorl_profile_gemm(int32_t enable);
int exorl_profile_gemm_read(double* flops_out_host, float* ms_out_host, int32_t cap, int32_t* n_out);
int exorl_profile_event_overhead(float* ms_out_host, void* stream);
int exorl_adam_step(float* p_dev, const float* g_dev, float* m_dev, float* v_dev, int64_t n, float lr,
                    float beta1, float beta2, float eps, int64_t t, float* target_dev, float tau, void* stream);
int exorl_soft_update(const float* p_dev, float* target_dev, int64_t n, float tau, void* stream);
int exorl_ln_tanh_fwd(const float* z_dev, const float* gain_dev, const float* beta_dev, float* h_dev,
                      float* xhat_dev, float* rstd_dev, int32_t rows, int32_t H, void* stream);
/* kNN particle-entropy building block: out[i][j] = j-th smallest L2 distance from src row i to the tgt rows (sorted). */
int exorl_knn_topk(const float* src_dev, int32_t n_src, const float* tgt_dev, int32_t n_tgt, int32_t dim,
                   int32_t k, float* out_dev, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Intrinsic-reward modules of the reward-free DDPG-backbone agents (states observations).
 * Replaces, per update() of the reference agent: the module's optimiser step and compute_intr_reward
 *   RND      agents/unsupervised_learning/rnd.py:79-108   (module rnd.py:13-60)
 *   ICM      agents/unsupervised_learning/icm.py:64-92    (module icm.py:12-45)
 *   ICM-APT  agents/unsupervised_learning/icm_apt.py:86-110 (module icm_apt.py:13-57; utils.PBE/RMS utils/utils.py:257-319)
 * The DDPG critic/actor update that follows is exorl_agent_update on the same batch with reward_out as its reward.
 * ------------------------------------------------------------------------------------------- */
#define EXORL_INTR_RND     0
#define EXORL_INTR_ICM     1
#define EXORL_INTR_ICM_APT 2
#define EXORL_INTR_DISAGREEMENT 3   /* agents/unsupervised_learning/disagreement.py:11-90 */
#define EXORL_INTR_DIAYN   4        /* agents/unsupervised_learning/diayn.py:15-127 */
#define EXORL_INTR_PROTO   5        /* agents/unsupervised_learning/proto.py:14-207 (state observations) */
#define EXORL_INTR_APS     6        /* agents/unsupervised_learning/aps.py:63-79,147-175 (the task rides in `skill`) */
#define EXORL_INTR_SMM     7        /* agents/unsupervised_learning/smm.py:27-112,173-246 (states; obs rows are [obs | z], z also in `skill`) */
#define EXORL_MAX_ENSEMBLE 8
#define EXORL_INTR_ENCODED 1       /* exorl_intr_cfg.flags */

typedef struct exorl_intr_cfg {
    int32_t kind;         /* EXORL_INTR_* */
    int32_t obs_dim, act_dim, hidden_dim;
    int32_t rep_dim;      /* rnd_rep_dim / icm_rep_dim / DIAYN skill_dim / APS sf_dim (unused by plain ICM and Disagreement) */
    int32_t batch;
    int32_t precision;    /* EXORL_PREC_* (MFMA operand type of the module's GEMMs) */
    int32_t knn_k, knn_avg, knn_rms;   /* ICM-APT: utils.PBE arguments (configs/agent/icm_apt.yaml) */
    int32_t n_models;     /* Disagreement ensemble size (0 -> 5, disagreement.py:12) */
    int32_t flags;        /* EXORL_INTR_ENCODED: the observation rows are encodings of pixel frames (obs_type == 'pixels') — SMM drops the goal prior
                             p*(s) from its reward (smm.py:232-235); RND takes the rows as already normalised (BatchNorm2d ran on the frames,
                             rnd.py:26-27,47-50) and feeds its frozen target net from next_obs, the frozen encoder copy's output (rnd.py:35-39) */
    float lr;             /* Adam, betas (0.9, 0.999), eps 1e-8 */
    float scale;          /* rnd_scale / icm_scale */
    float knn_clip;
    float clip_val;       /* RND: clamp of the BatchNorm-normalised observation (rnd.py:22, 5.0) */
    /* Proto (configs/agent/proto.yaml): rep_dim = pred_dim, hidden_dim = proj_dim, knn_k = topk */
    int32_t num_protos, queue_size;
    float tau;            /* softmax / Sinkhorn temperature */
    float target_tau;     /* encoder_target_tau: Polyak rate of predictor_target */
    /* SMM (configs/agent/smm.yaml, pretrain.yaml): rep_dim = z_dim; lr is unused, the two optimisers have their own rates */
    float sp_lr, vae_lr, vae_beta;
    float state_ent_coef, latent_ent_coef, latent_cond_ent_coef;
    float goal_x, goal_y; /* smm.py:139 self.goal = (150, 75): p*(s) is 1/dist of obs[:, :2] to it beyond distance 1 */
} exorl_intr_cfg;

typedef struct exorl_intr exorl_intr_t;

/* metric slots of exorl_intr_metrics */
#define EXORL_IM_LOSS        0   /* rnd_loss / icm_loss */
#define EXORL_IM_INTR_REWARD 1   /* mean intrinsic reward of the batch */
#define EXORL_IM_EXTR_REWARD 2   /* mean of the extrinsic reward passed in (0 if none) */
#define EXORL_IM_RMS_MEAN    3   /* running mean of the RMS (RND: pred_error_mean) */
#define EXORL_IM_RMS_STD     4   /* sqrt of its running variance (RND: pred_error_std) */
#define EXORL_IM_ACC         5   /* DIAYN: discriminator accuracy (diayn_acc) */
#define EXORL_IM_ENT_REWARD  6   /* APS: mean particle-entropy part of the reward (intr_ent_reward) */
#define EXORL_IM_SF_REWARD   7   /* APS: mean successor-feature part (intr_sf_reward) */
/* SMM reuses the slots: 0 loss_vae, 1 intr_reward, 2 extr_reward, 3 log_p_star (batch mean), 4 pred_log_ratios, 5 loss_pred,
 * 6 latent_cond_ent_coef term, 7 var_j(log_p_star_j) — the constant the reference's (B,B) reward broadcast adds to each critic's loss */
#define EXORL_N_INTR_METRICS 8

/* workspace: device memory of exorl_intr_workspace_bytes(cfg) bytes, 256-byte aligned, owned by the caller (so that the
 * parameters can be exposed as the caller's own tensors), or null to let the library allocate. */
size_t exorl_intr_workspace_bytes(const exorl_intr_cfg* cfg);
int exorl_intr_create(const exorl_intr_cfg* cfg, void* workspace, size_t workspace_bytes, exorl_intr_t** out);
int exorl_intr_destroy(exorl_intr_t* m);
/* Parameter tensors in the module's parameters() order (RND: predictor.{1,3,5}, target.{1,3,5}; ICM: forward_net.{0,2},
 * backward_net.{0,2}; ICM-APT: trunk.0, trunk.1 (LayerNorm), forward_net, backward_net; Disagreement: ensemble.{m}.{0,2};
 * DIAYN: skill_pred_net.{0,2,4}; APS: state_feat_net.{0,2,4}; SMM: z_pred_net.{0,2,4}, vae.enc.{0,2}, vae.enc_mu, vae.enc_logvar,
 * vae.dec.{0,2,4}; Proto: predictor, projector.trunk.{0,2}, protos (no bias), then the frozen predictor_target),
 * each weight then bias.
 * what = EXORL_T_*; RND's frozen target tensors have parameters only. */
int exorl_intr_num_tensors(exorl_intr_t* m, int32_t* n);
int exorl_intr_tensor(exorl_intr_t* m, int32_t index, int32_t what, void** ptr, int64_t* rows, int64_t* cols);
int exorl_intr_flat(exorl_intr_t* m, int32_t what, void** ptr, int64_t* numel);
/* Device state outside the parameters: rms = {float M, float S, double n} (utils.RMS); bn = running_mean[obs_dim],
 * running_var[obs_dim], num_batches_tracked (as float) of RND's BatchNorm1d, or null. */
int exorl_intr_state(exorl_intr_t* m, void** rms_dev, void** bn_dev, int64_t* bn_numel);
/* Proto's candidate queue (queue_size x pred_dim, device) and its write pointer (get: set == 0; restore: set != 0). */
int exorl_intr_queue(exorl_intr_t* m, void** queue_dev, int64_t* rows, int64_t* cols, int64_t* ptr_inout, int32_t set);
/* One sampled batch as device pointers with row strides in floats (so that columns of a wider matrix can be passed in place:
 * DIAYN's skill lives in the last columns of the agent's [obs | skill] rows). action / next_obs / skill / extr_reward may be
 * null where the module does not read them; reward_out (batch,) may alias extr_reward (the agent's reward slot). */
typedef struct exorl_intr_batch {
    const float* obs;      int64_t obs_ld;
    const float* action;   int64_t action_ld;
    const float* next_obs; int64_t next_obs_ld;
    const float* skill;    int64_t skill_ld;
    const float* extr_reward;
    float* reward_out;
    const float* next_obs_target; int64_t next_obs_target_ld;   /* Proto on pixels: encoder_target(next_obs) for the Sinkhorn branch
                                                                   (proto.py:144-148); null -> next_obs */
    float* dobs_out;            /* if set (train != 0): receives the loss gradient at the encoding the module's loss reaches the caller's encoder
                                   through, dense (batch, obs_dim), so that the caller continues the backward pass: d/d(obs rows) for Proto
                                   (proto_opt owns the encoder too, proto.py:75-78), ICM, ICM-APT and Disagreement (next_obs is encoded without a
                                   graph there, icm.py:97-99), SMM (through the VAE's input and its reconstruction target, smm.py:61-70) and RND with
                                   EXORL_INTR_ENCODED (the predictor's input); d/d(next_obs rows) for DIAYN and APS (diayn.py:78-92, aps.py:147-159) */
    const float* cat_uniform;   /* Proto: num_protos uniforms in [0,1) for Categorical(prob).sample() (proto.py:112); SMM: the VAE's
                                   epsilon, (batch, 128) standard normals (smm.py:62); null -> Philox */
} exorl_intr_batch;
/* train != 0: the module's optimiser step (update_rnd / update_icm / update_disagreement / update_diayn) then
 * compute_intr_reward under the updated module (rnd.py:121-124, icm.py:106-110, icm_apt.py:123-127, disagreement.py:106-112,
 * diayn.py:143-147); train == 0: compute_intr_reward only; train == 2 (Proto): the optimiser step only — with an encoder in front
 * the reward is computed from features re-encoded after that step (proto.py:173-177). */
int exorl_intr_update(exorl_intr_t* m, const exorl_intr_batch* batch, int32_t train, void* stream);
int exorl_intr_metrics(exorl_intr_t* m, float* host_out /* EXORL_N_INTR_METRICS */, void* stream);
/* optimiser step count of the module's Adam: set == 0 reads into *steps, else writes it (snapshot restore) */
int exorl_intr_opt_steps(exorl_intr_t* m, int64_t* steps, int32_t set);
/* Philox counter of the module's own random draws (Proto's categorical candidate picks, SMM's VAE epsilon): part of the pickled state */
int exorl_intr_counter(exorl_intr_t* m, uint64_t* counter_inout, int32_t set);

/* ---------------------------------------------------------------------------------------------
 * Pixel front end (building blocks; the pixel actor/critic heads are not wired into exorl_agent_* yet).
 *   utils.RandomShiftsAug  utils/utils.py:222-254      ddpg.Encoder  agents/unsupervised_learning/ddpg.py:12-39
 * ------------------------------------------------------------------------------------------- */
/* out (n, c, h, h) fp32 pixel values = RandomShiftsAug(pad)(x) for x (n, c, h, h) uint8. shifts_dev: (n, 2) int32 (x shift, y shift)
 * in [0, 2 pad] — what torch.randint draws at utils.py:244-248 — or null for Philox(seed, counter). */
int exorl_aug_shift(const unsigned char* x_dev, int32_t n, int32_t c, int32_t h, int32_t pad, const int32_t* shifts_dev, uint64_t seed,
                    uint64_t counter, float* out_dev, void* stream);
/* Encoder parameters live in one flat fp32 buffer in torch order convnet.{0,2,4,6}.{weight,bias} (each tensor padded to 4 floats). */
int64_t exorl_encoder_param_floats(int32_t c_in, int32_t hw);
int64_t exorl_encoder_out_dim(int32_t hw);                         /* repr_dim: 32*35*35 for 84x84, 32*25*25 for 64x64 */
int64_t exorl_encoder_workspace_floats(int32_t n, int32_t c_in, int32_t hw);
/* Encoder.forward on x (n, c_in, hw, hw) fp32 pixel values; *h_out_dev = the flattened features (n, repr_dim) inside ws_dev. */
int exorl_encoder_forward(const float* params_dev, int32_t c_in, int32_t hw, const float* x_dev, int32_t n, float* ws_dev, float** h_out_dev,
                          void* stream);
/* Backward after exorl_encoder_forward with the same x / ws: dh_dev (n, repr_dim) is overwritten; parameter gradients go to
 * grads_dev (flat layout of the parameters). */
int exorl_encoder_backward(const float* params_dev, int32_t c_in, int32_t hw, const float* x_dev, int32_t n, float* ws_dev, float* dh_dev,
                           float* grads_dev, void* stream);
/* The same with the 32-channel stride-1 layers (forward and dgrad) on the matrix cores: precision = EXORL_PREC_BF16 / _BF16X3 runs them
 * as an implicit GEMM on v_mfma_f32_32x32x16_bf16 (split mode: hi*hi + hi*lo + lo*hi); EXORL_PREC_F32 = the two calls above. The first
 * layer (3 or 9 input channels, stride 2, uint8 scaling) and the weight gradients stay fp32 FMA. */
int exorl_encoder_forward_prec(const float* params_dev, int32_t c_in, int32_t hw, const float* x_dev, int32_t n, float* ws_dev, float** h_out_dev,
                               int32_t precision, void* stream);
int exorl_encoder_backward_prec(const float* params_dev, int32_t c_in, int32_t hw, const float* x_dev, int32_t n, float* ws_dev, float* dh_dev,
                                float* grads_dev, int32_t precision, void* stream);
int exorl_u8_to_f32(const unsigned char* x_dev, int64_t n, float* out_dev, void* stream);

/* ---------------------------------------------------------------------------------------------
 * DDPG on pixel observations (agents/unsupervised_learning/ddpg.py with obs_type == 'pixels'): augmentation + encoder +
 * pixel Actor (:42-76) / Critic (:79-123) and update (:240-328; the encoder steps with the critic's loss).
 * ------------------------------------------------------------------------------------------- */
typedef struct exorl_pixel_cfg {
    int32_t c_in, hw;          /* obs_shape = (c_in, hw, hw), hw in {84, 64}, uint8 */
    int32_t act_dim, feature_dim, hidden_dim, batch;
    int32_t precision;         /* EXORL_PREC_*: Linear layers' GEMMs and (bf16 modes) the 32-channel convolutions on MFMA; fp32: fp32 FMA convolutions */
    int32_t meta_dim;          /* skill / task columns concatenated after the encoding in front of the actor's and the critic's trunk (ddpg.py:294-299,305-312) */
    float lr, tau, stddev_clip;
    int32_t sf_dim;            /* > 0: CriticSF (aps.py:17-60) — each Q head emits sf_dim successor features, Q = task . features, the task being the meta row */
    uint64_t seed;
} exorl_pixel_cfg;
typedef struct exorl_pixel_agent exorl_pixel_agent_t;
#define EXORL_PNET_ENCODER 0
#define EXORL_PNET_ACTOR   1
#define EXORL_PNET_CRITIC  2
#define EXORL_PNET_CRITIC_TARGET 3
size_t exorl_pixel_agent_workspace_bytes(const exorl_pixel_cfg* cfg);
int exorl_pixel_agent_create(const exorl_pixel_cfg* cfg, void* workspace_dev, size_t workspace_bytes, exorl_pixel_agent_t** out);
int exorl_pixel_agent_destroy(exorl_pixel_agent_t* a);
/* tensors in parameters() order: encoder convnet.{0,2,4,6}.{weight,bias} (weights as (32, ci*9)); actor trunk.0, trunk.1,
 * policy.{0,2,4}; critic trunk.0, trunk.1, Q1.{0,2,4}, Q2.{0,2,4}. what = EXORL_T_*; the target has parameters only. */
int exorl_pixel_agent_num_tensors(exorl_pixel_agent_t* a, int32_t net, int32_t* n);
int exorl_pixel_agent_tensor(exorl_pixel_agent_t* a, int32_t net, int32_t index, int32_t what, void** ptr_dev, int64_t* rows, int64_t* cols);
int exorl_pixel_agent_sync_target(exorl_pixel_agent_t* a, void* stream);       /* critic_target.load_state_dict(critic.state_dict()) */
int exorl_pixel_agent_batch_slots(exorl_pixel_agent_t* a, exorl_batch_out* out);   /* uint8 obs / next_obs rows for the HBM sampler */
int exorl_pixel_agent_set_batch(exorl_pixel_agent_t* a, const unsigned char* obs_dev, const float* action_dev, const float* reward_dev,
                                const float* discount_dev, const unsigned char* next_obs_dev, void* stream);
/* One update() on the batch in the slots. shifts_*: (batch, 2) int32 RandomShiftsAug draws for obs / next_obs or null -> Philox;
 * noise_*: (batch, act_dim) standard normals for the TruncatedNormal draws (critic target first, actor second) or null. */
int exorl_pixel_agent_update(exorl_pixel_agent_t* a, float stddev, const int32_t* shifts_obs_dev, const int32_t* shifts_next_dev,
                             const float* noise_critic_dev, const float* noise_actor_dev, void* stream);
int exorl_pixel_agent_metrics(exorl_pixel_agent_t* a, float* host_out /* EXORL_N_METRICS */, void* stream);
/* Primitives for agents that put a module between augmentation and the DDPG step (Proto on pixels, proto.py:159-207): augment once,
 * encode with the online or the target encoder, push a feature gradient back through the encoder with a chosen optimiser state,
 * maintain encoder_target; exorl_pixel_agent_update with shifts_obs_dev == (const int32_t*)-1 then reuses the augmented batch, and with
 * (const int32_t*)-2 also the encodings the last exorl_pixel_agent_encode(0, 0) / (1, 0) calls left (what the agents that step the encoder
 * through their own module hand the critic: computed before that step, detached — icm.py:97-131, diayn.py:137-170; needs
 * set_train_encoder(0)). */
int exorl_pixel_agent_augment(exorl_pixel_agent_t* a, const int32_t* shifts_obs_dev, const int32_t* shifts_next_dev, void* stream);
int exorl_pixel_agent_encode(exorl_pixel_agent_t* a, int32_t which, int32_t target, float** feat_out_dev, void* stream);
int exorl_pixel_agent_encoder_step(exorl_pixel_agent_t* a, int32_t which, float* dfeat_dev, int32_t opt /* 0 encoder_opt, 1 second state, 2 both */, void* stream);
int exorl_pixel_agent_encoder_target(exorl_pixel_agent_t* a, float tau, int32_t init, void* stream);
/* RND on pixels (rnd.py:26-27,35-39,47-53): x = clamp(BatchNorm2d(RandomShiftsAug(obs)), +-clip_val), then the agent's encoder on x
 * (*feat_pred_dev; exorl_pixel_agent_encoder_step(0, dfeat, 2) continues its backward pass and steps the encoder with rnd_opt's state and
 * then encoder_opt's, rnd.py:86-89) and the frozen encoder copy kept in the encoder_target slot (*feat_target_dev). Every call draws a new
 * augmentation and updates the BatchNorm running statistics, as RND.forward does. */
int exorl_pixel_agent_rnd_features(exorl_pixel_agent_t* a, const int32_t* shifts_dev, float clip_val, float** feat_pred_dev, float** feat_target_dev,
                                   void* stream);
int exorl_pixel_agent_bn_state(exorl_pixel_agent_t* a, void** ptr_dev, int64_t* n_floats);   /* running_mean[c] running_var[c] num_batches_tracked */
int exorl_pixel_agent_encoder_target_ptr(exorl_pixel_agent_t* a, void** ptr_dev);
/* Whole-agent pickling (pretrain.py:293-300): steps3 = {Adam step count of critic_opt/actor_opt, of proto_opt's encoder state, of encoder_opt},
 * counters4 = Philox counters of the update-noise, augmentation, act() and RND-augmentation streams; encoder_opt2 = proto_opt's Adam moments for the
 * encoder (n floats each, same layout as the encoder's flat parameters, which encoder_target_ptr also uses). */
int exorl_pixel_agent_state(exorl_pixel_agent_t* a, int64_t* steps3_out, uint64_t* counters4_out);
int exorl_pixel_agent_set_state(exorl_pixel_agent_t* a, const int64_t* steps3, const uint64_t* counters4);
int exorl_pixel_agent_encoder_opt2(exorl_pixel_agent_t* a, void** m_dev, void** v_dev, int64_t* n_floats);
/* enable == 0: update() treats the encoding as detached in update_critic (what the reward-free agents pass, proto.py:190-193):
 * no encoder backward, encoder_opt does not step. Default 1 (plain DDPG, ddpg.py:316-319). */
int exorl_pixel_agent_set_train_encoder(exorl_pixel_agent_t* a, int32_t enable);
int exorl_pixel_agent_act(exorl_pixel_agent_t* a, const unsigned char* obs_dev, const float* meta_dev /* (meta_dim,) or null */, float stddev, int32_t eval_mode, const float* noise_dev,
                          float* action_out_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EXORL_HIP_H */
