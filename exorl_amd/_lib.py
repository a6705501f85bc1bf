"""ctypes binding of libexorl_hip.so (include/exorl_hip.h). No fallback: if the HIP library is
missing or a call fails, this raises — the product path never routes around the GPU kernels."""
import ctypes as C
import os
from pathlib import Path

HERE = Path(__file__).resolve().parent
LIB_PATH = HERE / 'libexorl_hip.so'

c_void_p, c_int32, c_int64, c_float, c_size_t, c_uint64, c_uint32 = (
    C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t, C.c_uint64, C.c_uint32)
P = C.POINTER

# constants mirrored from include/exorl_hip.h
SAMPLER_MT19937, SAMPLER_PHILOX, SAMPLER_GIVEN = 0, 1, 2
AGENT_TD3_BC, AGENT_TD3, AGENT_BC, AGENT_DDPG, AGENT_CRR, AGENT_CQL, AGENT_APS = 0, 1, 2, 3, 4, 5, 6
M_CRITIC_CQL, M_CRITIC_CQL_LOGSUM, M_ACTOR_ALPHA, M_ACTOR_ALPHA_LOSS, M_ACTOR_ENT = 10, 11, 12, 13, 14
CRR_WEIGHT = {'identity': 0, 'indicator': 1, 'exp': 2}
PREC_F32, PREC_BF16, PREC_BF16X3, PREC_BF16X6 = 0, 1, 2, 3
NET_ACTOR, NET_CRITIC, NET_CRITIC_TARGET = 0, 1, 2
T_PARAM, T_GRAD, T_ADAM_M, T_ADAM_V = 0, 1, 2, 3
M_BATCH_REWARD, M_CRITIC_TARGET_Q, M_CRITIC_Q1, M_CRITIC_Q2, M_CRITIC_LOSS, M_ACTOR_LOSS, M_ACTOR_LOGPROB = range(7)
N_METRICS = 16
INTR_RND, INTR_ICM, INTR_ICM_APT, INTR_DISAGREEMENT, INTR_DIAYN, INTR_PROTO, INTR_APS, INTR_SMM = 0, 1, 2, 3, 4, 5, 6, 7
IM_LOSS, IM_INTR_REWARD, IM_EXTR_REWARD, IM_RMS_MEAN, IM_RMS_STD, IM_ACC, IM_ENT_REWARD, IM_SF_REWARD = range(8)
N_INTR_METRICS = 8


class ReplayCfg(C.Structure):
    _fields_ = [('obs_bytes', c_int32), ('act_dim', c_int32), ('meta_dim', c_int32), ('max_episodes', c_int32),
                ('capacity_rows', c_int64)]


class BatchOut(C.Structure):
    _fields_ = [('obs', c_void_p), ('obs_stride', c_int64), ('action', c_void_p), ('action_stride', c_int64),
                ('reward', c_void_p), ('discount', c_void_p), ('next_obs', c_void_p), ('next_obs_stride', c_int64),
                ('meta', c_void_p), ('meta_stride', c_int64)]


class AgentCfg(C.Structure):
    _fields_ = [('kind', c_int32), ('obs_dim', c_int32), ('act_dim', c_int32), ('hidden_dim', c_int32),
                ('batch', c_int32), ('precision', c_int32), ('world_size', c_int32), ('sf_dim', c_int32),
                ('lr', c_float), ('tau', c_float), ('alpha', c_float), ('stddev_clip', c_float), ('seed', c_uint64),
                ('num_value_samples', c_int32), ('weight_func', c_int32), ('n_samples', c_int32), ('use_critic_lagrange', c_int32),
                ('target_cql_penalty', c_float), ('reserved3', c_int32)]


class IntrCfg(C.Structure):
    _fields_ = [('kind', c_int32), ('obs_dim', c_int32), ('act_dim', c_int32), ('hidden_dim', c_int32), ('rep_dim', c_int32),
                ('batch', c_int32), ('precision', c_int32), ('knn_k', c_int32), ('knn_avg', c_int32), ('knn_rms', c_int32),
                ('n_models', c_int32), ('flags', c_int32), ('lr', c_float), ('scale', c_float), ('knn_clip', c_float), ('clip_val', c_float),
                ('num_protos', c_int32), ('queue_size', c_int32), ('tau', c_float), ('target_tau', c_float),
                ('sp_lr', c_float), ('vae_lr', c_float), ('vae_beta', c_float), ('state_ent_coef', c_float), ('latent_ent_coef', c_float),
                ('latent_cond_ent_coef', c_float), ('goal_x', c_float), ('goal_y', c_float)]


class IntrBatch(C.Structure):
    _fields_ = [('obs', c_void_p), ('obs_ld', c_int64), ('action', c_void_p), ('action_ld', c_int64), ('next_obs', c_void_p),
                ('next_obs_ld', c_int64), ('skill', c_void_p), ('skill_ld', c_int64), ('extr_reward', c_void_p), ('reward_out', c_void_p), ('next_obs_target', c_void_p),
                ('next_obs_target_ld', c_int64), ('dobs_out', c_void_p), ('cat_uniform', c_void_p)]


class PixelCfg(C.Structure):
    _fields_ = [('c_in', c_int32), ('hw', c_int32), ('act_dim', c_int32), ('feature_dim', c_int32), ('hidden_dim', c_int32), ('batch', c_int32),
                ('precision', c_int32), ('meta_dim', c_int32), ('lr', c_float), ('tau', c_float), ('stddev_clip', c_float), ('sf_dim', c_int32),
                ('seed', c_uint64)]


# name -> (restype, argtypes); every symbol declared in include/exorl_hip.h
PROTOTYPES = {
    'exorl_u8_to_f32': (C.c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    'exorl_pixel_agent_workspace_bytes': (c_size_t, [P(PixelCfg)]),
    'exorl_pixel_agent_create': (C.c_int, [P(PixelCfg), c_void_p, c_size_t, P(c_void_p)]),
    'exorl_pixel_agent_destroy': (C.c_int, [c_void_p]),
    'exorl_pixel_agent_num_tensors': (C.c_int, [c_void_p, c_int32, P(c_int32)]),
    'exorl_pixel_agent_tensor': (C.c_int, [c_void_p, c_int32, c_int32, c_int32, P(c_void_p), P(c_int64), P(c_int64)]),
    'exorl_pixel_agent_sync_target': (C.c_int, [c_void_p, c_void_p]),
    'exorl_pixel_agent_batch_slots': (C.c_int, [c_void_p, P(BatchOut)]),
    'exorl_pixel_agent_set_batch': (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'exorl_pixel_agent_update': (C.c_int, [c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'exorl_pixel_agent_augment': (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    'exorl_pixel_agent_encode': (C.c_int, [c_void_p, c_int32, c_int32, P(c_void_p), c_void_p]),
    'exorl_pixel_agent_encoder_step': (C.c_int, [c_void_p, c_int32, c_void_p, c_int32, c_void_p]),
    'exorl_pixel_agent_encoder_target': (C.c_int, [c_void_p, c_float, c_int32, c_void_p]),
    'exorl_pixel_agent_set_train_encoder': (C.c_int, [c_void_p, c_int32]),
    'exorl_pixel_agent_encoder_target_ptr': (C.c_int, [c_void_p, P(c_void_p)]),
    'exorl_pixel_agent_rnd_features': (C.c_int, [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p]),
    'exorl_pixel_agent_bn_state': (C.c_int, [c_void_p, c_void_p, c_void_p]),
    'exorl_pixel_agent_state': (C.c_int, [c_void_p, c_void_p, c_void_p]),
    'exorl_pixel_agent_set_state': (C.c_int, [c_void_p, c_void_p, c_void_p]),
    'exorl_pixel_agent_encoder_opt2': (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    'exorl_pixel_agent_metrics': (C.c_int, [c_void_p, c_void_p, c_void_p]),
    'exorl_pixel_agent_act': (C.c_int, [c_void_p, c_void_p, c_void_p, c_float, c_int32, c_void_p, c_void_p, c_void_p]),
    'exorl_aug_shift': (C.c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_uint64, c_uint64, c_void_p, c_void_p]),
    'exorl_encoder_param_floats': (c_int64, [c_int32, c_int32]),
    'exorl_encoder_out_dim': (c_int64, [c_int32]),
    'exorl_encoder_workspace_floats': (c_int64, [c_int32, c_int32, c_int32]),
    'exorl_encoder_forward': (C.c_int, [c_void_p, c_int32, c_int32, c_void_p, c_int32, c_void_p, P(c_void_p), c_void_p]),
    'exorl_encoder_backward': (C.c_int, [c_void_p, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    'exorl_encoder_forward_prec': (C.c_int, [c_void_p, c_int32, c_int32, c_void_p, c_int32, c_void_p, P(c_void_p), c_int32, c_void_p]),
    'exorl_encoder_backward_prec': (C.c_int, [c_void_p, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    'exorl_intr_workspace_bytes': (c_size_t, [P(IntrCfg)]),
    'exorl_intr_create': (C.c_int, [P(IntrCfg), c_void_p, c_size_t, P(c_void_p)]),
    'exorl_intr_destroy': (C.c_int, [c_void_p]),
    'exorl_intr_num_tensors': (C.c_int, [c_void_p, P(c_int32)]),
    'exorl_intr_tensor': (C.c_int, [c_void_p, c_int32, c_int32, P(c_void_p), P(c_int64), P(c_int64)]),
    'exorl_intr_flat': (C.c_int, [c_void_p, c_int32, P(c_void_p), P(c_int64)]),
    'exorl_intr_state': (C.c_int, [c_void_p, P(c_void_p), P(c_void_p), P(c_int64)]),
    'exorl_intr_queue': (C.c_int, [c_void_p, P(c_void_p), P(c_int64), P(c_int64), P(c_int64), c_int32]),
    'exorl_intr_update': (C.c_int, [c_void_p, P(IntrBatch), c_int32, c_void_p]),
    'exorl_intr_metrics': (C.c_int, [c_void_p, c_void_p, c_void_p]),
    'exorl_intr_opt_steps': (C.c_int, [c_void_p, P(c_int64), c_int32]),
    'exorl_intr_counter': (C.c_int, [c_void_p, c_void_p, c_int32]),
    'exorl_last_error': (C.c_char_p, []),
    'exorl_abi_version': (C.c_int, []),
    'exorl_device_info': (C.c_int, [C.c_char_p, C.c_int, P(C.c_int), P(c_int64)]),
    'exorl_replay_create': (C.c_int, [P(ReplayCfg), P(c_void_p)]),
    'exorl_replay_destroy': (C.c_int, [c_void_p]),
    'exorl_replay_append_episode': (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, P(c_int32)]),
    'exorl_replay_evict': (C.c_int, [c_void_p, c_int32]),
    'exorl_replay_set_order': (C.c_int, [c_void_p, c_void_p, c_int32]),
    'exorl_replay_num_rows': (C.c_int, [c_void_p, P(c_int64), P(c_int64)]),
    'exorl_replay_seed_mt': (C.c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_int32]),
    'exorl_replay_seed_mt_ints': (C.c_int, [c_void_p, c_uint64, c_uint32]),
    'exorl_replay_get_mt': (C.c_int, [c_void_p, c_void_p, P(c_int32), c_void_p, P(c_int32)]),
    'exorl_replay_seed_philox': (C.c_int, [c_void_p, c_uint64]),
    'exorl_replay_sample': (C.c_int, [c_void_p, c_int32, c_int32, c_float, c_int32, c_void_p, P(BatchOut), c_void_p, c_void_p]),
    'exorl_replay_last_pairs': (C.c_int, [c_void_p, c_int32, c_void_p, c_void_p]),
    'exorl_comm_unique_id': (C.c_int, [c_void_p]),
    'exorl_comm_init': (C.c_int, [c_int32, c_int32, c_void_p, P(c_void_p)]),
    'exorl_comm_destroy': (C.c_int, [c_void_p]),
    'exorl_comm_allreduce': (C.c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    'exorl_agent_set_comm': (C.c_int, [c_void_p, c_void_p]),
    'exorl_agent_workspace_bytes': (c_size_t, [P(AgentCfg)]),
    'exorl_agent_create': (C.c_int, [P(AgentCfg), c_void_p, c_size_t, P(c_void_p)]),
    'exorl_agent_destroy': (C.c_int, [c_void_p]),
    'exorl_agent_num_tensors': (C.c_int, [c_void_p, c_int32, P(c_int32)]),
    'exorl_agent_tensor': (C.c_int, [c_void_p, c_int32, c_int32, c_int32, P(c_void_p), P(c_int64), P(c_int64)]),
    'exorl_agent_flat': (C.c_int, [c_void_p, c_int32, c_int32, P(c_void_p), P(c_int64)]),
    'exorl_agent_params_changed': (C.c_int, [c_void_p, c_int32, c_void_p]),
    'exorl_agent_batch_slots': (C.c_int, [c_void_p, P(BatchOut)]),
    'exorl_agent_set_batch': (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'exorl_agent_update': (C.c_int, [c_void_p, c_float, c_void_p, c_void_p, c_void_p]),
    'exorl_agent_update_phase': (C.c_int, [c_void_p, c_int32, c_float, c_void_p, c_void_p, c_void_p]),
    'exorl_agent_stats_buffer': (C.c_int, [c_void_p, P(c_void_p), P(c_int64)]),
    'exorl_agent_cql_alpha': (C.c_int, [c_void_p, c_void_p, c_int32]),
    'exorl_agent_act': (C.c_int, [c_void_p, c_void_p, c_int32, c_float, c_int32, c_void_p, c_void_p, c_void_p]),
    'exorl_agent_act_host': (C.c_int, [c_void_p, c_void_p, c_int32, c_float, c_int32, c_void_p, c_void_p, c_void_p]),
    'exorl_agent_metrics': (C.c_int, [c_void_p, c_void_p, c_void_p]),
    'exorl_agent_set_metrics': (C.c_int, [c_void_p, c_int32]),
    'exorl_agent_set_parallel_branches': (C.c_int, [c_void_p, c_int32]),
    'exorl_agent_opt_steps': (C.c_int, [c_void_p, P(c_int64), P(c_int64)]),
    'exorl_agent_set_opt_steps': (C.c_int, [c_void_p, c_int64, c_int64]),
    'exorl_agent_enable_graph': (C.c_int, [c_void_p, c_void_p, c_int32, c_float, c_float, c_void_p]),
    'exorl_agent_step_graph': (C.c_int, [c_void_p, c_float, c_void_p]),
    'exorl_agent_noise_counter': (C.c_int, [c_void_p, P(c_uint64), c_void_p]),
    'exorl_debug_philox_normal': (C.c_int, [c_uint64, c_uint64, c_int64, c_void_p, c_void_p]),
    'exorl_agent_disable_graph': (C.c_int, [c_void_p]),
    'exorl_gemm': (C.c_int, [c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int64, c_void_p, c_int64,
                             c_void_p, c_int64, c_void_p, c_int32, c_int32, c_void_p]),
    'exorl_gemm_bf16': (C.c_int, [c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                  c_void_p, c_int32, c_int32, c_void_p]),
    'exorl_gemm_planes': (C.c_int, [c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64,
                                    c_void_p, c_int64, c_int32, c_void_p]),
    'exorl_gemm_tune': (C.c_int, [c_int32]),
    'exorl_debug_precision_override': (C.c_int, [c_int32]),
    'exorl_debug_gemm_stamps': (C.c_int, [c_void_p, c_int32]),
    'exorl_debug_conv_stamps': (C.c_int, [c_void_p, c_int32]),
    'exorl_profile_gemm': (C.c_int, [c_int32]),
    'exorl_profile_gemm_read': (C.c_int, [c_void_p, c_void_p, c_int32, P(c_int32)]),
    'exorl_profile_event_overhead': (C.c_int, [c_void_p, c_void_p]),
    'exorl_adam_step': (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float,
                                  c_int64, c_void_p, c_float, c_void_p]),
    'exorl_soft_update': (C.c_int, [c_void_p, c_void_p, c_int64, c_float, c_void_p]),
    'exorl_ln_tanh_fwd': (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    'exorl_knn_topk': (C.c_int, [c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
}


class ExorlError(RuntimeError):
    pass


_lib = None


def load():
    """Loads the in-tree HIP library (built by exorl_amd/build.py or __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ExorlError(f'{LIB_PATH} is missing: build it with `python exorl_amd/build.py` '
                         '(hipcc --offload-arch=gfx950). There is no CPU fallback.')
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)      # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    tune = os.environ.get('EXORL_GEMM_TUNE')        # kernel-variant experiments (tools/micro): bit mask for exorl_gemm_tune
    if tune:
        lib.exorl_gemm_tune(int(tune))
    return lib


def check(rc):
    if rc != 0:
        raise ExorlError(load().exorl_last_error().decode())


def ptr(t):
    """Device/host pointer of a torch tensor or numpy array (None -> NULL)."""
    if t is None:
        return None
    if hasattr(t, 'data_ptr'):
        return t.data_ptr()
    return t.ctypes.data


def current_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream
