"""update()/s of the DDPG-backbone reward-free agents on states at their shipped widths (configs/agent/*.yaml: hidden 1024,
batch 1024, nstep 3; walker shapes O=24, A=6), HBM replay + Philox sampler, per precision mode.

    python tools/micro/unsup_bench.py [agent ...] [--precision fp32,bf16x3] [--steps 300]

One update() = module step + intrinsic reward + DDPG step (ddpg.py:294-328 and the per-agent update methods)."""
import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from exorl_amd import agents
from exorl_amd.engine import ReplayEngine
from exorl_amd.replay_buffer import ArenaIterator

O, A, H, B = 24, 6, 1024, 1024
EPISODES, EP_LEN = 200, 1000


def ddpg_kw(kind, precision):
    return dict(name=kind, reward_free=True, obs_type='states', obs_shape=(O,), action_shape=(A,), device='cuda', lr=1e-4,
                feature_dim=50, hidden_dim=H, critic_target_tau=0.01, num_expl_steps=2000, update_every_steps=2,
                stddev_schedule=0.2, nstep=3, batch_size=B, stddev_clip=0.3, init_critic=True, use_tb=False, use_wandb=False,
                precision=precision)


def make(kind, precision):
    d = ddpg_kw(kind, precision)
    if kind == 'ddpg':
        return agents.DDPGAgent(**d)
    if kind == 'rnd':
        return agents.RNDAgent(rnd_rep_dim=512, update_encoder=True, rnd_scale=1.0, **d)
    if kind == 'icm':
        return agents.ICMAgent(icm_scale=1.0, update_encoder=True, **d)
    if kind == 'icm_apt':
        return agents.ICMAPTAgent(icm_scale=1.0, update_encoder=True, icm_rep_dim=512, knn_rms=True, knn_k=12, knn_avg=True, knn_clip=0.0, **d)
    if kind == 'disagreement':
        return agents.DisagreementAgent(update_encoder=True, **d)
    if kind == 'diayn':
        return agents.DIAYNAgent(update_skill_every_step=50, skill_dim=16, diayn_scale=1.0, update_encoder=True, skill_type='uniform', **d)
    if kind == 'aps':
        return agents.APSAgent(update_task_every_step=5, sf_dim=10, knn_rms=True, knn_k=12, knn_avg=True, knn_clip=0.0001,
                               num_init_steps=4096, lstsq_batch_size=4096, update_encoder=True, **d)
    if kind == 'smm':
        return agents.SMMAgent(z_dim=4, sp_lr=1e-3, vae_lr=1e-2, vae_beta=0.5, state_ent_coef=1.0, latent_ent_coef=1.0,
                               latent_cond_ent_coef=1.0, update_encoder=True, **d)
    if kind == 'proto':
        return agents.ProtoAgent(pred_dim=128, proj_dim=512, queue_size=2048, num_protos=512, tau=0.1, encoder_target_tau=0.05, topk=3,
                                 update_encoder=True, **d)
    raise SystemExit(f'unknown agent {kind}')


def replay_for(agent):
    specs = agent.get_meta_specs() if hasattr(agent, 'get_meta_specs') else ()
    mdim = sum(int(np.prod(s.shape)) for s in specs)
    eng = ReplayEngine((O,), np.float32, A, mdim, EPISODES * (EP_LEN + 1) + 64, EPISODES + 8, 'cuda')
    slots = []
    for e in range(EPISODES):
        rs = np.random.RandomState(7 + e)
        rows = EP_LEN + 1
        ep = dict(observation=rs.standard_normal((rows, O)).astype(np.float32), action=rs.uniform(-1, 1, (rows, A)).astype(np.float32),
                  reward=rs.uniform(0, 1, (rows, 1)).astype(np.float32), discount=np.ones((rows, 1), np.float32))
        for s in specs:
            v = rs.standard_normal((rows,) + tuple(s.shape)).astype(np.float32)
            if s.name in ('skill', 'z'):
                v = np.eye(s.shape[0], dtype=np.float32)[rs.randint(0, s.shape[0], rows)]
            else:
                v /= np.linalg.norm(v, axis=1, keepdims=True)
            ep[s.name] = v
        slots.append(eng.append_episode(ep, tuple(s.name for s in specs)))
    eng.set_order(slots)
    eng.seed_philox(3)
    return eng


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('agents', nargs='*', default=['ddpg', 'rnd', 'icm', 'icm_apt', 'disagreement', 'diayn', 'aps', 'smm', 'proto'])
    ap.add_argument('--precision', default='fp32,bf16x3')
    ap.add_argument('--steps', type=int, default=300)
    args = ap.parse_args()
    for kind in args.agents:
        for prec in args.precision.split(','):
            torch.manual_seed(1)
            np.random.seed(1)
            ag = make(kind, prec)
            it = ArenaIterator(replay_for(ag), B, 3, 0.99, 'philox')
            for i in range(30):
                ag.update(it, 2 * i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.steps):
                ag.update(it, 2 * (30 + i))
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f'{kind:13s} {prec:7s} {args.steps / dt:9.1f} update()/s  {1e3 * dt / args.steps:7.3f} ms', flush=True)
            del ag, it


if __name__ == '__main__':
    main()
