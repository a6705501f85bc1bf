"""update()/s at BASELINE config 4 shapes (Proto or plain DDPG on jaco pixels (3,84,84) uint8, A=9, batch 1024, nstep 3) on one MI355X.

    python tools/micro/pixel_bench.py [batch=1024] [proto|ddpg|rnd|icm|icm_apt|disagreement|diayn|aps|smm] [fp32|bf16x3|bf16]

Not the bench.py metric (that is TD3+BC on states); the numbers go to DESIGN.md's pixel section."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from exorl_amd import agents
from exorl_amd.engine import ReplayEngine
from exorl_amd.replay_buffer import ArenaIterator

C_, HW, A, B = 3, 84, 9, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
kind = sys.argv[2] if len(sys.argv) > 2 else 'proto'
precision = sys.argv[3] if len(sys.argv) > 3 else 'fp32'
EPISODES, EP_LEN = 40, 250
META_DIM = {'diayn': 16, 'aps': 10, 'smm': 4}.get(kind, 0)
eng = ReplayEngine((C_, HW, HW), np.uint8, A, META_DIM, EPISODES * (EP_LEN + 1) + 64, EPISODES + 8, 'cuda')
rs = np.random.RandomState(0)
slots = []
for e in range(EPISODES):
    rows = EP_LEN + 1
    ep = dict(observation=rs.randint(0, 256, (rows, C_, HW, HW)).astype(np.uint8), action=rs.uniform(-1, 1, (rows, A)).astype(np.float32),
              reward=rs.uniform(0, 1, (rows, 1)).astype(np.float32), discount=np.ones((rows, 1), np.float32))
    if META_DIM:
        m = np.zeros((rows, META_DIM), np.float32)
        m[:, e % META_DIM] = 1.0
        ep['skill'] = m
    slots.append(eng.append_episode(ep, ('skill',) if META_DIM else ()))
eng.set_order(slots)
eng.seed_philox(1)
it = ArenaIterator(eng, B, 3, 0.99, 'philox')
kw = dict(name=kind, reward_free=True, obs_type='pixels', obs_shape=(C_, HW, HW), action_shape=(A,), device='cuda', lr=1e-4, feature_dim=50,
          hidden_dim=1024, critic_target_tau=0.01, num_expl_steps=2000, update_every_steps=2, stddev_schedule=0.2, nstep=3, batch_size=B,
          stddev_clip=0.3, init_critic=True, use_tb=False, use_wandb=False, precision=precision)
torch.manual_seed(1)
META = {'diayn': 16, 'aps': 10, 'smm': 4}.get(kind, 0)          # configs/agent/{diayn,aps,smm}.yaml
if kind == 'proto':
    ag = agents.ProtoAgent(pred_dim=128, proj_dim=512, queue_size=2048, num_protos=512, tau=0.1, encoder_target_tau=0.05, topk=3, update_encoder=True, **kw)
elif kind == 'rnd':
    ag = agents.RNDAgent(rnd_rep_dim=512, update_encoder=True, rnd_scale=1.0, **kw)
elif kind == 'icm':
    ag = agents.ICMAgent(icm_scale=1.0, update_encoder=True, **kw)
elif kind == 'icm_apt':
    ag = agents.ICMAPTAgent(icm_scale=1.0, knn_rms=True, knn_k=12, knn_avg=True, knn_clip=0.0, update_encoder=True, icm_rep_dim=512, **kw)
elif kind == 'disagreement':
    ag = agents.DisagreementAgent(update_encoder=True, **kw)
elif kind == 'diayn':
    ag = agents.DIAYNAgent(update_skill_every_step=50, skill_dim=16, diayn_scale=1.0, update_encoder=True, skill_type='uniform', **kw)
elif kind == 'aps':
    ag = agents.APSAgent(update_task_every_step=5, sf_dim=10, knn_rms=True, knn_k=12, knn_avg=True, knn_clip=0.0001, num_init_steps=4096,
                         lstsq_batch_size=4096, update_encoder=True, **kw)
elif kind == 'smm':
    ag = agents.SMMAgent(z_dim=4, sp_lr=1e-3, vae_lr=1e-2, vae_beta=0.5, state_ent_coef=1.0, latent_ent_coef=1.0, latent_cond_ent_coef=1.0,
                         update_encoder=True, **kw)
else:
    ag = agents.DDPGAgent(**kw)
for i in range(3):
    ag.update(it, 2 * i)
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for i in range(n):
    ag.update(it, 2 * (i + 3))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f'{kind} pixels B={B} {precision}: {n / dt:.2f} update()/s, {1e3 * dt / n:.1f} ms per update', flush=True)
