"""Which split-bf16 products of a config-4 Proto update (jaco pixels, batch 1024, shipped dims) have to be exact for the per-step metrics to stay
inside 1e-4 of the reference. The product's own kernels, one product family at a time switched to exact fp32 products
(exorl_debug_precision_override), three update() calls against the reference's recorded fp64 trajectory (tests/golden/config4_proto_b1024.npz: `metrics_fp64`; its three fp32 runs give the scale).
    python tools/debug/config4_ablation.py [mask ...]        (default: the standard sweep)
Printed per mask: worst relative error per step over the metrics (|got - ref| / (|ref| + 1e-2)), and the worst metric's name; the reference's own
fp32 floor (all threads vs one thread) is printed first."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))
import _synth  # noqa: E402
from exorl_amd import _lib as L, agents  # noqa: E402

BITS = {1: 'fwd narrow', 2: 'fwd wide', 4: 'wgrad narrow', 8: 'wgrad wide', 16: 'dgrad narrow', 32: 'dgrad wide', 64: 'conv fwd', 128: 'conv dgrad', 256: 'conv wgrad'}


def make_agent(z, precision):
    C_, HW, A, F, H, B, N, PD, PJ, Q, NP = [int(v) for v in z['dims']]
    ag = agents.ProtoAgent(pred_dim=PD, proj_dim=PJ, queue_size=Q, num_protos=NP, tau=0.1, encoder_target_tau=0.05, topk=3, update_encoder=True,
                           name='proto', reward_free=True, obs_type='pixels', obs_shape=(C_, HW, HW), action_shape=(A,), device='cuda', lr=1e-4,
                           feature_dim=F, hidden_dim=H, critic_target_tau=0.01, num_expl_steps=2000, update_every_steps=2, stddev_schedule=0.2,
                           nstep=3, batch_size=B, stddev_clip=0.3, init_critic=True, use_tb=True, use_wandb=False, precision=precision)
    ps = _synth.config4_params(C_, A, F, H, PD, PJ, NP)
    for nm in ('encoder', 'actor', 'critic', 'predictor', 'projector', 'protos'):
        view = getattr(ag, nm)
        sd = view.state_dict()
        view.load_state_dict({k: torch.from_numpy(v).reshape(sd[k].shape) for k, v in ps[nm].items()})
    ag.engine.sync_target()
    for p, t in zip(ag.predictor.parameters(), ag.predictor_target.parameters()):
        t.copy_(p)
    ag.engine.encoder_target(init=True)
    return ag


def run(z, precision, mask):
    lib = L.load()
    C_, HW, A, F, H, B, N, PD, PJ, Q, NP = [int(v) for v in z['dims']]
    lib.exorl_debug_precision_override(mask)
    try:
        ag = make_agent(z, precision)
        ns = _synth.NoiseStream(22)
        ag.noise_hook = ns.draw
        keys = [str(k) for k in z['metric_keys']]
        errs, worst, ms = [], [], []
        for i in range(N):
            obs, nobs, act, rew, disc, so, sn, u = _synth.config4_inputs(i, B, C_, HW, A, NP)
            sh = [so, sn]
            ag.shift_hook = lambda n: sh.pop(0)
            ag.cat_hook = lambda n: u
            torch.cuda.synchronize()
            t0 = time.time()
            m = ag.update(iter([(obs, act, rew, disc, nobs)]), 2 * i)
            torch.cuda.synchronize()
            ms.append((time.time() - t0) * 1e3)
            e = np.array([abs(m[k] - v) / (abs(v) + 1e-2) for k, v in zip(keys, z['metrics_fp64'][i])])
            errs.append(float(e.max()))
            worst.append(keys[int(e.argmax())])
        return errs, worst, ms
    finally:
        lib.exorl_debug_precision_override(0)


if __name__ == '__main__':
    z = np.load(ROOT / 'tests' / 'golden' / 'config4_proto_b1024.npz')
    keys = [str(k) for k in z['metric_keys']]
    print('every line: worst |x - reference fp64| / (|reference fp64| + 1e-2) per update, over the metrics')
    for nm, label in (('metrics', 'reference fp32, oneDNN convolutions, all threads'), ('metrics_1thread', 'reference fp32, oneDNN convolutions, 1 thread'),
                      ('metrics_no_onednn', 'reference fp32, native convolutions')):
        floor = np.abs(z[nm] - z['metrics_fp64']) / (np.abs(z['metrics_fp64']) + 1e-2)
        print(f'{label:66s}: {" ".join(f"{e:.1e}" for e in floor.max(axis=1))}  worst {[keys[j] for j in floor.argmax(axis=1)]}', flush=True)
    masks = [int(a) for a in sys.argv[1:]] or ([0, 511] + list(BITS))
    for prec in ('fp32', 'bf16x6'):
        e, w, ms = run(z, prec, 0)
        print(f'precision {prec:16s}: {" ".join(f"{x:.1e}" for x in e)}  worst {w}  ms/update {ms[-1]:.1f}', flush=True)
    for mask in masks:
        e, w, ms = run(z, 'bf16x3', mask)
        name = ' + '.join(n for b, n in BITS.items() if mask & b) or 'none (pure split-bf16)'
        if mask and bin(mask).count('1') >= 8:
            name = 'all but: ' + (' + '.join(n for b, n in BITS.items() if not mask & b) or '(nothing: everything exact)')
        print(f'bf16x3, exact fp32 for [{mask:3d}] {name:40s}: {" ".join(f"{x:.1e}" for x in e)}  worst {w}  ms/update {ms[-1]:.1f}', flush=True)
