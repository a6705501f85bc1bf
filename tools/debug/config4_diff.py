"""Config 4 (Proto on pixels, batch 1024, shipped dims), HIP path next to the torch-CPU twin, update by update: every metric of both against the
reference's recorded values, then every parameter tensor's step (final - initial) compared element-wise. Where do they part?
    python tools/debug/config4_diff.py [precision] [steps]"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))
sys.path.insert(0, str(ROOT / 'tools' / 'debug'))
import _synth  # noqa: E402
from config4_ablation import make_agent  # noqa: E402
from oracle.torch_twin_pixels import TorchTwinProtoPixels  # noqa: E402

precision = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
z = np.load(ROOT / 'tests' / 'golden' / 'config4_proto_b1024.npz')
C_, HW, A, F, H, B, N, PD, PJ, Q, NP = [int(v) for v in z['dims']]
keys = [str(k) for k in z['metric_keys']]
ag = make_agent(z, precision)
tw = TorchTwinProtoPixels(C_, HW, A, F, H, PD, PJ, NP, Q)
params = _synth.config4_params(C_, A, F, H, PD, PJ, NP)
tw.load(params)
ns, ns2 = _synth.NoiseStream(22), _synth.NoiseStream(22)
ag.noise_hook = ns.draw
prev = {nm: {k: v.detach().clone() for k, v in getattr(tw, nm).state_dict().items()} for nm in params}
for i in range(steps):
    obs, nobs, act, rew, disc, so, sn, u = _synth.config4_inputs(i, B, C_, HW, A, NP)
    sh = [so, sn]
    ag.shift_hook = lambda n: sh.pop(0)
    ag.cat_hook = lambda n: u
    m = ag.update(iter([(obs, act, rew, disc, nobs)]), 2 * i)
    mt = tw.update((obs, act, rew, disc, nobs), so, sn, u, ns2.draw((B, A)), ns2.draw((B, A)))
    print(f'--- update {i}: metric, reference (fp64 run), twin rel.err, HIP rel.err')
    for k, v in zip(keys, z['metrics_fp64'][i]):
        print(f'  {k:18s} {v:+.7f}  twin {abs(mt[k] - v) / (abs(v) + 1e-2):.1e}  hip[{precision}] {abs(m[k] - v) / (abs(v) + 1e-2):.1e}')
    ri = ag.intr.reward_rows().cpu().numpy().reshape(-1) if hasattr(ag.intr, 'reward_rows') else None
    if ri is not None:
        print('  intrinsic reward rows: max abs diff vs twin', float(np.abs(ri - tw.last_intr.reshape(-1)).max()))
    for nm in params:
        sd_t, sd_h = getattr(tw, nm).state_dict(), getattr(ag, nm).state_dict()
        for k in sd_t:
            dt = (sd_t[k] - prev[nm][k]).reshape(-1).double()
            dh = (sd_h[k].cpu().reshape(-1).double() - prev[nm][k].reshape(-1).double())
            diff = (dt - dh).abs()
            flips = float(((dt * dh) < 0).double().mean())
            cos = float((dt @ dh) / (dt.norm() * dh.norm() + 1e-30))
            print(f'  step of {nm}.{k:16s} |twin step| mean {float(dt.abs().mean()):.2e}  max |diff| {float(diff.max()):.2e}  mean |diff| {float(diff.mean()):.2e}  '
                  f'sign flips {flips:.2e}  cos {cos:.6f}')
            prev[nm][k] = sd_t[k].detach().clone()
