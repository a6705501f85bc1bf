"""Which operand format do the H x H GEMMs need for the 1e-4 bar? The oracle's Linear forward / backward with the operands of every
1024-wide product rounded to a candidate format, 10 full-size steps, worst relative metric error against the reference's fp32 run
(tests/golden/full_*.json).   python tools/debug/prec_experiment.py td3_bc,td3 fp16,fp16s,bf16,bf16x2"""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))
import _synth
from oracle import nets
from oracle.agents import OracleAgent, param_shapes

F32 = np.float32


def bf16(x):
    u = np.ascontiguousarray(x, F32).view(np.uint32)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(F32)


def make_round(mode):
    if mode == 'fp32':
        return lambda x, grad=False: x
    if mode == 'bf16':
        return lambda x, grad=False: bf16(x)
    if mode == 'bf16x2':          # hi + lo (what bf16x3 feeds the MFMAs; products then drop lo*lo)
        return lambda x, grad=False: bf16(x) + bf16(x - bf16(x))
    if mode == 'fp16':            # plain fp16, no scaling: small gradients fall into subnormals
        return lambda x, grad=False: x.astype(np.float16).astype(F32)
    if mode == 'fp16s':           # fp16 with a power-of-two scale on gradient operands
        def r(x, grad=False):
            if not grad:
                return x.astype(np.float16).astype(F32)
            s = F32(2.0 ** 14)
            return ((x * s).astype(np.float16).astype(F32) / s).astype(F32)
        return r
    raise ValueError(mode)


def patch(mode, H=1024):
    rnd = make_round(mode)
    o_fwd, o_bwd = nets.linear_fwd, nets.linear_bwd

    def fwd(x, W, b):
        if W.shape[1] == H and W.shape[0] == H:
            return (rnd(x) @ rnd(W).T + b).astype(F32)
        return o_fwd(x, W, b)

    def bwd(x, W, dy, need_dx=True):
        if W.shape[1] == H and W.shape[0] == H:
            dyr = rnd(dy, True)
            dW = (dyr.T @ rnd(x)).astype(F32)
            db = dy.sum(0).astype(F32)
            dx = (dyr @ rnd(W)).astype(F32) if need_dx else None
            return dW, db, dx
        return o_bwd(x, W, dy, need_dx)
    nets.linear_fwd, nets.linear_bwd = fwd, bwd
    return o_fwd, o_bwd


def run(kind, mode):
    g = json.load(open(ROOT / 'tests' / 'golden' / f'full_{kind}.json'))
    O, A, H, B = g['dims']
    ash, csh = param_shapes(kind, O, A, H)
    pa = list(_synth.synth_params(ash, g['param_seed']).values())
    pc = list(_synth.synth_params(csh, g['param_seed'] + 1).values()) if csh else None
    orig = patch(mode, H)
    try:
        ag = OracleAgent(kind, pa, pc)
        ns = _synth.NoiseStream(g['noise_seed'])
        worst, where = 0.0, None
        for i in range(g['nsteps']):
            step = 2 * i if kind == 'ddpg' else i
            n1 = ns.draw((B, A)) if kind != 'bc' else None
            n2 = ns.draw((B * 10 if kind == 'crr' else B, A)) if kind != 'bc' else None
            m = ag.update(_synth.synth_batch(g['batch_seed'], i, B, O, A), step, n1, n2)
            for k, v in g['fp32']['metrics'][i].items():
                if k in m:
                    e = abs(m[k] - v) / (abs(v) + 1e-2)
                    if e > worst:
                        worst, where = e, (i, k)
    finally:
        nets.linear_fwd, nets.linear_bwd = orig
    return worst, where


if __name__ == '__main__':
    kinds = sys.argv[1].split(',') if len(sys.argv) > 1 else ['td3_bc', 'td3']
    modes = sys.argv[2].split(',') if len(sys.argv) > 2 else ['fp32', 'bf16x2', 'fp16s', 'fp16', 'bf16']
    for kind in kinds:
        for mode in modes:
            w, where = run(kind, mode)
            print(f'{kind:8s} {mode:7s} worst relative metric error {w:.2e} at {where}', flush=True)
