"""Seed-regenerable synthetic inputs shared by tools/gen_golden.py (fixture side) and the tests.

Everything here is plain numpy on np.random.RandomState so that the fixture generator (which
runs the reference in the build container) and the tests (which run the oracle / HIP path,
possibly on the GPU box where the reference does not exist) rebuild bit-identical inputs.
"""
import numpy as np


def synth_params(shapes, seed):
    """shapes: ordered list of (name, shape). Returns dict name -> float32 array.

    2-D  -> N(0,1)/sqrt(fan_in)   (Linear weight, stored (out, in))
    name endswith '.weight' and 1-D -> 1 + 0.1 N(0,1)   (LayerNorm gain)
    otherwise 1-D -> 0.1 N(0,1)   (biases)
    """
    rs = np.random.RandomState(seed)
    out = {}
    for name, shape in shapes:
        shape = tuple(int(s) for s in shape)
        if len(shape) == 2:
            w = rs.standard_normal(shape) / np.sqrt(shape[1])
        elif name.endswith('.weight'):
            w = 1.0 + 0.1 * rs.standard_normal(shape)
        else:
            w = 0.1 * rs.standard_normal(shape)
        out[name] = w.astype(np.float32)
    return out


def synth_conv_params(shapes, seed):
    """Encoder tensors with activations that stay O(1) through the four ReLU layers (He scaling): 4-D -> N(0,1) * sqrt(2 / fan_in),
    1-D -> 0.1 N(0,1). For the fixtures whose modules read the raw 39200-wide encoding (synth_params gives 4-D tensors its LayerNorm-gain
    branch, i.e. weights near 1 and encodings near 1e7 — harmless in front of a LayerNorm, useless in front of an MSE)."""
    rs = np.random.RandomState(seed)
    out = {}
    for name, shape in shapes:
        shape = tuple(int(s) for s in shape)
        if len(shape) == 4:
            w = rs.standard_normal(shape) * np.sqrt(2.0 / (shape[1] * shape[2] * shape[3]))
        else:
            w = 0.1 * rs.standard_normal(shape)
        out[name] = w.astype(np.float32)
    return out


def synth_batch(seed, step, batch, obs_dim, act_dim, gamma=0.99):
    """One minibatch as the sampler would emit it: (obs, action, reward, discount, next_obs)."""
    rs = np.random.RandomState(seed * 100003 + step)
    obs = rs.standard_normal((batch, obs_dim)).astype(np.float32)
    action = rs.uniform(-1.0, 1.0, (batch, act_dim)).astype(np.float32)
    reward = rs.uniform(0.0, 1.0, (batch, 1)).astype(np.float32)
    discount = np.full((batch, 1), np.float32(1.0) * np.float32(gamma), np.float32)
    next_obs = rs.standard_normal((batch, obs_dim)).astype(np.float32)
    return obs, action, reward, discount, next_obs


class NoiseStream:
    """Deterministic N(0,1) draws standing in for torch's _standard_normal (utils.py:142-145)."""

    def __init__(self, seed):
        self.rs = np.random.RandomState(seed)

    def draw(self, shape):
        return self.rs.standard_normal(tuple(shape)).astype(np.float32)


def synth_episodes(seed, lengths, obs_dim, act_dim, meta_dim=0, obs_u8=False):
    """Episodes in the on-disk layout of replay_buffer.py:115-150: len+1 rows, dummy first row."""
    rs = np.random.RandomState(seed)
    eps = []
    for L in lengths:
        rows = L + 1
        if obs_u8:
            obs = rs.randint(0, 256, (rows, obs_dim)).astype(np.uint8)
        else:
            obs = rs.standard_normal((rows, obs_dim)).astype(np.float32)
        act = rs.uniform(-1, 1, (rows, act_dim)).astype(np.float32)
        rew = rs.uniform(0, 1, (rows, 1)).astype(np.float32)
        disc = np.ones((rows, 1), np.float32)
        # sprinkle terminal / partial discounts so the n-step product is exercised
        flip = rs.uniform(size=rows)
        disc[flip < 0.15] = 0.0
        disc[(flip >= 0.15) & (flip < 0.3)] = 0.5
        act[0] = 0.0
        rew[0] = 0.0
        disc[0] = 1.0
        ep = dict(observation=obs, action=act, reward=rew, discount=disc)
        if meta_dim:
            ep['skill'] = rs.standard_normal((rows, meta_dim)).astype(np.float32)
        eps.append(ep)
    return eps


# ---- BASELINE config 4 (Proto on jaco pixels) at its own sizes: tests/golden/config4_proto_b1024.npz (tools/gen_golden.py::gen_config4)
def config4_inputs(step, B, C, HW, A, NP):
    """Inputs of update() number `step` of the config-4 fixture, regenerated from seeds on both sides (nothing batch-sized is stored):
    uint8 frames, the (action, reward, discount) rows, the two augmentation shift blocks and the Categorical uniforms."""
    rs = np.random.RandomState(4000 + step)
    obs = rs.randint(0, 256, (B, C, HW, HW)).astype(np.uint8)
    nobs = rs.randint(0, 256, (B, C, HW, HW)).astype(np.uint8)
    b = synth_batch(3, step, B, 4, A)
    so, sn = rs.randint(0, 9, (B, 2)).astype(np.int32), rs.randint(0, 9, (B, 2)).astype(np.int32)
    u = rs.uniform(size=NP).astype(np.float32)
    return obs, nobs, b[1], b[2], b[3], so, sn, u


def config4_params(C, A, F, H, PD, PJ, NP, R=39200):
    """Explicit weights of the config-4 fixture: He-scaled convolutions (encodings stay O(1), as under the reference's own orthogonal
    init), 1/sqrt(fan_in) Linear layers. Returns dict module -> ordered dict of arrays (reference state_dict keys)."""
    enc_keys = [f'convnet.{i}.{w}' for i in (0, 2, 4, 6) for w in ('weight', 'bias')]
    esh = list(zip(enc_keys, [s for l in range(4) for s in ((32, C if l == 0 else 32, 3, 3), (32,))]))
    tr = [('trunk.0.weight', (F, R)), ('trunk.0.bias', (F,)), ('trunk.1.weight', (F,)), ('trunk.1.bias', (F,))]
    head = lambda pre, i_, o_: [(f'{pre}.0.weight', (H, i_)), (f'{pre}.0.bias', (H,)), (f'{pre}.2.weight', (H, H)), (f'{pre}.2.bias', (H,)),
                                (f'{pre}.4.weight', (o_, H)), (f'{pre}.4.bias', (o_,))]
    ash = tr + head('policy', F, A)
    csh = tr + head('Q1', F + A, 1) + head('Q2', F + A, 1)
    return {'encoder': synth_conv_params(esh, 70), 'actor': synth_params(ash, 71), 'critic': synth_params(csh, 72),
            'predictor': synth_params([('weight', (PD, R)), ('bias', (PD,))], 73),
            'projector': synth_params([('trunk.0.weight', (PJ, PD)), ('trunk.0.bias', (PJ,)), ('trunk.2.weight', (PD, PJ)), ('trunk.2.bias', (PD,))], 74),
            'protos': synth_params([('weight', (NP, PD))], 75)}
