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
