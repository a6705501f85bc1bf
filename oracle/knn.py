"""ORACLE (test infrastructure only — never imported by the product path).

numpy restatement of the particle-entropy intrinsic rewards:
  utils.RMS / utils.PBE  (/root/reference/utils/utils.py:257-319, APT reward icm_apt.py:96-100)
  Proto kNN reward        (/root/reference/agents/unsupervised_learning/proto.py:114-119)
Pinned by tests/golden/utils_g2.npz.
"""
import numpy as np

F32 = np.float32


def pairwise_l2(src, tgt):
    d = src[:, None, :].astype(F32) - tgt[None, :, :].astype(F32)
    return np.sqrt((d * d).sum(-1, dtype=F32)).astype(F32)


def topk_smallest(dist, k):
    return np.sort(dist, axis=1)[:, :k]


class RMS:
    """utils.py:257-276 (running mean / variance, Chan update, n starts at epsilon)."""

    def __init__(self, epsilon=1e-4, shape=(1,)):
        self.M = np.zeros(shape, F32)
        self.S = np.ones(shape, F32)
        self.n = epsilon

    def __call__(self, x):
        bs = x.shape[0]
        delta = x.mean(0, dtype=F32) - self.M
        new_M = (self.M + delta * F32(bs) / F32(self.n + bs)).astype(F32)
        var = x.var(0, ddof=1, dtype=F32)
        new_S = ((self.S * F32(self.n) + var * F32(bs) + np.square(delta) * F32(self.n) * F32(bs) / F32(self.n + bs))
                 / F32(self.n + bs)).astype(F32)
        self.M, self.S = new_M, new_S
        self.n += bs
        return self.M, self.S


class PBE:
    """utils.py:279-319."""

    def __init__(self, rms, knn_clip, knn_k, knn_avg, knn_rms):
        self.rms, self.knn_clip, self.knn_k, self.knn_avg, self.knn_rms = rms, knn_clip, knn_k, knn_avg, knn_rms

    def __call__(self, rep):
        b1 = rep.shape[0]
        reward = topk_smallest(pairwise_l2(rep, rep), self.knn_k)
        if not self.knn_avg:
            reward = reward[:, -1].reshape(-1, 1)
            if self.knn_rms:
                reward = (reward / self.rms(reward)[0]).astype(F32)
            if self.knn_clip >= 0.0:
                reward = np.maximum(reward - F32(self.knn_clip), F32(0))
        else:
            reward = reward.reshape(-1, 1)
            if self.knn_rms:
                reward = (reward / self.rms(reward)[0]).astype(F32)
            if self.knn_clip >= 0.0:
                reward = np.maximum(reward - F32(self.knn_clip), F32(0))
            reward = reward.reshape(b1, self.knn_k).mean(1, keepdims=True, dtype=F32)
        return np.log(reward + F32(1.0)).astype(F32)


def proto_knn_reward(z, queue, topk=3):
    """proto.py:114-119: distance to the topk-th nearest queue row."""
    return topk_smallest(pairwise_l2(z, queue), topk)[:, -1:]
