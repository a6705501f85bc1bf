"""Oracle restatement of RandomShiftsAug and the conv encoder vs the reference's own outputs (tests/golden/pixels_g5.npz)."""
import numpy as np
import pytest

from oracle import pixels

ENC_KEYS = [f'convnet.{i}.{w}' for i in (0, 2, 4, 6) for w in ('weight', 'bias')]


@pytest.mark.parametrize('tag', ['small', 'full'])
def test_random_shifts_aug(gold, tag):
    z = np.load(gold / 'pixels_g5.npz')
    y = pixels.random_shifts_aug(z[f'aug_{tag}_x'], z[f'aug_{tag}_shift'])
    np.testing.assert_allclose(y, z[f"aug_{tag}_y"], rtol=0, atol=3e-3)      # pixel units (0..255): 1.2e-5 after /255 (see oracle note)
    assert (np.abs(y - z[f"aug_{tag}_y"]) > 1e-3).mean() < 5e-3
    # the augmentation is an integer-pixel shift of the replicate-padded image up to the grid's fp32 rounding
    x = z[f'aug_{tag}_x'].astype(np.float32)
    xp = np.pad(x, ((0, 0), (0, 0), (4, 4), (4, 4)), mode='edge')
    h = x.shape[2]
    for b, (sx, sy) in enumerate(z[f'aug_{tag}_shift']):
        np.testing.assert_allclose(y[b], xp[b][:, sy:sy + h, sx:sx + h], atol=2e-2)


@pytest.mark.parametrize('tag', ['c3', 'c9'])
def test_encoder_forward_backward(gold, tag):
    z = np.load(gold / 'pixels_g5.npz')
    p = [z[f'enc_{tag}_param/{k}'] for k in ENC_KEYS]
    h, cache = pixels.encoder_fwd(p, z[f'enc_{tag}_x'])
    assert h.shape == (2, 39200)
    np.testing.assert_allclose(h[:, ::97], z[f'enc_{tag}_h_sample'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose([h.astype(np.float64).sum(), (h.astype(np.float64) ** 2).sum()], z[f'enc_{tag}_h_sums'], rtol=1e-5)
    dh = np.random.RandomState(7).standard_normal(h.shape).astype(np.float32)
    grads, dx = pixels.encoder_bwd(p, cache, dh, need_dx=True)
    for k, g in zip(ENC_KEYS, grads):
        want = z[f'enc_{tag}_grad/{k}']
        np.testing.assert_allclose(g, want, rtol=2e-4, atol=2e-5 * np.abs(want).max(), err_msg=k)
    np.testing.assert_allclose(dx[0, 0], z[f'enc_{tag}_dx0'], rtol=2e-4, atol=1e-6 * max(1.0, np.abs(z[f'enc_{tag}_dx0']).max()))
