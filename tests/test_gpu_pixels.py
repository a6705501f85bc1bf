"""GPU parity of the pixel front end through the C ABI: RandomShiftsAug and the conv encoder (forward + backward) against the
reference's own outputs (tests/golden/pixels_g5.npz) and against the oracle at batch size."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import pixels

pytestmark = pytest.mark.gpu

ENC_KEYS = [f'convnet.{i}.{w}' for i in (0, 2, 4, 6) for w in ('weight', 'bias')]


@pytest.fixture(scope='module')
def lib():
    from exorl_amd import _lib as L
    return L.load()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def flat_params(lib, plist, c_in, hw=84):
    """torch-order tensors -> the library's flat layout (each tensor padded to 4 floats)."""
    n = lib.exorl_encoder_param_floats(c_in, hw)
    flat = np.zeros(n, np.float32)
    off, offs = 0, []
    for p in plist:
        flat[off:off + p.size] = p.reshape(-1)
        offs.append(off)
        off += (p.size + 3) // 4 * 4
    return flat, offs


def run_encoder(lib, plist, x, dh=None, prec=0):
    from exorl_amd import _lib as L
    n, c_in, hw, _ = x.shape
    flat, offs = flat_params(lib, plist, c_in, hw)
    P = dev(flat)
    X = dev(x.astype(np.float32))
    ws = torch.zeros(lib.exorl_encoder_workspace_floats(n, c_in, hw), device='cuda')
    hp = C.c_void_p()
    L.check(lib.exorl_encoder_forward_prec(P.data_ptr(), c_in, hw, X.data_ptr(), n, ws.data_ptr(), C.byref(hp), prec, None))
    D = lib.exorl_encoder_out_dim(hw)
    off = (hp.value - ws.data_ptr()) // 4
    h = ws[off:off + n * D].view(n, D).clone()
    grads = None
    if dh is not None:
        G = torch.zeros_like(P)
        DH = dev(dh.astype(np.float32))
        L.check(lib.exorl_encoder_backward_prec(P.data_ptr(), c_in, hw, X.data_ptr(), n, ws.data_ptr(), DH.data_ptr(), G.data_ptr(), prec, None))
        g = G.cpu().numpy()
        grads = [g[o:o + p.size].reshape(p.shape) for o, p in zip(offs, plist)]
    torch.cuda.synchronize()
    return h.cpu().numpy(), grads


@pytest.mark.parametrize('tag', ['small', 'full'])
def test_aug_vs_reference(lib, gold, tag):
    from exorl_amd import _lib as L
    z = np.load(gold / 'pixels_g5.npz')
    x, sh, want = z[f'aug_{tag}_x'], z[f'aug_{tag}_shift'], z[f'aug_{tag}_y']
    n, c, h, _ = x.shape
    X, S = dev(x), dev(sh.astype(np.int32))
    out = torch.empty(n, c, h, h, device='cuda')
    L.check(lib.exorl_aug_shift(X.data_ptr(), n, c, h, 4, S.data_ptr(), 0, 0, out.data_ptr(), None))
    got = out.cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=0, atol=3e-3)          # pixel units; see oracle/pixels.py on the grid's fp32 rounding
    assert (np.abs(got - want) > 1e-3).mean() < 5e-3
    np.testing.assert_allclose(got, pixels.random_shifts_aug(x, sh), rtol=0, atol=3e-3)
    # Philox shifts: every image is an integer shift in [0, 8]^2 of its replicate-padded self
    L.check(lib.exorl_aug_shift(X.data_ptr(), n, c, h, 4, None, 5, 9, out.data_ptr(), None))
    got = out.cpu().numpy()
    xp = np.pad(x.astype(np.float32), ((0, 0), (0, 0), (4, 4), (4, 4)), mode='edge')
    for b in range(n):
        errs = [np.abs(got[b] - xp[b][:, sy:sy + h, sx:sx + h]).max() for sy in range(9) for sx in range(9)]
        assert min(errs) < 2e-2


@pytest.mark.parametrize('prec', [0, 2, 3])        # 0: fp32 FMA convolutions; 2 / 3: split-bf16 / three-plane MFMA implicit GEMM for the 32-channel layers
@pytest.mark.parametrize('tag', ['c3', 'c9'])
def test_encoder_vs_reference(lib, gold, tag, prec):
    z = np.load(gold / 'pixels_g5.npz')
    p = [z[f'enc_{tag}_param/{k}'] for k in ENC_KEYS]
    x = z[f'enc_{tag}_x']
    dh = np.random.RandomState(7).standard_normal((2, 39200)).astype(np.float32)
    h, grads = run_encoder(lib, p, x, dh, prec)
    np.testing.assert_allclose(h[:, ::97], z[f'enc_{tag}_h_sample'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose([h.astype(np.float64).sum(), (h.astype(np.float64) ** 2).sum()], z[f'enc_{tag}_h_sums'], rtol=1e-5)
    for k, g in zip(ENC_KEYS, grads):
        want = z[f'enc_{tag}_grad/{k}']
        np.testing.assert_allclose(g, want, rtol=2e-4, atol=2e-5 * np.abs(want).max(), err_msg=k)


@pytest.mark.parametrize('prec', [0, 2, 3, 1])
def test_encoder_batch_vs_oracle(lib, prec):
    """A batch that spans several workgroups per layer and ragged output tiles (64x64 images: edges 31, 29, 27, 25)."""
    rs = np.random.RandomState(0)
    rt, at = {0: (1e-4, 1e-5), 2: (1e-4, 1e-5), 3: (1e-4, 1e-5), 1: (3e-2, 3e-3)}[prec]          # plain bf16 operands: 2^-9 per product
    for (n, c, hw) in ((5, 3, 64), (3, 9, 84)):
        p = []
        for l in range(4):
            ci = c if l == 0 else 32
            p += [(rs.standard_normal((32, ci, 3, 3)) / np.sqrt(ci * 9)).astype(np.float32), (0.1 * rs.standard_normal(32)).astype(np.float32)]
        x = rs.randint(0, 256, (n, c, hw, hw)).astype(np.uint8)
        ho, cache = pixels.encoder_fwd(p, x) if hw == 84 else _fwd_any(p, x)
        dh = rs.standard_normal(ho.shape).astype(np.float32)
        go, _ = pixels.encoder_bwd(p, cache, dh)
        h, g = run_encoder(lib, p, x, dh, prec)
        np.testing.assert_allclose(h, ho, rtol=rt, atol=at)
        for i, (a, b) in enumerate(zip(g, go)):
            if prec == 1:       # bf16 dgrad noise is per element; the gradient as a whole keeps its direction
                cos = float((a * b).sum() / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
                assert cos > 0.995, (f'grad {i}', cos)
            else:
                np.testing.assert_allclose(a, b, rtol=2 * rt, atol=2 * at * np.abs(b).max(), err_msg=f'grad {i}')


def test_encoder_three_plane_convolutions_are_fp32_grade(lib):
    """EXORL_PREC_BF16X6 against the fp32 FMA kernels on the same inputs: features and every gradient agree to fp32 rounding (a few 1e-7 of
    the tensor's scale), an order of magnitude closer than the two-plane split — the property that makes it the parity-grade mode of config 4."""
    rs = np.random.RandomState(3)
    p = []
    for l in range(4):
        ci = 3 if l == 0 else 32
        p += [(rs.standard_normal((32, ci, 3, 3)) * np.sqrt(2.0 / (ci * 9))).astype(np.float32), (0.1 * rs.standard_normal(32)).astype(np.float32)]
    x = rs.randint(0, 256, (6, 3, 84, 84)).astype(np.uint8)
    dh = rs.standard_normal((6, 39200)).astype(np.float32)
    h0, g0 = run_encoder(lib, p, x, dh, 0)
    dist = {}
    for prec in (2, 3):
        h, g = run_encoder(lib, p, x, dh, prec)
        e = [float(np.abs(h - h0).max() / np.abs(h0).max())]
        e += [float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30)) for a, b in zip(g[:6], g0[:6])]      # layers 0-2: their gradients pass through dgrad
        dist[prec] = max(e)
    print('encoder vs the fp32 kernels, worst tensor distance: split-bf16', dist[2], 'three-plane', dist[3])
    assert dist[3] < 2e-6 and dist[2] > 4 * dist[3], dist


@pytest.mark.parametrize('n,hw', [(3, 84), (12, 84), (12, 64), (9, 48), (300, 48)])
def test_wave_specialised_convolutions_match_the_strip_and_tile_kernels(lib, n, hw):
    """conv3x3_ws_kernel / conv_wgrad_ws_kernel (the product path of the 32 -> 32 layers) against the round-2 strip / tile kernels they replaced
    (exorl_gemm_tune bits 1073741824 and 64 keep those), same inputs, every precision mode: features, input-side gradients (through dgrad) and all
    weight / bias gradients. n <= 8 takes the one-pass-per-workgroup launch of the forward kernel, n > 8 the per-image loop (what batch 1024
    runs); 84-pixel frames give maps of 39 / 37 / 35, 64-pixel frames 29 / 27 / 25 (three passes where two batches cover a pass's rows), 48-pixel
    frames 21 / 19 / 17 (the weight-gradient kernel declines maps under 384 pixels: only forward / dgrad differ there); 300 images put two
    images on some workgroups of the persistent form and one on the others. Both sides form the
    same products and differ in summation order only (the new forward kernel starts its accumulator at the bias, the weight-gradient kernel walks
    the pixels in row-major passes instead of 16 x 16 tiles); a different last bit of an activation is then re-split into planes by the next
    layer, so the bars are a few units of the mode's own product accuracy relative to the tensor's scale: 5e-6 three-plane, 2e-5 two-plane,
    3e-2 plain bf16 (the bar that mode has against the oracle; one flipped bf16 rounding of an activation is 2^-8 of it). Measured over the four
    cases: <= 7e-7, <= 5e-6, 1e-3 .. 7e-3. Measured
    distances are printed."""
    rs = np.random.RandomState(11)
    p = []
    for l in range(4):
        ci = 3 if l == 0 else 32
        p += [(rs.standard_normal((32, ci, 3, 3)) * np.sqrt(2.0 / (ci * 9))).astype(np.float32), (0.1 * rs.standard_normal(32)).astype(np.float32)]
    x = rs.randint(0, 256, (n, 3, hw, hw)).astype(np.uint8)
    edge = (hw - 3) // 2 + 1 - 6
    dh = rs.standard_normal((n, 32 * edge * edge)).astype(np.float32)
    for prec, tol in ((3, 5e-6), (2, 2e-5), (1, 3e-2)):
        try:
            lib.exorl_gemm_tune(1073741824 | 64)
            h0, g0 = run_encoder(lib, p, x, dh, prec)
        finally:
            lib.exorl_gemm_tune(-1)
        h1, g1 = run_encoder(lib, p, x, dh, prec)
        assert np.isfinite(h1).all() and all(np.isfinite(g).all() for g in g1)
        d = [float(np.abs(h1 - h0).max() / np.abs(h0).max())] + [float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30)) for a, b in zip(g1, g0)]
        print(f'n={n} hw={hw} prec={prec}: ws vs strip/tile kernels, features {d[0]:.1e}, worst gradient {max(d[1:]):.1e}')
        # The gradients are compared on the small batches only. dh is random and independent of the activations, so one ReLU whose pre-activation
        # the two forward kernels put on different sides of zero (they differ by 4e-6 in the two-plane mode) moves a bias gradient by 1 / sqrt(n
        # pixels) of its size: from n = 200 on such flips exist (tools/debug/persist_probe.py: 1e-3 .. 4e-3 in the two-plane mode, none in the
        # three-plane mode) — a property of the comparison, not of either kernel.
        assert d[0] < tol and (n >= 200 or max(d) < tol), (prec, d)
        # the persistent form (n > 8: one workgroup per CU walks the images; n = 300 gives workgroups two images and others one) against one image
        # per workgroup (bit 8388608): the same sums in the same order, bit for bit
        try:
            lib.exorl_gemm_tune(8388608)
            h2, g2 = run_encoder(lib, p, x, dh, prec)
        finally:
            lib.exorl_gemm_tune(-1)
        assert np.array_equal(h1, h2) and all(np.array_equal(a, b) for a, b in zip(g1, g2)), (prec, 'persistent form differs')


def _fwd_any(p, x):
    return pixels.encoder_fwd(p, x)


# ---------------------------------------------------------------------------------------------------- DDPG on pixels
def make_pixel_agent(C_, HW, A, F, H, B, use_tb=True, precision='fp32'):
    from exorl_amd import agents
    return agents.DDPGAgent('ddpg', True, 'pixels', (C_, HW, HW), (A,), 'cuda', 1e-4, F, H, 0.01, 2000, 2, 0.2, 3, B, 0.3, True, use_tb, False,
                            precision=precision)


def load_pixel_params(ag, C_, A, F, H, R=39200):
    import _synth
    esh, ash, csh = pixels.pixel_param_shapes(C_, A, F, H, R)
    ps = [_synth.synth_params(sh, 50 + i) for i, sh in enumerate((esh, ash, csh))]
    for view, p in zip((ag.encoder, ag.actor, ag.critic), ps):
        view.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
    ag.engine.sync_target()
    return [list(p.values()) for p in ps]


def test_pixel_ddpg_vs_reference(gold):
    """3 update() calls of the reference's DDPGAgent(obs_type='pixels') (tests/golden/pixel_ddpg.npz): metrics and final weights."""
    import _synth
    z = np.load(gold / 'pixel_ddpg.npz')
    C_, HW, A, F, H, B, N = [int(v) for v in z['dims']]
    ag = make_pixel_agent(C_, HW, A, F, H, B)
    load_pixel_params(ag, C_, A, F, H)
    noise = _synth.NoiseStream(21)
    shifts = iter(z['shifts'])
    ag.noise_hook = noise.draw
    ag.shift_hook = lambda n: next(shifts)
    keys = [str(k) for k in z['metric_keys']]
    for i in range(N):
        batch = (z[f'batch/{i}/obs'], z[f'batch/{i}/action'], z[f'batch/{i}/reward'], z[f'batch/{i}/discount'], z[f'batch/{i}/next_obs'])
        assert ag.update(iter([]), 2 * i + 1) == {}
        m = ag.update(iter([batch]), 2 * i)
        assert sorted(m.keys()) == keys
        np.testing.assert_allclose(np.array([m[k] for k in keys]), z['metrics'][i], rtol=1e-4, atol=3e-6, err_msg=f'step {i} {keys}')
    for nm, view in (('encoder', ag.encoder), ('actor', ag.actor), ('critic', ag.critic), ('critic_target', ag.critic_target)):
        for k, v in view.state_dict().items():
            v = v.cpu().numpy()
            if f'final/{nm}/{k}' in z.files:
                np.testing.assert_allclose(v, z[f'final/{nm}/{k}'], rtol=1e-4, atol=2e-6, err_msg=f'{nm}.{k}')
            else:
                np.testing.assert_allclose(v.reshape(-1)[::997], z[f'final_sample/{nm}/{k}'], rtol=1e-4, atol=2e-6, err_msg=f'{nm}.{k}')
    # act(): eval = tanh(policy(trunk(encoder(obs)))) on the raw frame
    a = ag.act(z['batch/0/obs'][0], {}, 10**6, True)
    assert a.shape == (A,) and np.all(np.abs(a) <= 1.0)


def test_pixel_act_fused_path_matches_oracle_and_generic_path(lib):
    """act() on one raw frame (ddpg.py:221-238): the fused path (encoder, then trunk_one_kernel and act_fast_kernel<1>: two launches for
    Linear(39200 (+ meta), 50) + LayerNorm + tanh + the three-layer policy + tanh + the TruncatedNormal draw) against the oracle's fp32 actor and
    against the generic multi-launch path (exorl_gemm_tune bit 256), eval and sampling mode, with and without meta columns."""
    import _synth
    from exorl_amd import agents
    from oracle import nets
    C_, HW, A, F, H, B = 3, 84, 6, 50, 1024, 8
    rs = np.random.RandomState(5)
    obs = rs.randint(0, 256, (C_, HW, HW)).astype(np.uint8)
    noise = rs.standard_normal((1, A)).astype(np.float32)
    # plain DDPG: against the oracle
    ag = make_pixel_agent(C_, HW, A, F, H, B)
    enc, actor, critic = load_pixel_params(ag, C_, A, F, H)
    orc = pixels.OraclePixelDDPG(enc, actor, critic)
    feat, _ = pixels.encoder_fwd(orc.enc, obs[None])
    mu, _ = orc._actor(feat)
    a_eval = ag.act(obs, {}, 10**6, True)
    np.testing.assert_allclose(a_eval, mu[0], rtol=1e-4, atol=2e-6)
    ag.noise_hook = lambda shape: noise
    a_s = ag.act(obs, {}, 10**6, False)
    np.testing.assert_allclose(a_s, nets.truncated_normal_sample(mu, noise, 0.2, None)[0], rtol=1e-4, atol=2e-6)
    lib.exorl_gemm_tune(256)
    try:
        np.testing.assert_allclose(ag.act(obs, {}, 10**6, True), a_eval, rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(ag.act(obs, {}, 10**6, False), a_s, rtol=2e-5, atol=2e-6)
    finally:
        lib.exorl_gemm_tune(-1)
    # a meta-conditioned agent (DIAYN: 16 skill columns behind the encoding): fused against generic
    kw = dict(name='diayn', reward_free=True, obs_type='pixels', obs_shape=(C_, HW, HW), action_shape=(A,), device='cuda', lr=1e-4, feature_dim=F,
              hidden_dim=H, critic_target_tau=0.01, num_expl_steps=0, update_every_steps=2, stddev_schedule=0.2, nstep=3, batch_size=B,
              stddev_clip=0.3, init_critic=True, use_tb=True, use_wandb=False)
    torch.manual_seed(3)
    dg = agents.DIAYNAgent(update_skill_every_step=50, skill_dim=16, diayn_scale=1.0, update_encoder=True, skill_type='uniform', **kw)
    meta = {'skill': np.eye(16, dtype=np.float32)[3]}
    fused = dg.act(obs, meta, 10**6, True)
    lib.exorl_gemm_tune(256)
    try:
        generic = dg.act(obs, meta, 10**6, True)
    finally:
        lib.exorl_gemm_tune(-1)
    assert fused.shape == (A,) and np.all(np.abs(fused) <= 1.0)
    np.testing.assert_allclose(fused, generic, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize('precision', ['fp32', 'bf16x3', 'bf16x6'])
def test_pixel_ddpg_batch_vs_oracle(precision):
    """Shipped widths (feature_dim 50, hidden 1024) at a batch that spans many workgroups; 64x64 frames with 9 stacked channels.
    bf16x3: MFMA implicit-GEMM convolutions and split-bf16 Linear layers against the same fp32 oracle and bar."""
    import _synth
    for (C_, HW, A, F, H, B) in ((3, 84, 6, 50, 1024, 24), (9, 64, 4, 50, 256, 16)):
        R = 32 * ((HW - 3) // 2 + 1 - 6) ** 2
        ag = make_pixel_agent(C_, HW, A, F, H, B, precision=precision)
        enc, actor, critic = load_pixel_params(ag, C_, A, F, H, R)
        orc = pixels.OraclePixelDDPG(enc, actor, critic)
        rs = np.random.RandomState(1)
        ns, ns2 = _synth.NoiseStream(4), _synth.NoiseStream(4)
        ag.noise_hook = ns.draw
        sh = []
        ag.shift_hook = lambda n: sh[-1].pop(0)
        for i in range(2):
            obs, nobs = rs.randint(0, 256, (B, C_, HW, HW)).astype(np.uint8), rs.randint(0, 256, (B, C_, HW, HW)).astype(np.uint8)
            b = _synth.synth_batch(3, i, B, 4, A)
            so, sn = rs.randint(0, 9, (B, 2)).astype(np.int32), rs.randint(0, 9, (B, 2)).astype(np.int32)
            sh.append([so, sn])
            m = ag.update(iter([(obs, b[1], b[2], b[3], nobs)]), 2 * i)
            mo = orc.update((obs, b[1], b[2], b[3], nobs), 2 * i, so, sn, ns2.draw((B, A)), ns2.draw((B, A)))
            # Adam's first step moves every weight by lr * sign(g); of the 39200 x 50 trunk weights some have gradients at rounding-noise
            # level, whose sign depends on the summation order (a 32-way split-K of the trunk moved the split-bf16 run's step-1 actor_loss
            # by 2.6e-3 while the 16-way one, like fp32 in either, stays inside 2e-4): the bar below holds for the shipped configuration
            rt = 2e-4
            for k, v in mo.items():
                assert abs(m[k] - v) <= rt * abs(v) + 1e-5, (C_, i, k, m[k], v)
        for got, want in zip(ag.encoder.grads(), orc.last_enc_grads):
            got = got.cpu().numpy().reshape(want.shape)
            # second-step gradients: the two sides' weights already differ by Adam's rounding-noise moves, and ReLU pre-activations
            # next to zero (about 1.5 M of them per image batch here) fall on different sides
            np.testing.assert_allclose(got, want, rtol=5e-3, atol=3e-2 * np.abs(want).max() + 1e-9)


def test_pixel_proto_vs_reference(gold):
    """Proto on pixels (BASELINE config 4 in miniature): 3 update() calls of the reference agent (tests/golden/pixel_proto.npz)."""
    import _synth
    from exorl_amd import agents
    z = np.load(gold / 'pixel_proto.npz')
    C_, HW, A, F, H, B, N, PD, PJ, Q, NP = [int(v) for v in z['dims']]
    ag = agents.ProtoAgent(pred_dim=PD, proj_dim=PJ, queue_size=Q, num_protos=NP, tau=0.1, encoder_target_tau=0.05, topk=3, update_encoder=True,
                           name='proto', reward_free=True, obs_type='pixels', obs_shape=(C_, HW, HW), action_shape=(A,), device='cuda', lr=1e-4,
                           feature_dim=F, hidden_dim=H, critic_target_tau=0.01, num_expl_steps=2000, update_every_steps=2, stddev_schedule=0.2,
                           nstep=3, batch_size=B, stddev_clip=0.3, init_critic=True, use_tb=True, use_wandb=False)
    load_pixel_params(ag, C_, A, F, H)
    psh = [[('weight', (PD, 39200)), ('bias', (PD,))], [('trunk.0.weight', (PJ, PD)), ('trunk.0.bias', (PJ,)), ('trunk.2.weight', (PD, PJ)), ('trunk.2.bias', (PD,))],
           [('weight', (NP, PD))]]
    for i, (view, sh) in enumerate(zip((ag.predictor, ag.projector, ag.protos), psh)):
        view.load_state_dict({k: torch.from_numpy(v) for k, v in _synth.synth_params(sh, 53 + i).items()})
    utils_hard = lambda src, dst: [t.copy_(p) for p, t in zip(src.parameters(), dst.parameters())]
    utils_hard(ag.predictor, ag.predictor_target)
    ag.engine.encoder_target(init=True)
    noise = _synth.NoiseStream(21)
    shifts, us = iter(z['shifts']), iter(z['cat_uniform'])
    ag.noise_hook = noise.draw
    ag.shift_hook = lambda n: next(shifts)
    ag.cat_hook = lambda n: next(us)
    keys = [str(k) for k in z['metric_keys']]
    for i in range(N):
        batch = (z[f'batch/{i}/obs'], z[f'batch/{i}/action'], z[f'batch/{i}/reward'], z[f'batch/{i}/discount'], z[f'batch/{i}/next_obs'])
        m = ag.update(iter([batch]), 2 * i)
        assert sorted(m.keys()) == keys
        np.testing.assert_allclose(np.array([m[k] for k in keys]), z['metrics'][i], rtol=2e-4, atol=3e-6, err_msg=f'step {i} {keys}')
    for nm, view in (('encoder', ag.encoder), ('encoder_target', ag.encoder_target), ('critic', ag.critic), ('protos', ag.protos),
                     ('projector', ag.projector), ('predictor_target', ag.predictor_target)):
        for k, v in view.state_dict().items():
            v = v.cpu().numpy()
            if f'final/{nm}/{k}' in z.files:
                np.testing.assert_allclose(v, z[f'final/{nm}/{k}'], rtol=1e-4, atol=2e-6, err_msg=f'{nm}.{k}')
            else:
                np.testing.assert_allclose(v.reshape(-1)[::997], z[f'final_sample/{nm}/{k}'], rtol=1e-4, atol=2e-6, err_msg=f'{nm}.{k}')
    np.testing.assert_allclose(ag.queue.cpu().numpy(), z['final/queue'], rtol=1e-4, atol=1e-6)
    assert ag.queue_ptr == int(z['final/queue_ptr'])


def _pixel_intr_agent(kind, C_, HW, A, F, H, B, S, precision='fp32'):
    from exorl_amd import agents
    kw = dict(name=kind, reward_free=True, obs_type='pixels', obs_shape=(C_, HW, HW), action_shape=(A,), device='cuda', lr=1e-4, feature_dim=F,
              hidden_dim=H, critic_target_tau=0.01, num_expl_steps=2000, update_every_steps=2, stddev_schedule=0.2, nstep=3, batch_size=B,
              stddev_clip=0.3, init_critic=True, use_tb=True, use_wandb=False, precision=precision)
    if kind == 'icm':
        return agents.ICMAgent(icm_scale=1.0, update_encoder=True, **kw), 'icm'
    if kind == 'icm_apt':
        return agents.ICMAPTAgent(icm_scale=1.0, knn_rms=True, knn_k=3, knn_avg=True, knn_clip=0.0, update_encoder=True, icm_rep_dim=16, **kw), 'icm'
    if kind == 'disagreement':
        return agents.DisagreementAgent(update_encoder=True, **kw), 'disagreement'
    if kind == 'diayn':
        return agents.DIAYNAgent(update_skill_every_step=50, skill_dim=S, diayn_scale=1.0, update_encoder=True, skill_type='uniform', **kw), 'diayn'
    if kind == 'aps':
        return agents.APSAgent(update_task_every_step=5, sf_dim=S, knn_rms=True, knn_k=3, knn_avg=True, knn_clip=0.0001, num_init_steps=4096,
                               lstsq_batch_size=4096, update_encoder=True, **kw), 'aps'
    if kind == 'rnd':
        return agents.RNDAgent(rnd_rep_dim=16, update_encoder=True, rnd_scale=1.0, **kw), 'rnd'
    if kind == 'smm':
        return agents.SMMAgent(z_dim=S, sp_lr=1e-3, vae_lr=1e-4, vae_beta=0.5, state_ent_coef=1.0, latent_ent_coef=1.0, latent_cond_ent_coef=1.0,
                               update_encoder=True, **kw), 'smm'
    raise ValueError(kind)


def _frames(step, B, C_, HW):           # tools/gen_golden.py pixel_intr_frames
    rs = np.random.RandomState(1000 + step)
    return rs.randint(0, 256, (B, C_, HW, HW)).astype(np.uint8), rs.randint(0, 256, (B, C_, HW, HW)).astype(np.uint8)


@pytest.mark.parametrize('precision', ['fp32', 'bf16x6'])
@pytest.mark.parametrize('kind', ['icm', 'icm_apt', 'disagreement', 'diayn', 'aps', 'smm', 'rnd'])
def test_pixel_intrinsic_agents_vs_reference(gold, kind, precision):
    """The module agents on pixel observations against 3 update() calls of the reference's own classes (tests/golden/pixel_<kind>.npz,
    tools/gen_golden.py gen_pixel_intr): obs and next_obs are augmented and encoded once, the module and the encoder step on the module's
    loss, the reward comes from the updated module on the encodings made before that step, and the critic and actor see those encodings
    detached (icm.py:94-139, icm_apt.py:112-158, disagreement.py:88-136, diayn.py:125-176). Metrics per step and final weights."""
    import _synth
    z = np.load(gold / f'pixel_{kind}.npz')
    C_, HW, A, F, H, B, N, S = [int(v) for v in z['dims']]
    ag, mod = _pixel_intr_agent(kind, C_, HW, A, F, H, B, S, precision)       # bf16x6: the parity-grade fast mode of the pixel agents, same bars
    views = (('encoder', ag.encoder), ('actor', ag.actor), ('critic', ag.critic), (mod, getattr(ag, mod)))
    for i, (nm, view) in enumerate(views):
        shapes = [(k, tuple(v.shape)) for k, v in view.state_dict().items()]
        if nm == 'rnd':         # predictor.0 is the agent's encoder (loaded above); Linear layers seed 53, the frozen encoder copy seed 54
            lin = [(k, sh) for k, sh in shapes if len(sh) <= 2 and not k.startswith('normalize_obs') and '.0.convnet.' not in k]
            conv = [(k, sh) for k, sh in shapes if k.startswith('target.0.convnet.')]
            assert [k for k, _ in lin + conv] == [str(k) for k in z['keys/rnd']]
            params = dict(_synth.synth_params(lin, 53))
            params.update(_synth.synth_conv_params(conv, 54))
            view.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=False)
            continue
        assert [k for k, _ in shapes] == [str(k) for k in z[f'keys/{nm}']], nm
        params = (_synth.synth_conv_params if nm == 'encoder' else _synth.synth_params)(shapes, 50 + i)
        view.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    ag.engine.sync_target()
    noise = _synth.NoiseStream(21)
    shifts = iter(z['shifts'])
    ag.noise_hook = noise.draw
    ag.shift_hook = lambda n: next(shifts)
    if kind == 'smm':
        ag.eps_hook = _synth.NoiseStream(33).draw
    keys = [str(k) for k in z['metric_keys']]
    for i in range(N):
        b = _synth.synth_batch(61, i, B, 4, A)
        obs, nobs = _frames(i, B, C_, HW)
        batch = [obs, b[1], b[2], b[3], nobs] + ([z[f'batch/{i}/skill']] if kind in ('diayn', 'aps', 'smm') else [])
        assert ag.update(iter([]), 2 * i + 1) == {}
        m = ag.update(iter([tuple(batch)]), 2 * i)
        assert sorted(m.keys()) == keys
        np.testing.assert_allclose(np.array([m[k] for k in keys]), z['metrics'][i], rtol=2e-4, atol=3e-6, err_msg=f'{kind} step {i} {keys}')
    for nm, view in views + (('critic_target', ag.critic_target),):
        # SMM's skill predictor steps with lr 1e-3 (sp_lr): three Adam steps move a weight by up to 3e-3, and an element whose gradient is at
        # rounding level takes its first steps (lr * g / |g|) differently on the two sides — 1.5e-5 on one of 128 elements of z_pred_net.4
        atol = 2e-5 if (kind, nm) == ('smm', 'smm') else 2e-6
        for k, v in view.state_dict().items():
            v = v.cpu().numpy()
            if f'final/{nm}/{k}' in z.files:
                np.testing.assert_allclose(v, z[f'final/{nm}/{k}'], rtol=1e-4, atol=atol, err_msg=f'{kind} {nm}.{k}')
            else:
                np.testing.assert_allclose(v.reshape(-1)[::997], z[f'final_sample/{nm}/{k}'], rtol=1e-4, atol=atol, err_msg=f'{kind} {nm}.{k}')
    if 'final/rms' in z.files:
        r = ag.intrinsic_reward_rms if kind == 'rnd' else ag.pbe.rms
        M, S_, n = r.M, r.S, r.n
        np.testing.assert_allclose([float(M), float(S_), float(n)], z['final/rms'], rtol=1e-4)
    meta = {'skill': z['batch/0/skill'][0]} if kind in ('diayn', 'aps', 'smm') else {}
    a = ag.act(obs[0], meta, 10**6, True)
    assert a.shape == (A,) and np.all(np.abs(a) <= 1.0)


@pytest.mark.parametrize('kind', ['icm', 'disagreement', 'diayn', 'aps', 'rnd'])
def test_pixel_module_agents_split_bf16_tracks_fp32(kind):
    """The MFMA path of the module agents on pixels (split-bf16 convolutions, split-K trunks with the meta slab, 39200-wide module Linears) against
    the same agent in fp32 — which the reference fixtures pin — on identical inputs at feature_dim 50, hidden 256, batch 16: two updates,
    every metric within 2e-3 relative on the first and 5e-3 on the second (the fp32 path is the one held to the reference's numbers; this guards the operand planes and the
    extra GEMM problems of the bf16x3 build)."""
    import _synth
    C_, HW, A, F, H, B, S = 3, 84, 4, 50, 256, 16, 4
    res = {}
    for precision in ('fp32', 'bf16x3'):
        torch.manual_seed(11)
        ag, mod = _pixel_intr_agent(kind, C_, HW, A, F, H, B, S, precision=precision)
        shapes = [(k, tuple(v.shape)) for k, v in ag.encoder.state_dict().items()]
        ag.encoder.load_state_dict({k: torch.from_numpy(v) for k, v in _synth.synth_conv_params(shapes, 50).items()})
        if kind == 'rnd':
            ag.rnd_target_encoder.load_state_dict({k: torch.from_numpy(v) for k, v in _synth.synth_conv_params(shapes, 54).items()})
        noise, rs = _synth.NoiseStream(21), np.random.RandomState(3)
        ag.noise_hook = noise.draw
        ag.shift_hook = lambda n: rs.randint(0, 9, (n, 2)).astype(np.int32)
        if kind == 'smm':
            ag.eps_hook = _synth.NoiseStream(33).draw
        ms = []
        for i in range(2):
            b = _synth.synth_batch(61, i, B, 4, A)
            obs, nobs = _frames(i, B, C_, HW)
            batch = [obs, b[1], b[2], b[3], nobs]
            if kind in ('diayn', 'aps'):
                m = np.random.RandomState(70 + i).standard_normal((B, S)).astype(np.float32)
                if kind == 'diayn':
                    m = np.eye(S, dtype=np.float32)[np.argmax(m, 1)]
                else:
                    m /= np.linalg.norm(m, axis=1, keepdims=True)
                batch.append(m)
            ms.append(ag.update(iter([tuple(batch)]), 2 * i))
        res[precision] = ms
    for i in range(2):          # second update: both sides have taken Adam's sign-like first step from gradients that differ at rounding level
        rt, at = (2e-3, 2e-5) if i == 0 else (5e-3, 1e-4)
        for k, v in res['fp32'][i].items():
            assert abs(res['bf16x3'][i][k] - v) <= rt * abs(v) + at, (kind, i, k, res['bf16x3'][i][k], v)


def test_pixel_meta_agent_through_the_hbm_sampler_equals_the_tuple_path():
    """DIAYN on pixels fed by the device-resident sampler (frames, action, reward, discount AND the skill rows gathered straight into the
    pixel engine's slots) against the same agent fed the same rows as a tuple: identical metrics, bit for bit."""
    from exorl_amd.engine import ReplayEngine
    from exorl_amd.replay_buffer import ArenaIterator
    from exorl_amd import _lib as L
    C_, HW, A, F, H, B, S = 3, 84, 4, 50, 64, 8, 4
    eng = ReplayEngine((C_, HW, HW), np.uint8, A, S, 6 * 21 + 8, 8, 'cuda')
    rs = np.random.RandomState(0)
    slots = []
    for e in range(6):
        rows = 21
        skill = np.zeros((rows, S), np.float32)
        skill[:, e % S] = 1.0
        ep = dict(observation=rs.randint(0, 256, (rows, C_, HW, HW)).astype(np.uint8), action=rs.uniform(-1, 1, (rows, A)).astype(np.float32),
                  reward=rs.uniform(0, 1, (rows, 1)).astype(np.float32), discount=np.ones((rows, 1), np.float32), skill=skill)
        slots.append(eng.append_episode(ep, ('skill',)))
    eng.set_order(slots)
    eng.seed_philox(3)
    it = ArenaIterator(eng, B, 3, 0.99, 'philox')
    ags = []
    for _ in range(2):
        torch.manual_seed(9)
        ags.append(_pixel_intr_agent('diayn', C_, HW, A, F, H, B, S)[0])
    for i in range(2):
        m0 = ags[0].update(it, 2 * i)
        pairs = eng.last_pairs(B)
        batch = eng.sample(B, 3, 0.99, L.SAMPLER_GIVEN, pairs=pairs)
        assert batch[5].shape == (B, S) and torch.all(batch[5].sum(1) == 1.0)
        m1 = ags[1].update(iter([batch]), 2 * i)
        assert m0 == m1, (i, m0, m1)


@pytest.mark.parametrize('kind', ['ddpg', 'proto', 'rnd', 'diayn', 'aps'])
def test_pixel_agent_pickle_roundtrip_continues_bit_identically(kind):
    """pretrain.py:293-300 torch.save's the whole agent: the pixel agents carry encoder / actor / critic parameters and Adam moments,
    the step counts, the Philox counters of the noise and augmentation streams, and (Proto) encoder_target, proto_opt's second
    Adam state for the encoder, the module and its queue."""
    import io
    import pickle
    import _synth
    from exorl_amd import agents
    C_, HW, A, F, H, B = 3, 84, 4, 50, 64, 8
    kw = dict(name=kind, reward_free=True, obs_type='pixels', obs_shape=(C_, HW, HW), action_shape=(A,), device='cuda', lr=1e-4, feature_dim=F,
              hidden_dim=H, critic_target_tau=0.01, num_expl_steps=2000, update_every_steps=2, stddev_schedule=0.2, nstep=3, batch_size=B,
              stddev_clip=0.3, init_critic=True, use_tb=True, use_wandb=False)
    torch.manual_seed(5)
    S = 4
    if kind == 'proto':
        ag = agents.ProtoAgent(pred_dim=16, proj_dim=32, queue_size=32, num_protos=8, tau=0.1, encoder_target_tau=0.05, topk=3, update_encoder=True, **kw)
    elif kind == 'rnd':                       # + BatchNorm2d buffers, the frozen encoder copy, rnd_opt's Adam state for the encoder, the RMS
        ag = agents.RNDAgent(rnd_rep_dim=16, update_encoder=True, rnd_scale=1.0, **kw)
    elif kind == 'diayn':                     # + the skill columns in front of both trunks
        ag = agents.DIAYNAgent(update_skill_every_step=50, skill_dim=S, diayn_scale=1.0, update_encoder=True, skill_type='uniform', **kw)
    elif kind == 'aps':                       # + CriticSF heads, the kNN RMS
        ag = agents.APSAgent(update_task_every_step=5, sf_dim=S, knn_rms=True, knn_k=3, knn_avg=True, knn_clip=0.0001, num_init_steps=4096,
                             lstsq_batch_size=4096, update_encoder=True, **kw)
    else:
        ag = agents.DDPGAgent(**kw)

    def batch(i):
        rs = np.random.RandomState(100 + i)
        b = (rs.randint(0, 256, (B, C_, HW, HW)).astype(np.uint8), rs.uniform(-1, 1, (B, A)).astype(np.float32),
             rs.uniform(0, 1, (B, 1)).astype(np.float32), np.full((B, 1), 0.97, np.float32), rs.randint(0, 256, (B, C_, HW, HW)).astype(np.uint8))
        if kind in ('diayn', 'aps'):
            m = rs.standard_normal((B, S)).astype(np.float32)
            b += (m / np.linalg.norm(m, axis=1, keepdims=True),)
        return b
    for i in range(2):                       # Philox-drawn noise, shifts and categorical draws: their counters are part of the state
        ag.update(iter([batch(i)]), 2 * i)
    buf = io.BytesIO()
    torch.save({'agent': ag, '_global_step': 2}, buf)
    buf.seek(0)
    ag2 = torch.load(buf, weights_only=False)['agent']           # our own file (finetune.py:250)
    ag3 = pickle.loads(pickle.dumps(ag))
    ms = []
    for a in (ag, ag2, ag3):
        for i in range(2, 4):
            m = a.update(iter([batch(i)]), 2 * i)
        ms.append(m)
    assert ms[0] == ms[1] == ms[2]
    views = ['encoder', 'actor', 'critic', 'critic_target'] + (['encoder_target', 'predictor', 'predictor_target', 'projector', 'protos'] if kind == 'proto' else [])
    views += {'rnd': ['rnd'], 'diayn': ['diayn'], 'aps': ['aps']}.get(kind, [])
    for other in (ag2, ag3):
        assert type(other) is type(ag)
        for nm in views:
            for (k, p), q in zip(getattr(ag, nm).state_dict().items(), getattr(other, nm).state_dict().values()):
                assert torch.equal(p, q), (kind, nm, k)
        if kind == 'proto':
            assert torch.equal(ag.queue, other.queue) and ag.queue_ptr == other.queue_ptr


@pytest.mark.parametrize('precision', ['fp32', 'bf16x6', 'bf16x3'])
def test_config4_proto_pixels_shipped_dims_vs_oracle(precision):
    """BASELINE.json configs[3] at the sizes the 115 update()/s figure is quoted on — jaco frames (3, 84, 84) uint8, A = 9, feature_dim 50,
    hidden 1024, pred_dim 128, proj_dim 512, 512 prototypes, queue 2048, nstep 3 — except the batch: 256 instead of 1024, the largest the
    numpy oracle turns round inside the test budget (~10 s per oracle step; 2 steps x 2 precisions). Everything batch-shaped in the
    kernels (MFMA implicit-GEMM convolutions, 16-way split-K trunk, Sinkhorn over the batch, 3-NN against the queue) spans many
    workgroups at 256. Bar: every metric of both update() calls within 2e-4 of the oracle (pinned to the reference by pixel_proto.npz),
    the bar test_pixel_ddpg_batch_vs_oracle documents for the 39200-wide layers; step 2 runs against the queue step 1 filled."""
    import _synth
    from exorl_amd import agents
    from oracle.proto import OracleProto
    C_, HW, A, F, H, B, PD, PJ, Q, NP = 3, 84, 9, 50, 1024, 256, 128, 512, 2048, 512
    ag = agents.ProtoAgent(pred_dim=PD, proj_dim=PJ, queue_size=Q, num_protos=NP, tau=0.1, encoder_target_tau=0.05, topk=3, update_encoder=True,
                           name='proto', reward_free=True, obs_type='pixels', obs_shape=(C_, HW, HW), action_shape=(A,), device='cuda', lr=1e-4,
                           feature_dim=F, hidden_dim=H, critic_target_tau=0.01, num_expl_steps=2000, update_every_steps=2, stddev_schedule=0.2,
                           nstep=3, batch_size=B, stddev_clip=0.3, init_critic=True, use_tb=True, use_wandb=False, precision=precision)
    enc, actor, critic = load_pixel_params(ag, C_, A, F, H)
    psh = [[('weight', (PD, 39200)), ('bias', (PD,))], [('trunk.0.weight', (PJ, PD)), ('trunk.0.bias', (PJ,)), ('trunk.2.weight', (PD, PJ)), ('trunk.2.bias', (PD,))],
           [('weight', (NP, PD))]]
    pps = [_synth.synth_params(sh, 53 + i) for i, sh in enumerate(psh)]
    for view, p in zip((ag.predictor, ag.projector, ag.protos), pps):
        view.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
    for p, t in zip(ag.predictor.parameters(), ag.predictor_target.parameters()):
        t.copy_(p)
    ag.engine.encoder_target(init=True)
    orc = pixels.OracleProtoPixels(pixels.OraclePixelDDPG(enc, actor, critic), OracleProto([v for p in pps for v in p.values()], queue_size=Q))
    rs = np.random.RandomState(1)
    ns, ns2 = _synth.NoiseStream(4), _synth.NoiseStream(4)
    ag.noise_hook = ns.draw
    sh, us = [], []
    ag.shift_hook = lambda n: sh[-1].pop(0)
    ag.cat_hook = lambda n: us[-1]
    worst = 0.0
    for i in range(2):
        obs, nobs = rs.randint(0, 256, (B, C_, HW, HW)).astype(np.uint8), rs.randint(0, 256, (B, C_, HW, HW)).astype(np.uint8)
        b = _synth.synth_batch(3, i, B, 4, A)
        so, sn = rs.randint(0, 9, (B, 2)).astype(np.int32), rs.randint(0, 9, (B, 2)).astype(np.int32)
        sh.append([so, sn])
        us.append(rs.uniform(size=NP).astype(np.float32))
        m = ag.update(iter([(obs, b[1], b[2], b[3], nobs)]), 2 * i)
        mo = orc.update((obs, b[1], b[2], b[3], nobs), 2 * i, so, sn, us[-1], ns2.draw((B, A)), ns2.draw((B, A)))
        assert sorted(m.keys()) == sorted(mo.keys())
        errs = {k: abs(m[k] - v) / (abs(v) + 1e-2) for k, v in mo.items()}
        print(f'[config 4] proto pixels {precision} B={B} step {i}: ' + ' '.join(f'{k}={e:.1e}' for k, e in errs.items()))
        # fp32 (exact MFMA / FMA products): the 2e-4 bar of the 39200-wide layers. Split-bf16: every metric formed BEFORE the first optimiser
        # step agrees to 4e-6; what is evaluated through the once-stepped critic (step 0's actor_loss 4.1e-4, step 1's Q means 3.4e-4,
        # actor_loss 6.9e-4) does not: Adam's first steps move each of the 2 M trunk weights by lr * sign(g), and the sign of a gradient
        # below the split-bf16 error floor (~1e-5 of its sum of |terms|) is noise. Held to 1e-3 here; config 4 in this mode is documented
        # as outside the 1e-4 bar, the fp32 mode is inside it
        # round 3: `bf16x6` (three planes) is held to the fp32 bar; the test at the config's own batch against the reference itself is
        # test_config4_proto_pixels_b1024_vs_reference (this one keeps the numpy oracle in the loop at a batch it can turn round)
        bar = 1e-3 if precision == 'bf16x3' else 2e-4
        for k, v in mo.items():
            assert abs(m[k] - v) <= bar * abs(v) + 1e-5, (precision, i, k, m[k], v)
            worst = max(worst, errs[k])
    print(f'[config 4] proto pixels {precision} B={B}: worst relative metric error {worst:.2e}')
    np.testing.assert_allclose(ag.queue.cpu().numpy(), orc.proto.queue, rtol=2e-4, atol=1e-4 if precision == 'bf16x3' else 2e-5)


def _config4_agent(z, precision):
    import _synth
    from exorl_amd import agents
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


@pytest.mark.parametrize('precision', ['fp32', 'bf16x6', 'bf16x3'])
def test_config4_proto_pixels_b1024_vs_reference(gold, precision):
    """BASELINE.json configs[3] AT ITS OWN SIZES — jaco frames (3, 84, 84) uint8, A = 9, feature_dim 50, hidden 1024, pred_dim 128, proj_dim 512,
    512 prototypes, queue 2048, nstep 3, BATCH 1024 — three update() calls against the reference ITSELF (tests/golden/config4_proto_b1024.npz,
    tools/gen_golden.py::gen_config4; weights, frames, shifts, uniforms and noise regenerated from seeds).

    The bar. The fixture holds four runs of the reference: fp64, and fp32 three ways (oneDNN convolutions on all threads / on one thread, torch's
    native convolutions). Everything formed BEFORE a network has taken an optimiser step agrees between them to 1e-6, and is held to 1e-4 here.
    Everything evaluated through a stepped network does not: Adam's first steps move each weight by lr * sign(g), a gradient rounding
    difference becomes a whole +-lr step wherever it flips a sign (2 M trunk weights), and the reference's own fp32 runs sit 1.1e-3 (update 0),
    3.0e-3 and 7.3e-3 (updates 1, 2) from its fp64 run on actor_loss (oneDNN; 9e-5 / 7e-4 / 2e-4 on native convolutions) — the reference's fp32
    trajectory is not defined to 1e-4 at this size. So the reference here is its fp64 trajectory, and a metric passes when it is within 1e-4 of
    it OR within TWICE the largest distance any of the reference's own three fp32 runs has from it (`band`: three samples of a spread, hence the
    factor). Update 0 uses its own band (and, for the parity-grade modes, the flat 1e-4 below). Updates 1 and 2 are past the first optimiser
    step, where the spread of one metric at one update is a single draw of a chaotic quantity (critic_target_q: 1.8e-3 at update 1, 3.2e-4 at
    update 2 in the fixture): they share one band per metric, the larger of the two — a kernel change that moves a gradient by 1e-7 of its
    scale re-draws these numbers (the wave-specialised weight-gradient kernel did: update 2's critic_target_q went from 1.9e-4 to 7.5e-4 with
    update 0 and test_encoder_three_plane_convolutions_are_fp32_grade unchanged). Measured (round 3, fp32 mode): update 0 within 9.4e-6 on
    every metric — 100x closer to the fp64 run than the reference's own oneDNN fp32 run (1.1e-3) — updates 1 / 2 within 1.6e-3 / 4.5e-4.
    Measured distances are printed.
    `bf16x3` is NOT parity-grade here and is held to 10x the band, as a regression fence only: 3.2e-4 at update 0 (inside the band), 1.1e-2 at
    updates 1 and 2 (4x outside): its 2^-17 product error flips ~100x more Adam signs than fp32 rounding does. Config 4's parity-grade mode is
    fp32."""
    import _synth
    z = np.load(gold / 'config4_proto_b1024.npz')
    C_, HW, A, F, H, B, N, PD, PJ, Q, NP = [int(v) for v in z['dims']]
    keys = [str(k) for k in z['metric_keys']]
    ref = z['metrics_fp64']
    band = np.max([np.abs(z[nm] - ref) for nm in ('metrics', 'metrics_1thread', 'metrics_no_onednn')], axis=0)
    band[1:] = band[1:].max(axis=0)          # past the first optimiser step: one band per metric (docstring)
    ag = _config4_agent(z, precision)
    ns = _synth.NoiseStream(22)
    ag.noise_hook = ns.draw
    for i in range(N):
        obs, nobs, act, rew, disc, so, sn, u = _synth.config4_inputs(i, B, C_, HW, A, NP)
        sh = [so, sn]
        ag.shift_hook = lambda n: sh.pop(0)
        ag.cat_hook = lambda n: u
        m = ag.update(iter([(obs, act, rew, disc, nobs)]), 2 * i)
        assert sorted(m.keys()) == sorted(keys)
        rel = {k: abs(m[k] - v) / (abs(v) + 1e-12) for k, v in zip(keys, ref[i])}
        print(f'[config 4, B=1024, {precision}] update {i}: ' + ' '.join(f'{k}={e:.1e}' for k, e in rel.items()))
        print(f'[config 4, B=1024] update {i}: the reference\'s own fp32 runs vs its fp64 run: ' +
              ' '.join(f'{k}={b / (abs(v) + 1e-12):.1e}' for k, v, b in zip(keys, ref[i], band[i])))
        fence = 10.0 if precision == 'bf16x3' else 2.0
        for j, (k, v) in enumerate(zip(keys, ref[i])):
            assert abs(m[k] - v) <= max(1e-4 * abs(v) + 1e-6, fence * band[i][j]), (precision, i, k, m[k], v, band[i][j])
        if precision != 'bf16x3' and i == 0:      # before chaos sets in: every metric of the first update within 1e-4 of the fp64 reference, no band needed
            for k, v in zip(keys, ref[i]):
                assert abs(m[k] - v) <= 1e-4 * abs(v) + 1e-6, (i, k, m[k], v)
    # parameters after three updates against the reference's fp64 run, every 997th element: the direction of the accumulated step agrees, and
    # no more steps went the other way than in the reference's own fp32 run (Adam's sign-like first steps: a flipped sign is a 2 lr error)
    worst = 0.0
    for nm in ('encoder', 'actor', 'critic', 'predictor', 'projector', 'protos'):
        for k, t in getattr(ag, nm).state_dict().items():
            got = t.detach().cpu().numpy().reshape(-1)
            got = got if got.size <= 4096 else got[::997]
            init, r64, r32 = (z[f'{tag}/{nm}/{k}'].reshape(-1).astype(np.float64) for tag in ('init_sample', 'final_sample_fp64', 'final_sample'))
            d, d64, d32 = got - init, r64 - init, r32 - init
            if np.linalg.norm(d64) < 1e-12:
                continue
            cos = float(d @ d64 / (np.linalg.norm(d) * np.linalg.norm(d64) + 1e-30))
            cos32 = float(d32 @ d64 / (np.linalg.norm(d32) * np.linalg.norm(d64) + 1e-30))
            assert cos >= min(0.999, cos32 - (2e-2 if precision == 'bf16x3' else 2e-3)), (precision, nm, k, cos, cos32)
            worst = max(worst, 1 - cos)
    print(f'[config 4, B=1024, {precision}] parameter steps vs the reference fp64 run: worst 1 - cos = {worst:.2e}')
