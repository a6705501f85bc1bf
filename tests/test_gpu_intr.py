"""GPU parity of the reward-free agents (RND / ICM / ICM-APT): reference-shaped Python classes -> C ABI -> HIP,
against the reference's recorded trajectories (tests/golden/tiny_{rnd,icm,icm_apt*}.npz) and, at the shipped
widths (H=1024, rep 512, B=1024), against the oracle on the same seeded inputs."""
import numpy as np
import pytest
import torch

import _synth
from oracle.agents import OracleAgent, param_shapes
from oracle.intr import OracleICM, OracleICMAPT, OracleRND, OracleUnsupAgent, intr_param_shapes

pytestmark = pytest.mark.gpu


def ddpg_kw(kind, O, A, H, B, use_tb=True, precision='fp32'):
    return dict(name=kind, reward_free=True, obs_type='states', obs_shape=(O,), action_shape=(A,), device='cuda', lr=1e-4,
                feature_dim=50, hidden_dim=H, critic_target_tau=0.01, num_expl_steps=2000, update_every_steps=2,
                stddev_schedule=0.2, nstep=3, batch_size=B, stddev_clip=0.3, init_critic=True, use_tb=use_tb, use_wandb=False,
                precision=precision)


def make(kind, O, A, H, B, R, use_tb=True, precision='fp32', **kw):
    from exorl_amd import agents
    base = kind.partition('-')[0]
    d = ddpg_kw(base, O, A, H, B, use_tb, precision)
    if base == 'rnd':
        return agents.RNDAgent(rnd_rep_dim=R, update_encoder=True, rnd_scale=1.0, **d)
    if base == 'icm':
        return agents.ICMAgent(icm_scale=1.0, update_encoder=True, **d)
    apt = dict(knn_rms=True, knn_k=3, knn_avg=True, knn_clip=0.0)
    if kind.endswith('kth'):
        apt.update(knn_avg=False, knn_clip=0.0005)
    apt.update(kw)
    return agents.ICMAPTAgent(icm_scale=1.0, update_encoder=True, icm_rep_dim=R, **apt, **d)


def nets_of(ag):
    mod = ('rnd', ag.rnd) if hasattr(ag, 'rnd') else ('icm', ag.icm)
    return [('actor', ag.actor), ('critic', ag.critic), ('critic_target', ag.critic_target), mod]


@pytest.mark.parametrize('kind', ['rnd', 'icm', 'icm_apt', 'icm_apt-kth'])
def test_tiny_trajectory_vs_reference(gold, kind):
    z = np.load(gold / f'tiny_{kind}.npz')
    torch.manual_seed(21)
    ag = make(kind, 5, 3, 32, 8, 16)
    for nm, net in nets_of(ag):
        sd = net.state_dict()
        for k, v in sd.items():
            np.testing.assert_allclose(v.cpu().numpy(), z[f'init/{nm}/{k}'], rtol=0, atol=2e-6, err_msg=f'{nm}.{k}')
        net.load_state_dict({k: torch.from_numpy(z[f'init/{nm}/{k}']) for k in sd})
    noise = iter([z[f'noise/{i}'] for i in range(10)])
    ag.noise_hook = lambda shape: next(noise)
    keys = [str(k) for k in z['metric_keys']]
    for i in range(5):
        batch = tuple(z[f'batch/{i}/{j}'] for j in range(5))
        assert ag.update(iter([]), 2 * i + 1) == {}
        m = ag.update(iter([batch]), 2 * i)
        assert sorted(m.keys()) == keys
        intr = ag.engine._view(ag.engine.batch_slots().reward, 8).cpu().numpy().reshape(-1, 1)
        np.testing.assert_allclose(intr, z['intr_reward'][i], rtol=1e-4, atol=2e-6, err_msg=f'{kind} intr reward step {i}')
        got = np.array([m[k] for k in keys])
        np.testing.assert_allclose(got, z['metrics'][i], rtol=1e-4, atol=2e-6, err_msg=f'{kind} step {i} {keys}')
    for nm, net in nets_of(ag):
        for k, v in net.state_dict().items():
            np.testing.assert_allclose(v.cpu().numpy().astype(np.float64), z[f'final/{nm}/{k}'], rtol=1e-4, atol=2e-6, err_msg=f'{nm}.{k}')
    if 'final/rms' in z.files:
        rms = ag.intrinsic_reward_rms if kind == 'rnd' else ag.pbe.rms
        np.testing.assert_allclose([float(rms.M), float(rms.S), rms.n], z['final/rms'], rtol=2e-5)


def build_pair(kind, O, A, H, B, R, precision='fp32', **kw):
    """exorl_amd agent and oracle agent with the same synthetic parameters."""
    base = kind.partition('-')[0]
    ag = make(kind, O, A, H, B, R, precision=precision, **kw)
    ash, csh = param_shapes('ddpg', O, A, H)
    pa, pc = _synth.synth_params(ash, 3), _synth.synth_params(csh, 4)
    ag.actor.load_state_dict({k: torch.from_numpy(v) for k, v in pa.items()})
    ag.critic.load_state_dict({k: torch.from_numpy(v) for k, v in pc.items()})
    ag.critic_target.load_state_dict(ag.critic.state_dict())
    ish = intr_param_shapes(base, O, A, H, R)
    pi = _synth.synth_params(ish, 5)
    mod = ag.rnd if base == 'rnd' else ag.icm
    sd = {k: torch.from_numpy(v) for k, v in pi.items()}
    if base == 'rnd':
        sd.update({k: v for k, v in mod.state_dict().items() if k.startswith('normalize_obs')})
    mod.load_state_dict(sd)
    ddpg = OracleAgent('ddpg', list(pa.values()), list(pc.values()))
    if base == 'rnd':
        om = OracleRND(list(pi.values()))
    elif base == 'icm':
        om = OracleICM(list(pi.values()))
    else:
        o = dict(knn_rms=True, knn_k=3, knn_avg=True, knn_clip=0.0)
        o.update(kw)
        om = OracleICMAPT(list(pi.values()), **o)
    return ag, OracleUnsupAgent(base, ddpg, om), ish


@pytest.mark.parametrize('kind,dims,kw', [
    ('rnd', (24, 6, 1024, 1024, 512), {}),                                   # configs/agent/rnd.yaml widths, walker shapes
    ('icm', (24, 6, 1024, 1024, 0), {}),
    ('icm_apt', (24, 6, 1024, 1024, 512), dict(knn_k=12)),                   # configs/agent/icm_apt.yaml: k=12, avg, rms, clip 0
    ('icm_apt', (17, 6, 256, 512, 128), dict(knn_k=5, knn_avg=False, knn_clip=0.0005)),
    ('rnd', (9, 2, 136, 100, 40), {}),                                       # unaligned widths -> scalar-load GEMM path
    ('icm', (9, 2, 136, 100, 0), {}),
])
def test_shipped_widths_vs_oracle(kind, dims, kw):
    O, A, H, B, R = dims
    ag, orc, ish = build_pair(kind, O, A, H, B, R, **kw)
    ns = _synth.NoiseStream(11)
    ag.noise_hook = ns.draw
    ns2 = _synth.NoiseStream(11)
    for i in range(3):
        batch = _synth.synth_batch(17, i, B, O, A)
        m = ag.update(iter([batch]), 2 * i)
        mo = orc.update(batch, 2 * i, ns2.draw((B, A)), ns2.draw((B, A)))
        intr = ag.engine._view(ag.engine.batch_slots().reward, B).cpu().numpy().reshape(-1, 1)
        np.testing.assert_allclose(intr, orc.last_intr, rtol=2e-4, atol=1e-5, err_msg=f'{kind} intr reward step {i}')
        for k, v in mo.items():
            assert abs(m[k] - v) <= 1e-4 * abs(v) + 2e-6, (kind, i, k, m[k], v)
    mod = ag.rnd if kind == 'rnd' else ag.icm
    for (k, _), p, want in zip(ish, mod.parameters(), orc.module.p):
        got = p.cpu().numpy().reshape(want.shape)
        # Adam moves every element by ~lr per step whatever its gradient's size, so an element whose gradient is rounding
        # noise can land up to 2*lr*steps away; all but a handful must agree tightly, none may exceed that bound
        bad = np.abs(got - want) > 2e-6 + 1e-4 * np.abs(want)
        assert bad.mean() <= 1e-3, (k, bad.mean())
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=2 * 1e-4 * 3, err_msg=k)
    # module gradients of the last step, tensor by tensor
    n_train = 6 if kind == 'rnd' else len(ish)
    for i in range(n_train):
        got, want = ag.intr.tensor(None, i, 1).cpu().numpy(), orc.module.last_grads[i]
        np.testing.assert_allclose(got.reshape(want.shape), want, rtol=2e-4, atol=1e-8 + 2e-4 * np.abs(want).max(), err_msg=f'grad {ish[i][0]}')


def test_reward_only_and_state_roundtrip():
    """compute_intr_reward alone (train=0) leaves the parameters untouched; RMS / optimiser counters can be saved and restored."""
    O, A, H, B, R = 24, 6, 256, 128, 64
    ag, orc, ish = build_pair('rnd', O, A, H, B, R)
    batch = _synth.synth_batch(3, 0, B, O, A)
    ag.engine.set_batch(*batch)
    s = ag.engine.batch_slots()
    before = ag.intr.flat().clone()
    ag.intr.update(s.obs, s.action, s.next_obs, s.reward, s.reward, train=False)
    torch.cuda.synchronize()
    assert torch.equal(before, ag.intr.flat())
    want = orc.module.reward(batch[0])
    np.testing.assert_allclose(ag.engine._view(s.reward, B).cpu().numpy().reshape(-1, 1), want, rtol=1e-4, atol=1e-6)
    M, S, n = ag.intr.rms_state()
    assert abs(n - (1e-4 + B)) < 1e-9
    ag.intr.set_rms_state(0.25, 2.0, 77.5)
    assert ag.intr.rms_state() == (0.25, 2.0, 77.5)
    ag.intr.set_opt_steps(41)
    assert ag.intr.opt_steps() == 41
    sd = ag.rnd.state_dict()
    assert list(sd)[:3] == ['normalize_obs.running_mean', 'normalize_obs.running_var', 'normalize_obs.num_batches_tracked']
    assert int(sd['normalize_obs.num_batches_tracked']) == 1


def test_bf16_mode_tracks_fp32():
    O, A, H, B, R = 24, 6, 512, 256, 128
    res = {}
    for prec in ('fp32', 'bf16'):
        ag, _, _ = build_pair('icm_apt', O, A, H, B, R, precision=prec, knn_k=4)
        ag.noise_hook = _synth.NoiseStream(5).draw
        for i in range(3):
            m = ag.update(iter([_synth.synth_batch(19, i, B, O, A)]), 2 * i)
        res[prec] = m
    for k, v in res['fp32'].items():
        assert abs(res['bf16'][k] - v) <= 3e-2 * abs(v) + 3e-2, (k, res['bf16'][k], v)
