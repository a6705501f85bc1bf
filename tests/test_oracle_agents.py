"""Oracle (numpy restatement) vs the reference's own outputs for utils.* single ops and the
agent.update() trajectories (tests/golden/utils_g2.npz, tiny_*.npz, full_*.json)."""
import json

import numpy as np
import pytest
import torch

import _synth
from oracle import nets, knn
from oracle.agents import OracleAgent, param_shapes


def test_truncated_normal(gold):
    z = np.load(gold / 'utils_g2.npz')
    for tag, clip in (('clip', 0.3), ('noclip', None)):
        x = nets.truncated_normal_sample(z['tn_mu'], z['tn_noise'], 0.2, clip)
        assert np.array_equal(x, z[f'tn_{tag}_x'])
        # straight-through: d(sum(x*w))/dmu == w exactly
        assert np.array_equal(z[f'tn_{tag}_grad'], np.arange(24, dtype=np.float32).reshape(6, 4))
    np.testing.assert_allclose(nets.normal_log_prob(z['tn_a'], z['tn_mu'], 0.2), z['tn_logprob'], rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(np.full((6, 4), nets.normal_entropy(0.2)), z['tn_entropy'], rtol=1e-6)


def test_schedule(gold):
    z = np.load(gold / 'utils_g2.npz')
    sch = ['0.2', 'linear(1.0,0.1,100)', 'step_linear(1.0,0.5,50,0.1,100)']
    steps = [0, 10, 50, 75, 100, 1000]
    got = np.array([[nets.schedule(s, t) for t in steps] for s in sch])
    assert np.array_equal(got, z['schedule'])


def test_soft_update_and_adam_bit_exact(gold):
    z = np.load(gold / 'utils_g2.npz')
    tgt = [z['soft_tw0'].copy(), z['soft_tb0'].copy()]
    nets.soft_update([z['soft_w'], z['soft_b']], tgt, 0.01)
    assert np.array_equal(tgt[0], z['soft_tw1']) and np.array_equal(tgt[1], z['soft_tb1'])
    p = [z['adam_p0'].copy()]
    opt = nets.Adam(p, 1e-4)
    for g, want in zip(z['adam_grads'], z['adam_p']):
        opt.step(p, [g])
        np.testing.assert_allclose(p[0], want, rtol=0, atol=1e-9)
        assert np.mean(p[0] == want) > 0.95     # same op order: bit-equal except rare libm sqrt/div ulp


def test_rms_pbe_knn(gold):
    z = np.load(gold / 'utils_g2.npz')
    rms = knn.RMS()
    for x, want in zip(z['rms_x'], z['rms_MS']):
        M, S = rms(x)
        np.testing.assert_allclose(np.stack([M, S]), want, rtol=2e-6)
    for tag, (avg, use_rms, clip, k) in dict(avg=(True, False, 0.0, 3), kth=(False, False, 0.0, 3),
                                            avg_rms=(True, True, 0.0005, 4), kth_rms_noclip=(False, True, -1.0, 2)).items():
        r = knn.RMS()
        pbe = knn.PBE(r, clip, k, avg, use_rms)
        np.testing.assert_allclose(pbe(z['pbe_rep']), z[f'pbe_{tag}_r1'], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(pbe(z['pbe_rep'] * np.float32(1.5)), z[f'pbe_{tag}_r2'], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(r.M, z[f'pbe_{tag}_M'], rtol=1e-5)
        np.testing.assert_allclose(r.S, z[f'pbe_{tag}_S'], rtol=1e-5)
    np.testing.assert_allclose(knn.proto_knn_reward(z['knn_z'], z['knn_queue'], 3), z['knn_reward'], rtol=1e-6)


def _load_tiny(gold, kind):
    z = np.load(gold / f'tiny_{kind}.npz')
    O, A, H = 5, 3, 32
    ash, csh = param_shapes(kind.partition('-')[0], O, A, H)
    actor = [z[f'init/actor/{k}'] for k, _ in ash]
    for (k, s), p in zip(ash, actor):
        assert tuple(p.shape) == tuple(s), (k, p.shape, s)
    critic = [z[f'init/critic/{k}'] for k, _ in csh] if csh else None
    return z, ash, csh, actor, critic


@pytest.mark.parametrize('kind', ['td3_bc', 'td3', 'bc', 'ddpg', 'crr', 'crr-exp', 'crr-identity'])
def test_tiny_trajectory(gold, kind):
    z, ash, csh, actor, critic = _load_tiny(gold, kind)
    base, _, wf = kind.partition('-')
    ag = OracleAgent(base, actor, critic, **({'weight_func': wf} if wf else {}))
    kind = base
    keys = [str(k) for k in z['metric_keys']]
    ni = 0
    for i in range(5):
        batch = [z[f'batch/{i}/{j}'] for j in range(5)]
        step = 2 * i if kind == 'ddpg' else i
        if kind == 'bc':
            m = ag.update(batch, step)
        else:
            m = ag.update(batch, step, z[f'noise/{ni}'], z[f'noise/{ni + 1}'])
            ni += 2
        got = np.array([m[k] for k in keys])
        np.testing.assert_allclose(got, z['metrics'][i], rtol=2e-5, atol=2e-6, err_msg=f'{kind} step {i} {keys}')
    for (k, _), p in zip(ash, ag.actor):
        np.testing.assert_allclose(p, z[f'final/actor/{k}'], rtol=1e-4, atol=2e-6, err_msg=k)
    if csh:
        for (k, _), p, t in zip(csh, ag.critic, ag.critic_target):
            np.testing.assert_allclose(p, z[f'final/critic/{k}'], rtol=1e-4, atol=2e-6, err_msg=k)
            np.testing.assert_allclose(t, z[f'final/critic_target/{k}'], rtol=1e-4, atol=2e-6, err_msg=k)


def test_ddpg_skips_odd_steps(gold):
    z, ash, csh, actor, critic = _load_tiny(gold, 'ddpg')
    ag = OracleAgent('ddpg', actor, critic)
    assert ag.update(None, 1) == {}        # ddpg.py:302-303, no batch consumed


@pytest.mark.parametrize('kind', ['td3_bc', 'bc', 'crr'])
def test_full_size_trajectory(gold, kind):
    """BASELINE dims (H=1024). North-star tolerance: per-step losses to 1e-4 rtol vs the reference fp32 path."""
    g = json.load(open(gold / f'full_{kind}.json'))
    O, A, H, B = g['dims']
    ash, csh = param_shapes(kind, O, A, H)
    actor = list(_synth.synth_params(ash, g['param_seed']).values())
    critic = list(_synth.synth_params(csh, g['param_seed'] + 1).values()) if csh else None
    ag = OracleAgent(kind, actor, critic)
    noise = _synth.NoiseStream(g['noise_seed'])
    nsteps = 4
    for i in range(nsteps):
        batch = _synth.synth_batch(g['batch_seed'], i, B, O, A)
        if kind == 'bc':
            m = ag.update(batch, i)
        else:
            m = ag.update(batch, i, noise.draw((B, A)), noise.draw((10 * B if kind == 'crr' else B, A)))
        ref = g['fp32']['metrics'][i]
        for k, v in ref.items():
            assert abs(m[k] - v) <= 1e-4 * abs(v) + 1e-6, (kind, i, k, m[k], v)


@pytest.mark.parametrize('variant', ['cql', 'cql-lagrange'])
def test_cql_tiny_trajectory(gold, variant):
    """Oracle CQL vs the reference's CQLAgent (tiny_cql.npz, and use_critic_lagrange=True in tiny_cql-lagrange.npz): all 11
    metrics, final weights, log_actor_alpha and log_critic_alpha."""
    from oracle.agents import OracleCQL
    z, ash, csh, actor, critic = _load_tiny(gold, variant)
    ag = OracleCQL(actor, critic, use_critic_lagrange=variant.endswith('lagrange'))
    keys = [str(k) for k in z['metric_keys']]
    ni = 0
    for i in range(5):
        batch = [z[f'batch/{i}/{j}'] for j in range(5)]
        m = ag.update(batch, i, *[z[f'noise/{ni + k}'] for k in range(5)])
        ni += 5
        got = np.array([m[k] for k in keys])
        np.testing.assert_allclose(got, z['metrics'][i], rtol=3e-5, atol=3e-6, err_msg=f'cql step {i} {keys}')
    for (k, _), p in zip(ash, ag.actor):
        np.testing.assert_allclose(p, z[f'final/actor/{k}'], rtol=1e-4, atol=2e-6, err_msg=k)
    for (k, _), p, t in zip(csh, ag.critic, ag.critic_target):
        np.testing.assert_allclose(p, z[f'final/critic/{k}'], rtol=1e-4, atol=2e-6, err_msg=k)
        np.testing.assert_allclose(t, z[f'final/critic_target/{k}'], rtol=1e-4, atol=2e-6, err_msg=k)
    np.testing.assert_allclose(ag.log_actor_alpha[0], z['final/log_actor_alpha'], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(ag.log_critic_alpha[0], z['final/log_critic_alpha'], rtol=1e-5, atol=1e-8)


def test_torch_twin_matches_reference_trajectory(gold):
    """bench.py's cpu_baseline times oracle/torch_twin.py (nn.Linear / LayerNorm / Adam, the library ops the reference uses). It must BE
    the reference's update: the reference's own recorded TD3+BC trajectory at BASELINE dims, first 3 steps, every metric to 2e-5
    (same torch, same op order -> differences are thread-count summation order only), and its fp64 twin against the fp64 recording."""
    import json
    from oracle.torch_twin import TorchTwinTD3BC
    g = json.load(open(gold / 'full_td3_bc.json'))
    O, A, H, B = g['dims']
    ash, csh = param_shapes('td3_bc', O, A, H)
    torch.set_num_threads(min(8, torch.get_num_threads()))
    for tag, dt, tol in (('fp32', torch.float32, 6e-5), ('fp64', torch.float64, 1e-9)):
        tw = TorchTwinTD3BC(O, A, H, dtype=dt)
        tw.load(list(_synth.synth_params(ash, g['param_seed']).values()), list(_synth.synth_params(csh, g['param_seed'] + 1).values()))
        ns = _synth.NoiseStream(g['noise_seed'])
        for i in range(3):
            m = tw.update(_synth.synth_batch(g['batch_seed'], i, B, O, A), ns.draw((B, A)), ns.draw((B, A)))
            for k, v in m.items():
                want = g[tag]['metrics'][i][k]
                assert abs(v - want) <= tol * abs(want) + tol * 1e-2, (tag, i, k, v, want)
