"""Oracle restatement of the RND / ICM / ICM-APT agents vs the reference's own outputs (tests/golden/tiny_{rnd,icm,icm_apt*}.npz:
5 update() calls of the reference agents on CPU, reward_free=True)."""
import numpy as np
import pytest

from oracle.agents import OracleAgent, param_shapes
from oracle.intr import OracleAPS, OracleDIAYN, OracleDisagreement, OracleICM, OracleICMAPT, OracleRND, OracleUnsupAgent, intr_param_shapes

O, A, H, R = 5, 3, 32, 16
KINDS = ['rnd', 'icm', 'icm_apt', 'icm_apt-kth', 'disagreement', 'diayn', 'aps']
MODULE = {'rnd': 'rnd', 'disagreement': 'disagreement', 'diayn': 'diayn', 'aps': 'aps'}


def build_oracle(z, kind):
    base = kind.partition('-')[0]
    S = 4 if base in ('diayn', 'aps') else 0
    ash, csh = param_shapes('aps' if base == 'aps' else 'ddpg', O + S, A, H, sf_dim=4 if base == 'aps' else None)
    ddpg = OracleAgent('aps' if base == 'aps' else 'ddpg', [z[f'init/actor/{k}'] for k, _ in ash], [z[f'init/critic/{k}'] for k, _ in csh],
                       sf_dim=4 if base == 'aps' else None)
    mod_name = MODULE.get(base, 'icm')
    ish = intr_param_shapes(base, O, A, H, S or R)
    params = [z[f'init/{mod_name}/{k}'] for k, _ in ish]
    for (k, s), p in zip(ish, params):
        assert tuple(p.shape) == tuple(s), (k, p.shape, s)
    if base == 'rnd':
        mod = OracleRND(params)
    elif base == 'icm':
        mod = OracleICM(params)
    elif base == 'disagreement':
        mod = OracleDisagreement(params)
    elif base == 'diayn':
        mod = OracleDIAYN(params)
    elif base == 'aps':
        mod = OracleAPS(params, knn_k=3)
    else:
        mod = OracleICMAPT(params, knn_k=3, **(dict(knn_avg=False, knn_clip=0.0005) if kind.endswith('kth') else {}))
    return OracleUnsupAgent(base, ddpg, mod), ash, csh, ish, mod_name


@pytest.mark.parametrize('kind', KINDS)
def test_tiny_unsup_trajectory(gold, kind):
    z = np.load(gold / f'tiny_{kind}.npz')
    ag, ash, csh, ish, mod_name = build_oracle(z, kind)
    keys = [str(k) for k in z['metric_keys']]
    for i in range(5):
        batch = [z[f'batch/{i}/{j}'] for j in range(6 if kind in ('diayn', 'aps') else 5)]
        assert ag.update(batch, 2 * i + 1, None, None) == {}
        m = ag.update(batch, 2 * i, z[f'noise/{2 * i}'], z[f'noise/{2 * i + 1}'])
        np.testing.assert_allclose(ag.last_intr, z['intr_reward'][i], rtol=2e-5, atol=2e-6, err_msg=f'{kind} intr step {i}')
        got = np.array([m[k] for k in keys])
        np.testing.assert_allclose(got, z['metrics'][i], rtol=5e-5, atol=2e-6, err_msg=f'{kind} step {i} {keys}')
    for (k, _), p in zip(ish, ag.module.p):
        np.testing.assert_allclose(p, z[f'final/{mod_name}/{k}'], rtol=1e-4, atol=2e-6, err_msg=k)
    for (k, _), p in zip(ash, ag.ddpg.actor):
        np.testing.assert_allclose(p, z[f'final/actor/{k}'], rtol=1e-4, atol=2e-6, err_msg=k)
    for (k, _), p, t in zip(csh, ag.ddpg.critic, ag.ddpg.critic_target):
        np.testing.assert_allclose(p, z[f'final/critic/{k}'], rtol=1e-4, atol=2e-6, err_msg=k)
        np.testing.assert_allclose(t, z[f'final/critic_target/{k}'], rtol=1e-4, atol=2e-6, err_msg=k)
    rms = ag.module.rms if kind == 'rnd' else getattr(ag.module, 'pbe', None) and ag.module.pbe.rms
    if rms is not None and 'final/rms' in z:
        np.testing.assert_allclose([rms.M[0], rms.S[0], rms.n], z['final/rms'], rtol=2e-5)
    if kind == 'rnd':      # BatchNorm buffers (two forward passes per update: update_rnd + compute_intr_reward)
        np.testing.assert_allclose(ag.module.running_mean, z['final/rnd/normalize_obs.running_mean'], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(ag.module.running_var, z['final/rnd/normalize_obs.running_var'], rtol=1e-5, atol=1e-6)
        assert ag.module.num_batches == int(z['final/rnd/normalize_obs.num_batches_tracked'])


def test_tiny_proto_trajectory(gold):
    from oracle.proto import OracleProto, OracleProtoAgent, proto_param_shapes, uniform_from_normal, sinkhorn_knopp
    z = np.load(gold / 'tiny_proto.npz')
    ash, csh = param_shapes('ddpg', O, A, H)
    ddpg = OracleAgent('ddpg', [z[f'init/actor/{k}'] for k, _ in ash], [z[f'init/critic/{k}'] for k, _ in csh])
    psh = proto_param_shapes(O, 8, 16, 6)
    params = [z['init/' + k.replace('predictor.', 'predictor/').replace('projector.', 'projector/').replace('protos.', 'protos/')] for k, _ in psh]
    for (k, s), p in zip(psh, params):
        assert tuple(p.shape) == tuple(s), (k, p.shape, s)
    ag = OracleProtoAgent(ddpg, OracleProto(params, queue_size=24))
    keys = [str(k) for k in z['metric_keys']]
    for i in range(5):
        batch = [z[f'batch/{i}/{j}'] for j in range(5)]
        assert ag.update(batch, 2 * i + 1, None, None, None) == {}
        m = ag.update(batch, 2 * i, uniform_from_normal(z[f'noise/{3 * i}']), z[f'noise/{3 * i + 1}'], z[f'noise/{3 * i + 2}'])
        np.testing.assert_allclose(ag.last_intr, z['intr_reward'][i], rtol=2e-5, atol=2e-6, err_msg=f'intr step {i}')
        got = np.array([m[k] for k in keys])
        np.testing.assert_allclose(got, z['metrics'][i], rtol=5e-5, atol=2e-6, err_msg=f'step {i} {keys}')
    fin = {'predictor.weight': 'predictor/weight', 'predictor.bias': 'predictor/bias', 'projector.trunk.0.weight': 'projector/trunk.0.weight',
           'projector.trunk.0.bias': 'projector/trunk.0.bias', 'projector.trunk.2.weight': 'projector/trunk.2.weight',
           'projector.trunk.2.bias': 'projector/trunk.2.bias', 'protos.weight': 'protos/weight'}
    for (k, _), p in zip(psh, ag.module.p):
        np.testing.assert_allclose(p, z['final/' + fin[k]], rtol=1e-4, atol=2e-6, err_msg=k)
    np.testing.assert_allclose(ag.module.pt[0], z['final/predictor_target/weight'], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(ag.module.queue, z['final/queue'], rtol=1e-5, atol=1e-6)
    assert ag.module.queue_ptr == int(z['final/queue_ptr'])
    # sinkhorn: every sample's assignment sums to 1, prototypes are used equally up to the last column normalisation
    q = sinkhorn_knopp(np.random.RandomState(0).standard_normal((32, 6)).astype(np.float32) * 3)
    np.testing.assert_allclose(q.sum(1), 1.0, rtol=1e-5)


def test_tiny_smm_trajectory(gold):
    """SMM incl. the reference's (B,B) reward broadcast (see OracleSMMAgent)."""
    from oracle.intr import OracleSMM, OracleSMMAgent, smm_param_shapes
    z = np.load(gold / 'tiny_smm.npz')
    Z = 4
    ash, csh = param_shapes('ddpg', O + Z, A, H)
    ddpg = OracleAgent('ddpg', [z[f'init/actor/{k}'] for k, _ in ash], [z[f'init/critic/{k}'] for k, _ in csh])
    ssh = smm_param_shapes(O, Z, H)
    params = [z[f'init/smm/{k}'] for k, _ in ssh]
    for (k, s), p in zip(ssh, params):
        assert tuple(p.shape) == tuple(s), (k, p.shape, s)
    ag = OracleSMMAgent(ddpg, OracleSMM(params))
    keys = [str(k) for k in z['metric_keys']]
    for i in range(5):
        batch = [z[f'batch/{i}/{j}'] for j in range(6)]
        assert ag.update(batch, 2 * i + 1, None, None, None) == {}
        m = ag.update(batch, 2 * i, z[f'noise/{3 * i}'], z[f'noise/{3 * i + 1}'], z[f'noise/{3 * i + 2}'])
        got = np.array([m[k] for k in keys])
        np.testing.assert_allclose(got, z['metrics'][i], rtol=5e-5, atol=2e-6, err_msg=f'step {i} {keys}')
    for (k, _), p in zip(ssh, ag.module.p):
        # vae_lr = 1e-2: Adam's normalised step turns last-bit gradient differences into ~1e-5 parameter differences
        want = z[f'final/smm/{k}']
        bad = np.abs(p - want) > 2e-6 + 1e-4 * np.abs(want)
        assert bad.mean() <= 1e-3, (k, bad.mean())
        np.testing.assert_allclose(p, want, rtol=2e-3, atol=2e-4, err_msg=k)
    for (k, _), p in zip(ash, ag.ddpg.actor):
        np.testing.assert_allclose(p, z[f'final/actor/{k}'], rtol=1e-4, atol=2e-6, err_msg=k)
    for (k, _), p, t in zip(csh, ag.ddpg.critic, ag.ddpg.critic_target):
        np.testing.assert_allclose(p, z[f'final/critic/{k}'], rtol=1e-4, atol=2e-6, err_msg=k)
        np.testing.assert_allclose(t, z[f'final/critic_target/{k}'], rtol=1e-4, atol=2e-6, err_msg=k)
