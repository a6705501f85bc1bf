"""GPU parity of the reward-free agents (RND / ICM / ICM-APT): reference-shaped Python classes -> C ABI -> HIP,
against the reference's recorded trajectories (tests/golden/tiny_{rnd,icm,icm_apt*}.npz) and, at the shipped
widths (H=1024, rep 512, B=1024), against the oracle on the same seeded inputs."""
import numpy as np
import pytest
import torch

import _synth
from oracle.agents import OracleAgent, param_shapes
from oracle.intr import OracleAPS, OracleDIAYN, OracleDisagreement, OracleICM, OracleICMAPT, OracleRND, OracleUnsupAgent, intr_param_shapes

pytestmark = pytest.mark.gpu


def assert_mostly_close(got, want, rtol, atol, hard_atol, frac=5e-3, err_msg=''):
    """fp32 parity at full width has two knife edges the reference shares with any other fp32 implementation: Adam moves an
    element whose gradient is rounding noise by +-lr either way, and a ReLU pre-activation within an ulp of zero flips its mask
    (about one of the 2M hidden activations per step). All but `frac` of the elements must agree to (rtol, atol); none may be off
    by more than hard_atol."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    bad = np.abs(got - want) > atol + rtol * np.abs(want)
    assert bad.mean() <= frac, (err_msg, float(bad.mean()))
    np.testing.assert_allclose(got, want, rtol=rtol, atol=hard_atol, err_msg=err_msg)


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
    if base == 'disagreement':
        return agents.DisagreementAgent(update_encoder=True, **d)
    if base == 'aps':
        return agents.APSAgent(update_task_every_step=5, sf_dim=R, knn_rms=True, knn_k=kw.get('knn_k', 3), knn_avg=True, knn_clip=0.0001,
                               num_init_steps=4096, lstsq_batch_size=4096, update_encoder=True, **d)
    if base == 'diayn':
        return agents.DIAYNAgent(update_skill_every_step=50, skill_dim=R, diayn_scale=1.0, update_encoder=True, skill_type='uniform', **d)
    apt = dict(knn_rms=True, knn_k=3, knn_avg=True, knn_clip=0.0)
    if kind.endswith('kth'):
        apt.update(knn_avg=False, knn_clip=0.0005)
    apt.update(kw)
    return agents.ICMAPTAgent(icm_scale=1.0, update_encoder=True, icm_rep_dim=R, **apt, **d)


def module_of(ag):
    for nm in ('rnd', 'icm', 'disagreement', 'diayn', 'aps'):
        if hasattr(ag, nm):
            return nm, getattr(ag, nm)


def nets_of(ag):
    mod = module_of(ag)
    return [('actor', ag.actor), ('critic', ag.critic), ('critic_target', ag.critic_target), mod]


@pytest.mark.parametrize('kind', ['rnd', 'icm', 'icm_apt', 'icm_apt-kth', 'disagreement', 'diayn', 'aps'])
def test_tiny_trajectory_vs_reference(gold, kind):
    z = np.load(gold / f'tiny_{kind}.npz')
    torch.manual_seed(21)
    ag = make(kind, 5, 3, 32, 8, 4 if kind in ('diayn', 'aps') else 16)
    for nm, net in nets_of(ag):
        sd = net.state_dict()
        for k, v in sd.items():
            np.testing.assert_allclose(v.cpu().numpy(), z[f'init/{nm}/{k}'], rtol=0, atol=2e-6, err_msg=f'{nm}.{k}')
        net.load_state_dict({k: torch.from_numpy(z[f'init/{nm}/{k}']) for k in sd})
    noise = iter([z[f'noise/{i}'] for i in range(10)])
    ag.noise_hook = lambda shape: next(noise)
    keys = [str(k) for k in z['metric_keys']]
    for i in range(5):
        batch = tuple(z[f'batch/{i}/{j}'] for j in range(6 if kind in ('diayn', 'aps') else 5))
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
    ash, csh = param_shapes('aps' if base == 'aps' else 'ddpg', O + (R if base in ('diayn', 'aps') else 0), A, H, sf_dim=R if base == 'aps' else None)
    pa, pc = _synth.synth_params(ash, 3), _synth.synth_params(csh, 4)
    ag.actor.load_state_dict({k: torch.from_numpy(v) for k, v in pa.items()})
    ag.critic.load_state_dict({k: torch.from_numpy(v) for k, v in pc.items()})
    ag.critic_target.load_state_dict(ag.critic.state_dict())
    ish = intr_param_shapes(base, O, A, H, R)
    pi = _synth.synth_params(ish, 5)
    mod = module_of(ag)[1]
    sd = {k: torch.from_numpy(v) for k, v in pi.items()}
    if base == 'rnd':
        sd.update({k: v for k, v in mod.state_dict().items() if k.startswith('normalize_obs')})
    mod.load_state_dict(sd)
    ddpg = OracleAgent('aps' if base == 'aps' else 'ddpg', list(pa.values()), list(pc.values()), sf_dim=R if base == 'aps' else None)
    if base == 'rnd':
        om = OracleRND(list(pi.values()))
    elif base == 'icm':
        om = OracleICM(list(pi.values()))
    elif base == 'disagreement':
        om = OracleDisagreement(list(pi.values()))
    elif base == 'diayn':
        om = OracleDIAYN(list(pi.values()))
    elif base == 'aps':
        om = OracleAPS(list(pi.values()), knn_k=kw.get('knn_k', 3))
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
    ('disagreement', (24, 6, 1024, 1024, 0), {}),                            # configs/agent/disagreement.yaml widths
    ('diayn', (24, 6, 1024, 1024, 16), {}),                                  # configs/agent/diayn.yaml: skill_dim 16
    ('diayn', (9, 2, 136, 100, 5), {}),
    ('aps', (24, 6, 1024, 1024, 10), dict(knn_k=12)),                        # configs/agent/aps.yaml: sf_dim 10, knn_k 12
    ('aps', (9, 2, 136, 100, 5), {}),
    ('icm_apt', (24, 6, 256, 128, 64), dict(knn_k=12)),                      # mid size: strict gradient comparison
    ('disagreement', (24, 6, 256, 128, 0), {}),
])
def test_shipped_widths_vs_oracle(kind, dims, kw):
    _shipped_widths(kind, dims, kw, 'fp32')


@pytest.mark.parametrize('kind,dims,kw', [
    ('rnd', (24, 6, 1024, 1024, 512), {}),
    ('icm_apt', (24, 6, 1024, 1024, 512), dict(knn_k=12)),
    ('disagreement', (24, 6, 1024, 1024, 0), {}),
    ('diayn', (24, 6, 1024, 1024, 16), {}),
])
def test_shipped_widths_vs_oracle_bf16x3(kind, dims, kw):
    """Split-bf16 MFMA operands (DDPG step on hi/lo planes, module MLPs split inside the generic GEMM) against the fp32 oracle:
    metrics at the same 1e-4, intrinsic rewards row by row."""
    _shipped_widths(kind, dims, kw, 'bf16x3')


def _shipped_widths(kind, dims, kw, precision):
    O, A, H, B, R = dims
    ag, orc, ish = build_pair(kind, O, A, H, B, R, precision=precision, **kw)
    ns = _synth.NoiseStream(11)
    ag.noise_hook = ns.draw
    ns2 = _synth.NoiseStream(11)
    for i in range(3):
        batch = _synth.synth_batch(17, i, B, O, A)
        if kind == 'diayn':
            batch = batch + (np.eye(R, dtype=np.float32)[np.random.RandomState(i).randint(0, R, B)],)
        if kind == 'aps':
            t = np.random.RandomState(i).standard_normal((B, R)).astype(np.float32)
            batch = batch + ((t / np.linalg.norm(t, axis=1, keepdims=True)).astype(np.float32),)
        m = ag.update(iter([batch]), 2 * i)
        mo = orc.update(batch, 2 * i, ns2.draw((B, A)), ns2.draw((B, A)))
        intr = ag.engine._view(ag.engine.batch_slots().reward, B).cpu().numpy().reshape(-1, 1)
        # APS: the reward is a 12-NN distance in a 10-d feature space divided by its running mean — the rounding-noise-sized moves
        # Adam makes on the feature net (see assert_mostly_close) shift neighbour distances by ~1e-4 relative after the first step
        rt, at = (1e-3, 1e-4) if kind == 'aps' else (2e-4, 5e-5)
        if precision != 'fp32':      # split-bf16 products carry ~1e-5 of the logit / feature scale; a reward is a difference of those
            rt, at = 1e-3, 2e-4
        assert_mostly_close(intr, orc.last_intr, rt, at, 2e-3 * np.abs(orc.last_intr).max() + 10 * at, 2e-2, f'{kind} intr reward step {i}')
        # actor_loss = -mean_b Q(s_b, pi(s_b)) is a cancelling mean here (4e-3 against |Q| ~ 0.1-0.5 per sample): in split-bf16 mode
        # the 1e-4 is taken against a 0.1 scale for it, as for any signed mean; fp32 mode keeps the tight absolute floor
        at_m = 1e-5 if precision != 'fp32' else 2e-6
        for k, v in mo.items():
            assert abs(m[k] - v) <= 1e-4 * abs(v) + at_m, (kind, i, k, m[k], v)
    mod = module_of(ag)[1]
    for (k, _), p, want in zip(ish, mod.parameters(), orc.module.p):
        assert_mostly_close(p.cpu().numpy().reshape(want.shape), want, 1e-4, 2e-6, 2 * 1e-4 * 3, err_msg=k)
    if precision != 'fp32':
        return
    # module gradients of the last step, tensor by tensor
    n_train = 6 if kind == 'rnd' else len(ish)
    for i in range(n_train):
        got, want = ag.intr.tensor(None, i, 1).cpu().numpy(), orc.module.last_grads[i]
        # after the first Adam step the two sides' weights differ by rounding noise, so a ReLU pre-activation next to zero can
        # fall on different sides: that changes one hidden unit's row of dW for one sample (seen: 30 of 7680 elements)
        big = B * H >= 1 << 19
        assert_mostly_close(got.reshape(want.shape), want, 2e-4, 1e-8 + (1e-2 if big else 2e-4) * np.abs(want).max(),
                            0.1 * np.abs(want).max(), frac=1e-2, err_msg=f'grad {ish[i][0]}')


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


@pytest.mark.parametrize('kind', ['td3_bc', 'cql', 'rnd', 'icm_apt', 'proto', 'smm'])
def test_pickle_roundtrip_continues_bit_identically(kind):
    """pretrain.py:293-300 / finetune.py:222-252 torch.save and torch.load the whole agent object: a restored agent must
    continue the run exactly (parameters, Adam moments and step counts, running statistics)."""
    import io
    import pickle
    O, A, H, B, R = 17, 6, 128, 64, 32
    if kind in ('rnd', 'icm_apt'):
        ag = make(kind, O, A, H, B, R)
    elif kind == 'proto':                    # categorical candidate picks come from the module's own Philox stream
        ag = make_proto(O, A, H, B, 16, 32, 8, 32)
    elif kind == 'smm':                      # so does the VAE's epsilon
        ag = make_smm(O, A, H, B, 4)
    else:
        from test_gpu_agent import make as make_offline
        ag = make_offline(kind, O, A, H, B)
    extra = lambda i: ((np.eye(4, dtype=np.float32)[np.random.RandomState(i).randint(0, 4, B)],) if kind == 'smm' else ())
    upd = (lambda a, i: a.update(iter([_synth.synth_batch(23, i, B, O, A) + extra(i)]), 2 * i))

    def hook(agent, seed):
        ns = _synth.NoiseStream(seed)
        if kind == 'cql':
            agent.noise_hook = lambda shape, dist='normal': (ns.draw(shape) if dist == 'normal' else np.tanh(ns.draw(shape)))
        else:
            agent.noise_hook = ns.draw
    hook(ag, 1)
    for i in range(3):
        upd(ag, i)
    buf = io.BytesIO()
    torch.save({'agent': ag, '_global_step': 3}, buf)            # what pretrain.py:297-300 does
    buf.seek(0)
    payload = torch.load(buf, weights_only=False)                 # our own file (finetune.py:250)
    ag2 = payload['agent']
    assert type(ag2) is type(ag) and payload['_global_step'] == 3
    ag3 = pickle.loads(pickle.dumps(ag))
    for a in (ag, ag2, ag3):
        hook(a, 2)
        for i in range(3, 5):
            m = upd(a, i)
    nets = [ag.actor] + ([ag.critic, ag.critic_target] if hasattr(ag, 'critic') else [])
    for other in (ag2, ag3):
        onets = [other.actor] + ([other.critic, other.critic_target] if hasattr(other, 'critic') else [])
        for n1, n2 in zip(nets, onets):
            for (k, p), q in zip(n1.named_parameters(), n2.parameters()):
                assert torch.equal(p, q), (kind, k)
        if hasattr(ag, 'intr'):
            assert torch.equal(ag.intr.flat(), other.intr.flat())
            assert ag.intr.rms_state() == other.intr.rms_state() and ag.intr.opt_steps() == other.intr.opt_steps() == 5
        assert ag.engine.opt_steps() == other.engine.opt_steps()
        assert other.use_tb == ag.use_tb and other.hidden_dim == ag.hidden_dim


def test_diayn_zero_copy_sampler_path_matches_iterator_path():
    """DIAYN with the HBM sampler: obs and the stored skill land directly in the agent's [obs | skill] rows; the result must
    equal feeding the same sampled batch through the plain iterator path."""
    from exorl_amd.engine import ReplayEngine
    from exorl_amd.replay_buffer import ArenaIterator
    O, A, H, B, S = 11, 3, 64, 32, 4
    rs = np.random.RandomState(0)
    eng = ReplayEngine((O,), np.float32, A, S, 4096, 16, 'cuda')
    slots = []
    for e in range(6):
        rows = 20 + e
        ep = dict(observation=rs.standard_normal((rows, O)).astype(np.float32), action=rs.uniform(-1, 1, (rows, A)).astype(np.float32),
                  reward=rs.uniform(0, 1, (rows, 1)).astype(np.float32), discount=np.ones((rows, 1), np.float32),
                  skill=np.eye(S, dtype=np.float32)[rs.randint(0, S, rows)])
        slots.append(eng.append_episode(ep, ('skill',)))
    eng.set_order(slots)
    agents = []
    for path in ('zero_copy', 'iterator'):
        torch.manual_seed(3)
        ag = make('diayn', O, A, H, B, S)
        ag.noise_hook = _synth.NoiseStream(9).draw
        eng.seed_philox(5)
        it = ArenaIterator(eng, B, 3, 0.99, 'philox')
        for i in range(3):
            if path == 'zero_copy':
                m = ag.update(it, 2 * i)
            else:
                m = ag.update(iter([tuple(t.cpu().numpy() for t in next(it))]), 2 * i)
        agents.append((ag, m))
    (a0, m0), (a1, m1) = agents
    assert m0 == m1
    for p, q in zip(a0.actor.parameters() + a0.critic.parameters() + a0.diayn.parameters(),
                    a1.actor.parameters() + a1.critic.parameters() + a1.diayn.parameters()):
        assert torch.equal(p, q)


# ---------------------------------------------------------------------------------------------------- Proto (states)
def make_proto(O, A, H, B, pred, proj, protos, queue, use_tb=True, precision='fp32'):
    from exorl_amd import agents
    return agents.ProtoAgent(pred_dim=pred, proj_dim=proj, queue_size=queue, num_protos=protos, tau=0.1, encoder_target_tau=0.05, topk=3,
                             update_encoder=True, **ddpg_kw('proto', O, A, H, B, use_tb, precision))


PROTO_VIEWS = ['predictor', 'predictor_target', 'projector', 'protos']


def test_proto_tiny_trajectory_vs_reference(gold):
    from oracle.proto import uniform_from_normal
    z = np.load(gold / 'tiny_proto.npz')
    torch.manual_seed(21)
    ag = make_proto(5, 3, 32, 8, 8, 16, 6, 24)
    nets = [('actor', ag.actor), ('critic', ag.critic), ('critic_target', ag.critic_target)] + [(n, getattr(ag, n)) for n in PROTO_VIEWS]
    for nm, net in nets:
        sd = net.state_dict()
        for k, v in sd.items():
            np.testing.assert_allclose(v.cpu().numpy(), z[f'init/{nm}/{k}'], rtol=0, atol=2e-6, err_msg=f'{nm}.{k}')
        net.load_state_dict({k: torch.from_numpy(z[f'init/{nm}/{k}']) for k in sd})
    stream = iter([z[f'noise/{i}'] for i in range(15)])
    ag.cat_hook = lambda n: uniform_from_normal(next(stream))
    ag.noise_hook = lambda shape: next(stream)
    keys = [str(k) for k in z['metric_keys']]
    for i in range(5):
        batch = tuple(z[f'batch/{i}/{j}'] for j in range(5))
        assert ag.update(iter([]), 2 * i + 1) == {}
        m = ag.update(iter([batch]), 2 * i)
        assert sorted(m.keys()) == keys
        intr = ag.engine._view(ag.engine.batch_slots().reward, 8).cpu().numpy().reshape(-1, 1)
        np.testing.assert_allclose(intr, z['intr_reward'][i], rtol=1e-4, atol=2e-6, err_msg=f'intr reward step {i}')
        got = np.array([m[k] for k in keys])
        np.testing.assert_allclose(got, z['metrics'][i], rtol=1e-4, atol=2e-6, err_msg=f'step {i} {keys}')
    for nm, net in nets:
        for k, v in net.state_dict().items():
            np.testing.assert_allclose(v.cpu().numpy(), z[f'final/{nm}/{k}'], rtol=1e-4, atol=2e-6, err_msg=f'{nm}.{k}')
    np.testing.assert_allclose(ag.queue.cpu().numpy(), z['final/queue'], rtol=1e-5, atol=1e-6)
    assert ag.queue_ptr == int(z['final/queue_ptr'])


@pytest.mark.parametrize('dims', [(24, 6, 1024, 1024, 128, 512, 512, 2048),      # configs/agent/proto.yaml widths, walker states
                                  (9, 2, 72, 100, 20, 36, 10, 40)])
def test_proto_shipped_widths_vs_oracle(dims):
    from oracle.proto import OracleProto, OracleProtoAgent, proto_param_shapes
    O, A, H, B, pred, proj, protos, queue = dims
    ag = make_proto(O, A, H, B, pred, proj, protos, queue)
    ash, csh = param_shapes('ddpg', O, A, H)
    pa, pc = _synth.synth_params(ash, 3), _synth.synth_params(csh, 4)
    ag.actor.load_state_dict({k: torch.from_numpy(v) for k, v in pa.items()})
    ag.critic.load_state_dict({k: torch.from_numpy(v) for k, v in pc.items()})
    ag.critic_target.load_state_dict(ag.critic.state_dict())
    psh = proto_param_shapes(O, pred, proj, protos)
    pp = list(_synth.synth_params(psh, 5).values())
    for view, ts in ((ag.predictor, pp[0:2]), (ag.projector, pp[2:6]), (ag.protos, pp[6:7]), (ag.predictor_target, pp[0:2])):
        for p, t in zip(view.parameters(), ts):
            p.copy_(torch.from_numpy(t).reshape(p.shape))
    orc = OracleProtoAgent(OracleAgent('ddpg', list(pa.values()), list(pc.values())), OracleProto(pp, queue_size=queue))
    rs = np.random.RandomState(2)
    ns, ns2 = _synth.NoiseStream(11), _synth.NoiseStream(11)
    ag.noise_hook = ns.draw
    us = []
    ag.cat_hook = lambda n: us[-1]
    for i in range(3):
        us.append(rs.uniform(0, 1, protos))
        batch = _synth.synth_batch(17, i, B, O, A)
        m = ag.update(iter([batch]), 2 * i)
        mo = orc.update(batch, 2 * i, us[-1], ns2.draw((B, A)), ns2.draw((B, A)))
        intr = ag.engine._view(ag.engine.batch_slots().reward, B).cpu().numpy().reshape(-1, 1)
        assert_mostly_close(intr, orc.last_intr, 2e-4, 2e-5, 0.05 * np.abs(orc.last_intr).max() + 0.05, 2e-2, f'proto intr reward step {i}')
        for k, v in mo.items():
            assert abs(m[k] - v) <= 2e-4 * abs(v) + 2e-6, (i, k, m[k], v)
    views = [ag.predictor, ag.projector, ag.protos]
    got = [p for v in views for p in v.parameters()]
    for (k, _), p, want in zip(psh, got, orc.module.p):
        assert_mostly_close(p.cpu().numpy().reshape(want.shape), want, 1e-4, 2e-6, 2 * 1e-4 * 3, err_msg=k)
    grads = [g for v in views for g in v.grads()]
    for (k, _), g, want in zip(psh, grads, orc.module.last_grads):
        assert_mostly_close(g.cpu().numpy().reshape(want.shape), want, 5e-4, 1e-8 + 1e-3 * np.abs(want).max(), 0.1 * np.abs(want).max(),
                            frac=2e-2, err_msg=f'grad {k}')
    np.testing.assert_allclose(ag.predictor_target.parameters()[0].cpu().numpy(), orc.module.pt[0], rtol=1e-4, atol=2e-6)
    assert ag.queue_ptr == orc.module.queue_ptr


# ---------------------------------------------------------------------------------------------------- SMM (states)
def make_smm(O, A, H, B, Z, use_tb=True, precision='fp32'):
    from exorl_amd import agents
    return agents.SMMAgent(z_dim=Z, sp_lr=1e-3, vae_lr=1e-2, vae_beta=0.5, state_ent_coef=1.0, latent_ent_coef=1.0, latent_cond_ent_coef=1.0,
                           update_encoder=True, **ddpg_kw('smm', O, A, H, B, use_tb, precision))


def test_smm_tiny_trajectory_vs_reference(gold):
    """Includes the reference's (B,B) reward broadcast (see SMMAgent's docstring): metrics and weights of its own 5-step run."""
    z = np.load(gold / 'tiny_smm.npz')
    torch.manual_seed(21)
    ag = make_smm(5, 3, 32, 8, 4)
    nets = [('actor', ag.actor), ('critic', ag.critic), ('critic_target', ag.critic_target), ('smm', ag.smm)]
    for nm, net in nets:
        sd = net.state_dict()
        for k, v in sd.items():      # same RNG draws; the 150x150 QR differs in the last bits across host CPUs
            np.testing.assert_allclose(v.cpu().numpy(), z[f'init/{nm}/{k}'], rtol=0, atol=1e-5, err_msg=f'{nm}.{k}')
        net.load_state_dict({k: torch.from_numpy(z[f'init/{nm}/{k}']) for k in sd})
    stream = iter([z[f'noise/{i}'] for i in range(15)])
    ag.eps_hook = lambda shape: next(stream)
    ag.noise_hook = lambda shape: next(stream)
    keys = [str(k) for k in z['metric_keys']]
    for i in range(5):
        batch = tuple(z[f'batch/{i}/{j}'] for j in range(6))
        assert ag.update(iter([]), 2 * i + 1) == {}
        m = ag.update(iter([batch]), 2 * i)
        assert sorted(m.keys()) == keys
        got = np.array([m[k] for k in keys])
        np.testing.assert_allclose(got, z['metrics'][i], rtol=1e-4, atol=3e-6, err_msg=f'step {i} {keys}')
    for nm, net in nets:
        for k, v in net.state_dict().items():
            want = z[f'final/{nm}/{k}']
            assert_mostly_close(v.cpu().numpy(), want, 1e-4, 2e-6, 2e-3 * max(1.0, float(np.abs(want).max())), frac=2e-3, err_msg=f'{nm}.{k}')


@pytest.mark.parametrize('dims', [(24, 6, 1024, 1024, 4), (9, 2, 72, 100, 5)])     # configs/agent/smm.yaml: hidden 1024, z_dim 4
def test_smm_shipped_widths_vs_oracle(dims):
    from oracle.intr import OracleSMM, OracleSMMAgent, smm_param_shapes
    O, A, H, B, Z = dims
    ag = make_smm(O, A, H, B, Z)
    ash, csh = param_shapes('ddpg', O + Z, A, H)
    pa, pc = _synth.synth_params(ash, 3), _synth.synth_params(csh, 4)
    ag.actor.load_state_dict({k: torch.from_numpy(v) for k, v in pa.items()})
    ag.critic.load_state_dict({k: torch.from_numpy(v) for k, v in pc.items()})
    ag.critic_target.load_state_dict(ag.critic.state_dict())
    ssh = smm_param_shapes(O, Z, H)
    ps = _synth.synth_params(ssh, 5)
    ag.smm.load_state_dict({k: torch.from_numpy(v) for k, v in ps.items()})
    orc = OracleSMMAgent(OracleAgent('ddpg', list(pa.values()), list(pc.values())), OracleSMM(list(ps.values())))
    ns, ns2 = _synth.NoiseStream(11), _synth.NoiseStream(11)
    ag.eps_hook = ns.draw
    ag.noise_hook = ns.draw
    for i in range(2):
        batch = _synth.synth_batch(17, i, B, O, A)
        batch = batch + (np.eye(Z, dtype=np.float32)[np.random.RandomState(i).randint(0, Z, B)],)
        m = ag.update(iter([batch]), 2 * i)
        mo = orc.update(batch, 2 * i, ns2.draw((B, 128)), ns2.draw((B, A)), ns2.draw((B, A)))
        intr = ag.engine._view(ag.engine.batch_slots().reward, B).cpu().numpy().reshape(-1, 1)
        # vae_lr = 1e-2: after the first step the two VAEs differ by Adam's rounding-noise moves (+-1e-2 on near-zero gradients)
        rt = 2e-4 if i == 0 else 2e-2
        assert_mostly_close(intr, orc.last_intr, rt, 1e-4, 0.1 * np.abs(orc.last_intr).max(), 2e-2, f'smm reward step {i}')
        for k, v in mo.items():
            assert abs(m[k] - v) <= (2e-4 if i == 0 else 5e-3) * abs(v) + 1e-5, (i, k, m[k], v)
    for i, ((k, _), want) in enumerate(zip(ssh, orc.module.last_pred_grads + orc.module.last_vae_grads)):
        got = ag.intr.tensor(None, i, 1).cpu().numpy().reshape(want.shape)
        assert_mostly_close(got, want, 5e-2, 1e-8 + 2e-2 * np.abs(want).max(), 0.5 * np.abs(want).max() + 1e-6, frac=2e-2, err_msg=f'grad {k}')
