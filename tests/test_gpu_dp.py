"""The multi-process data-parallel branch of the PRODUCT (exorl_amd.agents._run_update under torch.distributed) executed for real:
two fresh child processes (tests/_dp_worker.py, started by conftest.py before this process touches the GPU), both on cuda:0 over
gloo, each an exorl_amd agent of batch B/2 fed by make_replay_loader(..., num_workers=2, worker_ids=[rank]). Bars: the replicas end
bit-identical, and equal the single-process update of the concatenated batch (mean over the global batch: td3_bc.py:133-137,154)."""
import json

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dp_run(request):
    run = getattr(request.config, '_dp_children', None)
    if run is None:
        pytest.skip('the data-parallel children were not started (conftest.py starts them when this test is selected on a GPU box)')
    for p, log in run['procs']:
        try:
            rc = p.wait(timeout=600)
        except Exception:
            p.kill()
            raise
        assert rc == 0, open(log).read()[-3000:]
    return run


def test_ranks_draw_different_device_noise(dp_run):
    """Without a noise hook every rank draws the TruncatedNormal noise of ITS rows on the device: the ranks' Philox streams differ (the rank is
    folded into the seed in _AgentBase._build), so the global batch has iid noise rows as the reference's single process draws them."""
    n0, n1 = (np.load(dp_run['out'] / f'noise_rank{r}.npz') for r in (0, 1))
    assert int(n0['seed']) != int(n1['seed'])
    assert n0['block'].shape == n1['block'].shape and not np.array_equal(n0['block'], n1['block'])
    assert abs(float(np.corrcoef(n0['block'].reshape(-1), n1['block'].reshape(-1))[0, 1])) < 0.2


@pytest.mark.parametrize('tag', ['td3_bc_fp32', 'td3_bc_bf16x3', 'bc_fp32'])
def test_two_process_dp_equals_single_process(dp_run, tag):
    import _dp_worker as W
    import _synth
    from exorl_amd.replay_buffer import ReplayBufferStorage, make_replay_loader
    out = dp_run['out']
    r0, r1 = np.load(out / f'{tag}_rank0.npz'), np.load(out / f'{tag}_rank1.npz')
    for k in r0.files:
        assert np.array_equal(r0[k], r1[k]), (tag, k)                  # replicas stay bit-identical
    m0, m1 = (json.load(open(out / f'metrics_rank{r}.json'))[tag] for r in (0, 1))
    assert m0 == m1 and len(m0) == W.STEPS                             # every rank reports the global means
    # the same update in ONE process on the concatenated batch: the two ranks' shards are the two reference workers
    kind, precision = tag.rsplit('_', 1)
    ag = W.build_agent(kind, precision, W.B_GLOBAL, True)
    st = ReplayBufferStorage((), (), dp_run['data'])
    its = [iter(make_replay_loader(st, 10**6, W.B_GLOBAL // 2, 2, True, 1, 0.99, worker_ids=[r], seed=77)) for r in (0, 1)]
    ns = _synth.NoiseStream(9)
    draws = []
    ag.noise_hook = lambda shape: draws.pop(0)
    tol_m, tol_p = {'fp32': (2e-5, (2e-5, 2e-7)), 'bf16x3': (1e-4, (1e-4, 2e-6))}[precision]
    for step in range(W.STEPS):
        draws[:] = [ns.draw((W.B_GLOBAL, W.A)), ns.draw((W.B_GLOBAL, W.A))]
        halves = [next(it) for it in its]
        batch = tuple(torch.cat([h[j] for h in halves]) for j in range(5))
        m = ag.update(iter([batch]), step)
        for k, v in m.items():
            assert abs(m0[step][k] - v) <= tol_m * abs(v) + 1e-6, (tag, step, k, m0[step][k], v)
    nets = [('actor', ag.actor)] + ([('critic', ag.critic), ('critic_target', ag.critic_target)] if hasattr(ag, 'critic') else [])
    for n, net in nets:
        want = torch.cat([p.reshape(-1) for p in net.parameters()]).cpu().numpy()
        if precision == 'fp32':
            np.testing.assert_allclose(r0[n], want, rtol=tol_p[0], atol=tol_p[1], err_msg=f'{tag} {n}')
        else:       # Adam moves rounding-noise gradients by +-lr either way: compare what the steps changed
            d = np.abs(r0[n] - want)
            assert np.mean(d > tol_p[1] + tol_p[0] * np.abs(want)) <= 5e-3 and d.max() <= 6.5e-4, (tag, n, d.max())


@pytest.mark.parametrize('tag', ['td3_fp32', 'cql_fp32', 'cqll_fp32', 'crr_fp32'])
def test_two_process_dp_equals_single_process_hooked(dp_run, tag):
    """TD3 (no batch-global statistic), CRR (value samples repeated n times per row) and CQL — whose _run_update has its own torch.distributed branch (critic gradients, the summed log pi
    behind the entropy temperature, actor gradients) and five noise tensors per step — as two ranks x B/2 against one process x B.
    `cqll` = CQL with use_critic_lagrange: the multiplier steps on the penalty of the GLOBAL batch before any critic gradient exists
    (cql.py:199-213), so phase 0 runs as phases 4 and 5 around a sum-all-reduce of the two penalty sums; the multiplier and its Adam moments
    must come out as in one process."""
    import _dp_worker as W
    import _synth
    from exorl_amd.replay_buffer import ReplayBufferStorage, make_replay_loader
    out = dp_run['out']
    r0, r1 = np.load(out / f'{tag}_rank0.npz'), np.load(out / f'{tag}_rank1.npz')
    for k in r0.files:
        assert np.array_equal(r0[k], r1[k]), (tag, k)
    m0, m1 = (json.load(open(out / f'metrics_rank{r}.json'))[tag] for r in (0, 1))
    assert m0 == m1 and len(m0) == W.STEPS
    kind, precision = tag.rsplit('_', 1)
    ag = W.build_agent(kind, precision, W.B_GLOBAL, True)
    st = ReplayBufferStorage((), (), dp_run['data'])
    its = [iter(make_replay_loader(st, 10**6, W.B_GLOBAL // 2, 2, True, 1, 0.99, worker_ids=[r], seed=78)) for r in (0, 1)]
    ag.noise_hook = W.sliced_noise_hook(_synth.NoiseStream(11), slice(None), W.B_GLOBAL)
    for step in range(W.STEPS):
        halves = [next(it) for it in its]
        batch = tuple(torch.cat([h[j] for h in halves]) for j in range(5))
        m = ag.update(iter([batch]), step)
        assert set(m) == set(m0[step])
        for k, v in m.items():
            assert abs(m0[step][k] - v) <= 5e-5 * abs(v) + 2e-6, (tag, step, k, m0[step][k], v)
    for n, net in (('actor', ag.actor), ('critic', ag.critic), ('critic_target', ag.critic_target)):
        want = torch.cat([p.reshape(-1) for p in net.parameters()]).cpu().numpy()
        # cqll: one critic weight of 41730 came out 4.6e-6 apart (a near-zero gradient element whose two-rank and one-process sums round
        # differently moves by a fraction of lr = 1e-4 under Adam's normalisation); everything else inside the common bar
        np.testing.assert_allclose(r0[n], want, rtol=5e-5, atol=1e-5 if kind == 'cqll' else 5e-7, err_msg=f'{tag} {n}')
    if kind in ('cql', 'cqll'):
        want = np.asarray(ag.engine.cql_alpha_state(), np.float64)          # log_actor_alpha, m, v, log_critic_alpha, m, v
        np.testing.assert_allclose(r0['cql_scalars'], want, rtol=5e-5, atol=1e-9, err_msg=f'{tag} temperature / multiplier state')
        if kind == 'cqll':
            assert abs(want[3]) > 1e-5          # log_critic_alpha has moved off its initial 0


@pytest.mark.parametrize('kind', ['rnd', 'icm', 'icm_apt', 'disagreement', 'diayn', 'aps', 'smm', 'proto'])
def test_two_process_sharded_reward_free_agents_equal_single_process(dp_run, kind):
    """SURVEY 8e, last row: the reward-free agents shard their actor / critic step and run the module step (BatchNorm, RMS, kNN, Sinkhorn,
    queue, the module optimiser) on the all-gathered batch. Two ranks x B/2 must leave bit-identical replicas, give every row the reward the
    single-process run on the concatenated batch gives it (module state included: same kernels on the same B rows -> bit-equal), and land
    on that run's actor / critic to fp32 summation order."""
    import _dp_worker as W
    import _synth
    out = dp_run['out']
    r0, r1 = np.load(out / f'unsup_{kind}_rank0.npz'), np.load(out / f'unsup_{kind}_rank1.npz')
    for k in r0.files:
        if k != 'reward':
            assert np.array_equal(r0[k], r1[k]), (kind, k)
    m0, m1 = (json.load(open(out / f'metrics_rank{r}.json'))[f'unsup_{kind}'] for r in (0, 1))
    assert m0 == m1 and len(m0) == W.USTEPS
    ag = W.build_unsup(kind, W.UB_GLOBAL)
    W.unsup_hooks(ag, kind, 17)
    ns = _synth.NoiseStream(9)
    draws = []
    ag.noise_hook = lambda shape: draws.pop(0)
    Br = W.UB_GLOBAL // 2
    module_names = [n for n, _ in W.unsup_views(ag)][3:]
    for i in range(W.USTEPS):
        draws[:] = [ns.draw((W.UB_GLOBAL, W.UA)), ns.draw((W.UB_GLOBAL, W.UA))]
        m = ag.update(iter([W.unsup_batch(kind, i)]), 2 * i)
        rew = ag.engine._view(ag.engine.batch_slots().reward, W.UB_GLOBAL).cpu().numpy()
        # the module saw the same rows in the same order through the same kernels: its rewards are the single-process ones bit for bit
        assert np.array_equal(np.concatenate([r0['reward'][i], r1['reward'][i]]), rew), (kind, i)
        for k, v in m.items():
            assert abs(m0[i][k] - v) <= 5e-5 * abs(v) + 2e-6, (kind, i, k, m0[i][k], v)
    for n, v in W.unsup_views(ag):
        want = torch.cat([p.reshape(-1) for p in v.parameters()]).cpu().numpy()
        if n in module_names:
            assert np.array_equal(r0[n], want), (kind, n)
        else:
            np.testing.assert_allclose(r0[n], want, rtol=5e-5, atol=5e-7, err_msg=f'{kind} {n}')


@pytest.mark.parametrize('kind,precision', [('td3_bc', 'fp32'), ('td3_bc', 'bf16x3'), ('cql', 'fp32'), ('cqll', 'fp32'), ('bc', 'fp32')])
def test_native_comm_single_rank_runs_the_dp_step(kind, precision):
    """exorl_comm_* and exorl_agent_set_comm on the box's one GPU: a 1-rank RCCL communicator makes every all-reduce an identity, so
    the LIBRARY-driven data-parallel step (gradients finalised into the flat buffers -> ncclAllReduce on the step's stream -> unfused
    Adam; TD3+BC's statistic reduced on the side stream behind the critic backward) must equal the hand-driven phase sequence bit
    for bit with metrics on, and the fused single-GPU step to rounding with metrics off. More ranks cannot share a GPU under RCCL:
    the N-rank behaviour is covered by the gloo two-process test above (same phases, collectives in Python) and by construction."""
    import _synth
    from exorl_amd import _lib as L
    from exorl_amd.comm import Comm
    from exorl_amd.engine import AgentEngine
    from oracle.agents import param_shapes
    O, A, H, B = 24, 6, 128, 64
    comm = Comm(0, 1, Comm.unique_id())
    probe = torch.arange(8, dtype=torch.float32, device='cuda')
    comm.allreduce(probe)
    assert torch.equal(probe.cpu(), torch.arange(8, dtype=torch.float32))
    lagrange = kind == 'cqll'               # CQL with use_critic_lagrange: with a communicator the library runs phase 0 as 4 | all-reduce | 5
    kind = 'cql' if lagrange else kind
    ash, csh = param_shapes(kind, O, A, H)
    pa, pc = list(_synth.synth_params(ash, 1).values()), (list(_synth.synth_params(csh, 2).values()) if csh else None)

    def engine(metrics):
        e = AgentEngine(kind, O, A, H, B, precision=precision, alpha=0.01 if kind == 'cql' else 2.5, use_critic_lagrange=lagrange)
        for i, w in enumerate(pa):
            e.tensor(L.NET_ACTOR, i).copy_(torch.from_numpy(w).reshape(e.tensor(L.NET_ACTOR, i).shape))
        if pc:
            for i, w in enumerate(pc):
                e.tensor(L.NET_CRITIC, i).copy_(torch.from_numpy(w).reshape(e.tensor(L.NET_CRITIC, i).shape))
        e.params_changed(sync_target=True)
        e.set_metrics(metrics)
        return e
    native, hand, native_fast, fused = engine(True), engine(True), engine(False), engine(False)
    native.set_comm(comm)
    native_fast.set_comm(comm)
    ns = _synth.NoiseStream(4)
    n = 3
    for step in range(3):
        batch = _synth.synth_batch(6, step, B, O, A)
        if kind == 'cql':
            nc = np.concatenate([ns.draw((B, A)).ravel(), np.tanh(ns.draw((n, B, A))).ravel(), ns.draw((n, B, A)).ravel(), ns.draw((n, B, A)).ravel()])
            na = ns.draw((B, A))
        else:
            nc, na = ns.draw((B, A)), ns.draw((B, A))
        for e in (native, hand, native_fast, fused):
            e.set_batch(*batch)
        native.update(0.2, nc, na)
        for ph in range(4):
            hand.update_phase(ph, 0.2, nc, na)
        native_fast.update(0.2, nc, na)
        fused.update(0.2, nc, na)
        assert np.array_equal(native.metrics_raw(), hand.metrics_raw())
    nets = [L.NET_ACTOR] + ([L.NET_CRITIC, L.NET_CRITIC_TARGET] if pc else [])
    if kind == 'cql':                       # temperature (and, with use_critic_lagrange, the multiplier) and their Adam moments
        assert np.array_equal(native.cql_alpha_state(), hand.cql_alpha_state())
        assert not lagrange or abs(float(native.cql_alpha_state()[3])) > 1e-6
    for net in nets:
        assert torch.equal(native.flat(net), hand.flat(net)), net
        d = (native_fast.flat(net) - fused.flat(net)).abs()
        tol = 2e-6 + 1e-5 * fused.flat(net).abs()
        assert float((d > tol).float().mean()) <= 5e-3 and float(d.max()) <= 6.5e-4, (net, float(d.max()))
    native.set_comm(None)
    native_fast.set_comm(None)


@pytest.mark.parametrize('precision', ['fp32', 'bf16x3'])
def test_dp_step_with_the_library_communicator_is_capturable(precision):
    """The data-parallel step as ONE hipGraph: device Philox sampler, finalised gradients, RCCL all-reduces (a 1-rank communicator on the
    box's one GPU; captured on the step's stream and its side stream), unfused Adam — replayed four times and compared bit for bit with
    the same agent stepping eagerly over an identically seeded replay engine."""
    import _synth
    from exorl_amd import agents
    from exorl_amd.comm import Comm
    from exorl_amd.engine import ReplayEngine
    from exorl_amd.replay_buffer import ArenaIterator
    O, A, H, B, E, T = 24, 6, 256, 64, 12, 30
    comm = Comm(0, 1, Comm.unique_id())

    def side():
        eng = ReplayEngine((O,), np.float32, A, 0, E * (T + 1) + 8, E + 4, 'cuda')
        slots = []
        for e in range(E):
            ep = _synth.synth_episodes(40 + e, [T], O, A)[0]
            slots.append(eng.append_episode(ep))
        eng.set_order(slots)
        eng.seed_philox(7)
        torch.manual_seed(3)
        ag = agents.TD3BCAgent('td3_bc', (O,), (A,), 'cuda', 1e-4, H, 0.01, '0.2', 1, B, 0.3, False, 2.5, precision=precision, seed=5)
        ag.engine.set_comm(comm)
        return ag, ArenaIterator(eng, B, 1, 0.99, 'philox')
    (g, git), (e, eit) = side(), side()
    assert g.enable_graph(git) and g.engine.graph_captures == 1
    for step in range(4):
        assert g.update(git, step) == {} and e.update(eit, step) == {}
    for net in ('actor', 'critic', 'critic_target'):
        for (k, p), q in zip(getattr(g, net).state_dict().items(), getattr(e, net).state_dict().values()):
            assert torch.equal(p, q), (net, k)
    assert g.engine.opt_steps() == e.engine.opt_steps() == (4, 4)
    g.disable_graph()
    g.engine.set_comm(None)
    e.engine.set_comm(None)
