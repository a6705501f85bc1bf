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
