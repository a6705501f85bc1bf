"""End to end: a directory of reference-format episodes (.npz, replay_buffer.py:18-29) -> HBM replay -> agent.update() loop
(exorl_amd.train_offline, the loop of train_offline.py:90-123)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

O, A, H, B = 11, 3, 128, 64


def write_dataset(path, n_eps=12, ep_len=60, seed=0):
    from exorl_amd.replay_buffer import ReplayBufferStorage
    import _synth
    rs = np.random.RandomState(seed)
    W = rs.standard_normal((O, A)).astype(np.float32) / np.sqrt(O)
    path.mkdir(parents=True, exist_ok=True)
    for e in range(n_eps):
        rows = ep_len + 1
        obs = rs.standard_normal((rows, O)).astype(np.float32)
        act = np.tanh(obs @ W).astype(np.float32)            # the "expert": next action is a function of the PREVIOUS row's obs
        act = np.vstack([np.zeros((1, A), np.float32), act[:-1]])
        ep = dict(observation=obs, action=act, reward=rs.uniform(0, 1, (rows, 1)).astype(np.float32), discount=np.ones((rows, 1), np.float32))
        np.savez_compressed(path / f'20240101T000000_{e}_{ep_len}.npz', **ep)
    return W


def test_bc_learns_the_expert_from_a_directory(tmp_path):
    from exorl_amd import agents
    from exorl_amd.train_offline import train_offline
    write_dataset(tmp_path / 'buffer')
    torch.manual_seed(0)
    ag = agents.BCAgent('bc', (O,), (A,), 'cuda', 1e-3, H, B, '0.2', True)
    logged = []
    evals = []
    rows = train_offline(ag, tmp_path / 'buffer', 600, B, 0.99, eval_every_steps=200, log_every_steps=100,
                         eval_fn=lambda step, a: evals.append(step), log_fn=lambda step, m: logged.append((step, m['actor_loss'])))
    assert evals == [0, 200, 400] and [s for s, _ in logged] == [0, 100, 200, 300, 400, 500]
    assert logged[-1][1] < logged[0][1] - 1.0, logged           # -log N(a; mu, 0.2) falls as mu approaches the expert action
    assert all('fps' in r and 'total_time' in r for _, r in rows)


def test_td3_bc_offline_loop_runs_through_the_captured_graph(tmp_path):
    from exorl_amd import agents
    from exorl_amd.train_offline import train_offline
    write_dataset(tmp_path / 'buffer')
    torch.manual_seed(0)
    ag = agents.TD3BCAgent('td3_bc', (O,), (A,), 'cuda', 1e-4, H, 0.01, '0.2', 1, B, 0.3, False, 2.5)
    before = [p.clone() for p in ag.actor.parameters()]
    rows = train_offline(ag, tmp_path / 'buffer', 50, B, 0.99, log_every_steps=10)
    assert ag._graph_iter is not None                           # static directory + Philox sampler: sample+update captured once
    assert ag.engine.opt_steps() == (50, 50)
    assert any(not torch.equal(p, q) for p, q in zip(before, ag.actor.parameters()))
    assert len(rows) == 5 and rows[0][1]['step'] == 0


def test_offline_loop_equals_the_oracle_loop(tmp_path):
    """the whole offline harness — dataset directory -> OfflineReplayBuffer-semantics loader (MT19937 streams seeded
    like `utils.set_seed_everywhere`) -> TD3+BC agent.update() — against the oracle doing the same loop on the CPU: the oracle's
    OfflineReplayBuffer restatement draws the batches (pinned to the reference by replay_offline_*.npz), OracleAgent makes the
    updates (pinned by tiny_/full_td3_bc fixtures). Logged metrics must agree step by step to 1e-4 and the final actor too."""
    import random
    import _synth
    from exorl_amd import agents
    from exorl_amd.replay_buffer import load_episode
    from exorl_amd.train_offline import train_offline
    from oracle.agents import OracleAgent
    from oracle.replay import OracleOfflineReplay
    write_dataset(tmp_path / 'buffer', n_eps=7, ep_len=40)
    torch.manual_seed(0)
    ag = agents.TD3BCAgent('td3_bc', (O,), (A,), 'cuda', 1e-4, H, 0.01, '0.2', 1, B, 0.3, True, 2.5)
    pa = [p.cpu().numpy().copy() for p in ag.actor.parameters()]
    pc = [p.cpu().numpy().copy() for p in ag.critic.parameters()]
    ns_gpu, ns_cpu = _synth.NoiseStream(5), _synth.NoiseStream(5)
    ag.noise_hook = ns_gpu.draw
    random.seed(3)
    np.random.seed(3)
    steps = 6
    rows = train_offline(ag, tmp_path / 'buffer', steps, B, 0.99, replay_buffer_size=10**6, log_every_steps=1, sampler='mt19937', use_graph=False)
    # the same loop on the CPU
    files = sorted((tmp_path / 'buffer').glob('*.npz'))
    directory = {f'episode_{fn.stem.split("_")[1]}_{fn.stem.split("_")[2]}.npz': load_episode(fn) for fn in files}
    rb = OracleOfflineReplay(None, 10**6, 0, 0.99, relabel=False)
    rb.seed(3, 3)
    orc = OracleAgent('td3_bc', [p.reshape(q.shape) for p, q in zip(pa, pa)], pc)
    for step in range(steps):
        _, batch = rb.sample_batch(directory, B)
        mo = orc.update(batch, step, ns_cpu.draw((B, A)), ns_cpu.draw((B, A)))
        got = rows[step][1]
        for k, v in mo.items():
            assert abs(got[k] - v) <= 1e-4 * abs(v) + 1e-6, (step, k, got[k], v)
    for p, q in zip(ag.actor.parameters(), orc.actor):
        np.testing.assert_allclose(p.cpu().numpy().reshape(q.shape), q, rtol=1e-4, atol=2e-6)
