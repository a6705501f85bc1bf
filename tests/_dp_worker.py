"""One data-parallel rank of the product path, run as a fresh child process (tests/test_gpu_dp.py starts two of them before the
parent touches the GPU; bench.py --gpus N does the same with its own children). Every rank sits on cuda:0 and talks gloo, so the
multi-process branch of exorl_amd.agents (_run_update's phase split + torch.distributed.all_reduce, _metrics' all-reduce,
make_replay_loader(..., worker_ids=[rank])) executes for real on a one-GPU box."""
import json
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))

O, A, H, B_GLOBAL, STEPS = 24, 6, 128, 128, 3


def write_dataset(path):
    import _synth
    from exorl_amd.replay_buffer import ReplayBufferStorage
    st = ReplayBufferStorage((), (), Path(path))
    for ep in _synth.synth_episodes(4, [50, 60, 70, 40, 55, 65], O, A):
        st._store_episode(ep)
    return st


def build_agent(kind, precision, batch, use_tb):
    import _synth
    from exorl_amd import agents
    from oracle.agents import param_shapes
    if kind == 'td3_bc':
        ag = agents.TD3BCAgent('td3_bc', (O,), (A,), 'cuda:0', 1e-4, H, 0.01, '0.2', 1, batch, 0.3, use_tb, 2.5, precision=precision)
    elif kind == 'td3':
        ag = agents.TD3Agent('td3', (O,), (A,), 'cuda:0', 1e-4, H, 0.01, '0.2', 1, batch, 0.3, use_tb, precision=precision)
    elif kind == 'crr':
        ag = agents.CRRAgent('crr', (O,), (A,), 'cuda:0', 1e-4, H, 0.01, 4, 'exp', '0.2', 1, batch, 0.3, use_tb, precision=precision)
    elif kind in ('cql', 'cqll'):           # cqll: use_critic_lagrange=True (the penalty weight is learnt, cql.py:199-213)
        ag = agents.CQLAgent('cql', (O,), (A,), 'cuda:0', 1e-4, H, 0.01, 1, batch, use_tb, 0.01, 3, 5.0, kind == 'cqll', precision=precision)
    else:
        ag = agents.BCAgent('bc', (O,), (A,), 'cuda:0', 1e-4, H, batch, '0.2', use_tb, precision=precision)
    ash, csh = param_shapes('cql' if kind == 'cqll' else kind, O, A, H)
    ag.actor.load_state_dict({k: torch.from_numpy(v) for k, v in _synth.synth_params(ash, 1).items()})
    if csh:
        ag.critic.load_state_dict({k: torch.from_numpy(v) for k, v in _synth.synth_params(csh, 2).items()})
        ag.critic_target.load_state_dict(ag.critic.state_dict())
    return ag


def sliced_noise_hook(ns, rows, local_rows):
    """noise_hook over the GLOBAL batch: every draw is made for all B_GLOBAL rows (so that every process consumes the stream alike) and the
    caller keeps its row slice; CQL's (n, B, A) blocks slice their middle axis and its uniform(-1, 1) actions are a squashed normal draw;
    CRR's (B n, A) block of repeated samples is laid out b * n + sample (crr.py:125, einops 'b x -> (b n) x'): n rows per batch row."""
    def hook(shape, dist='normal'):
        if len(shape) == 2:
            k = shape[0] // local_rows
            start, stop, _ = rows.indices(B_GLOBAL)
            out = ns.draw((k * B_GLOBAL, shape[1]))[k * start:k * stop]
        else:
            out = ns.draw((shape[0], B_GLOBAL, shape[2]))[:, rows]
        return np.ascontiguousarray(np.tanh(out) if dist == 'uniform' else out)
    return hook


HOOKED = [('td3', 'fp32'), ('cql', 'fp32'), ('cqll', 'fp32'), ('crr', 'fp32')]          # the kinds run through sliced_noise_hook (CQL: its own _run_update branch under torch.distributed)


# ---- reward-free agents (sharded actor / critic step, module step on the all-gathered batch: agents._IntrAgent._intr_step_dp) ----------
UO, UA, UH, UB_GLOBAL, USTEPS = 12, 4, 64, 64, 3
UNSUP = ['rnd', 'icm', 'icm_apt', 'disagreement', 'diayn', 'aps', 'smm', 'proto']
U_META = {'diayn': 6, 'aps': 5, 'smm': 4}


def build_unsup(kind, batch):
    """Same seed -> same initial weights in every process (the reference's RNG consumption does not depend on the batch size)."""
    import test_gpu_intr as T
    torch.manual_seed(33)
    if kind == 'proto':
        return T.make_proto(UO, UA, UH, batch, 16, 32, 16, 256)
    if kind == 'smm':
        return T.make_smm(UO, UA, UH, batch, U_META['smm'])
    return T.make(kind, UO, UA, UH, batch, U_META.get(kind, 16))


def unsup_views(ag):
    views = [('actor', ag.actor), ('critic', ag.critic), ('critic_target', ag.critic_target)]
    for nm in ('rnd', 'icm', 'disagreement', 'diayn', 'aps', 'smm', 'predictor', 'predictor_target', 'projector', 'protos'):
        if hasattr(ag, nm):
            views.append((nm, getattr(ag, nm)))
    return views


def unsup_batch(kind, step, rows=slice(None)):
    """Global batch of one update (+ the meta rows of DIAYN / APS / SMM); a rank takes its row slice."""
    import _synth
    b = list(_synth.synth_batch(41, step, UB_GLOBAL, UO, UA))
    if kind in U_META:
        rs = np.random.RandomState(500 + step)
        Z = U_META[kind]
        if kind == 'aps':
            m = rs.standard_normal((UB_GLOBAL, Z)).astype(np.float32)
            m /= np.linalg.norm(m, axis=1, keepdims=True)
        else:
            m = np.eye(Z, dtype=np.float32)[rs.randint(0, Z, UB_GLOBAL)]
        b.append(m)
    return tuple(np.ascontiguousarray(x[rows]) for x in b)


def unsup_hooks(ag, kind, seed):
    """Module-side draws (SMM's VAE epsilon, Proto's Categorical uniforms) are over the module's — global — batch: every rank makes the
    same ones. The device generators of the library are seeded rank-independently and would do; explicit streams keep the parent's
    single-process run on the same numbers."""
    import _synth
    ms = _synth.NoiseStream(seed)
    if kind == 'smm':
        ag.eps_hook = lambda shape: ms.draw(shape)
    if kind == 'proto':
        rs = np.random.RandomState(seed)
        ag.cat_hook = lambda n: rs.uniform(0, 1, n).astype(np.float32)


def run_unsup(rank, world, out_dir, result):
    import _synth
    Br = UB_GLOBAL // world
    sl = slice(rank * Br, (rank + 1) * Br)
    for kind in UNSUP:
        ag = build_unsup(kind, Br)
        assert ag.world_size == world and ag.intr.batch == UB_GLOBAL
        unsup_hooks(ag, kind, 17)
        ns = _synth.NoiseStream(9)
        draws = []
        ag.noise_hook = lambda shape: draws.pop(0)
        metrics, rewards = [], []
        for i in range(USTEPS):
            draws[:] = [ns.draw((UB_GLOBAL, UA))[sl], ns.draw((UB_GLOBAL, UA))[sl]]
            m = ag.update(iter([unsup_batch(kind, i, sl)]), 2 * i)
            metrics.append({k: float(v) for k, v in m.items()})
            rewards.append(ag.engine._view(ag.engine.batch_slots().reward, Br).cpu().numpy().copy())
        torch.cuda.synchronize()
        if kind == 'icm_apt':                  # a snapshot taken inside a data-parallel run (pretrain.py:293-300 pickles the agent): the gathered-batch
            import pickle                      # buffers must not ride along, the clone must carry the module state
            clone = pickle.loads(pickle.dumps(ag))
            assert clone.intr.rms_state() == ag.intr.rms_state() and clone.intr.batch == ag.intr.batch
            for pa, pb in zip(clone.icm.parameters(), ag.icm.parameters()):
                assert torch.equal(pa, pb)
            del clone
        np.savez(out_dir / f'unsup_{kind}_rank{rank}.npz', reward=np.stack(rewards),
                 **{n: torch.cat([p.reshape(-1) for p in v.parameters()]).cpu().numpy() for n, v in unsup_views(ag)})
        result[f'unsup_{kind}'] = metrics
        del ag


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    data_dir, out_dir = Path(sys.argv[1]), Path(sys.argv[2])
    dist.init_process_group('gloo', init_method=f"tcp://127.0.0.1:{os.environ['MASTER_PORT']}", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import _synth
    from exorl_amd.replay_buffer import ReplayBufferStorage, make_replay_loader
    result = {}
    Br = B_GLOBAL // world
    for kind, precision in (('td3_bc', 'fp32'), ('td3_bc', 'bf16x3'), ('bc', 'fp32')):
        ag = build_agent(kind, precision, Br, True)
        assert ag.world_size == world
        st = ReplayBufferStorage((), (), data_dir)               # the directory the parent wrote
        it = iter(make_replay_loader(st, 10**6, Br, world, True, 1, 0.99, worker_ids=[rank], seed=77))
        ns = _synth.NoiseStream(9)
        sl = slice(rank * Br, (rank + 1) * Br)
        draws = []
        ag.noise_hook = lambda shape: draws.pop(0)
        metrics = []
        for step in range(STEPS):
            draws[:] = [ns.draw((B_GLOBAL, A))[sl], ns.draw((B_GLOBAL, A))[sl]]
            m = ag.update(it, step)
            metrics.append({k: float(v) for k, v in m.items()})
        torch.cuda.synchronize()
        tag = f'{kind}_{precision}'
        nets = [('actor', ag.actor)] + ([('critic', ag.critic), ('critic_target', ag.critic_target)] if hasattr(ag, 'critic') else [])
        np.savez(out_dir / f'{tag}_rank{rank}.npz', **{n: torch.cat([p.reshape(-1) for p in net.parameters()]).cpu().numpy() for n, net in nets})
        result[tag] = metrics
        del ag, it
    for kind, precision in HOOKED:
        ag = build_agent(kind, precision, Br, True)
        assert ag.world_size == world
        st = ReplayBufferStorage((), (), data_dir)
        it = iter(make_replay_loader(st, 10**6, Br, world, True, 1, 0.99, worker_ids=[rank], seed=78))
        ag.noise_hook = sliced_noise_hook(_synth.NoiseStream(11), slice(rank * Br, (rank + 1) * Br), Br)
        metrics = [{k: float(v) for k, v in ag.update(it, step).items()} for step in range(STEPS)]
        torch.cuda.synchronize()
        tag = f'{kind}_{precision}'
        nets = [('actor', ag.actor), ('critic', ag.critic), ('critic_target', ag.critic_target)]
        extra = {'cql_scalars': np.asarray(ag.engine.cql_alpha_state(), np.float64)} if kind in ('cql', 'cqll') else {}      # temperature (+ multiplier) and their Adam moments
        np.savez(out_dir / f'{tag}_rank{rank}.npz', **{n: torch.cat([p.reshape(-1) for p in net.parameters()]).cpu().numpy() for n, net in nets}, **extra)
        result[tag] = metrics
        del ag, it
    run_unsup(rank, world, out_dir, result)
    # no noise hook: the device Philox stream of each rank (the ranks of one global batch must not repeat each other's noise rows)
    ag = build_agent('td3_bc', 'fp32', Br, False)
    seed = int(ag.engine.cfg.seed)
    np.savez(out_dir / f'noise_rank{rank}.npz', seed=np.uint64(seed), block=ag.engine.philox_normal(seed, 2, (Br, A)).cpu().numpy())
    del ag
    json.dump(result, open(out_dir / f'metrics_rank{rank}.json', 'w'))
    dist.barrier()
    dist.destroy_process_group()
    print(f'rank {rank} done')


if __name__ == '__main__':
    main()
