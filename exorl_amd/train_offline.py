"""The gradient-step loop of the reference's offline entry script (train_offline.py:61-123) as a function, for callers
that bring their own environment / logger: load a directory of `.npz` episodes into the HBM replay, then drive
agent.update(replay_iter, step) — through the captured hipGraph when the agent and sampler allow it.

The reference script cannot run as shipped (it calls make_replay_loader with the offline argument shape);
exorl_amd.replay_buffer.make_replay_loader accepts both shapes, so the loop below is the script's loop, not a patch of it.
Environment construction, evaluation roll-outs, video and the CSV/TensorBoard logger stay the caller's (eval_fn / log_fn).
"""
import time

from .replay_buffer import make_replay_loader


def train_offline(agent, replay_dir, num_grad_steps, batch_size, discount, replay_buffer_size=10**7, eval_every_steps=10000,
                  log_every_steps=1000, eval_fn=None, log_fn=None, env=None, sampler='philox', use_graph=True, start_step=0):
    """train_offline.py:90-123. Returns the list of (step, metrics) rows that were logged.

    eval_fn(step, agent): called every eval_every_steps (train_offline.py:108-112).
    log_fn(step, metrics): called with the agent's metrics (only non-empty when the agent was built with use_tb=True)
    plus fps / total_time every log_every_steps (train_offline.py:114-121)."""
    loader = make_replay_loader(env, replay_dir, replay_buffer_size, batch_size, 0, discount, sampler=sampler)
    replay_iter = iter(loader)
    if use_graph and hasattr(agent, 'enable_graph'):
        agent.enable_graph(replay_iter, start_step)      # False (and eager launches) when the pairing cannot be captured
    import torch
    rows = []
    t_start = t_last = time.time()
    for global_step in range(start_step, start_step + num_grad_steps):
        if eval_fn is not None and eval_every_steps and global_step % eval_every_steps == 0:
            eval_fn(global_step, agent)
        metrics = agent.update(replay_iter, global_step)
        if log_every_steps and global_step % log_every_steps == 0:
            torch.cuda.synchronize()
            now = time.time()
            row = dict(metrics, fps=log_every_steps / max(now - t_last, 1e-9), total_time=now - t_start, step=global_step)
            t_last = now
            rows.append((global_step, row))
            if log_fn is not None:
                log_fn(global_step, row)
    return rows
