"""Snapshot interchange with the reference.

The reference saves snapshots by pickling the whole agent object (pretrain.py:293-300) and loads them with torch.load
(finetune.py:222-252); such a file names the reference's classes, so it can only be opened where those classes import, and opening
it executes whatever the pickle says. What crosses between the two backends is therefore the part that is data: every network's
state_dict under the reference's own parameter names (td3_bc.py:12-56 `policy.N.*` / `q1_net.N.*`, ddpg.py:42-123 `trunk.*` /
`policy.*` / `Q1.*`, the intrinsic modules' names), as a plain dict of tensors that `torch.load(..., weights_only=True)` accepts.

    exorl_amd side                                        reference side (a maintainer's three lines)
    save_state_dicts(agent, 'snap.pt')                    d = torch.load('snap.pt', weights_only=True)
                                                          agent.actor.load_state_dict(d['actor']); agent.critic.load_state_dict(d['critic'])
    load_state_dicts(agent, 'ref.pt')                     torch.save({'actor': agent.actor.state_dict(), 'critic': agent.critic.state_dict(),
                                                                      'critic_target': agent.critic_target.state_dict()}, 'ref.pt')
Whole-agent pickles of THIS backend (its own format) are _AgentBase.__getstate__/__setstate__ in agents.py.
"""
from collections import OrderedDict

import torch

NETS = ('encoder', 'actor', 'critic', 'critic_target', 'rnd', 'icm', 'disagreement', 'diayn', 'aps', 'smm', 'predictor',
        'predictor_target', 'projector', 'protos', 'encoder_target')


def state_dicts(agent):
    """{net name: state_dict on the CPU} for every network the agent has, names and keys as in the reference."""
    out = OrderedDict()
    for name in NETS:
        net = getattr(agent, name, None)
        if net is not None and hasattr(net, 'state_dict'):
            out[name] = OrderedDict((k, v.detach().cpu().clone()) for k, v in net.state_dict().items())
    return out


def save_state_dicts(agent, path):
    torch.save(dict(state_dicts(agent)), path)


def load_state_dicts(agent, path_or_dict, strict=True):
    """Loads what save_state_dicts (or the reference-side snippet above) wrote. Only tensors are read: weights_only=True."""
    d = path_or_dict if isinstance(path_or_dict, dict) else torch.load(path_or_dict, map_location='cpu', weights_only=True)
    loaded = []
    for name, sd in d.items():
        net = getattr(agent, name, None)
        if net is None:
            if strict:
                raise KeyError(f'snapshot has a network {name!r} this agent ({type(agent).__name__}) does not')
            continue
        net.load_state_dict(sd)
        loaded.append(name)
    if 'critic' in d and 'critic_target' not in d and hasattr(agent, 'critic_target'):
        agent.critic_target.load_state_dict(agent.critic.state_dict())       # td3_bc.py:93
    return loaded
