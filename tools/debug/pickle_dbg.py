import pickle, sys
import numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import _synth
from exorl_amd import agents, _lib as L
O, A, H, B = 17, 6, 128, 64
ag = agents.TD3BCAgent('td3_bc', (O,), (A,), 'cuda', 1e-4, H, 0.01, 0.2, 1, B, 0.3, True, 2.5)
ns = _synth.NoiseStream(1); ag.noise_hook = ns.draw
for i in range(3):
    ag.update(iter([_synth.synth_batch(23, i, B, O, A)]), i)
ag3 = pickle.loads(pickle.dumps(ag))
def cmp(tag):
    for net in (0, 1, 2):
        for w in ((0,) if net == 2 else (0, 1, 2, 3)):
            a, b = ag.engine.flat(net, w), ag3.engine.flat(net, w)
            print(tag, net, w, bool(torch.equal(a, b)), float((a - b).abs().max()))
    print(tag, 'steps', ag.engine.opt_steps(), ag3.engine.opt_steps())
cmp('after load')
for a in (ag, ag3):
    a.noise_hook = _synth.NoiseStream(2).draw
    m = a.update(iter([_synth.synth_batch(23, 3, B, O, A)]), 3)
    print(m)
cmp('after 1 step')
