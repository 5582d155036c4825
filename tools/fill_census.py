#!/usr/bin/env python
"""Which python lines of this package issue the fill / zero / copy launches of one training step (TorchDispatchMode
over one step of the bench model; the C++ autograd engine's own zero-fills show up as 'autograd / torch')."""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.utils._python_dispatch as pd
from neural_sp_amd import ops, parallel
from neural_sp_amd.configs import conformer_rnnt_args, synthetic_batch
from neural_sp_amd.speech2text import Speech2Text

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device('cuda:0')
ops.set_compute_mode('bf16')
torch.manual_seed(1)
model = Speech2Text(conformer_rnnt_args('L', n_layers=12, vocab=1000, dropout=0.1, ctc_weight=0.3)).to(dev)
params = list(model.parameters())
opt = torch.optim.Adam(params, lr=1e-4, betas=(0.9, 0.98), eps=1e-9, fused=True)
batch = synthetic_batch(B=B, t_range=(1200, 1600), u_range=(120, 200), vocab=1000, seed=0)
sites = collections.Counter()


class Mode(pd.TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        n = str(func)
        if any(k in n for k in ('fill_', 'zero_', 'zeros', 'copy_', 'aten.full', 'aten.clone', 'aten.cat', 'aten.add', 'aten.mul')):
            fr = [f for f in traceback.extract_stack(limit=24) if '/neural_sp_amd/' in f.filename]
            key = (n, '%s:%d' % (os.path.basename(fr[-1].filename), fr[-1].lineno) if fr else 'autograd / torch')
            sites[key] += 1
        return func(*args, **(kwargs or {}))


def step():
    loss, obs = model(batch, task='all')
    loss.backward()
    parallel.clip_grad_norm_(params, 5.0)
    opt.step()
    opt.zero_grad(set_to_none=True)


for _ in range(3):
    step()
with Mode():
    step()
torch.cuda.synchronize()
for k, v in sites.most_common(45):
    print('%4d  %-34s %s' % (v, k[0], k[1]))
print('total', sum(sites.values()))
