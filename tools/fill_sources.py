#!/usr/bin/env python
"""Count python-initiated zero fills of one training step by call site (monkeypatched torch.zeros*)."""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sp_amd import ops, parallel
from neural_sp_amd.configs import conformer_rnnt_args, synthetic_batch
from neural_sp_amd.speech2text import Speech2Text
ops.set_compute_mode('bf16')
margs = conformer_rnnt_args('L', n_layers=12, vocab=1000, dropout=0.1, ctc_weight=0.3)
model = Speech2Text(margs).cuda(0)
params = list(model.parameters())
opt = torch.optim.Adam(params, lr=1e-4, fused=True)
batch = synthetic_batch(B=16, t_range=(1200, 1600), u_range=(120, 200), vocab=1000, seed=0)
def step():
    loss, _ = model(batch, task='all'); loss.backward()
    parallel.clip_grad_norm_(params, 5.0); opt.step(); opt.zero_grad(set_to_none=True)
for _ in range(3): step()
torch.cuda.synchronize()
cnt = collections.Counter()
def wrap(mod, name):
    orig = getattr(mod, name)
    def f(*a, **k):
        fr = [x for x in traceback.extract_stack()[:-1] if 'neural_sp_amd' in x.filename or 'tools' in x.filename]
        cnt[(name, '%s:%d' % (os.path.basename(fr[-1].filename), fr[-1].lineno) if fr else '?')] += 1
        return orig(*a, **k)
    setattr(mod, name, f)
for n in ('zeros', 'zeros_like', 'ones', 'full'):
    wrap(torch, n)
for n in ('new_zeros', 'zero_', 'fill_'):
    wrap(torch.Tensor, n)
step()
torch.cuda.synchronize()
print('python-initiated fills:', sum(cnt.values()))
for (n, s), k in cnt.most_common(40):
    print('%4d %-12s %s' % (k, n, s))
