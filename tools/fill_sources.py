#!/usr/bin/env python
"""Which ops issue the small fill kernels of a training step?  (torch.profiler event tree:
every aten::fill_/zero_ is attributed to its chain of parent ops, across threads.)"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from neural_sp_amd import ops, parallel
from neural_sp_amd.configs import conformer_rnnt_args, synthetic_batch
from neural_sp_amd.speech2text import Speech2Text
ops.set_compute_mode('bf16')
margs = conformer_rnnt_args('L', n_layers=12, vocab=1000, dropout=0.1, ctc_weight=0.3)
model = Speech2Text(margs).cuda(0)
params = list(model.parameters())
opt = torch.optim.Adam(params, lr=1e-4, fused=True)
batch = synthetic_batch(B=int(os.environ.get('NSP_B', '16')), t_range=(1200, 1600), u_range=(120, 200), vocab=1000, seed=0)
def step():
    loss, _ = model(batch, task='all'); loss.backward()
    parallel.clip_grad_norm_(params, 5.0); opt.step(); opt.zero_grad(set_to_none=True)
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU]) as prof:
    step()
torch.cuda.synchronize()
cnt = collections.Counter()
WANT = tuple(sys.argv[1:]) or ('aten::fill_', 'aten::zero_')
for ev in prof.events():
    if ev.name in WANT:
        chain, p = [], ev.cpu_parent
        while p is not None and len(chain) < 4:
            chain.append(p.name); p = p.cpu_parent
        cnt[(ev.name, ' <- '.join(chain))] += 1
for (n, s), k in cnt.most_common(25):
    print('%4d %-12s %s' % (k, n, s[:150]))
