#!/usr/bin/env python
"""cProfile of the host side of the FORWARD (main thread) and of the BACKWARD (autograd thread,
profiled through threading.setprofile) of one training step."""
import cProfile, io, os, pstats, sys, threading
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
which = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
sort = sys.argv[2] if len(sys.argv) > 2 else 'tottime'
def step(pr=None):
    if pr and which == 'fwd': pr.enable()
    loss, _ = model(batch, task='all')
    if pr and which == 'fwd': pr.disable()
    loss.backward()
    parallel.clip_grad_norm_(params, 5.0); opt.step(); opt.zero_grad(set_to_none=True)
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
if which == 'bwd':
    # the autograd engine's device thread runs the python backward functions
    threading.setprofile(lambda *a: None)
    import torch.autograd
    orig = torch.autograd.Function.backward
    prs = {}
    def hook_thread():
        pass
    # simplest: profile in the engine thread by wrapping every Function.backward call
    import functools
    from neural_sp_amd import ops as O
    for name in dir(O):
        obj = getattr(O, name)
        if isinstance(obj, type) and issubclass(obj, torch.autograd.Function) and obj is not torch.autograd.Function:
            b = obj.backward
            def mk(b):
                @functools.wraps(b)
                def w(ctx, *g):
                    pr.enable()
                    try:
                        return b(ctx, *g)
                    finally:
                        pr.disable()
                return staticmethod(w)
            obj.backward = mk(b)
for _ in range(3): step(pr)
torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(sort).print_stats(45); print(s.getvalue()[:9000])
