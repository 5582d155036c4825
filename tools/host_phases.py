#!/usr/bin/env python
"""Host-side time of the phases of a training step in a free-running loop (no syncs inside):
is the step host-bound or GPU-bound?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sp_amd import ops, parallel, _lib as nsp_lib


class NoLaunch(object):
    """HOST_ONLY=1 python tools/host_phases.py: every kernel entry point of libnsp_hip.so becomes a no-op, so
    that what is left is the host cost of the step (allocations, autograd, argument marshalling).  Results are
    garbage by construction; the wrapper lives here, in the tool, and nowhere in the package."""

    def __init__(self, real):
        self._real = real

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if name.startswith('nsp_') and name != 'nsp_version':
            return lambda *a: 0
        return fn


if os.environ.get('HOST_ONLY') == '1':
    nsp_lib._lib = NoLaunch(nsp_lib.lib())
from neural_sp_amd.configs import conformer_rnnt_args, synthetic_batch
from neural_sp_amd.speech2text import Speech2Text
ops.set_compute_mode('bf16')
margs = conformer_rnnt_args('L', n_layers=12, vocab=1000, dropout=0.1, ctc_weight=0.3)
model = Speech2Text(margs).cuda(0)
params = list(model.parameters())
opt = torch.optim.Adam(params, lr=1e-4, fused=True)
batches = [synthetic_batch(B=16, t_range=(1200, 1600), u_range=(120, 200), vocab=1000, seed=i) for i in range(4)]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
acc = [0.0] * 4
def step(i, rec):
    t0 = time.perf_counter()
    loss, obs = model(batches[i % 4], task='all')
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    parallel.clip_grad_norm_(params, 5.0); opt.step(); opt.zero_grad(set_to_none=True)
    t3 = time.perf_counter()
    if rec:
        acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2
for i in range(4): step(i, False)
torch.cuda.synchronize()
T0 = time.perf_counter()
for i in range(N): step(i, True)
T1 = time.perf_counter()
torch.cuda.synchronize()
T2 = time.perf_counter()
print('host per step: fwd %.2f  bwd %.2f  clip+adam %.2f  = %.2f ms | loop wall %.2f ms/step, + final drain %.2f ms total'
      % (acc[0] / N * 1e3, acc[1] / N * 1e3, acc[2] / N * 1e3, sum(acc[:3]) / N * 1e3, (T1 - T0) / N * 1e3, (T2 - T1) * 1e3))
