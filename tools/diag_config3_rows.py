#!/usr/bin/env python
"""Where is the fp32-mode gradient error of one tensor of the config-3 model? (rows = hidden units)"""
import os
import sys
import importlib.util
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
spec = importlib.util.spec_from_file_location('fs', os.path.join(os.path.dirname(__file__), '..', 'tests', 'test_fullsize_parity_gpu.py'))
fs = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fs)
model, margs, batch = fs._config3()
ref, robs, rg = fs._oracle(model, margs, batch)
loss, obs, g = fs._hip(model, batch, sys.argv[1] if len(sys.argv) > 1 else 'f32')
for name in sys.argv[2:] or ['dec_fwd.layers.2.feed_forward.w_1.weight', 'dec_fwd.layers.2.feed_forward.w_1.bias']:
    a, r = g[name].double(), rg[name].double()
    d = (a - r).abs()
    if d.dim() == 1:
        d = d[:, None]
    rows = d.max(dim=1).values
    top = torch.topk(rows, 6)
    print(name, 'max|g| %.3e' % r.abs().max().item(), 'rows with error > 1e-3 of max: %d of %d' % ((rows > 1e-3 * r.abs().max()).sum().item(), rows.numel()))
    print('   top rows', [(int(i), '%.2e' % (v / r.abs().max()).item()) for v, i in zip(top.values, top.indices)])
