"""Round-5 diagnosis, pure torch (no kernel of this repository in the probing process): kernel 1 overwrites a recycled
block, kernel 2 (same stream) reads it back and compares.  Alone on the device this never fails; does it beside a second
process?  (tools/conv_first_kernel_stress.py saw the repository's first-layer kernel pair fail ~1 in 2500 that way, by
exactly one 128-byte line, and never alone.)

  python tools/conv_first_kernel_stress.py --load-seconds 60 &
  python tools/stale_line_probe.py --trials 30000 [--gap-ms 0.2]
"""
import argparse
import time

import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument('--trials', type=int, default=20000)
ap.add_argument('--gap-ms', type=float, default=0.0)
ap.add_argument('--n', type=int, default=921600)      # elements of the block under test (the conv output of the failing test: 2 x 180 x 80 x 32)
a = ap.parse_args()
dev = torch.device('cuda', 0)
rng = np.random.RandomState(1)
src = torch.randn(a.n, device=dev).to(torch.bfloat16)
bad_trials = torch.zeros((), device=dev, dtype=torch.int64)
bad_elems = torch.zeros((), device=dev, dtype=torch.int64)
junk = []
torch.cuda.synchronize()
t0 = time.time()
for i in range(a.trials):
    if a.gap_ms > 0 and i % 4 == 0:
        torch.cuda.synchronize()
        time.sleep(a.gap_ms * 1e-3)
    junk.append(torch.full((int(rng.randint(1000, 400000)),), float(i), device=dev))
    if len(junk) > 3:
        junk.pop(int(rng.randint(0, len(junk))))
    y = torch.empty(a.n, device=dev, dtype=torch.bfloat16)      # recycled block, old contents = something else
    torch.mul(src, 1.0, out=y)                                    # kernel 1: overwrite all of it
    d = (y != src)                                                # kernel 2: read it back
    bad_trials += d.any().long()
    bad_elems += d.sum()
    del y
torch.cuda.synchronize()
print('[pure-torch probe] %d trials in %.1f s (gap %.1f ms): read-back differs %d times (%d elements = %d bytes in all)'
      % (a.trials, time.time() - t0, a.gap_ms, bad_trials.item(), bad_elems.item(), 2 * bad_elems.item()), flush=True)
