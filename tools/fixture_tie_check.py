"""Does a golden fixture contain a max-pool window whose arg-max is decided by fp32 rounding?

The conv front-end ends in MaxPool2d (conv.py:347-396); the gradient of a pooled value goes to ONE input
position.  When the two largest entries of a window differ by less than fp32 rounding (a few 1e-7 relative),
the reference's own fp32 run and its fp64 restatement pick different positions and the front-end weight
gradients move by ~1e-3 of their max -- a discrete choice, not an accuracy property of either side.  (Found when
the first conformer_concat / conformer_conv1d fixtures showed 2.2e-3 / 7.5e-4 against the fp64 oracle while
every other fixture sits below 1.2e-5: one such window each, among 368,640.)  oracle/gen_golden.py seeds are
chosen so that no window has a relative gap in (0, 1e-5); exact ties (gap 0: the constant padded region, windows
that are all zero after the ReLU) are resolved identically by everyone (first position in scan order).

    python tools/fixture_tie_check.py [fixture names ...]
"""
import glob
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def near_ties(state_dict, xs_list, poolings, lo=0.0, hi=1e-5, relu_abs=2e-6):
    """(pool windows whose two largest entries have a relative gap in (lo, hi), smallest non-zero gap,
    front-end ReLU pre-activations with 0 < |x| < relu_abs, smallest non-zero |x|): the two kinds of discrete
    choice (which position gets a pooled gradient; whether a unit passes gradient at all) that fp32 rounding decides"""
    sd = {k: v.double() for k, v in state_dict.items() if k.startswith('enc.conv.layers')}
    xl = [len(x) for x in xs_list]
    xs = torch.zeros(len(xl), max(xl), xs_list[0].shape[1], dtype=torch.float64)
    for b, x in enumerate(xs_list):
        xs[b, :len(x)] = torch.as_tensor(x, dtype=torch.float64)
    ci = sd['enc.conv.layers.0.conv1.weight'].shape[1]      # conv_in_channel (conv.py:167-175)
    x = xs.view(xs.shape[0], xs.shape[1], ci, xs.shape[2] // ci).transpose(2, 1)
    count, smallest = 0, float('inf')
    n_relu, smallest_pre = 0, float('inf')
    for i, pool in enumerate(poolings):
        p = 'enc.conv.layers.%d' % i
        for cv in ('.conv1', '.conv2'):
            pre = F.conv2d(x, sd[p + cv + '.weight'], sd[p + cv + '.bias'], padding=1)
            a = pre.abs()
            n_relu += int(((a > 0) & (a < relu_abs)).sum())
            smallest_pre = min(smallest_pre, float(a[a > 0].min()))
            x = torch.relu(pre)
        if pool[0] * pool[1] > 1:
            B, C, T, Fq = x.shape
            xp = F.pad(x, (0, (-Fq) % pool[1], 0, (-T) % pool[0]), value=float('-inf'))
            w = xp.unfold(2, pool[0], pool[0]).unfold(3, pool[1], pool[1]).reshape(B, C, -1, pool[0] * pool[1])
            top = w.topk(2, dim=-1).values
            gap = (top[..., 0] - top[..., 1]) / top[..., 0].abs().clamp(min=1e-30)
            sel = (gap > lo) & (gap < hi) & torch.isfinite(top[..., 1])
            count += int(sel.sum())
            if (gap > 0).any():
                smallest = min(smallest, float(gap[gap > 0].min()))
            x = F.max_pool2d(x, tuple(pool), tuple(pool), ceil_mode=True)
    return count, smallest, n_relu, smallest_pre


def main():
    names = sys.argv[1:] or sorted(os.path.basename(p)[:-3] for p in glob.glob(os.path.join(GOLDEN, '*_xs.pt')))
    for name in names:
        fix = torch.load(os.path.join(GOLDEN, name + '.pt'), weights_only=False)
        pools = [[int(v) for v in p.strip('()').split(',')] for p in fix['args']['conv_poolings'].split('_')]
        n, smallest, nr, spre = near_ties(fix['state_dict'], fix['batch']['xs'], pools)
        print('%-30s pool windows with relative top-2 gap in (0, 1e-5): %d (smallest %.2e) | ReLU pre-activations with '
              '0 < |x| < 2e-6: %d (smallest %.2e)' % (name, n, smallest, nr, spre))


if __name__ == '__main__':
    main()
