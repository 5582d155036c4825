"""Predict, on CPU, how far the bf16 throughput mode can move a golden fixture's loss and gradients.

The oracle (oracle/model_ref.py, fp64) is run with every Linear / pointwise-conv operand rounded to bf16
(fp64 accumulation), which is what the MFMA GEMMs of the bf16 mode do to their operands; attention products
and the conv front-end are left exact, so this is a LOWER bound of the deviation the GPU test will see.
Used to set the stated gates of tests/test_golden_gpu.py::test_golden_bf16 for fixtures added without GPU
time (round 2): every variant fixture sits at 2e-6 .. 3e-4 except conformer_concat_ctc_xs (1.3e-3 for any
seed / batch size: the un-normalised ReLU(Linear(3d -> d)) of ConcatSubsampler with torch's default init).

    python tools/bf16_sensitivity.py [fixture names ...]
"""
import argparse
import glob
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import model_ref  # noqa: E402

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def _r16(t):
    return t.to(torch.bfloat16).to(t.dtype)


def main():
    names = sys.argv[1:] or sorted(os.path.basename(p)[:-3] for p in glob.glob(os.path.join(GOLDEN, '*_xs.pt')))
    lin, c1 = F.linear, F.conv1d

    def lin16(x, w, b=None):
        return lin(_r16(x), _r16(w), b)

    def conv1d16(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
        if groups == 1:
            return c1(_r16(x), _r16(w), b, stride, padding, dilation, groups)
        return c1(x, w, b, stride, padding, dilation, groups)

    for name in names:
        fix = torch.load(os.path.join(GOLDEN, name + '.pt'), weights_only=False)
        args = argparse.Namespace(**fix['args'])
        sd = {k: v.clone().double().requires_grad_(v.is_floating_point() and 'inv_freq' not in k and k != 'enc.pos_enc.pe')
              if v.is_floating_point() else v for k, v in fix['state_dict'].items()}
        qw = args.mocha_quantity_loss_weight if fix['meta'].get('trigger_quantity_loss') else 0.0
        ss_seed = fix['meta'].get('scheduled_sampling_seed')
        kw = dict(quantity_weight=qw, scheduled_sampling=ss_seed is not None,
                  stableemit=bool(fix['meta'].get('trigger_stableemit')), ctc_trigger_points=fix.get('ctc_trigger_points'),
                  latency_weight=getattr(args, 'mocha_latency_loss_weight', 0.0) if fix['meta'].get('trigger_quantity_loss') else 0.0)

        def seed():
            if ss_seed is not None:
                import random
                random.seed(ss_seed)
        F.linear, F.conv1d = lin16, conv1d16
        try:
            seed()
            loss = model_ref.speech2text_loss(sd, args, fix['batch'], torch.float64, **kw)[0]
        finally:
            F.linear, F.conv1d = lin, c1
        ref = fix['loss'].item()
        gn = list(fix['grads'])
        grads = torch.autograd.grad(loss, [sd[n] for n in gn], allow_unused=True)
        m = sorted(g.abs().max().item() for g in fix['grads'].values())
        gmax = m[int(0.9 * (len(m) - 1))]
        cos = {}
        for n, g in zip(gn, grads):
            r = fix['grads'][n]
            if g is None or r.numel() < 16 or r.abs().max() < 1e-5 * gmax:
                continue
            cos[n] = F.cosine_similarity(g.float().flatten(), r.flatten(), dim=0).item()
        worst = min(cos.items(), key=lambda kv: kv[1])
        # how far the REFERENCE's own fp32 gradients (the fixture) are from the fp64 oracle: the noise floor of any
        # fp32-vs-fixture comparison (2.2e-3 of max on conformer_concat_ctc_xs, 7.5e-4 on conformer_conv1d_ctc_xs,
        # <= 1.2e-5 elsewhere) -> the per-fixture gates FP32_GRAD_GATE of tests/test_golden_gpu.py
        sd64 = {k: v.clone().double().requires_grad_(v.is_floating_point() and 'inv_freq' not in k and k != 'enc.pos_enc.pe')
                if v.is_floating_point() else v for k, v in fix['state_dict'].items()}
        seed()
        l64 = model_ref.speech2text_loss(sd64, args, fix['batch'], torch.float64, **kw)[0]
        g64 = torch.autograd.grad(l64, [sd64[n] for n in gn], allow_unused=True)
        noise = max(((fix['grads'][n].double() - g).abs().max() / max(g.abs().max().item(), 1e-5 * gmax)).item()
                    for n, g in zip(gn, g64) if g is not None and not (
                        fix['args'].get('conformer_normalization') == 'batch_norm' and n.endswith('.conv.depthwise_conv.bias')))   # true gradient zero
        print('%-30s bf16 operands: loss rel %.2e, min cosine %.5f (%s) | reference fp32 vs fp64 gradients: %.2e of max'
              % (name, abs(loss.item() - ref) / abs(ref), worst[1], worst[0], noise))


if __name__ == '__main__':
    main()
