"""Round-5 diagnosis of the round-4 stock-DDP failure (rank 1's loss differed between two identical iterations).
One process, no process group: the rank-1 sub-batch of tests/ddp_hip_worker.py through Speech2Text N times, no
optimizer step, dropout 0.  Every iteration must give the same loss bit for bit and the same gradients.  Prints, per
iteration, the loss, a checksum of every sub-module's output (first module whose checksum moves = where it starts)
and the worst per-tensor gradient difference against iteration 0.

  python tools/ddp_repro.py [--single-stream] [--iters 3] [--rank 1] [--poison] [--no-refresh]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--single-stream', action='store_true')
    ap.add_argument('--iters', type=int, default=3)
    ap.add_argument('--rank', type=int, default=1)
    ap.add_argument('--poison', action='store_true')
    ap.add_argument('--no-refresh', action='store_true')
    ap.add_argument('--hooks', action='store_true')
    ap.add_argument('--warm-full', action='store_true', help='rank 0 of the test: one pass over all four utterances first')
    a = ap.parse_args()
    import torch
    if a.poison:
        from tests import poison
        poison.enable()
    from neural_sp_amd import ops
    from neural_sp_amd.speech2text import Speech2Text
    from tests import ddp_hip_worker as W
    torch.cuda.set_device(0)
    args = W.model_args(small=False)
    torch.manual_seed(7)
    model = Speech2Text(args).cuda(0)
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() == 1:
                p.add_(torch.empty_like(p).uniform_(-0.1, 0.1))
    full = W.global_batch(args.vocab)
    ops.set_compute_mode('bf16')
    if a.single_stream:
        def guard(self=model):
            for name in ('dec_fwd', 'dec_fwd_sub1', 'dec_fwd_sub2'):
                dec = getattr(self, name, None)
                if dec is not None and hasattr(dec, 'ensure_streams'):
                    dec._nsp_single_stream = True
            return True
        model._ddp_guard = guard
    if a.no_refresh:
        real = ops.refresh_weight_shadows
        ops.refresh_weight_shadows = lambda force=False: real(force=False)
    if a.warm_full:
        model.zero_grad(set_to_none=True)
        loss, _ = model(full, task='all')
        loss.backward()
        torch.cuda.synchronize()
        model.zero_grad(set_to_none=True)
    local = W.sub_batch(full, list(range(a.rank, 4, 2)))
    sums = []
    cur = {}

    def hook(name):
        def f(mod, inp, out):
            outs = out if isinstance(out, (tuple, list)) else (out,)
            tot = []
            for o in outs:
                if torch.is_tensor(o) and o.is_floating_point() and o.numel() > 1:
                    tot.append(o.detach().double().sum().item())
                    o16 = getattr(o, '_nsp16', None)
                    if torch.is_tensor(o16):
                        tot.append(o16.detach().double().sum().item())
            cur[name] = tuple(tot)
        return f
    if a.hooks:
        for n, m in model.named_modules():
            if n:
                m.register_forward_hook(hook(n))
    g0 = None
    losses = []
    for it in range(a.iters):
        cur.clear()
        model.zero_grad(set_to_none=True)
        loss, obs = model(local, task='all')
        loss.backward()
        torch.cuda.synchronize()
        ops.lstm_check()
        losses.append(loss.item())
        g = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        sums.append(dict(cur))
        msg = 'it %d loss %.7f' % (it, losses[-1])
        if g0 is None:
            g0 = g
        else:
            worst, wn = 0.0, ''
            for n in g0:
                d = (g[n] - g0[n]).abs().max().item() / max(g0[n].abs().max().item(), 1e-20)
                if not d == d:
                    d = float('inf')
                if d > worst:
                    worst, wn = d, n
            msg += ' | worst grad diff vs it0 %.3e (%s)' % (worst, wn)
            if a.hooks:
                moved = [n for n in sums[0] if sums[it].get(n) != sums[0][n]]
                msg += ' | modules whose output checksum moved: %d, first %s' % (len(moved), moved[:4])
        print(msg, flush=True)
    ok = all(l == losses[0] for l in losses)
    print('RESULT %s single_stream=%s poison=%s no_refresh=%s warm_full=%s rank=%d env=%s' % (
        'DETERMINISTIC' if ok else 'LOSS-MOVED', a.single_stream, a.poison, a.no_refresh, a.warm_full, a.rank,
        {k: v for k, v in os.environ.items() if k.startswith('NSP_')}), flush=True)


if __name__ == '__main__':
    main()
