"""Round-5 diagnosis: the first kernels of a step (H2D -> nsp_pad_batch -> nsp_conv2d3x3_fwd, C_in = 1) on one process
while ANOTHER process keeps the device busy -- the situation in which rank 1 of the stock-DDP test saw a first-layer output
with ~100 scattered wrong pixels although its inputs (checked before and after by torch reductions) were right.

  python tools/conv_first_kernel_stress.py --load-seconds 40 &      # the neighbour: XS transducer steps in a loop
  python tools/conv_first_kernel_stress.py --probe 4000 [--gap-ms 2] [--bcast] [--twice]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load_torch(seconds):
    """a neighbour that runs no kernel of this repository: library GEMMs + elementwise ops"""
    import torch
    dev = torch.device('cuda', 0)
    a = torch.randn(2048, 2048, device=dev, dtype=torch.bfloat16)
    b = torch.randn(2048, 2048, device=dev, dtype=torch.bfloat16)
    c = torch.randn(4 << 20, device=dev)
    t0, n = time.time(), 0
    while time.time() - t0 < seconds:
        for _ in range(20):
            d = a @ b
            e = torch.relu(c * 1.0001 + 0.5)
            f = torch.softmax(d.float(), dim=-1)
        n += 20
        torch.cuda.synchronize()
    print('[load torch] %d rounds in %.1f s' % (n, time.time() - t0), flush=True)


def probe_mm(a):
    """a probe that runs no kernel of this repository: one library GEMM per trial, compared with its first result"""
    import torch
    dev = torch.device('cuda', 0)
    x = torch.randn(1024, 768, device=dev, dtype=torch.bfloat16)
    w = torch.randn(768, 1024, device=dev, dtype=torch.bfloat16)
    ref = x @ w
    for _ in range(3):
        assert torch.equal(x @ w, ref)
    torch.cuda.synchronize()
    bad = torch.zeros(2, device=dev, dtype=torch.int64)
    junk = []
    import numpy as np
    rng = np.random.RandomState(5)
    t0 = time.time()
    for i in range(a.probe):
        if a.gap_ms > 0:
            torch.cuda.synchronize()
            time.sleep(a.gap_ms * 1e-3)
        if a.churn:
            junk.append(torch.full((int(rng.randint(1000, 400000)),), float(i), device=dev))
            if len(junk) > 3:
                junk.pop(int(rng.randint(0, len(junk))))
        y = x @ w
        d = (y != ref)
        bad += torch.stack([d.any().long(), d.sum()])
    torch.cuda.synchronize()
    print('[probe mm, pure torch] %d trials in %.1f s (gap %.1f ms, churn %s): GEMM result differs in %d trials (%d elements)'
          % (a.probe, time.time() - t0, a.gap_ms, a.churn, bad[0].item(), bad[1].item()), flush=True)


def load(seconds):
    import torch
    from neural_sp_amd import ops
    from neural_sp_amd.speech2text import Speech2Text
    from tests import ddp_hip_worker as W
    args = W.model_args(small=False)
    torch.manual_seed(7)
    model = Speech2Text(args).cuda(0)
    full = W.global_batch(args.vocab)
    ops.set_compute_mode('bf16')
    t0, n = time.time(), 0
    while time.time() - t0 < seconds:
        model.zero_grad(set_to_none=True)
        loss, _ = model(W.sub_batch(full, [0, 2]), task='all')
        loss.backward()
        n += 1
        if n % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print('[load] %d steps in %.1f s' % (n, time.time() - t0), flush=True)


def probe(a):
    import numpy as np
    import torch
    from neural_sp_amd import ops
    dev = torch.device('cuda', 0)
    ops.set_compute_mode('bf16')
    rng = np.random.RandomState(3)
    B, T, F = 2, 180, 80
    xs = [rng.randn(T, F).astype(np.float32), rng.randn(T - 37, F).astype(np.float32)]
    xlens = torch.IntTensor([len(x) for x in xs])
    offs = torch.zeros(B, dtype=torch.int64)
    offs[1:] = torch.cumsum(xlens[:-1].long() * F, 0)
    w = torch.randn(32, 1, 3, 3, device=dev) * 0.3
    bias = torch.randn(32, device=dev) * 0.1
    w2 = torch.randn(32, 32, 3, 3, device=dev) * 0.05
    inv_freq = torch.randn(64, device=dev)
    pin = torch.randn(64).pin_memory()
    side = torch.cuda.Stream(device=dev)

    def once():
        out = ops.pad_batch(ops.h2d_packed(xs, dev), ops.h2d(offs, dev), ops.h2d(xlens, dev), B, T, F, 0.)
        y = ops._conv3x3_fwd(out.view(B, T, F, 1), w.permute(0, 2, 3, 1).contiguous(), bias, True, out16=True)
        return out, y

    if a.raw:
        return probe_raw(a, once, xs, w, bias, dev, rng)
    out_ref, y_ref = once()
    for _ in range(3):
        o, y = once()
        assert torch.equal(o, out_ref) and torch.equal(y, y_ref)
    torch.cuda.synchronize()
    print('[probe] reference fixed; waiting for the neighbour', flush=True)
    time.sleep(a.wait)
    bad_y = torch.zeros((), device=dev, dtype=torch.int64)
    bad_y2 = torch.zeros((), device=dev, dtype=torch.int64)
    bad_out = torch.zeros((), device=dev, dtype=torch.int64)
    bad_pix = torch.zeros((), device=dev, dtype=torch.int64)
    junk = []
    t0 = time.time()
    for i in range(a.probe):
        if a.gap_ms > 0:
            torch.cuda.synchronize()
            time.sleep(a.gap_ms * 1e-3)
        if a.bcast:           # what gloo's broadcast does on a receiving rank: H2D on a pool stream between two event waits
            ev = torch.cuda.Event()
            ev.record()
            side.wait_event(ev)
            with torch.cuda.stream(side):
                inv_freq.copy_(pin, non_blocking=True)
                ev2 = torch.cuda.Event()
                ev2.record(side)
            torch.cuda.current_stream().wait_event(ev2)
        if a.churn:           # move the allocator: blocks of other sizes with other contents come and go
            junk.append(torch.full((int(rng.randint(1000, 400000)),), float(i), device=dev))
            if len(junk) > 3:
                junk.pop(int(rng.randint(0, len(junk))))
        out, y = once()
        bad_out += (out != out_ref).any().long()
        d = (y != y_ref).any(dim=3)
        bad_y += d.any().long()
        bad_pix += d.sum()
        if a.twice:
            y2 = ops._conv3x3_fwd(out.view(B, T, F, 1), w.permute(0, 2, 3, 1).contiguous(), bias, True, out16=True)
            bad_y2 += (y2 != y_ref).any().long()
        if a.dump and i % 50 == 49:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print('[probe] %d trials in %.1f s (gap %.1f ms, bcast %s, churn %s): first conv output wrong %d times (%d pixels in all), '
          'immediate re-run wrong %d times, padded input wrong %d times'
          % (a.probe, time.time() - t0, a.gap_ms, a.bcast, a.churn, bad_y.item(), bad_pix.item(), bad_y2.item(), bad_out.item()), flush=True)


def explain(y, dd, x, w, bias, shown):
    """for each wrong pixel: which subset of the 9 taps (and which per-tap scaling) reproduces what the kernel stored?"""
    import itertools
    import numpy as np
    import torch
    B, T, F = x.shape[:3]
    xc = x.reshape(B, T, F).float().cpu().numpy()
    wc = w.float().cpu().numpy().reshape(32, 3, 3)
    bc = bias.float().cpu().numpy()
    pix = sorted(set((int(r[0]), int(r[1]), int(r[2])) for r in dd.tolist()))
    for (b, t, f) in pix[:4]:
        got = y[b, t, f].float().cpu().numpy()
        contrib = np.zeros((9, 32), np.float32)
        for dt in (-1, 0, 1):
            for df in (-1, 0, 1):
                tt, ff = t + dt, f + df
                xv = xc[b, tt, ff] if (0 <= tt < T and 0 <= ff < F) else 0.0
                contrib[(dt + 1) * 3 + (df + 1)] = xv * wc[:, dt + 1, df + 1]
        # per channel group e = c % 4 (acc.x / .y / .z / .w of a thread: x and z are the LOW halves of the packed accumulators)
        res = {}
        for e in range(4):
            ch = np.arange(e, 32, 4)
            want = torch.tensor(np.maximum(bc + contrib.sum(0), 0.0)).to(torch.bfloat16).float().numpy()
            if np.all(got[ch] == want[ch]):
                res[e] = 'right'
                continue
            found = None
            for nmiss in range(1, 5):
                for miss in itertools.combinations(range(9), nmiss):
                    keep = [k for k in range(9) if k not in miss]
                    for relu in (True, False):
                        v = bc + contrib[keep].sum(0)
                        if relu:
                            v = np.maximum(v, 0.0)
                        for mode in ('rne', 'trunc'):
                            if mode == 'rne':
                                vb = torch.tensor(v).to(torch.bfloat16).float().numpy()
                            else:
                                vb = (v.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
                            if np.all(np.abs(vb[ch] - got[ch]) <= 1e-6 + 0.004 * np.abs(got[ch])):
                                found = 'taps %s missing (relu %s, %s)' % (list(miss), relu, mode)
                                break
                        if found:
                            break
                    if found:
                        break
                if found:
                    break
            res[e] = found or ('unexplained: got %s want %s' % (got[ch][:4].tolist(), want[ch][:4].tolist()))
        print('[explain] pixel (b %d, t %d, f %d), channel groups c %% 4 = 0..3 (tap = 3*(dt+1) + (df+1)): %s' % (b, t, f, res), flush=True)
        shown[0] += 1


def probe_raw(a, once, xs, w, bias, dev, rng):
    """the conv launched on NaN-prefilled output buffers of this probe's own: a line that is wrong right after the launch is
    either still NaN (the conv's store is not there: lost, or overtaken by the older fill) or some other value (computed
    wrong); a second look after a device synchronisation says whether the first look was merely early / stale"""
    import torch
    from neural_sp_amd import _lib, ops
    L = _lib.lib()
    B, T, F = 2, 180, 80
    out_ref, y_ref = once()
    torch.cuda.synchronize()
    x = out_ref.clone()
    wcl = w.permute(0, 2, 3, 1).contiguous()
    pool = [torch.empty_like(y_ref) for _ in range(6)]
    st = torch.cuda.current_stream().cuda_stream
    ev_first = torch.zeros(6, device=dev, dtype=torch.int64)     # [wrong at first look, of which NaN, wrong at second look, of which NaN, trials with any, -]
    junk = []
    seen = 0
    shown = [0]
    t0 = time.time()
    for i in range(a.probe):
        if a.gap_ms > 0:
            torch.cuda.synchronize()
            time.sleep(a.gap_ms * 1e-3)
        if a.churn:
            junk.append(torch.full((int(rng.randint(1000, 400000)),), float(i), device=dev))
            if len(junk) > 3:
                junk.pop(int(rng.randint(0, len(junk))))
        if a.h2d:
            out, _ = once()
        y = pool[i % len(pool)]
        y.fill_(float('nan'))
        rc = L.nsp_conv2d3x3_fwd(x.data_ptr(), wcl.data_ptr(), bias.data_ptr(), y.data_ptr(), B, T, F, 1, 32, 1, None, 0, 1, st)
        assert rc == 0
        d1 = (y != y_ref)
        n1 = torch.isnan(y)
        if a.second_look:
            torch.cuda.synchronize()
            d2 = (y != y_ref)
            n2 = torch.isnan(y)
        else:
            d2, n2 = d1, n1
        ev_first += torch.stack([d1.sum(), n1.sum(), d2.sum(), n2.sum(), d1.any().long(), d2.any().long()])
        if a.dump and i % 500 == 499:
            if ev_first[4].item() > seen:
                seen = ev_first[4].item()
                for k in range(len(pool)):
                    dd = (pool[k] != y_ref).nonzero()
                    if dd.numel():
                        print('[probe raw] wrong elements (b, t, f, channel) in buffer %d: %s' % (k, dd.tolist()[:24]), flush=True)
                        if shown[0] < 12:
                            explain(pool[k], dd, x, w, bias, shown)
    torch.cuda.synchronize()
    e = ev_first.tolist()
    print('[probe raw] %d trials in %.1f s (gap %.1f ms, churn %s, h2d %s, second look %s): trials with a wrong first look %d '
          '(%d elements, %d of them still NaN); wrong at the second look %d trials (%d elements, %d NaN)'
          % (a.probe, time.time() - t0, a.gap_ms, a.churn, a.h2d, a.second_look, e[4], e[0], e[1], e[5], e[2], e[3]), flush=True)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--load-seconds', type=float, default=0)
    ap.add_argument('--probe', type=int, default=0)
    ap.add_argument('--gap-ms', type=float, default=0)
    ap.add_argument('--wait', type=float, default=0)
    ap.add_argument('--bcast', action='store_true')
    ap.add_argument('--churn', action='store_true')
    ap.add_argument('--twice', action='store_true')
    ap.add_argument('--dump', action='store_true')
    ap.add_argument('--raw', action='store_true')
    ap.add_argument('--load-kind', default='nsp')
    ap.add_argument('--kind', default='conv')
    ap.add_argument('--h2d', action='store_true')
    ap.add_argument('--second-look', action='store_true')
    a = ap.parse_args()
    if a.load_seconds > 0:
        (load_torch if a.load_kind == 'torch' else load)(a.load_seconds)
    if a.probe > 0:
        (probe_mm if a.kind == 'mm' else probe)(a)
