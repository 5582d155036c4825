#!/usr/bin/env python
"""The training step's main GEMM configurations (Conformer-L, M = B*T' rows) timed back to back, for A/B runs of two
builds of the library: `python tools/gemm_step_ab.py` runs itself as two subprocesses per round (NSP_LIB_OVERRIDE =
tools/probe/ab/libnsp_hip_prev.so vs the tree's library), three interleaved rounds, and prints min times side by side."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import torch
    from neural_sp_amd import ops
    ops.set_compute_mode('bf16')
    dev = torch.device('cuda:0')
    M = int(os.environ.get('GM', '102400'))
    res = {}

    def bench(name, M, N, K, odt=torch.float32, **kw):
        x = (torch.randn(M, K, device=dev)).bfloat16()
        w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
        out = torch.empty(M, N, device=dev, dtype=odt)
        a = dict(kw)
        if a.pop('bias', False): a['bias'] = torch.randn(N, device=dev)
        if a.pop('pre', False): a['pre_out'] = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        if a.pop('res', False): a['res'] = torch.randn(M, N, device=dev)
        if a.pop('dsrc', False): a['dact_src'] = torch.randn(M, N, device=dev).bfloat16()
        if a.pop('slabs', False): a['colsum_slabs'] = torch.zeros(((M + 127) // 128 * 4, N), device=dev)
        for _ in range(2): ops._gemm_raw_untimed(M, N, K, x, K, 1, w, 1, K, out, N, **a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops._gemm_raw_untimed(M, N, K, x, K, 1, w, 1, K, out, N, **a)
        e1.record(); torch.cuda.synchronize()
        res[name] = (e0.elapsed_time(e1) * 100, 2.0 * M * N * K)
    bf = torch.bfloat16
    bench('FFN1 fwd  [M,2048,512] bias swish pre16 drop -> bf16', M, 2048, 512, bf, bias=True, act=2, pre=True, dropout_p=0.1, seed=1, offset=8)
    bench('FFN2 fwd  [M,512,2048] bias drop res -> fp32', M, 512, 2048, bias=True, dropout_p=0.1, seed=1, offset=8, res=True, alpha=0.5)
    bench('QKV       [M,1536,512] plain -> bf16', M, 1536, 512, bf)
    bench('att out   [M,512,512] drop res -> fp32', M, 512, 512, dropout_p=0.1, seed=1, offset=8, res=True)
    bench('pw1       [M,1024,512] bias -> fp32', M, 1024, 512, bias=True)
    bench('pw2       [M,512,512] bias drop res -> fp32', M, 512, 512, bias=True, dropout_p=0.1, seed=1, offset=8, res=True)
    bench('dgrad FFN2 [M,2048,512] swish\' src16 drop slabs -> bf16', M, 2048, 512, bf, dsrc=True, dact=2, dropout_p=0.1, seed=1, offset=8, slabs=True)
    bench('dgrad FFN1 [M,512,2048] plain -> fp32', M, 512, 2048)
    bench('dgrad QKV [M,512,1536] plain -> fp32', M, 512, 1536)
    bench('dgrad d   [M,512,512] plain -> fp32', M, 512, 512)
    print(json.dumps(res))


if __name__ == '__main__':
    if os.environ.get('GEMM_AB_CHILD'):
        child()
        sys.exit(0)
    # arms: "tree" (the in-tree library) plus GEMM_AB_LIBS="name=path,..." (default: prev = tools/probe/ab/libnsp_hip_prev.so)
    arms = [('tree', None)]
    spec = os.environ.get('GEMM_AB_LIBS', 'prev=' + os.path.join(ROOT, 'tools', 'probe', 'ab', 'libnsp_hip_prev.so'))
    for item in spec.split(','):
        n, pth = item.split('=')
        arms.insert(0, (n, pth if os.path.isabs(pth) else os.path.join(ROOT, pth)))
    runs = {n: [] for n, _ in arms}
    for rnd in range(3):
        for which, pth in arms:
            env = dict(os.environ, GEMM_AB_CHILD='1')
            if pth:
                env['NSP_LIB_OVERRIDE'] = pth
            out = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
            if out.returncode != 0:
                print(which, 'failed:', out.stderr[-2000:])
                sys.exit(1)
            runs[which].append(json.loads(out.stdout.strip().splitlines()[-1]))
    names = list(runs['tree'][0])
    print('GEMM (M = %s rows): min us over 3 interleaved rounds' % os.environ.get('GM', '102400'))
    print('%-64s ' % '' + ' '.join('%10s' % n for n, _ in arms))
    for n in names:
        row = [min(r[n][0] for r in runs[a]) for a, _ in arms]
        print('%-64s ' % n + ' '.join('%10.1f' % v for v in row) + '   | vs tree: ' + ' '.join('x%.3f' % (row[-1] / v) for v in row[:-1]))
