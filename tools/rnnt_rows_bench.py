#!/usr/bin/env python
"""The RNN-T joint's two logit passes alone, at the bench workload's size (3.6 M lattice nodes x 1000 words x joint
width 512): node-stationary kernel (nsp_rnnt_joint_rows) against the tiled GEMM epilogues (nsp_rnnt_joint_gemm +
lse_merge / record packing), interleaved in one process; results of the two paths compared."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sp_amd import ops, _lib
from neural_sp_amd.ops import _p, _stream

dev = torch.device('cuda:0')
L = _lib.lib()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 3601837
J, V, Vp, blank = 512, 1000, 1024, 0
torch.manual_seed(0)
h = torch.empty(M, J, device=dev, dtype=torch.bfloat16)
for r0 in range(0, M, 1 << 18):
    h[r0:r0 + (1 << 18)] = torch.tanh(torch.randn(min(1 << 18, M - r0), J, device=dev)).bfloat16()
w = torch.zeros(Vp, J, device=dev, dtype=torch.bfloat16)
w[:V] = (torch.randn(V, J, device=dev) * (2.0 / J ** 0.5)).bfloat16()
bias = torch.zeros(Vp, device=dev); bias[:V] = torch.randn(V, device=dev)
lab = torch.randint(-1, V, (M,), device=dev, dtype=torch.int32)
aux_r = torch.empty(3, M, device=dev); aux_g = torch.empty(3, M, device=dev)
part = torch.empty(M, Vp // 64, 2, device=dev)
gb = torch.rand(M, device=dev) * 0.01; gl = torch.rand(M, device=dev) * 0.01
d_r = torch.empty(M, Vp, device=dev, dtype=torch.bfloat16); d_g = torch.empty(M, Vp, device=dev, dtype=torch.bfloat16)
db_r = torch.empty((M + 255) // 256, Vp, device=dev); db_g = torch.zeros((M + 127) // 128 * 2, Vp, device=dev)
rec = torch.empty(M, 4, device=dev)


def lse_rows():
    assert L.nsp_rnnt_joint_rows(1, _p(h), _p(w), _p(bias), M, V, Vp, J, blank, _p(lab), _p(aux_r[0]), _p(aux_r[1]), _p(aux_r[2]), None, None, 1.0, None, _stream()) == 0
def lse_gemm():
    assert L.nsp_rnnt_joint_gemm(1, _p(h), _p(w), _p(bias), M, V, Vp, J, blank, _p(lab), _p(part), _p(aux_g[1]), _p(aux_g[2]), None, None, 1.0, None, None, _stream()) == 0
    assert L.nsp_rnnt_lse_merge(_p(part), Vp // 64, _p(aux_g[0]), _p(aux_g[1]), _p(aux_g[2]), _p(lab), M, _stream()) == 0
def dl_rows():
    assert L.nsp_rnnt_joint_rows(2, _p(h), _p(w), _p(bias), M, V, Vp, J, blank, _p(lab), _p(aux_g[0]), _p(gb), _p(gl), _p(db_r), _p(d_r), 0.5, None, _stream()) == 0
def dl_gemm():
    assert L.nsp_rnnt_joint_gemm(2, _p(h), _p(w), _p(bias), M, V, Vp, J, blank, _p(lab), _p(aux_g[0]), _p(gb), _p(gl), _p(db_g), _p(d_g), 0.5, None, _p(rec), _stream()) == 0


def dl_rows_noslabs():
    assert L.nsp_rnnt_joint_rows(2, _p(h), _p(w), _p(bias), M, V, Vp, J, blank, _p(lab), _p(aux_g[0]), _p(gb), _p(gl), None, _p(d_r), 0.5, None, _stream()) == 0
def dl_rows_nostore():
    os.environ['NSP_RNNT_ROWS_DEBUG'] = '1'
    try:
        assert L.nsp_rnnt_joint_rows(2, _p(h), _p(w), _p(bias), M, V, Vp, J, blank, _p(lab), _p(aux_g[0]), _p(gb), _p(gl), _p(db_r), _p(d_r), 0.5, None, _stream()) == 0
    finally:
        os.environ.pop('NSP_RNNT_ROWS_DEBUG')


def timeit(fn, reps=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ops.set_compute_mode('bf16')
for fn in (lse_gemm, lse_rows, dl_gemm, dl_rows): fn()
torch.cuda.synchronize()
print('lse      rows vs gemm: max |d lse| %.3g, |d lp_blank| %.3g, |d lp_label| %.3g (finite labels)' % (
    (aux_r[0] - aux_g[0]).abs().max().item(), (aux_r[1] - aux_g[1]).abs().max().item(),
    (aux_r[2] - aux_g[2])[lab >= 0].abs().max().item()))
rows = torch.randint(0, M, (4096,), device=dev)
print('dlogits  rows vs gemm on 4096 sampled nodes: max |diff| %.3g (values up to %.3g); bias-gradient sums rel diff %.3g' % (
    (d_r[rows].float() - d_g[rows].float()).abs().max().item(), d_g[rows].float().abs().max().item(),
    ((db_r.sum(0) - db_g.sum(0)).abs().max() / db_g.sum(0).abs().max()).item()))
fl = 2.0 * M * Vp * J
best = {}
for rnd in range(3):
    for name, fn in (('lse gemm+merge', lse_gemm), ('lse rows', lse_rows), ('dlogits gemm', dl_gemm), ('dlogits rows', dl_rows),
                     ('dlogits rows, no bias-gradient sums', dl_rows_noslabs), ('dlogits rows, image stores dropped', dl_rows_nostore)):
        best[name] = min(best.get(name, 1e9), timeit(fn))
for name, t in best.items():
    print('%-40s %7.3f ms  (%5.0f TFLOP/s executed)' % (name, t, fl / t / 1e9))
