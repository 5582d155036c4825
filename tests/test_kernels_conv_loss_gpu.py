"""GPU parity of conv / pooling / CTC / RNN-T kernels against torch fp32 (same device) and
the fp64 lattice oracle (oracle/)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device('cuda:0')


def _rel(a, b):
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def test_glu_dwconv_maxpool1d():
    from neural_sp_amd import ops
    torch.manual_seed(0)
    B, T, C, k = 3, 53, 64, 15
    x = torch.randn(B, T, 2 * C, device=_dev(), requires_grad=True)
    y = ops.glu(x)
    ref = F.glu(x, dim=-1)
    g = torch.randn_like(ref)
    assert _rel(y, ref) < 1e-5
    assert _rel(torch.autograd.grad(y, x, g)[0], torch.autograd.grad(ref, x, g)[0]) < 1e-4
    for causal in (False, True):
        xc = torch.randn(B, T, C, device=_dev(), requires_grad=True)
        w = torch.randn(C, 1, k, device=_dev(), requires_grad=True)
        b = torch.randn(C, device=_dev(), requires_grad=True)
        y = ops.depthwise_conv1d(xc, w, b, causal)
        pad = k - 1 if causal else (k - 1) // 2
        ref = F.conv1d(xc.transpose(1, 2), w, b, padding=pad, groups=C)
        if causal:
            ref = ref[:, :, :-pad]
        ref = ref.transpose(1, 2)
        g = torch.randn_like(ref)
        assert _rel(y, ref) < 1e-5
        for a, r in zip(torch.autograd.grad(y, (xc, w, b), g), torch.autograd.grad(ref, (xc, w, b), g)):
            assert _rel(a, r) < 1e-4
    for T2 in (53, 54):
        xm = torch.randn(B, T2, C, device=_dev(), requires_grad=True)
        y = ops.maxpool1d_time(xm, 2)
        ref = F.max_pool1d(xm.transpose(1, 2), 2, 2, ceil_mode=True).transpose(1, 2)
        g = torch.randn_like(ref)
        assert _rel(y, ref) < 1e-6
        assert _rel(torch.autograd.grad(y, xm, g)[0], torch.autograd.grad(ref, xm, g)[0]) < 1e-6


@pytest.mark.parametrize('mode', ['f32', 'bf16', 'bf16maps'])
@pytest.mark.parametrize('Ci', [1, 32])
def test_conv3x3(mode, Ci, monkeypatch):
    """bf16 mode is checked on bf16-representable inputs so that products are exact and
    ReLU masks cannot flip: any residual error is indexing, not rounding.  'bf16' keeps fp32 feature
    maps (NSP_CONV_BF16_MAPS=0: exact comparison); 'bf16maps' is the default throughput layout with the
    [B,T,F,32] maps and their gradients stored as bf16 (one output rounding: 8e-3 of max)."""
    from neural_sp_amd import ops
    maps16 = mode == 'bf16maps'
    monkeypatch.setenv('NSP_CONV_BF16_MAPS', '1' if maps16 else '0')
    mode = 'bf16' if maps16 else mode
    tol = 8e-3 if maps16 else 1e-4
    torch.manual_seed(1)
    B, T, Fq, Co = 2, 37, 40, 32

    def q(t):
        return t.bfloat16().float() if mode == 'bf16' else t
    x = q(torch.randn(B, Ci, T, Fq, device=_dev())).requires_grad_(Ci > 1)
    w = q(torch.randn(Co, Ci, 3, 3, device=_dev()) / math.sqrt(9 * Ci)).requires_grad_()
    b = torch.randn(Co, device=_dev(), requires_grad=True)
    ref = torch.relu(F.conv2d(x, w, b, padding=1))
    g = q(torch.randn_like(ref))
    ins = (x, w, b) if Ci > 1 else (w, b)
    rgrads = torch.autograd.grad(ref, ins, g)
    xcl = x.detach().permute(0, 2, 3, 1).contiguous().requires_grad_(Ci > 1)
    with ops.compute_mode(mode):
        y = ops.conv3x3_relu(xcl, w, b)
        assert y.dtype == (torch.bfloat16 if maps16 else torch.float32)
        assert _rel(y.float().permute(0, 3, 1, 2), ref) < tol
        grads = torch.autograd.grad(y, (xcl, w, b) if Ci > 1 else (w, b), g.permute(0, 2, 3, 1).contiguous().to(y.dtype))
    if Ci > 1:
        assert _rel(grads[0].float().permute(0, 3, 1, 2), rgrads[0]) < tol
    assert _rel(grads[-2], rgrads[-2]) < 1e-3
    assert _rel(grads[-1], rgrads[-1]) < 1e-4


def test_maxpool2d():
    from neural_sp_amd import ops
    torch.manual_seed(2)
    B, T, Fq, C = 2, 37, 41, 32
    for pt, pf in ((2, 2), (1, 1), (2, 1)):
        x = torch.randn(B, C, T, Fq, device=_dev(), requires_grad=True)
        ref = F.max_pool2d(x, (pt, pf), (pt, pf), ceil_mode=True) if pt * pf > 1 else x * 1
        g = torch.randn_like(ref)
        rx, = torch.autograd.grad(ref, x, g)
        xcl = x.detach().permute(0, 2, 3, 1).contiguous().requires_grad_()
        for btcf in (False, True):
            y = ops.maxpool2d(xcl, pt, pf, btcf)
            yr = y.permute(0, 2, 1, 3) if btcf else y.permute(0, 3, 1, 2)
            assert _rel(yr, ref) < 1e-6
            gg = g.permute(0, 2, 1, 3) if btcf else g.permute(0, 2, 3, 1)
            gx, = torch.autograd.grad(y, xcl, gg.contiguous())
            assert _rel(gx.permute(0, 3, 1, 2), rx) < 1e-6
            # bf16 maps: values are selected, never computed -> bit-exact against the fp32 path on the same
            # (bf16-representable) input, forward and backward
            x16 = xcl.detach().bfloat16().requires_grad_()
            x32 = x16.detach().float().requires_grad_()
            y16 = ops.maxpool2d(x16, pt, pf, btcf)
            y32 = ops.maxpool2d(x32, pt, pf, btcf)
            assert y16.dtype == torch.bfloat16 and torch.equal(y16.float(), y32)
            g16, = torch.autograd.grad(y16, x16, gg.contiguous())
            g32, = torch.autograd.grad(y32, x32, gg.contiguous())
            assert g16.dtype == torch.bfloat16 and torch.equal(g16, g32.bfloat16())


def _ctc_ref(logits, ys, elens, lsm):
    B, T, V = logits.shape
    ylens = torch.tensor([len(y) for y in ys], dtype=torch.int32)
    ys_cat = torch.cat([torch.tensor(y, dtype=torch.int32) for y in ys])
    lp = logits.transpose(0, 1).log_softmax(2)
    loss = F.ctc_loss(lp, ys_cat, elens.cpu(), ylens, reduction='sum', zero_infinity=True) / B
    if lsm > 0:
        p = torch.softmax(logits, -1)
        lpp = torch.log_softmax(logits, -1)
        kl = p * (lpp - math.log(1 / (V - 1)))
        kl = sum(kl[b, :elens[b]].sum() for b in range(B)) / elens.sum()
        loss = loss * (1 - lsm) + kl * lsm
    return loss


@pytest.mark.parametrize('lsm', [0.0, 0.1])
def test_ctc_loss(lsm):
    from neural_sp_amd import ops
    torch.manual_seed(3)
    B, T, V = 4, 50, 37
    logits = torch.randn(B, T, V, device=_dev()) * 2
    elens = torch.tensor([50, 41, 30, 8], dtype=torch.int32)
    # last utterance: labels longer than frames allow -> infinite NLL -> zero_infinity
    ys = [[5, 5, 7, 9, 11], [4, 6, 6, 6, 8, 10, 12, 3], [20] * 3, list(range(4, 16))]
    Lmax = max(len(y) for y in ys)
    lab = torch.zeros(B, Lmax, dtype=torch.int32)
    for b, y in enumerate(ys):
        lab[b, :len(y)] = torch.tensor(y)
    ylens = torch.tensor([len(y) for y in ys], dtype=torch.int32)
    lg = logits.clone().requires_grad_()
    loss, nll = ops.ctc_loss(lg, lab.to(_dev()), elens.to(_dev()), ylens.to(_dev()), lsm, int(elens.sum()), 0)
    gl, = torch.autograd.grad(loss, lg)
    lref = logits.detach().cpu().double().requires_grad_()
    ref = _ctc_ref(lref, ys, elens, lsm)
    gr, = torch.autograd.grad(ref, lref)
    assert abs(loss.item() - ref.item()) / abs(ref.item()) < 1e-5, (loss.item(), ref.item())
    assert _rel(gl.cpu().double(), gr) < 1e-4


def test_rnnt_joint_loss():
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
    from oracle.rnnt_ref import rnnt_loss_ref
    from neural_sp_amd import ops
    torch.manual_seed(4)
    B, T, U, J, V = 3, 13, 6, 32, 29
    e = torch.randn(B, T, J, device=_dev(), requires_grad=True)
    gq = torch.randn(B, U + 1, J, device=_dev(), requires_grad=True)
    w = (torch.randn(V, J, device=_dev()) * 0.3).requires_grad_()
    bo = torch.randn(V, device=_dev(), requires_grad=True)
    elens = torch.tensor([13, 9, 5], dtype=torch.int32)
    ylens = torch.tensor([6, 3, 0], dtype=torch.int32)
    lab = torch.randint(1, V, (B, U), dtype=torch.int32)
    for b in range(B):
        lab[b, ylens[b]:] = 0
    with ops.compute_mode('f32'):
        loss, nll = ops.rnnt_joint_loss(e, gq, w, bo, lab.to(_dev()), elens.to(_dev()), ylens.to(_dev()), 0)
        grads = torch.autograd.grad(loss, (e, gq, w, bo))
    # fp64 oracle on CPU
    e64, g64, w64, b64 = [t.detach().cpu().double().requires_grad_() for t in (e, gq, w, bo)]
    logits = torch.tanh(e64[:, :, None] + g64[:, None]) @ w64.t() + b64
    ref = rnnt_loss_ref(torch.log_softmax(logits, -1), lab.long(), elens.long(), ylens.long(), blank=0).mean()
    rg = torch.autograd.grad(ref, (e64, g64, w64, b64))
    assert abs(loss.item() - ref.item()) / abs(ref.item()) < 1e-5, (loss.item(), ref.item())
    for a, r in zip(grads, rg):
        assert _rel(a.cpu().double(), r) < 2e-4


@pytest.mark.parametrize('path', ['padded', 'compact'])
def test_rnnt_lattice_kernels_reproduce_the_published_warp_transducer_answers(path):
    """device twin of tests/test_kernels_emu_cpu.py: nsp_rnnt_logsoftmax_gather / nsp_rnnt_lattice / nsp_rnnt_grad_logits and
    nsp_rnnt_lattice_compact through the C ABI against warp-transducer's published known answers (third-party pin of a17)"""
    from neural_sp_amd import _lib
    from tests import rnnt_known_answers as K
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    for ka in K.CASES:
        nll, g = (K.through_padded_kernels if path == 'padded' else K.through_compact_lattice)(L, ka, lambda t: t.to(_dev()), st)
        K.check(ka, nll, g)


@pytest.mark.parametrize('U,J,V', [(6, 32, 32), (40, 64, 40), (70, 96, 1000)])
def test_rnnt_joint_loss_bf16_fused_backward(U, J, V):
    """bf16 mode: tanh' applied in the data-gradient GEMM epilogue (bf16 dz image) and the single
    reduction pass for both joint-input gradients, against the fp64 oracle (bf16 tolerances)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
    from oracle.rnnt_ref import rnnt_loss_ref_diag
    from neural_sp_amd import ops
    torch.manual_seed(U)
    B, T = 3, 17
    e = (torch.randn(B, T, J, device=_dev()) * 0.7).requires_grad_()
    gq = (torch.randn(B, U + 1, J, device=_dev()) * 0.7).requires_grad_()
    w = (torch.randn(V, J, device=_dev()) * 0.2).requires_grad_()
    bo = torch.randn(V, device=_dev(), requires_grad=True)
    elens = torch.tensor([T, T - 4, 5], dtype=torch.int32)
    ylens = torch.tensor([U, max(0, U - 3), 0], dtype=torch.int32)
    lab = torch.randint(1, V, (B, U), dtype=torch.int32)
    for b in range(B):
        lab[b, ylens[b]:] = 0
    with ops.compute_mode('bf16'):
        loss, nll = ops.rnnt_joint_loss(e, gq, w, bo, lab.to(_dev()), elens.to(_dev()), ylens.to(_dev()), 0)
        grads = torch.autograd.grad(loss, (e, gq, w, bo))
    e64, g64, w64, b64 = [t.detach().cpu().double().requires_grad_() for t in (e, gq, w, bo)]
    logits = torch.tanh(e64[:, :, None] + g64[:, None]) @ w64.t() + b64
    ref = rnnt_loss_ref_diag(torch.log_softmax(logits, -1), lab.long(), elens.long(), ylens.long(), blank=0).mean()
    rg = torch.autograd.grad(ref, (e64, g64, w64, b64))
    assert abs(loss.item() - ref.item()) / abs(ref.item()) < 5e-3, (loss.item(), ref.item())
    for a, r in zip(grads, rg):
        assert _rel(a.cpu().double(), r) < 3e-2


def test_ctc_forced_align_vs_reference_golden():
    """trigger points must equal the reference CTCForcedAligner's output (fixture generated by
    oracle/gen_golden.py run_align from neural_sp/models/seq2seq/decoders/ctc.py:628-753)."""
    import os
    from neural_sp_amd.decoders import CTCForcedAligner
    fix = torch.load(os.path.join(os.path.dirname(__file__), 'golden', 'ctc_align.pt'), weights_only=False)
    tp = CTCForcedAligner()(fix['logits'].to(_dev()), fix['elens'], fix['ys'])
    assert tp.dtype == torch.int32
    assert torch.equal(tp.cpu(), fix['trigger_points'].int()), (tp.cpu().tolist(), fix['trigger_points'].tolist())


@pytest.mark.parametrize('mode,tol', [('f32', 1e-4), ('bf16', 3e-2)])
def test_lstm_vs_torch(mode, tol):
    from neural_sp_amd import ops
    torch.manual_seed(5)
    B, L, I, H = 5, 23, 48, 64
    ref = torch.nn.LSTM(I, H, 1, batch_first=True).to(_dev())
    x = torch.randn(B, L, I, device=_dev(), requires_grad=True)
    dy = torch.randn(B, L, H, device=_dev())
    yr, _ = ref(x)
    params = [ref.weight_ih_l0, ref.weight_hh_l0, ref.bias_ih_l0, ref.bias_hh_l0]
    gr = torch.autograd.grad(yr, [x] + params, dy)
    with ops.compute_mode(mode):
        y = ops.lstm(x, *params)
        assert _rel(y, yr) < tol
        g = torch.autograd.grad(y, [x] + params, dy)
    for a, r in zip(g, gr):
        assert _rel(a, r) < tol


@pytest.mark.gpu
@pytest.mark.parametrize('nl,p_drop,B,L', [(1, 0.0, 5, 9), (2, 0.0, 5, 23), (2, 0.3, 18, 11), (3, 0.2, 4, 6)])
def test_lstm_stack_wavefront_vs_torch(nl, p_drop, B, L):
    """The (layer, time) wavefront (all layers in one launch per stage, input projection of the
    upper layers and the inter-layer dropout fused) against torch.nn.LSTM layers with the SAME
    dropout masks (regenerated from the op's seed/offset)."""
    from neural_sp_amd import ops
    torch.manual_seed(11)
    I, H = 48, 64
    refs = [torch.nn.LSTM(I if l == 0 else H, H, 1, batch_first=True).to(_dev()) for l in range(nl)]
    x = torch.randn(B, L, I, device=_dev(), requires_grad=True)
    dy = torch.randn(B, L, H, device=_dev())
    layers = [(r.weight_ih_l0, r.weight_hh_l0, r.bias_ih_l0, r.bias_hh_l0) for r in refs]
    flat = [t for lay in layers for t in lay]
    seeds = []
    orig = ops.next_dropout_seed

    def recording():
        sd = orig()
        seeds.append(sd)
        return sd
    ops.next_dropout_seed = recording
    try:
        with ops.compute_mode('bf16'):
            assert ops.lstm_stack_supported(layers, x)
            y = ops.lstm_stack(x, layers, p_drop)
            g = torch.autograd.grad(y, [x] + flat, dy)
    finally:
        ops.next_dropout_seed = orig
    assert len(seeds) == (nl - 1 if p_drop > 0 else 0)
    h = x
    for l, r in enumerate(refs):
        h, _ = r(h)
        if l < nl - 1 and p_drop > 0:
            mask = ops.dropout_raw(torch.ones(B, L, H, device=_dev()), p_drop, seeds[l][0], seeds[l][1])
            assert 0.0 < float((mask == 0).float().mean()) < 2 * p_drop
            h = h * mask
    gr = torch.autograd.grad(h, [x] + flat, dy)
    assert _rel(y, h) < 2e-2
    for a, r_ in zip(g, gr):
        assert _rel(a, r_) < 3e-2


@pytest.mark.gpu
@pytest.mark.parametrize('nl,p_drop,B,L,H', [(2, 0.0, 16, 40, 256), (2, 0.25, 7, 33, 512), (1, 0.0, 3, 5, 256),
                                             (3, 0.1, 16, 12, 256), (2, 0.2, 37, 21, 256), (2, 0.0, 64, 9, 256),
                                             (2, 0.2, 83, 7, 256)])
def test_lstm_stack_persistent_matches_wavefront(nl, p_drop, B, L, H, monkeypatch):
    """The persistent (grid-barrier) recurrence must reproduce the per-stage-launch wavefront:
    same bf16 operands and the same fp32 accumulation order -> (near) identical outputs and grads;
    a barrier that timed out would poison the result with NaN."""
    from neural_sp_amd import ops
    torch.manual_seed(13)
    I = 64
    refs = [torch.nn.LSTM(I if l == 0 else H, H, 1, batch_first=True).to(_dev()) for l in range(nl)]
    x = torch.randn(B, L, I, device=_dev(), requires_grad=True)
    dy = torch.randn(B, L, H, device=_dev())
    layers = [(r.weight_ih_l0, r.weight_hh_l0, r.bias_ih_l0, r.bias_hh_l0) for r in refs]
    flat = [t for lay in layers for t in lay]
    outs = {}
    # persistent launches layer by layer (opt-in), the persistent (layer, time) wavefront (default), per-stage launches
    for mode, (flag, layerwise) in {'layers': ('1', '1'), 'wavefront': ('1', '0'), 'stages': ('0', '0')}.items():
        monkeypatch.setenv('NSP_LSTM_PERSISTENT', flag)
        monkeypatch.setenv('NSP_LSTM_LAYERWISE', layerwise)
        ops._DROPOUT_STATE['counter'] = 77
        with ops.compute_mode('bf16'):
            y = ops.lstm_stack(x, layers, p_drop)
            outs[mode] = [y] + list(torch.autograd.grad(y, [x] + flat, dy))
    for mode in ('layers', 'wavefront'):
        for a, b in zip(outs[mode], outs['stages']):
            assert torch.isfinite(a).all()
            assert _rel(a, b) < 2e-3, mode


@pytest.mark.gpu
@pytest.mark.parametrize('M,J,V,blank', [(1, 128, 70, 69), (300, 256, 1000, 0), (70001, 512, 257, 5), (513, 512, 64, 63)])
def test_rnnt_joint_rows_matches_tiled_gemm_path(M, J, V, blank):
    """nsp_rnnt_joint_rows (node-stationary: 256 nodes x whole vocabulary per workgroup) against nsp_rnnt_joint_gemm
    (+ nsp_rnnt_lse_merge), the path it replaces, on the same operands: one node, a ragged last workgroup, a blank
    index that is not 0 (last column of the vocabulary; of a vocabulary that fills its last 64-column slice exactly),
    vocabulary padding inside the last slice (V = 257: 63 padded columns), labels -1 (no label) mixed in.  Forward:
    log-sum-exp and both log-probabilities; backward: the bf16 gradient image and the bias-gradient sums."""
    from neural_sp_amd import ops, _lib
    from neural_sp_amd.ops import _p, _stream
    L = _lib.lib()
    dev = _dev()
    Vp = (V + 63) // 64 * 64
    torch.manual_seed(M + V)
    h = torch.tanh(torch.randn(M, J, device=dev)).bfloat16()
    w = torch.zeros(Vp, J, device=dev, dtype=torch.bfloat16)
    w[:V] = (torch.randn(V, J, device=dev) * (2.0 / J ** 0.5)).bfloat16()
    bias = torch.zeros(Vp, device=dev)
    bias[:V] = torch.randn(V, device=dev)
    lab = torch.randint(-1, V, (M,), device=dev, dtype=torch.int32)
    aux_r = torch.full((3, M), float('nan'), device=dev)
    aux_g = torch.full((3, M), float('nan'), device=dev)
    part = torch.empty(M, Vp // 64, 2, device=dev)
    with ops.compute_mode('bf16'):
        assert L.nsp_rnnt_joint_rows(1, _p(h), _p(w), _p(bias), M, V, Vp, J, blank, _p(lab), _p(aux_r[0]), _p(aux_r[1]),
                                     _p(aux_r[2]), None, None, 1.0, None, _stream()) == 0
        assert L.nsp_rnnt_joint_gemm(1, _p(h), _p(w), _p(bias), M, V, Vp, J, blank, _p(lab), _p(part), _p(aux_g[1]),
                                     _p(aux_g[2]), None, None, 1.0, None, None, _stream()) == 0
        assert L.nsp_rnnt_lse_merge(_p(part), Vp // 64, _p(aux_g[0]), _p(aux_g[1]), _p(aux_g[2]), _p(lab), M, _stream()) == 0
        # against torch as well: fp32 logits of the bf16 operands
        logits = h.float() @ w[:V].float().t() + bias[:V]
        lse = torch.logsumexp(logits, -1)
        assert (aux_r[0] - lse).abs().max().item() < 2e-4
        assert (aux_r[1] - (logits[:, blank] - lse)).abs().max().item() < 2e-4
        has = lab >= 0
        pick = logits.gather(1, lab.clamp(min=0).long()[:, None])[:, 0] - lse
        assert (aux_r[2][has] - pick[has]).abs().max().item() < 2e-4 if has.any() else True
        assert torch.isinf(aux_r[2][~has]).all() and (aux_r[2][~has] < 0).all()
        assert (aux_r[0] - aux_g[0]).abs().max().item() < 1e-5 and (aux_r[1] - aux_g[1]).abs().max().item() < 1e-5
        gb = torch.rand(M, device=dev) * 0.01
        gl = torch.rand(M, device=dev) * 0.01 * has
        d_r = torch.full((M, Vp), float('nan'), device=dev, dtype=torch.bfloat16)
        d_g = torch.full((M, Vp), float('nan'), device=dev, dtype=torch.bfloat16)
        db_r = torch.full(((M + 255) // 256, Vp), float('nan'), device=dev)
        db_g = torch.zeros((M + 127) // 128 * 2, Vp, device=dev)
        rec = torch.empty(M, 4, device=dev)
        sd = torch.tensor([2.0], device=dev)
        assert L.nsp_rnnt_joint_rows(2, _p(h), _p(w), _p(bias), M, V, Vp, J, blank, _p(lab), _p(aux_g[0]), _p(gb), _p(gl),
                                     _p(db_r), _p(d_r), 0.25, _p(sd), _stream()) == 0
        assert L.nsp_rnnt_joint_gemm(2, _p(h), _p(w), _p(bias), M, V, Vp, J, blank, _p(lab), _p(aux_g[0]), _p(gb), _p(gl),
                                     _p(db_g), _p(d_g), 0.25, _p(sd), _p(rec), _stream()) == 0
    want = -(0.5 * (gb + gl))[:, None] * torch.exp(logits - aux_g[0][:, None])
    want[:, blank] += 0.5 * gb
    want[has, lab[has].long()] += 0.5 * gl[has]
    assert torch.isfinite(d_r.float()).all() and (d_r[:, V:] == 0).all()
    assert (d_r[:, :V].float() - want).abs().max().item() < 1e-2 * want.abs().max().item() + 1e-6
    assert (d_r.float() - d_g.float()).abs().max().item() <= 1e-2 * want.abs().max().item()
    assert _rel(db_r.sum(0)[:V], want.sum(0)) < 2e-3 and _rel(db_r.sum(0), db_g.sum(0)) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize('B,T,U,J,V', [(3, 21, 6, 32, 29), (4, 37, 40, 64, 43), (3, 50, 70, 96, 1000),
                                       (5, 120, 33, 512, 1000), (2, 9, 0, 64, 130), (3, 30, 9, 128, 130), (5, 100, 33, 256, 1000),
                                       (2, 12, 4, 128, 38000)])     # (a vocabulary whose bias does not fit beside the node-stationary kernel's ring: tiled path)
@pytest.mark.parametrize('rows', ['1', '0'])
def test_rnnt_joint_loss_fused_compact_no_logit_tensor(B, T, U, J, V, rows, monkeypatch):
    """The fused / compacted RNN-T joint (csrc/rnnt_fused.hip + the NSP_EPI_RNNT_* GEMM epilogues):
    loss and all four gradients against the fp64 lattice oracle on the materialised joint, ragged
    lengths incl. an empty label sequence and T_b = 1, V % 64 != 0.  Memory: no fp32 [.,V] tensor exists;
    the only V-wide buffer is the bf16 gradient image over the M COMPACTED nodes that the weight-gradient
    GEMM reduces over (2 Vp bytes per node), next to h and dz (2 J bytes each): the peak must fit that
    budget, which is less than half of what logits(fp32) + log-softmax gradient(bf16) + h + dz cost on
    the padded [B,T,U+1] grid."""
    from oracle.rnnt_ref import rnnt_loss_ref_diag
    from neural_sp_amd import ops
    # rows = '1': J = 128 / 256 / 512 take the node-stationary kernel (nsp_rnnt_joint_rows: whole vocabulary per workgroup,
    # no partials / merge / packed records); '0' keeps every width on the tiled GEMM epilogues
    monkeypatch.setenv('NSP_RNNT_ROWS', rows)
    if rows == '1' and (J not in (128, 256, 512) or V > 3000):
        pytest.skip('outside the node-stationary kernel: same path as rows = 0')
    torch.manual_seed(B * 1000 + T)
    e = (torch.randn(B, T, J, device=_dev()) * 0.7).requires_grad_()
    gq = (torch.randn(B, U + 1, J, device=_dev()) * 0.7).requires_grad_()
    w = (torch.randn(V, J, device=_dev()) * (2.0 / J ** 0.5)).requires_grad_()
    bo = torch.randn(V, device=_dev(), requires_grad=True)
    el = [T, max(1, T - 4), 1, max(1, T // 2), T][:B]
    yl = [U, max(0, U - 3), 0, U // 2, U][:B]
    elens, ylens = torch.tensor(el, dtype=torch.int32), torch.tensor(yl, dtype=torch.int32)
    lab = torch.randint(1, V, (B, max(U, 1)), dtype=torch.int32)
    for b in range(B):
        lab[b, yl[b]:] = 0
    with ops.compute_mode('bf16'):
        M = sum(t * (u + 1) for t, u in zip(el, yl))
        assert ops.rnnt_joint_fused_supported(J, U + 1, M)
        on_device = _dev().type == 'cuda'      # (the CPU tier runs this body on the emulator, without the memory part)
        if on_device:
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats()
            base = torch.cuda.memory_allocated()
        loss, nll = ops.rnnt_joint_loss(e, gq, w, bo, lab.to(_dev()), elens.to(_dev()), ylens.to(_dev()), 0,
                                        elens_host=el, ylens_host=yl)
        assert loss.grad_fn.__class__.__name__.startswith('RNNTJointLossFusedFn')
        grads = torch.autograd.grad(loss, (e, gq, w, bo))
        if on_device:
            torch.cuda.synchronize()
            peak = torch.cuda.max_memory_allocated() - base
    Vp = (V + 63) // 64 * 64
    budget = M * (2 * Vp + 4 * J + 256) + 20 * V * J * 4 + (8 << 20)      # d16 + h + dz + per-node scalars; dW slabs
    padded_old = B * T * (U + 1) * (4 * V + 2 * Vp + 4 * J)
    if on_device:
        assert peak < budget, (peak, budget)
        if padded_old > (64 << 20):
            assert peak < 0.6 * padded_old, (peak, padded_old)
    e64, g64, w64, b64 = [t.detach().cpu().double().requires_grad_() for t in (e, gq, w, bo)]
    logits = torch.tanh(e64[:, :, None] + g64[:, None]) @ w64.t() + b64
    refs = rnnt_loss_ref_diag(torch.log_softmax(logits, -1), lab.long(), elens.long(), ylens.long(), blank=0)
    ref = refs.mean()
    rg = torch.autograd.grad(ref, (e64, g64, w64, b64))
    assert torch.allclose(nll.cpu().double(), refs.detach(), rtol=5e-3, atol=1e-2), (nll, refs)
    assert abs(loss.item() - ref.item()) / abs(ref.item()) < 5e-3, (loss.item(), ref.item())
    for a, r, name in zip(grads, rg, ('d enc_proj', 'd dec_proj', 'd W_out', 'd b_out')):
        assert a.shape == r.shape
        assert _rel(a.cpu().double(), r) < 3e-2, name
        cos = torch.nn.functional.cosine_similarity(a.cpu().double().flatten(), r.flatten(), dim=0).item()
        assert cos > 0.9995, (name, cos)


@pytest.mark.gpu
@pytest.mark.parametrize('V,lsm', [(43, 0.1), (10000, 0.1), (1000, 0.0)])
def test_xe_lsm_loss_matches_reference_formula(V, lsm):
    """nsp_xe_lsm_fwd_bwd vs criterion.py:45-86 written out in torch fp64 (dense target distribution,
    log_softmax, masked sum / batch size), its autograd gradient, and compute_accuracy's arg-max."""
    from neural_sp_amd import ops
    torch.manual_seed(V)
    B, L, pad = 5, 17, 3
    logits = (torch.randn(B, L, V, device=_dev()) * 2).requires_grad_()
    ys = torch.randint(4, V, (B, L), device=_dev())
    for b in range(B):
        ys[b, L - 2 * b:] = pad
    loss, rows, correct = ops.xe_lsm_loss(logits, ys.reshape(-1).int(), lsm, pad, B)
    g, = torch.autograd.grad(loss, logits)
    x = logits.detach().double().cpu().view(-1, V).requires_grad_()
    yo = ys.cpu().view(-1)
    mask = yo == pad
    lp = torch.log_softmax(x, -1)
    tgt = torch.full_like(lp, lsm / (V - 1))
    tgt.scatter_(1, yo.masked_fill(mask, 0).unsqueeze(1), 1 - lsm)
    ref_rows = -(tgt * lp).sum(1).masked_fill(mask, 0)
    ref = ref_rows.sum() / B
    rg, = torch.autograd.grad(ref, x)
    assert abs(loss.item() - ref.item()) / abs(ref.item()) < 1e-5
    assert _rel(rows.cpu().double(), ref_rows.detach()) < 1e-5
    assert _rel(g.cpu().double().view(-1, V), rg) < 1e-4
    assert torch.equal(correct.cpu().bool(), (x.argmax(1) == yo) & ~mask)
