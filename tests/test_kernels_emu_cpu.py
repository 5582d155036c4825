"""Existing HIP kernels without inline asm / gfx950 builtins (elementwise, LayerNorm, CTC, label-smoothed XE,
depthwise conv / GLU / max-pool, greedy-decode helpers) executed on the HOST EMULATOR (tests/hipemu) through the
real ctypes glue and autograd Functions of neural_sp_amd.ops, in fp32 ('f32' compute mode).  These kernels are
pinned on the device by tests/test_kernels_*_gpu.py; this file keeps their arithmetic under test in the CPU tier
too (the tier that runs when no GPU is at hand).  MFMA GEMM / attention / conv2d / LSTM kernels cannot be emulated."""
import pytest
import torch
import torch.nn.functional as F

from tests.hipemu import build_emu

pytestmark = pytest.mark.skipif(not build_emu.available(), reason='no host clang++ for the HIP emulator')


@pytest.fixture
def emu():
    from neural_sp_amd import ops
    from tests.hipemu.shim import emulated_kernels
    with emulated_kernels(), ops.compute_mode('f32'):
        yield ops


def _rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp(min=1e-12)).item()


@pytest.mark.parametrize('act', ['none', 'swish', 'relu'])
def test_layer_norm_forward_backward(emu, act):
    torch.manual_seed(0)
    x = torch.randn(37, 64, requires_grad=True)
    g = (torch.rand(64) + 0.5).requires_grad_(True)
    b = (torch.rand(64) - 0.5).requires_grad_(True)
    w = torch.randn(37, 64)

    def f(y):
        return {'none': y, 'swish': y * torch.sigmoid(y), 'relu': torch.relu(y)}[act]
    yr = f(F.layer_norm(x, (64,), g, b, 1e-12))
    gr = torch.autograd.grad((yr * w).sum(), [x, g, b])
    y = emu.layer_norm(x, g, b, 1e-12, act=act)
    go = torch.autograd.grad((y * w).sum(), [x, g, b])
    assert _rel(y, yr) < 1e-5
    for a, r in zip(go, gr):
        assert _rel(a, r) < 1e-4


def test_layer_norm_split_sums_the_residual_gradient(emu):
    torch.manual_seed(1)
    x = torch.randn(5, 9, 32, requires_grad=True)
    g, b = torch.ones(32, requires_grad=True), torch.zeros(32, requires_grad=True)
    xn, xres = emu.layer_norm_split(x, g, b, 1e-12)
    w1, w2 = torch.randn(5, 9, 32), torch.randn(5, 9, 32)
    (gx,) = torch.autograd.grad((xn * w1).sum() + (xres * w2).sum(), [x])
    xr = x.detach().clone().requires_grad_(True)
    (gr,) = torch.autograd.grad((F.layer_norm(xr, (32,), g, b, 1e-12) * w1).sum() + (xr * w2).sum(), [xr])
    assert _rel(gx, gr) < 1e-4


@pytest.mark.parametrize('lsm', [0.0, 0.1])
def test_ctc_loss_matches_torch(emu, lsm):
    torch.manual_seed(2)
    B, T, V = 4, 30, 11
    logits = torch.randn(B, T, V, requires_grad=True)
    ys = [[3, 4, 4, 5], [1, 2], [7], [2, 2, 2, 9, 1]]
    elens = torch.tensor([30, 21, 9, 17], dtype=torch.int32)
    ylens = torch.tensor([len(y) for y in ys], dtype=torch.int32)
    lab = torch.zeros(B, 5, dtype=torch.int32)
    for i, y in enumerate(ys):
        lab[i, :len(y)] = torch.tensor(y)
    loss, nll = emu.ctc_loss(logits, lab, elens, ylens, lsm, int(elens.sum()), 0)
    (g,) = torch.autograd.grad(loss.sum(), [logits])
    lr = logits.detach().double().requires_grad_(True)
    lp = torch.log_softmax(lr, -1)
    ref = F.ctc_loss(lp.transpose(0, 1), torch.cat([torch.tensor(y) for y in ys]), elens.long(), ylens.long(),
                     blank=0, reduction='sum', zero_infinity=True) / B
    if lsm > 0:       # criterion.py:110-127
        import math
        kl = sum((torch.softmax(lr[b, :int(elens[b])], -1) * (lp[b, :int(elens[b])] - math.log(1 / (V - 1)))).sum()
                 for b in range(B)) / int(elens.sum())
        ref = ref * (1 - lsm) + kl * lsm
    (gr,) = torch.autograd.grad(ref, [lr])
    assert abs(loss.item() - ref.item()) / abs(ref.item()) < 1e-5
    assert _rel(g, gr.float()) < 1e-4
    assert g[1, 21:].abs().max() == 0           # no gradient past an utterance's end


def test_xe_lsm_loss_and_accuracy(emu):
    torch.manual_seed(3)
    rows, V, B = 23, 43, 4
    logits = torch.randn(rows, V, requires_grad=True)
    ys = torch.randint(4, V, (rows,), dtype=torch.int32)
    ys[[2, 9, 22]] = 3                          # pad
    loss, loss_rows, correct = emu.xe_lsm_loss(logits, ys, 0.1, 3, B)
    (g,) = torch.autograd.grad(loss.sum(), [logits])
    lr = logits.detach().double().requires_grad_(True)
    lp = torch.log_softmax(lr, -1)
    tgt = torch.full_like(lp, 0.1 / (V - 1))
    tgt.scatter_(1, ys.long().clamp(min=0).unsqueeze(1), 0.9)
    rows_ref = -(tgt * lp).sum(1).masked_fill(ys == 3, 0)
    (gr,) = torch.autograd.grad(rows_ref.sum() / B, [lr])
    assert abs(loss.item() - rows_ref.sum().item() / B) < 1e-4
    assert _rel(g, gr.float()) < 1e-4
    assert torch.equal(correct.bool(), (logits.argmax(1) == ys.long()) & (ys != 3))


def test_depthwise_conv_glu_maxpool(emu):
    torch.manual_seed(4)
    B, T, C, k = 2, 19, 8, 7
    x = torch.randn(B, T, 2 * C, requires_grad=True)
    conv = torch.nn.Conv1d(C, C, k, padding=(k - 1) // 2, groups=C)
    h = emu.glu(x)
    y = emu.depthwise_conv1d(h, conv.weight, conv.bias, False)
    p = emu.maxpool1d_time(y, 2)
    w = torch.randn_like(p)
    go = torch.autograd.grad((p * w).sum(), [x, conv.weight, conv.bias])
    xr = x.detach().clone().requires_grad_(True)
    yr = conv(F.glu(xr, -1).transpose(2, 1)).transpose(2, 1)
    pr = F.max_pool1d(yr.transpose(2, 1), 2, 2, ceil_mode=True).transpose(2, 1)
    gr = torch.autograd.grad((pr * w).sum(), [xr, conv.weight, conv.bias])
    assert _rel(p, pr) < 1e-5
    for a, r in zip(go, gr):
        assert _rel(a, r) < 1e-4


def test_causal_depthwise_conv(emu):
    torch.manual_seed(5)
    B, T, C, k = 2, 12, 4, 3
    x = torch.randn(B, T, C)
    conv = torch.nn.Conv1d(C, C, k, padding=k - 1, groups=C)
    y = emu.depthwise_conv1d(x, conv.weight, conv.bias, True)
    yr = conv(x.transpose(2, 1))[:, :, :-(k - 1)].transpose(2, 1)
    assert _rel(y, yr) < 1e-5


def test_dropout_mask_is_a_pure_function_of_seed_and_offset(emu):
    x = torch.ones(64, 32)
    a = emu.dropout_raw(x, 0.3, 1234, 0)
    b = emu.dropout_raw(x, 0.3, 1234, 0)
    c = emu.dropout_raw(x, 0.3, 1234, 64 * 32)
    assert torch.equal(a, b) and not torch.equal(a, c)
    keep = (a != 0).float().mean().item()
    assert abs(keep - 0.7) < 0.05 and torch.allclose(a[a != 0], torch.tensor(1 / 0.7))


def test_elementwise_and_input_glue(emu):
    torch.manual_seed(6)
    x, z = torch.randn(3, 5, 8), torch.randn(5, 8)
    assert _rel(emu.scale_add_bcast(x, z, 2.0), 2.0 * x + z) < 1e-6
    assert _rel(emu.axpby(x, x * 2, 0.5, 0.25), 0.5 * x + 0.5 * x) < 1e-6
    pre = torch.randn(4, 8)
    assert _rel(emu.dact_mul(torch.ones(4, 8), pre, emu.ACT['swish']),
                torch.sigmoid(pre) * (1 + pre * (1 - torch.sigmoid(pre)))) < 1e-5
    xs = torch.randn(2, 10, 8)
    ref = xs.clone()
    ref[:, :, 2:5] = 0
    ref[:, 3:6] = 0
    assert torch.equal(emu.specaug_apply_(xs.clone(), [(2, 5)], [(3, 6)]), ref)
    # ragged -> padded
    a, b = torch.randn(4, 8), torch.randn(7, 8)
    packed = torch.cat([a.reshape(-1), b.reshape(-1)])
    out = emu.pad_batch(packed, torch.tensor([0, 32]), torch.tensor([4, 7], dtype=torch.int32), 2, 7, 8)
    assert torch.equal(out[0, :4], a) and out[0, 4:].abs().sum() == 0 and torch.equal(out[1], b)
    # the Transformer-XL position table (positional_embedding.py:131-138)
    inv = 1 / (10000 ** (torch.arange(0.0, 16, 2.0) / 16))
    pos = torch.arange(-1, -10, -1.0)
    s = torch.einsum('i,j->ij', pos, inv)
    assert _rel(emu.xl_pos_table(inv, 9), torch.cat([s.sin(), s.cos()], -1)) < 1e-5


def test_greedy_decode_helpers(emu):
    torch.manual_seed(7)
    x = torch.randn(11, 23)
    x[3, 5] = x[3, 9] = 10.0                    # tie: first index wins
    assert torch.equal(emu.argmax_rows(x).long(), x.argmax(1)) and int(emu.argmax_rows(x)[3]) == 5
    B, H = 3, 8
    gates, h0, c0 = torch.randn(B, 4 * H), torch.randn(B, H), torch.randn(B, H)
    upd = torch.tensor([1, 0, 1], dtype=torch.int32)
    h, c = emu.lstm_cell_step(gates, h0, c0, upd)
    i, f, g, o = gates.chunk(4, 1)
    cr = torch.sigmoid(f) * c0 + torch.sigmoid(i) * torch.tanh(g)
    hr = torch.sigmoid(o) * torch.tanh(cr)
    assert _rel(h[0], hr[0]) < 1e-5 and _rel(c[2], cr[2]) < 1e-5
    assert torch.equal(h[1], h0[1]) and torch.equal(c[1], c0[1])     # rows with update == 0 keep their state


def test_rnnt_lattice_kernels_against_the_fp64_oracle():
    """nsp_rnnt_logsoftmax_gather -> nsp_rnnt_lattice -> nsp_rnnt_grad_logits (rnnt.hip) through the C ABI of the
    emulated library: per-utterance -log P and the gradient w.r.t. the logits vs oracle/rnnt_ref.py (Graves 2012)."""
    from neural_sp_amd import _lib
    from oracle.rnnt_ref import rnnt_loss_ref
    from tests.hipemu.shim import emulated_kernels
    torch.manual_seed(8)
    B, T, U, V = 3, 7, 4, 9
    U1 = U + 1
    logits = torch.randn(B, T, U1, V)
    labels = torch.randint(1, V, (B, U), dtype=torch.int32)
    elens = torch.tensor([7, 5, 3], dtype=torch.int32)
    ylens = torch.tensor([4, 2, 0], dtype=torch.int32)
    lr = logits.double().requires_grad_(True)
    nll_ref = rnnt_loss_ref(torch.log_softmax(lr, -1), labels.long(), elens.long(), ylens.long(), blank=0)
    (g_ref,) = torch.autograd.grad(nll_ref.sum(), [lr])
    f = lambda *s: torch.empty(*s, dtype=torch.float32)
    lse, lpb, lpl, alpha, beta, gb, gl = (f(B, T, U1) for _ in range(7))
    nll = f(B)
    work = logits.clone()
    with emulated_kernels() as L:
        p = lambda t: t.data_ptr()
        assert L.nsp_rnnt_logsoftmax_gather(p(work), p(labels), p(elens), p(ylens), p(lse), p(lpb), p(lpl), B, T, U1, V, 0, 0) == 0
        assert L.nsp_rnnt_lattice(p(lpb), p(lpl), p(elens), p(ylens), p(alpha), p(beta), p(nll), p(gb), p(gl), B, T, U1, 0) == 0
        assert L.nsp_rnnt_grad_logits(p(work), p(lse), p(labels), p(gb), p(gl), p(elens), p(ylens), 1.0, None,
                                      B, T, U1, V, 0, None, 0, None, 0) == 0
    assert _rel(nll, nll_ref.float()) < 1e-5
    assert _rel(work, g_ref.float()) < 1e-4
    assert work[1, 5:].abs().max() == 0 and work[2, :, 1:].abs().max() == 0      # padded lattice nodes: no gradient
    assert _lib.prototypes()['nsp_rnnt_lattice'][1][-1] is not None


@pytest.mark.parametrize('path', ['padded', 'compact'])
def test_rnnt_lattice_kernels_reproduce_the_published_warp_transducer_answers(path):
    """the RNN-T lattice kernels (csrc/rnnt.hip padded grid; csrc/rnnt_fused.hip compact layout) on the host emulator against
    the known-answer cases of warp-transducer's unit tests (tests/rnnt_known_answers.py), directly -- not via the oracle"""
    from tests import rnnt_known_answers as K
    from tests.hipemu.shim import emulated_kernels
    with emulated_kernels() as L:
        for ka in K.CASES:
            nll, g = (K.through_padded_kernels if path == 'padded' else K.through_compact_lattice)(L, ka)
            K.check(ka, nll, g)


def _lstm_case(seed=7, B=3, n=6, I=16, H=32):
    torch.manual_seed(seed)
    ref = torch.nn.LSTM(I, H, 1, batch_first=True)
    x = torch.randn(B, n, I, requires_grad=True)
    h0, c0 = torch.randn(B, H, requires_grad=True), torch.randn(B, H, requires_grad=True)
    params = [ref.weight_ih_l0, ref.weight_hh_l0, ref.bias_ih_l0, ref.bias_hh_l0]
    return ref, x, h0, c0, params, (torch.randn(B, n, H), torch.randn(B, H), torch.randn(B, H))


def test_lstm_step_kernels_vs_torch(emu):
    """nsp_lstm_fwd / nsp_lstm_bwd (fp32 MFMA step kernels) through ops.lstm: outputs and every gradient"""
    ref, x, _, _, params, (wy, _, _) = _lstm_case()
    yr, _ = ref(x)
    gr = torch.autograd.grad((yr * wy).sum(), [x] + params)
    y = emu.lstm(x, *params)
    g = torch.autograd.grad((y * wy).sum(), [x] + params)
    assert _rel(y, yr) < 1e-5
    for a, r in zip(g, gr):
        assert _rel(a, r) < 1e-5


def test_lstm_with_initial_and_final_state(emu):
    """ops.lstm_state (nsp_lstm_*_range): the step kernels started from (h0, c0), differentiable through the initial and
    the final state -- what carries the forward direction of the latency-controlled BLSTM from chunk to chunk"""
    ref, x, h0, c0, params, (wy, wh, wc) = _lstm_case()
    yr, (hn, cn) = ref(x, (h0[None], c0[None]))
    gr = torch.autograd.grad((yr * wy).sum() + (hn[0] * wh).sum() + (cn[0] * wc).sum(), [x, h0, c0] + params)
    y, h, c = emu.lstm_state(x, *params, h0, c0)
    g = torch.autograd.grad((y * wy).sum() + (h * wh).sum() + (c * wc).sum(), [x, h0, c0] + params)
    assert _rel(y, yr) < 1e-5 and _rel(h, hn[0]) < 1e-5 and _rel(c, cn[0]) < 1e-5
    for a, r in zip(g, gr):
        assert _rel(a, r) < 1e-5
    # two chained calls == one call over the concatenation (state hand-over, gradients through it)
    y1, h1, c1 = emu.lstm_state(x[:, :4], *params, h0, c0)
    y2, h2, c2 = emu.lstm_state(x[:, 4:], *params, h1, c1)
    g2 = torch.autograd.grad((torch.cat([y1, y2], 1) * wy).sum() + (h2 * wh).sum() + (c2 * wc).sum(), [x, h0, c0] + params)
    for a, r in zip(g2, gr):
        assert _rel(a, r) < 1e-5
