"""GPU parity of the basic HIP kernels against plain torch fp32 ops on the same device."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device('cuda:0')


def _rel(a, b):
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


@pytest.mark.parametrize('mode,tol', [('f32', 2e-5), ('bf16', 2e-2)])
@pytest.mark.parametrize('M,N,K', [(128, 128, 32), (300, 1000, 512), (77, 130, 68), (1024, 512, 2048)])
def test_gemm_nt(mode, tol, M, N, K):
    from neural_sp_amd import ops
    torch.manual_seed(0)
    x = torch.randn(M, K, device=_dev())
    w = torch.randn(N, K, device=_dev()) / math.sqrt(K)
    b = torch.randn(N, device=_dev())
    r = torch.randn(M, N, device=_dev())
    with ops.compute_mode(mode):
        pre = torch.empty(M, N, device=_dev())
        y = ops.linear_fwd(x, w, b, act=ops.ACT['swish'], res=r, alpha=0.5, pre_out=pre)
    ref_pre = x @ w.t() + b
    ref = r + 0.5 * ref_pre * torch.sigmoid(ref_pre)
    assert _rel(pre, ref_pre) < tol, ('pre', _rel(pre, ref_pre))
    assert _rel(y, ref) < tol, ('y', _rel(y, ref))


def test_gemm_layout_asymmetric():
    """A = I against an asymmetric B catches transposed fragments / outputs."""
    from neural_sp_amd import ops
    n = 128
    eye = torch.eye(n, device=_dev())
    w = torch.arange(n * n, device=_dev(), dtype=torch.float32).view(n, n) / (n * n)
    for mode in ('f32', 'bf16'):
        with ops.compute_mode(mode):
            y = ops.linear_fwd(eye, w)  # = w^T
        assert _rel(y, w.t()) < (1e-6 if mode == 'f32' else 5e-3), mode


@pytest.mark.parametrize('mode,tol', [('f32', 2e-5), ('bf16', 2e-2)])
@pytest.mark.parametrize('M,N,K', [(256, 128, 96), (531, 1000, 512), (12800, 512, 2048)])
def test_gemm_dgrad_wgrad(mode, tol, M, N, K):
    from neural_sp_amd import ops
    torch.manual_seed(1)
    x = torch.randn(M, K, device=_dev())
    w = torch.randn(N, K, device=_dev()) / math.sqrt(K)
    dy = torch.randn(M, N, device=_dev())
    with ops.compute_mode(mode):
        dx = ops.linear_dgrad(dy, w)
        dw = ops.linear_wgrad(dy, x)
        db = ops.colsum(dy)
    assert _rel(dx, dy @ w) < tol
    assert _rel(dw, dy.t() @ x) < tol
    assert _rel(db, dy.sum(0)) < 1e-4


def test_gemm_batched_strided():
    """q.k^T and P.v in the [B,T,H,dk] layout used by the attention path."""
    from neural_sp_amd import ops
    torch.manual_seed(2)
    B, T, H, dk = 3, 70, 4, 64
    d = H * dk
    q = torch.randn(B, T, H, dk, device=_dev())
    k = torch.randn(B, T, H, dk, device=_dev())
    v = torch.randn(B, T, H, dk, device=_dev())
    S = torch.empty(B, H, T, T, device=_dev())
    with ops.compute_mode('f32'):
        ops.gemm_raw(T, T, dk, q, d, 1, k, 1, d, S, T, batch=(B, H), a_b=(T * d, dk),
                     b_b=(T * d, dk), c_b=(H * T * T, T * T))
        ref = torch.einsum('bihd,bjhd->bhij', q, k)
        assert _rel(S, ref) < 2e-5
        P = torch.softmax(ref, -1).contiguous()
        O = torch.empty(B, T, H, dk, device=_dev())
        ops.gemm_raw(T, dk, T, P, T, 1, v, d, 1, O, d, batch=(B, H), a_b=(H * T * T, T * T),
                     b_b=(T * d, dk), c_b=(T * d, dk))
        assert _rel(O, torch.einsum('bhij,bjhd->bihd', P, v)) < 2e-5
        # dV = P^T dO   (TN), reduction over queries
        dO = torch.randn(B, T, H, dk, device=_dev())
        dV = torch.empty(B, T, H, dk, device=_dev())
        ops.gemm_raw(T, dk, T, P, 1, T, dO, d, 1, dV, d, batch=(B, H), a_b=(H * T * T, T * T),
                     b_b=(T * d, dk), c_b=(T * d, dk))
        assert _rel(dV, torch.einsum('bhij,bihd->bjhd', P, dO)) < 2e-5


@pytest.mark.parametrize('rows,d', [(5, 256), (1000, 512), (33, 1024), (64, 144)])
def test_layernorm(rows, d):
    from neural_sp_amd import ops
    torch.manual_seed(3)
    x = (torch.randn(rows, d, device=_dev()) * 3 + 1).requires_grad_()
    g = torch.randn(d, device=_dev()).requires_grad_()
    b = torch.randn(d, device=_dev()).requires_grad_()
    dy = torch.randn(rows, d, device=_dev())
    for act in ('none', 'swish'):
        y = ops.layer_norm(x, g, b, 1e-12, act)
        ref = torch.nn.functional.layer_norm(x, (d,), g, b, 1e-12)
        if act == 'swish':
            ref = ref * torch.sigmoid(ref)
        assert _rel(y, ref) < 1e-5
        gx, gg, gb = torch.autograd.grad(y, (x, g, b), dy)
        rx, rg, rb = torch.autograd.grad(ref, (x, g, b), dy)
        assert _rel(gx, rx) < 1e-4 and _rel(gg, rg) < 1e-4 and _rel(gb, rb) < 1e-4


def _ref_attn_probs(S, QP, klens, clamp, scale, causal, lookahead, nl, nc):
    B, H, Tq, Tk = S.shape
    i = torch.arange(Tq, device=S.device)[:, None]
    j = torch.arange(Tk, device=S.device)[None, :]
    e = S.clone()
    if QP is not None:
        rel = (i - j).abs()
        if clamp > 0:
            rel = rel.clamp(max=clamp)
        bd = torch.gather(QP.permute(0, 2, 1, 3), 3, rel[None, None].expand(B, H, Tq, Tk))
        e = e + bd
    e = e * scale
    vis = (j[None] < klens[:, None, None])
    if causal:
        vis = vis & (j <= i + lookahead)[None]
    if nc > 0:
        c0 = (i // nc) * nc
        vis = vis & ((j >= (c0 - nl).clamp(min=0)) & (j < c0 + nc))[None]
    e = e.masked_fill(~vis[:, None], torch.finfo(torch.float32).min)
    return torch.softmax(e, -1), vis


@pytest.mark.parametrize('clamp,causal,nc', [(10, False, 0), (-1, False, 0), (10, True, 0), (4, False, 8)])
def test_attn_softmax(clamp, causal, nc):
    from neural_sp_amd import ops
    torch.manual_seed(4)
    B, H, T = 3, 4, 45
    R = clamp + 1 if clamp > 0 else T
    S0 = torch.randn(B, H, T, T, device=_dev())
    QP0 = torch.randn(B, T, H, R, device=_dev())
    klens = torch.tensor([45, 30, 17], device=_dev(), dtype=torch.int32)
    scale = 0.125
    nl = 8 if nc else 0
    S = S0.clone().requires_grad_()
    QP = QP0.clone().requires_grad_()
    Pref, vis = _ref_attn_probs(S, QP, klens, clamp, scale, causal, 1, nl, nc)
    dP = torch.randn_like(Pref)
    gS, gQP = torch.autograd.grad(Pref, (S, QP), dP)
    mp = ops._mask_params(B, H, T, T, R, clamp, scale, klens, causal, 1, nl, nc)
    P = S0.clone()
    ops.attn_softmax_fwd_raw(P, QP0, mp)
    assert _rel(P, Pref.detach()) < 1e-5
    dS = dP.clone()
    dQP = torch.empty_like(QP0)
    ops.attn_softmax_bwd_raw(P, dS, dQP, mp)
    assert _rel(dS, gS) < 1e-4
    assert _rel(dQP, gQP) < 1e-4


def test_elementwise():
    from neural_sp_amd import ops
    torch.manual_seed(5)
    x = torch.randn(1000, 37, device=_dev())
    z = torch.randn(1000, 37, device=_dev())
    assert _rel(ops.axpby(x, z, 0.5, 2.0), 0.5 * x + 2 * z) < 1e-6
    for name, f in [('relu', torch.relu), ('swish', lambda t: t * torch.sigmoid(t)), ('tanh', torch.tanh),
                    ('gelu_accurate', lambda t: torch.nn.functional.gelu(t)),
                    ('gelu', lambda t: torch.nn.functional.gelu(t, approximate='tanh'))]:
        xr = x.clone().requires_grad_()
        ref = f(xr)
        assert _rel(ops.act_fwd(x, ops.ACT[name]), ref.detach()) < 1e-5, name
        g, = torch.autograd.grad(ref, xr, z)
        assert _rel(ops.dact_mul(z, x, ops.ACT[name]), g) < 1e-4, name


@pytest.mark.parametrize('M,N,K', [(128, 128, 64), (300, 1000, 512), (1024, 512, 2048), (77, 136, 72)])
def test_gemm_bf16_operands_all_layouts(M, N, K):
    """bf16-operand MFMA kernel (KC glds-free loader and RC ds_read_tr loader) vs fp32 matmul of
    the same bf16-rounded operands (products exact => only accumulation order differs)."""
    from neural_sp_amd import ops
    torch.manual_seed(7)
    x = torch.randn(M, K, device=_dev()).bfloat16()
    w = (torch.randn(N, K, device=_dev()) / math.sqrt(K)).bfloat16()
    dy = torch.randn(M, N, device=_dev()).bfloat16()
    b = torch.randn(N, device=_dev())
    with ops.compute_mode('bf16'):
        y = torch.empty(M, N, device=_dev())
        ops.gemm_raw(M, N, K, x, K, 1, w, 1, K, y, N, bias=b)                      # KC x KC
        assert _rel(y, x.float() @ w.float().t() + b) < 1e-5
        yb = torch.empty(M, N, device=_dev(), dtype=torch.bfloat16)
        ops.gemm_raw(M, N, K, x, K, 1, w, 1, K, yb, N, bias=b)                     # bf16 output
        assert _rel(yb.float(), x.float() @ w.float().t() + b) < 1e-2
        dx = torch.empty(M, K, device=_dev())
        ops.gemm_raw(M, K, N, dy, N, 1, w, K, 1, dx, K)                            # KC x RC (dgrad)
        assert _rel(dx, dy.float() @ w.float()) < 1e-5
        dw = torch.zeros(N, K, device=_dev())
        ops.gemm_raw(N, K, M, dy, 1, N, x, K, 1, dw, K, splitk=2)                   # RC x RC (wgrad)
        assert _rel(dw, dy.float().t() @ x.float()) < 1e-5
        if M % 8 == 0:  # the contiguous extent of an RC operand must be a multiple of 8
            xt = x.t().contiguous()                                                  # [K, M]
            y2 = torch.empty(M, N, device=_dev())
            ops.gemm_raw(M, N, K, xt, 1, M, w, 1, K, y2, N)                         # RC x KC
            assert _rel(y2, x.float() @ w.float().t()) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['bf16', 'f32'])
def test_splitk_slabs_with_an_empty_split_are_fully_written(mode):
    """K = 2091 rows cut into 8 splits of ceil(33/8) = 5 k-tiles leaves the last split empty: its
    slab must still be written (zeros), or the slab reduction sums garbage left in the buffer."""
    from neural_sp_amd import ops
    torch.manual_seed(0)
    dev = _dev()
    M, N, K, sk = 40, 64, 2091, 8
    a = torch.randn(K, M, device=dev)
    b = torch.randn(K, N, device=dev)
    with ops.compute_mode(mode):
        aa, bb = (ops.to_bf16(a), ops.to_bf16(b)) if mode == 'bf16' else (a, b)
        part = torch.full((sk, M, N), float('nan'), device=dev)
        ops.gemm_raw(M, N, K, aa, 1, aa.stride(0), bb, bb.stride(0), 1, part, N, splitk=sk, c_ss=M * N)
    assert torch.isfinite(part).all()
    ref = aa.float().t() @ bb.float()
    assert ((part.sum(0) - ref).abs().max() / ref.abs().max()).item() < (2e-3 if mode == 'bf16' else 1e-4)


@pytest.mark.gpu
def test_weight_gradient_gemm_on_the_lds_dma_ring(monkeypatch, shapes=((2048, 640, 512), (1280, 1000, 264), (4096, 512, 2048), (512, 128, 384))):
    """gemm_bf16_rr_ring_kernel<2> -- what a weight gradient runs on when the 8-phase kernel does not take it (forced here by
    NSP_GEMM_RR8P=0; by default: an output extent <= 128): unpadded k-major LDS images written by LDS-DMA, operands formed by
    swizzled transposed reads; split-K slabs; ragged M/N edges.  The last shape (3 output tiles < 24) stays on the
    register-staged kernel.  (The 1- / 3-stage and 256 x 256 variants of rounds 1-3 were removed in round 5.)"""
    from neural_sp_amd import ops
    torch.manual_seed(1)
    dev = _dev()
    monkeypatch.setenv('NSP_GEMM_RR8P', '0')
    for rows, N, K in shapes:
        dy = torch.randn(rows, N, device=dev).bfloat16()
        x = torch.randn(rows, K, device=dev).bfloat16()
        ref = dy.float().t() @ x.float()
        with ops.compute_mode('bf16'):
            dw = ops.linear_wgrad(dy, x)
        assert ((dw - ref).abs().max() / ref.abs().max()).item() < 1e-4, (rows, N, K)


@pytest.mark.gpu
@pytest.mark.parametrize('rows,N,K', [(2048, 640, 512), (1291, 1000, 264), (4096, 512, 2048), (333, 136, 1280), (260, 256, 256), (20000, 2048, 512)])
def test_weight_gradient_gemm_on_the_phase_interleaved_kernel(rows, N, K, monkeypatch):
    """gemm_bf16_kk8p_kernel<.., RR = true> (the default for weight gradients with both output extents > 128): k-major
    sub-images [64 k][64 columns] filled by BUFFER LDS-DMA (k-rows beyond the reduction are out of range and land as zeros:
    1291, 333, 260 rows; an odd number of k-tiles is padded to an even one the same way), transposed reads from inline
    asm, split count from nsp_wgrad_splitk, explicit split counts incl. splits without k-tiles, ragged output edges."""
    from neural_sp_amd import ops, _lib
    torch.manual_seed(rows)
    dev = _dev()
    dy = torch.randn(rows, N, device=dev).bfloat16()
    x = torch.randn(rows, K, device=dev).bfloat16()
    ref = dy.float().t() @ x.float()
    monkeypatch.setenv('NSP_GEMM_RR8P', '1')
    sk = _lib.lib().nsp_wgrad_splitk(N, K, rows)
    assert sk > 0
    with ops.compute_mode('bf16'):
        dw = ops.linear_wgrad(dy, x)
        monkeypatch.setenv('NSP_GEMM_RR8P', '0')
        base = ops.linear_wgrad(dy, x)
    scale = ref.abs().max()
    assert ((base - ref).abs().max() / scale).item() < 1e-4
    assert ((dw - ref).abs().max() / scale).item() < 1e-4
    with ops.compute_mode('bf16'):
        monkeypatch.setenv('NSP_GEMM_RR8P', '1')
        for sk in (1, 3, 8):
            part = torch.full((sk, N, K), float('nan'), device=dev)
            ops.gemm_raw(N, K, rows, dy, 1, N, x, K, 1, part, K, splitk=sk, c_ss=N * K, alpha=0.5)
            assert ((2 * part.sum(0) - ref).abs().max() / scale).item() < 1e-4, sk


@pytest.mark.gpu
@pytest.mark.parametrize('M,N,K,grid', [(1000, 384, 128, 0), (2051, 1000, 512, 8), (4096, 512, 2048, 0), (300, 4352, 256, 8), (300, 2304, 128, 8), (70000, 512, 384, 0),
                                        (4300, 4600, 256, 0)])     # 17 x 18 tiles: the 4 x 8 super-tile order of wide outputs, ragged super-tiles
def test_phase_interleaved_256_tile_gemm(M, N, K, grid, monkeypatch):
    """gemm_bf16_kk8p_kernel (256 x 256 tiles, 8 waves, load units six phases ahead with counted waits, persistent over
    tiles) forced onto small problems: every fused epilogue against torch, ragged M / N, one / two / sixteen
    loop iterations (K = 128 ... 2048), workgroups with 0, 1 and several tiles (70000 x 512: 548 tiles for 256
    workgroups, the unit stream crossing tile boundaries), column-sum slabs, and the dropout mask of the 128 x 128 kernel."""
    from neural_sp_amd import ops
    monkeypatch.setenv('NSP_GEMM_8P', '2')                    # (2 = also where the launcher's rule prefers the 128 x 128 kernels)
    monkeypatch.setenv('NSP_GEMM_8P_MIN_TILES', '1')
    if grid:
        monkeypatch.setenv('NSP_GEMM_8P_GRID', str(grid))     # few workgroups: every one walks several tiles
    torch.manual_seed(M + N)
    dev = _dev()
    a = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.5).bfloat16()
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev)
    src = torch.randn(M, N, device=dev).bfloat16()
    ref = a.float() @ w.float().t()
    with ops.compute_mode('bf16'):
        c = torch.empty(M, N, device=dev)
        ops.gemm_raw(M, N, K, a, K, 1, w, 1, K, c, N)
        assert _rel(c, ref) < 1e-5
        c16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        pre = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        ops.gemm_raw(M, N, K, a, K, 1, w, 1, K, c16, N, bias=bias, act=2, pre_out=pre)
        z = ref + bias
        assert _rel(pre.float(), z) < 1e-2 and _rel(c16.float(), z * torch.sigmoid(z)) < 1e-2
        ops.gemm_raw(M, N, K, a, K, 1, w, 1, K, c, N, bias=bias, res=res, alpha=0.5)
        assert _rel(c, 0.5 * (ref + bias) + res) < 1e-5
        slabs = torch.zeros(((M + 127) // 128 * 4, N), device=dev)
        ops.gemm_raw(M, N, K, a, K, 1, w, 1, K, c16, N, dact_src=src, dact=2, colsum_slabs=slabs)
        s = torch.sigmoid(src.float())
        want = ref * (s * (1 + src.float() * (1 - s)))
        assert _rel(c16.float(), want) < 1e-2
        assert _rel(slabs.sum(0), want.sum(0)) < 1e-4      # (the slabs sum the fp32 values in front of the bf16 rounding)
        # dropout: same mask as the 128 x 128 kernels (pure function of seed / element offset)
        ops.gemm_raw(M, N, K, a, K, 1, w, 1, K, c, N, dropout_p=0.3, seed=11, offset=0)
        monkeypatch.setenv('NSP_GEMM_8P', '0')
        c2 = torch.empty(M, N, device=dev)
        ops.gemm_raw(M, N, K, a, K, 1, w, 1, K, c2, N, dropout_p=0.3, seed=11, offset=0)
        assert torch.equal(c == 0, c2 == 0) and _rel(c, c2) < 1e-6
        assert 0.2 < (c == 0).float().mean().item() < 0.4


@pytest.mark.gpu
def test_layernorm_bf16_only_output_refuses_an_fp32_consumer():
    """Throughput mode: a pre-norm LayerNorm writes only the bf16 image of its output (`_nsp16`); the fp32-typed tensor autograd
    sees is a stride-0 NaN.  A GEMM consumer takes the image; a consumer that would read the fp32 tensor as data (an FFN
    whose inner width is not a multiple of 8 runs on the fp32 path) must RAISE, not train on NaN."""
    from neural_sp_amd import ops
    torch.manual_seed(2)
    d = 64
    x = torch.randn(3, 7, d, device=_dev(), requires_grad=True)
    g, b = torch.ones(d, device=_dev(), requires_grad=True), torch.zeros(d, device=_dev(), requires_grad=True)
    w = (torch.randn(128, d, device=_dev()) * 0.1).requires_grad_()
    w1, b1 = (torch.randn(100, d, device=_dev()) * 0.1).requires_grad_(), torch.zeros(100, device=_dev(), requires_grad=True)
    w2, b2 = (torch.randn(d, 100, device=_dev()) * 0.1).requires_grad_(), torch.zeros(d, device=_dev(), requires_grad=True)
    with ops.compute_mode('bf16'):
        xn, res = ops.layer_norm_split(x, g, b)
        assert getattr(xn, '_nsp_placeholder', False) and xn._nsp16.dtype == torch.bfloat16
        y = ops.linear(xn, w, None)                                 # reads the bf16 image
        ref = torch.nn.functional.linear(torch.nn.functional.layer_norm(x, (d,), g, b, 1e-12), w)
        assert torch.isfinite(y).all() and _rel(y, ref) < 2e-2
        with pytest.raises(RuntimeError, match='fp32 image'):
            ops.ffn(xn, w1, b1, w2, b2, 'relu')                     # inner width 100: fp32 path -> would read the placeholder


@pytest.mark.gpu
def test_residual_gradient_prepared_by_the_layer_norm_backward(monkeypatch):
    """The pre-norm chain  y1 = x + drop(0.5 FFN(LN(x)));  y2 = y1 + drop(Linear(LN(y1)));  out = LN(y2)  in bf16 mode: the
    LayerNorm backward kernels of the second and third norm hand the prepared bf16 gradient image (scale, dropout mask,
    bias-gradient sums) to the FFN's / Linear's backward (ops._PREP; nsp_layernorm_bwd_prep).  Same gradients as with the
    hand-over switched off (the image is bit-identical: same arithmetic on the same dx), both hand-overs made and taken,
    and a gradient that is ACCUMULATED on the way (a second consumer of y1) falls back to grad_prep."""
    from neural_sp_amd import ops
    dev = _dev()
    d, dff, rows = 256, 512, 300
    torch.manual_seed(3)
    P = lambda *sh: torch.nn.Parameter(torch.randn(*sh, device=dev) / sh[-1] ** 0.5)
    w1, b1, w2, b2, wl, bl = P(dff, d), P(dff), P(d, dff), P(d), P(d, d), P(d)
    g1, be1, g2, be2, g3, be3 = [torch.nn.Parameter(torch.rand(d, device=dev) + 0.5) for _ in range(6)]
    x0 = torch.randn(2, rows // 2, d, device=dev)
    dout = torch.randn(2, rows // 2, d, device=dev)

    def run(extra_consumer):
        ops._DROPOUT_STATE['counter'] = 0          # the same dropout streams in every run
        x = x0.clone().requires_grad_()
        with ops.compute_mode('bf16'):
            ops.optimizer_stepped()
            z1, r1 = ops.layer_norm_split(x, g1, be1)
            y1 = ops.ffn(z1, w1, b1, w2, b2, 'swish', p_h=0.1, res=r1, alpha=0.5, p_o=0.1)
            z2, r2 = ops.layer_norm_split(y1, g2, be2)
            y2 = ops.linear(z2, wl, bl, res=r2, dropout_p=0.1)
            out = ops.layer_norm(y2, g3, be3)
            loss = (out * dout).sum()
            if extra_consumer:
                loss = loss + y1.sum() * 1e-3
            return torch.autograd.grad(loss, (x, w1, b1, w2, b2, wl, bl, g1, g2, g3))
    outs = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('NSP_LN_PREP', mode)
        torch.manual_seed(11)
        ops._PREP_STATS.update(made=0, taken=0)
        outs[mode] = run(False)
        if mode == '1':
            assert ops._PREP_STATS == {'made': 2, 'taken': 2}, ops._PREP_STATS
        else:
            assert ops._PREP_STATS == {'made': 0, 'taken': 0}
    for a, b in zip(outs['1'], outs['0']):
        assert _rel(a, b) < 3e-4        # (the bf16 images are identical; bias / LayerNorm-parameter sums are fp32 atomics in both arms)
    monkeypatch.setenv('NSP_LN_PREP', '1')
    ops._PREP_STATS.update(made=0, taken=0)
    run(True)
    assert ops._PREP_STATS['made'] == 2 and ops._PREP_STATS['taken'] == 1, ops._PREP_STATS      # y1's gradient was accumulated: grad_prep


@pytest.mark.gpu
@pytest.mark.parametrize('rows,d', [(333, 512), (70, 256)])
def test_layer_norm_swish_for_a_gemm_only_consumer(rows, d):
    """ops.layer_norm(act='swish', gemm_only=True) in bf16 mode: forward writes only the bf16 image (the fp32-typed result
    is a NaN placeholder that a linear layer never reads), backward recomputes the Swish pre-image from xhat, gamma,
    beta (nsp_layernorm_bwd_recompute) -- against torch through the consuming linear layer."""
    from neural_sp_amd import ops
    torch.manual_seed(rows)
    dev = _dev()
    x = (torch.randn(rows, d, device=dev) * 2 + 0.3).requires_grad_()
    g = torch.nn.Parameter(torch.rand(d, device=dev) + 0.5)
    b = torch.nn.Parameter(torch.randn(d, device=dev) * 0.2)
    w = torch.nn.Parameter(torch.randn(64, d, device=dev) / d ** 0.5)
    dy = torch.randn(rows, 64, device=dev)
    with ops.compute_mode('bf16'):
        h = ops.layer_norm(x, g, b, 1e-12, act='swish', gemm_only=True)
        assert torch.isnan(h).all() and h._nsp16.dtype == torch.bfloat16
        y = ops.linear(h, w)
        got = (y.detach(), h._nsp16.float()) + torch.autograd.grad(y, (x, g, b), dy)
    xr = x.detach().clone().requires_grad_()
    gr = g.detach().clone().requires_grad_()
    br = b.detach().clone().requires_grad_()
    z = torch.nn.functional.layer_norm(xr, (d,), gr, br, 1e-12)
    hr = z * torch.sigmoid(z)
    yr = hr @ w.detach().t()
    ref = (yr.detach(), hr.detach()) + torch.autograd.grad(yr, (xr, gr, br), dy)
    for a, r, name in zip(got, ref, ('y', 'h16', 'dx', 'dgamma', 'dbeta')):
        assert _rel(a, r) < 2e-2, name


@pytest.mark.gpu
@pytest.mark.parametrize('M,C', [(1000, 64), (2051, 512), (37, 256)])
def test_linear_glu_on_the_bf16_image(M, C, monkeypatch):
    """ops.linear_glu in bf16 mode (LinearGLUFn: the pointwise conv's [M, 2C] output only ever exists as the bf16 image
    the GEMM epilogue writes; backward's GLU kernel emits the bf16 gradient operand + bias-gradient slabs) against torch
    on the bf16-rounded operands, and against the unfused linear + glu pair (NSP_LINEAR_GLU=0)."""
    from neural_sp_amd import ops
    torch.manual_seed(M + C)
    dev = _dev()
    x = (torch.randn(M, C, device=dev) * 0.7).requires_grad_()
    w = torch.nn.Parameter(torch.randn(2 * C, C, 1, device=dev) / C ** 0.5)        # Conv1d(kernel 1) layout
    b = torch.nn.Parameter(torch.randn(2 * C, device=dev) * 0.3)
    dy = torch.randn(M, C, device=dev)
    outs = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('NSP_LINEAR_GLU', mode)
        with ops.compute_mode('bf16'):
            y = ops.linear_glu(x, w, b)
            assert y.grad_fn.__class__.__name__.startswith('LinearGLUFn' if mode == '1' else 'GLUFn')
            outs[mode] = (y.detach(),) + torch.autograd.grad(y, (x, w, b), dy)
    xr = x.detach().bfloat16().float().requires_grad_()
    wr = w.detach().bfloat16().float().requires_grad_()
    br = b.detach().clone().requires_grad_()
    h = xr @ wr[:, :, 0].t() + br
    yr = h[:, :C] * torch.sigmoid(h[:, C:])
    ref = (yr.detach(),) + torch.autograd.grad(yr, (xr, wr, br), dy)
    for a, r, name in zip(outs['1'], ref, ('y', 'dx', 'dw', 'db')):
        assert a.shape == r.shape, name
        assert _rel(a, r) < 2e-2, name
    for a, u, name in zip(outs['1'], outs['0'], ('y', 'dx', 'dw', 'db')):
        assert _rel(a, u) < 2e-2, name


@pytest.mark.gpu
@pytest.mark.parametrize('rows,d,offer', [(70, 256, True), (131, 512, False), (33, 64, True)])
def test_layer_norm_pair_at_a_block_boundary(rows, d, offer):
    """Round 6: ops.layer_norm_pair = the last LayerNorm of a Conformer block + the first one of the next block in one
    kernel per direction, against the two ops it replaces (layer_norm, then layer_norm_split): outputs bit-equal, all five
    gradients and -- when the input carries a linear layer's offer -- the prepared bf16 image and its column sums."""
    from neural_sp_amd import ops
    torch.manual_seed(rows + d)
    dev = _dev()
    ga, ba, gb, bb = [torch.nn.Parameter(torch.randn(d, device=dev) * 0.5 + (1.0 if i % 2 == 0 else 0.0)) for i in range(4)]
    x0 = torch.randn(2, rows, d, device=dev)
    dz = torch.randn(2, rows, d, device=dev)
    dres = torch.randn(2, rows, d, device=dev)
    res = {}
    with ops.compute_mode('bf16'):
        for mode in ('pair', 'two'):
            x = x0.clone().requires_grad_()
            xin = ops.scale(x, 1.0)                     # (a non-leaf input that can carry an offer)
            tok = None
            if offer:
                tok = ops._prep_offer(True, 0.5, 0.1, 1234, 77, d)
                ops.tag_prep(xin)
            if mode == 'pair':
                zn, y = ops.layer_norm_pair(xin, ga, ba, 1e-12, gb, bb, 1e-12)
            else:
                y0 = ops.layer_norm(xin, ga, ba, 1e-12)
                zn, y = ops.layer_norm_split(y0, gb, bb, 1e-12)
            z16 = zn._nsp16.clone()
            grads = torch.autograd.grad([zn, y], [x, ga, ba, gb, bb], [dz, dres])
            got = ops._PREP.pop(tok, None) if tok is not None else None
            res[mode] = (y.detach().clone(), z16, grads, got)
    assert torch.equal(res['pair'][0], res['two'][0]) and torch.equal(res['pair'][1], res['two'][1])
    for a, b, name in zip(res['pair'][2], res['two'][2], ('dx', 'dga', 'dba', 'dgb', 'dbb')):
        assert _rel(a, b) < 2e-5, (name, _rel(a, b))
    if offer:
        (_, _, g16a, gsa), (_, _, g16b, gsb) = res['pair'][3], res['two'][3]
        assert _rel(g16a.float(), g16b.float()) < 1e-2 and _rel(gsa, gsb) < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize('B,T,C,k', [(3, 70, 64, 15), (2, 37, 256, 7), (2, 130, 512, 15)])
def test_linear_glu_depthwise_conv_as_one_node(B, T, C, k, monkeypatch):
    """Round 6: depthwise_conv(glu(pointwise_conv(x))) of the Conformer conv module (conformer_convolution.py:107-113) as
    one autograd node (LinearGLUDwconvFn: GLU applied by the depthwise kernels to the bf16 image, the GLU's backward inside
    the conv's data-gradient kernel) against the two-node form it replaces (same arithmetic: the forward is bit-equal) and
    against torch on the bf16-rounded operands."""
    from neural_sp_amd import ops
    torch.manual_seed(B * T + C)
    dev = _dev()
    x = (torch.randn(B, T, C, device=dev) * 0.7).requires_grad_()
    w = torch.nn.Parameter(torch.randn(2 * C, C, 1, device=dev) / C ** 0.5)
    b = torch.nn.Parameter(torch.randn(2 * C, device=dev) * 0.3)
    wd = torch.nn.Parameter(torch.randn(C, 1, k, device=dev) / k ** 0.5)
    bd = torch.nn.Parameter(torch.randn(C, device=dev) * 0.2)
    dy = torch.randn(B, T, C, device=dev)
    outs = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('NSP_GLU_DWCONV', mode)
        with ops.compute_mode('bf16'):
            y = ops.linear_glu_dwconv(x, w, b, wd, bd)
            assert y.grad_fn.__class__.__name__.startswith('LinearGLUDwconvFn' if mode == '1' else 'DepthwiseConv1dFn')
            outs[mode] = (y.detach(),) + torch.autograd.grad(y, (x, w, b, wd, bd), dy)
    assert torch.equal(outs['1'][0], outs['0'][0])
    names = ('y', 'dx', 'dw', 'db', 'dw_dw', 'db_dw')
    for a, u, name in zip(outs['1'], outs['0'], names):
        assert a.shape == u.shape, name
        assert _rel(a, u) < 3e-3, (name, _rel(a, u))
    xr = x.detach().bfloat16().float().requires_grad_()
    wr = w.detach().bfloat16().float().requires_grad_()
    br = b.detach().clone().requires_grad_()
    wdr = wd.detach().clone().requires_grad_()
    bdr = bd.detach().clone().requires_grad_()
    hh = xr @ wr[:, :, 0].t() + br
    h = hh + (hh.bfloat16().float() - hh).detach()          # the GEMM's bf16 image, straight-through
    gl = h[..., :C] * torch.sigmoid(h[..., C:])
    yr = torch.nn.functional.conv1d(gl.transpose(1, 2), wdr, bdr, padding=(k - 1) // 2, groups=C).transpose(1, 2)
    ref = (yr.detach(),) + torch.autograd.grad(yr, (xr, wr, br, wdr, bdr), dy)
    for a, r, name in zip(outs['1'], ref, names):
        assert _rel(a, r) < 2e-2, (name, _rel(a, r))


@pytest.mark.gpu
@pytest.mark.parametrize('M,N,K,grid', [(6144, 256, 512, 16), (6000, 256, 256, 16), (6144, 512, 256, 32), (102400, 512, 2048, 0), (51200, 512, 1536, 0),
                                        (40960, 512, 256, 0)])
def test_phase_interleaved_gemm_stream_k(M, N, K, grid, monkeypatch):
    """The 8-phase kernel's stream-K schedule (round 6): an XCD's workgroups take equal shares of its tiles' loop
    iterations, a tile on a share boundary is started by one workgroup (raw accumulators + flag through the workspace) and
    finished by its right neighbour, which starts from that partial -- the same accumulation order, so every output must
    be BIT-EQUAL to the tile-list schedule's (NSP_GEMM_8P_STREAMK=0), dropout mask included.  Small problems on 16 / 32
    workgroups (3 tiles for 2 workgroups per XCD, ragged M; 6 tiles in two columns for 4), and the step's N = 512 shapes on the full grid
    (800 / 400 / 320 tiles on 256 workgroups; 320 tiles x 2 iterations: every workgroup gives and takes)."""
    from neural_sp_amd import ops
    monkeypatch.setenv('NSP_GEMM_8P', '2')
    monkeypatch.setenv('NSP_GEMM_8P_MIN_TILES', '1')
    if grid:
        monkeypatch.setenv('NSP_GEMM_8P_GRID', str(grid))
    torch.manual_seed(M + K)
    dev = _dev()
    a = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.5).bfloat16()
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev)
    src = torch.randn(M, N, device=dev).bfloat16()
    outs = {}
    with ops.compute_mode('bf16'):
        for mode in ('0', '2'):
            monkeypatch.setenv('NSP_GEMM_8P_STREAMK', mode)
            c = torch.full((M, N), float('nan'), device=dev)
            ops.gemm_raw(M, N, K, a, K, 1, w, 1, K, c, N)
            cr = torch.full((M, N), float('nan'), device=dev)
            ops.gemm_raw(M, N, K, a, K, 1, w, 1, K, cr, N, bias=bias, res=res, alpha=0.5)
            cd = torch.full((M, N), float('nan'), device=dev)
            ops.gemm_raw(M, N, K, a, K, 1, w, 1, K, cd, N, bias=bias, res=res, alpha=0.5, dropout_p=0.2, seed=5, offset=64)
            c16 = torch.full((M, N), float('nan'), device=dev, dtype=torch.bfloat16)      # (bf16 outputs: no stream-K twin, the same kernel in both modes)
            slabs = torch.zeros(((M + 127) // 128 * 4, N), device=dev)
            ops.gemm_raw(M, N, K, a, K, 1, w, 1, K, c16, N, dact_src=src, dact=2, colsum_slabs=slabs)
            for rep in range(3):            # the workspace's flags carry the launch epoch: back-to-back launches must not see stale ones
                c2 = torch.full((M, N), float('nan'), device=dev)
                ops.gemm_raw(M, N, K, a, K, 1, w, 1, K, c2, N)
                assert torch.equal(c2, c), (mode, rep)
            outs[mode] = (c, cr, cd, c16, slabs.sum(0))
    ref = a.float() @ w.float().t()
    assert _rel(outs['2'][0], ref) < 1e-5
    assert _rel(outs['2'][1], 0.5 * (ref + bias) + res) < 1e-5
    for x, y, name in zip(outs['0'], outs['2'], ('plain', 'bias + residual', 'bias + dropout + residual', "act' source", 'slabs')):
        assert torch.equal(x, y), name


@pytest.mark.gpu
def test_phase_interleaved_gemm_with_an_operand_beyond_4_gb(monkeypatch):
    """The RNN-T joint's data gradient reads a [3.6 M, 1024] bf16 operand (7.4 GB): the 8-phase kernel addresses A with a
    scalar base per tile + 32-bit lane offsets.  2.2 M x 1024 (4.5 GB) x a [512, 1024] weight with the tanh' epilogue,
    checked on the row blocks at both ends and either side of the 4-GB line."""
    from neural_sp_amd import ops
    monkeypatch.setenv('NSP_GEMM_8P', '1')
    dev = _dev()
    M, N, K = 2200003, 512, 1024
    torch.manual_seed(5)
    a = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
    for r0 in range(0, M, 262144):
        a[r0:r0 + 262144] = (torch.randn(min(262144, M - r0), K, device=dev) * 0.5).bfloat16()
    w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    h = torch.tanh(torch.randn(M, N, device=dev)).bfloat16()
    out = torch.full((M, N), float('nan'), device=dev, dtype=torch.bfloat16)
    with ops.compute_mode('bf16'):
        ops.gemm_raw(M, N, K, a, K, 1, w, 1, K, out, N, dact_src=h, dact=6)
    line = (1 << 32) // (2 * K)                      # first row whose byte offset needs a 33rd bit
    for r0 in (0, line - 300, line + 5, M - 700):
        rows = slice(r0, r0 + 700)
        want = (a[rows].float() @ w.float().t()) * (1 - h[rows].float() ** 2)
        assert _rel(out[rows].float(), want) < 1e-2, r0
    assert torch.isfinite(out.float()).all()


@pytest.mark.gpu
@pytest.mark.parametrize('M,N,K', [(8200, 1412, 128), (16389, 768, 64)])
def test_specialised_epilogues_on_the_single_stage_kernel(M, N, K):
    """Grids above 640 workgroups run gemm_bf16_kk_glds_kernel<0>, whose epilogue is one of the compile-time
    specialisations of gemm_epilogue_fast (FFN first linear, act' source, residual, plain).  Each against
    torch, with ragged M and a ragged last column tile (the side operand is loaded with clamped addresses
    there); dropout masks must equal those of the run-time epilogue (small grid -> ring kernel) at the same
    element offsets; column-sum slabs must equal the column sums of the stored values."""
    from neural_sp_amd import ops
    torch.manual_seed(M)
    dev = _dev()
    a = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.5).bfloat16()
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev)
    src16 = torch.randn(M, N, device=dev).bfloat16()
    src = src16.float()
    ref = a.float() @ w.float().t()
    small = 128            # rows of the small-grid comparison run (same N / ldc -> same element offsets)
    with ops.compute_mode('bf16'):
        g = lambda C, rows=M, **k: ops.gemm_raw(rows, N, K, a, K, 1, w, 1, K, C, N, **k)
        c32 = torch.empty(M, N, device=dev)
        c16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        pre = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        g(c32)
        assert _rel(c32, ref) < 1e-5
        g(c16, bias=bias)
        assert _rel(c16.float(), ref + bias) < 1e-2
        for act, f in ((2, lambda z: z * torch.sigmoid(z)), (1, torch.relu)):
            g(c16, bias=bias, act=act, pre_out=pre)
            z = ref + bias
            assert _rel(pre.float(), z) < 1e-2 and _rel(c16.float(), f(z)) < 1e-2
            g(c16, bias=bias, act=act, pre_out=pre, dropout_p=0.25, seed=9)
            s16 = torch.empty(small, N, device=dev, dtype=torch.bfloat16)
            sp = torch.empty(small, N, device=dev, dtype=torch.bfloat16)
            g(s16, rows=small, bias=bias, act=act, pre_out=sp, dropout_p=0.25, seed=9)
            assert torch.equal(c16[:small], s16)
            keep = c16.float() != 0
            assert _rel(c16.float(), torch.where(keep, f(z) / 0.75, torch.zeros_like(z))) < 1e-2
        s = torch.sigmoid(src)
        for dact, d in ((2, s * (1 + src * (1 - s))), (1, (src > 0).float()), (6, 1 - src * src)):
            slabs = torch.zeros(((M + 127) // 128 * 4, N), device=dev)
            g(c16, dact_src=src16, dact=dact, colsum_slabs=slabs)
            assert _rel(c16.float(), ref * d) < 1e-2
            assert _rel(slabs.sum(0), (ref * d).sum(0)) < 2e-3
            if dact != 6:
                g(c16, dact_src=src16, dact=dact, dropout_p=0.25, seed=4)
                s16 = torch.empty(small, N, device=dev, dtype=torch.bfloat16)
                g(s16, rows=small, dact_src=src16, dact=dact, dropout_p=0.25, seed=4)
                assert torch.equal(c16[:small], s16)
        g(c32, bias=bias, res=res, alpha=0.5)
        assert _rel(c32, 0.5 * (ref + bias) + res) < 1e-5
        g(c32, bias=bias, res=res, alpha=0.5, dropout_p=0.25, seed=3)
        s32 = torch.empty(small, N, device=dev)
        g(s32, rows=small, bias=bias, res=res, alpha=0.5, dropout_p=0.25, seed=3)
        # (fp32 output: the two kernels may contract mul+add differently -> compare the masks and the values)
        assert torch.equal(c32[:small] == res[:small], s32 == res[:small]) and _rel(c32[:small], s32) < 1e-6
        dropped = (c32 == res)
        assert 0.2 < dropped.float().mean().item() < 0.3
        assert _rel(torch.where(dropped, torch.zeros_like(c32), c32 - res), torch.where(dropped, torch.zeros_like(c32), 0.5 * (ref + bias) / 0.75)) < 1e-5
