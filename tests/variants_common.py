"""Shared bodies of tests/test_variants_emu_cpu.py (host emulator) and tests/test_variants_gpu.py (device): the
kernels of csrc/norm_subsample.hip through the real ctypes glue / autograd Functions of neural_sp_amd.ops, against
torch-CPU restatements of the reference modules (encoders/subsampling.py, conformer_convolution.py:58-66,119-124).
`where`: 'emu' = host-emulated kernels on CPU tensors, 'gpu' = libnsp_hip.so on cuda:0."""
import contextlib
import copy
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

SUM_CASES = [(T, f, kind) for kind in ('drop', 'add', 'mean_pool') for (T, f) in [(11, 2), (12, 2), (13, 3), (5, 4), (1, 2)]
             if not (kind == 'add' and f != 2)]            # AddSubsampler asserts factor <= 2
GATHER_CASES = [(12, 2, 2, 0), (13, 3, 3, 0), (13, 3, 2, 1), (14, 3, 3, 1), (7, 3, 4, 1)]
BN_CASES = [(5, 8), (150, 64), (700, 260)]
GN_CASES = [(3, 4), (90, 64), (300, 132)]


def _env(where):
    if where == 'emu':
        from tests.hipemu.shim import emulated_kernels
        return emulated_kernels(), torch.device('cpu')
    return contextlib.nullcontext(), torch.device('cuda', 0)


def _swish(x):
    return x * torch.sigmoid(x)


def _weights(y):
    return (torch.linspace(-1.0, 1.0, y.numel()).view_as(y) * 0.7 + 0.1).to(y.dtype)


def _ref_grads(fn, *inputs):
    y = fn(*inputs)
    (y * _weights(y)).sum().backward()
    return y.detach(), [i.grad.clone() for i in inputs]


def _our_grads(where, fn, *inputs):
    ctx, dev = _env(where)
    with ctx:
        ins = [i.detach().clone().to(dev).requires_grad_(True) for i in inputs]
        y = fn(*ins)
        (y * _weights(y.detach().cpu()).to(dev)).sum().backward()
        return y.detach().cpu(), [i.grad.detach().cpu() for i in ins]


def check_window_sum(where, T, f, kind):
    from neural_sp_amd import ops
    torch.manual_seed(T * 10 + f)
    B, C = 3, 8
    x = torch.randn(B, T, C)

    def ref(x):
        if kind == 'drop':       # subsampling.py:118-121
            return x[:, ::f]
        if kind == 'add':        # subsampling.py:153-167
            xe = x[:, ::2]
            xo = x[:, 1::2] if T % 2 == 0 else torch.cat([x, x.new_zeros(B, 1, C)], dim=1)[:, 1::2]
            return xo + xe
        return F.avg_pool1d(x.transpose(2, 1), f, f, 0, ceil_mode=True).transpose(2, 1)   # :239-240

    To = math.ceil(T / f)
    k = {'drop': 1, 'add': 2, 'mean_pool': f}[kind]
    yr, (gr,) = _ref_grads(ref, x.clone().requires_grad_(True))
    yo, (go,) = _our_grads(where, lambda t: ops.time_window_sum(t, k, f, 0, To, mean=(kind == 'mean_pool')), x)
    assert yo.shape == yr.shape
    torch.testing.assert_close(yo, yr, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(go, gr, rtol=1e-6, atol=1e-6)


def check_window_gather(where, T, k, stride, pad):
    from neural_sp_amd import ops
    torch.manual_seed(T)
    B, C = 2, 12
    x = torch.randn(B, T, C)
    To = (T + 2 * pad - (k - 1) - 1) // stride + 1

    def ref(x):
        xp = F.pad(x, (0, 0, pad, pad))
        return torch.stack([xp[:, to * stride:to * stride + k].reshape(B, k * C) for to in range(To)], dim=1)

    yr, (gr,) = _ref_grads(ref, x.clone().requires_grad_(True))
    yo, (go,) = _our_grads(where, lambda t: ops.time_window_gather(t, k, stride, pad, To), x)
    torch.testing.assert_close(yo, yr, rtol=0, atol=0)
    torch.testing.assert_close(go, gr, rtol=1e-6, atol=1e-6)


def check_conv1d_weight_view(where):
    """The [Co, k*Ci] view of the Conv1d weight used with the gathered windows equals F.conv1d
    (Conv1dSubsampler, subsampling.py:55-94) -- with a torch matmul in place of the GEMM."""
    from neural_sp_amd import ops
    torch.manual_seed(5)
    B, T, C, f = 2, 17, 8, 3
    conv = nn.Conv1d(C, C, 3, stride=f, padding=1)
    x = torch.randn(B, T, C)
    ref = conv(x.transpose(2, 1)).transpose(2, 1)
    To = (T + 2 - 2 - 1) // f + 1
    ctx, dev = _env(where)
    with ctx:
        g = ops.time_window_gather(x.to(dev), 3, f, 1, To).cpu()
    w2 = conv.weight.permute(0, 2, 1).contiguous().view(C, 3 * C)
    torch.testing.assert_close(F.linear(g, w2, conv.bias), ref, rtol=1e-5, atol=1e-5)


def check_batch_norm(where, M, C):
    from neural_sp_amd import ops
    torch.manual_seed(M)
    x = torch.randn(M, C) * 1.7 + torch.linspace(-3, 3, C)       # non-zero channel means
    bn_r = nn.BatchNorm1d(C)
    with torch.no_grad():
        bn_r.weight.uniform_(0.5, 1.5)
        bn_r.bias.uniform_(-0.5, 0.5)
        bn_r.running_mean.uniform_(-1, 1)
        bn_r.running_var.uniform_(0.5, 2)
    ctx, dev = _env(where)
    bn_o = nn.BatchNorm1d(C)
    bn_o.load_state_dict(bn_r.state_dict())
    bn_o.to(dev)
    for training in (True, False):
        bn_r.train(training)
        bn_o.train(training)
        for p in list(bn_r.parameters()) + list(bn_o.parameters()):
            p.grad = None
        xr = x.clone().requires_grad_(True)
        yr = _swish(bn_r(xr.view(M, C, 1))).view(M, C)   # conformer_convolution.py:119-122: [B*T, C, 1], then Swish
        (yr * _weights(yr)).sum().backward()
        with _env(where)[0]:
            xo = x.clone().to(dev).requires_grad_(True)
            yo = ops.batch_norm_act(xo, bn_o, bn_o.training, act='swish')
            (yo * _weights(yr).to(dev)).sum().backward()
        torch.testing.assert_close(yo.detach().cpu(), yr.detach(), rtol=2e-5, atol=2e-5)
        torch.testing.assert_close(xo.grad.cpu(), xr.grad, rtol=1e-4, atol=2e-5)
        torch.testing.assert_close(bn_o.weight.grad.cpu(), bn_r.weight.grad, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(bn_o.bias.grad.cpu(), bn_r.bias.grad, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(bn_o.running_mean.cpu(), bn_r.running_mean, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(bn_o.running_var.cpu(), bn_r.running_var, rtol=1e-5, atol=1e-5)
        assert int(bn_o.num_batches_tracked) == int(bn_r.num_batches_tracked) == 1


def check_group_norm(where, M, C):
    from neural_sp_amd import ops
    torch.manual_seed(C)
    x = torch.randn(M, C)
    gn = nn.GroupNorm(max(1, C // 2), C)                          # conformer_convolution.py:61-63
    with torch.no_grad():
        gn.weight.uniform_(0.5, 1.5)
        gn.bias.uniform_(-0.5, 0.5)

    def ref(x, w, b):
        return _swish(F.group_norm(x.view(M, C, 1), gn.num_groups, w, b, gn.eps)).view(M, C)

    # fp64 reference: with pairs whose two channels nearly coincide rstd approaches 1/sqrt(eps) = 316 and torch's
    # own fp32 GroupNorm backward is off by 7e-4 absolute here (the kernel, differencing the pair first, by 1.5e-5)
    yr, (gx, gw, gb) = _ref_grads(ref, x.double().requires_grad_(True), gn.weight.detach().double().requires_grad_(True),
                                  gn.bias.detach().double().requires_grad_(True))
    yo, (ox, ow, ob) = _our_grads(where, lambda t, w, b: ops.group_norm2_act(t, w, b, gn.eps, act='swish'),
                                  x, gn.weight.detach(), gn.bias.detach())
    torch.testing.assert_close(yo, yr.float(), rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(ox, gx.float(), rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(ow, gw.float(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(ob, gb.float(), rtol=1e-4, atol=1e-4)


def check_flip_mask_and_merge(where):
    """nsp_time_flip_mask: what packing does around a BLSTM layer (encoders/rnn.py:534-541), incl. strided halves"""
    from neural_sp_amd import ops
    torch.manual_seed(0)
    B, T, H = 4, 9, 8
    lens = torch.tensor([9, 5, 1, 7], dtype=torch.int32)
    yf, yr = torch.randn(B, T, H), torch.randn(B, T, H)

    def ref(yf, yr):
        out = torch.zeros(B, T, 2 * H)
        for b in range(B):
            n = int(lens[b])
            out[b, :n, :H] = yf[b, :n]
            out[b, :n, H:] = yr[b, :n].flip(0)
        return out

    yr_, (gf, gr) = _ref_grads(ref, yf.clone().requires_grad_(True), yr.clone().requires_grad_(True))
    ctx, dev = _env(where)
    with ctx:
        ld = lens.to(dev)
        a, b = yf.clone().to(dev).requires_grad_(True), yr.clone().to(dev).requires_grad_(True)
        out = ops.bidir_merge(a, b, ld)
        (out * _weights(yr_).to(dev)).sum().backward()
        x = torch.randn(B, T, H)
        xf = ops.time_flip_mask(x.to(dev), ld, True).cpu()
        xm = ops.time_flip_mask(x.to(dev), ld, False).cpu()
    assert torch.equal(out.detach().cpu(), yr_) and torch.equal(a.grad.cpu(), gf) and torch.equal(b.grad.cpu(), gr)
    for bb in range(B):
        n = int(lens[bb])
        assert torch.equal(xf[bb, :n], x[bb, :n].flip(0)) and torch.equal(xm[bb, :n], x[bb, :n])
        assert xf[bb, n:].abs().sum() == 0 and xm[bb, n:].abs().sum() == 0


def check_im2col_conv(where):
    """nsp_im2col3x3 + the [Co, Ci*9] weight view == F.conv2d(padding=1) for 3 input channels (conv.py:167-175,303-307)"""
    from neural_sp_amd import ops
    torch.manual_seed(12)
    B, T, Fq, Ci, Co = 2, 11, 9, 3, 32
    conv = nn.Conv2d(Ci, Co, 3, padding=1)
    x = torch.randn(B, T, Fq, Ci)
    ref = torch.relu(conv(x.permute(0, 3, 1, 2))).permute(0, 2, 3, 1)
    # reference gradients first: Module.to(cuda) replaces the Parameter objects, so the device side
    # runs on its own copy of the module
    rw, rb = torch.autograd.grad((ref * _weights(ref)).sum(), [conv.weight, conv.bias])
    ctx, dev = _env(where)
    with ctx, ops.compute_mode('f32'):
        dconv = copy.deepcopy(conv).to(dev)
        y = ops.conv3x3_relu(x.to(dev), dconv.weight, dconv.bias)
        w = _weights(ref).to(dev)
        gw, gb = torch.autograd.grad((y * w).sum(), [dconv.weight, dconv.bias])
    torch.testing.assert_close(y.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(gw.cpu(), rw, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(gb.cpu(), rb, rtol=1e-3, atol=1e-4)


# ---- MoChA / MMA training scans (csrc/mocha.hip) against the torch-op restatement of hma_train.py:12-67 and
# mocha_train.py:13-83 (the form the reference itself uses: clamp / log / cumsum / exp, ones-filter conv1d)
def _ref_mono_alpha(e, aw_prev, eps, no_denom, lam):
    p = (1 - lam) * torch.sigmoid(e)
    x = torch.log(torch.clamp(1 - p, min=eps, max=1.0))
    excl = torch.cumsum(torch.cat([x.new_zeros(x.shape[:-1] + (1,)), x[..., :-1]], dim=-1), dim=-1)
    c = torch.exp(excl)
    den = 1 if no_denom else torch.clamp(c, min=eps, max=1.0)
    return p * c * torch.cumsum(aw_prev / den, dim=-1), p


def _ref_moving_sum(x, back, forward):
    shape = x.shape
    y = F.conv1d(F.pad(x.reshape(-1, 1, shape[-1]), [back, forward]), x.new_ones(1, 1, back + forward + 1))
    return y.view(shape[:-1] + (y.shape[-1],))


def _ref_chunk_beta(u, alpha, w, sf):
    klen = u.shape[-1]
    u = u - torch.max(u, dim=-1, keepdim=True)[0]
    ex = torch.clamp(torch.exp(u), min=1e-5)
    if w == -1:
        den = torch.cumsum(ex, dim=-1)
        return ex * _ref_moving_sum(alpha * sf / den, back=0, forward=klen - 1)
    den = _ref_moving_sum(ex, back=w - 1, forward=0)
    return ex * _ref_moving_sum(alpha * sf / den, back=0, forward=w - 1)


MOCHA_CASES = [(3, 50, 4, False, 0.0), (2, 200, 8, False, 0.1), (5, 1, 2, True, 0.0), (4, 129, -1, False, 0.0),
               (2, 333, 16, True, 0.2)]


def check_mocha_scans(where, rows, klen, w, no_denom, lam):
    from neural_sp_amd import ops
    torch.manual_seed(rows * 1000 + klen)
    NEG = float(torch.finfo(torch.float32).min)
    e = torch.randn(rows, klen) * 2 - 1.0
    u = torch.randn(rows, klen) * 3
    if klen > 8:                                     # padded frames: masked energies, as the decoders pass them
        e[0, klen - 5:] = NEG
        u[0, klen - 5:] = NEG
        u[1, 3] = u[1].max() + 0.5                   # a unique maximum away from index 0
    aw = torch.softmax(torch.randn(rows, klen), -1)
    if rows > 1:
        aw[1] = 0
        aw[1, 0] = 1.0                               # the decoders' first step: [1, 0, 0, ...]
    eps = 1e-6
    er, awr, ur = e.clone().requires_grad_(True), aw.clone().requires_grad_(True), u.clone().requires_grad_(True)
    a_ref, p_ref = _ref_mono_alpha(er, awr, eps, no_denom, lam)
    b_ref = _ref_chunk_beta(ur, a_ref, w, 1.3)
    wa, wb = _weights(a_ref.detach()), _weights(b_ref.detach()) * 0.7
    (a_ref * wa).sum().backward(retain_graph=True)
    ge_a, gaw_a = er.grad.clone(), awr.grad.clone()
    er.grad = None; awr.grad = None
    (b_ref * wb).sum().backward()
    ctx, dev = _env(where)
    with ctx:
        ed, awd, ud = [t.clone().to(dev).requires_grad_(True) for t in (e, aw, u)]
        a, p = ops.mono_alpha(ed, awd, eps, no_denom, lam)
        (a * wa.to(dev)).sum().backward(retain_graph=True)
        ge, gaw = ed.grad.clone().cpu(), awd.grad.clone().cpu()
        ed.grad = None; awd.grad = None
        b = ops.chunk_beta(ud, a, w, 1.3)
        (b * wb.to(dev)).sum().backward()
        got = [t.detach().cpu() for t in (a, p, b, ed.grad, awd.grad, ud.grad)]

    def close(x, y, what, tol=2e-5):
        scale = max(y.abs().max().item(), 1e-30)
        err = (x - y).abs().max().item() / scale
        assert err < tol, (what, err)
    close(got[0], a_ref.detach(), 'alpha'); close(got[1], p_ref.detach(), 'p_choose'); close(ge, ge_a, 'd e (alpha only)', 2e-4)
    close(gaw, gaw_a, 'd aw_prev (alpha only)', 2e-4)
    close(got[2], b_ref.detach(), 'beta', 1e-4)
    close(got[3], er.grad, 'd e through beta', 5e-4); close(got[4], awr.grad, 'd aw_prev through beta', 5e-4)
    close(got[5], ur.grad, 'd u', 5e-4)


# ---- decoder-step kernels (csrc/decoder_step.hip) against their torch-op forms
def check_decoder_step_kernels(where, act='tanh', with_loc=True):
    from neural_sp_amd import ops
    torch.manual_seed(5)
    B, T, A, H = 3, 37, 70, 24
    K, Q, v = torch.randn(B, T, A), torch.randn(B, 1, A), torch.randn(1, A) * 0.3
    C = torch.randn(B, T, A) * 0.5 if with_loc else None
    mask = torch.ones(B, 1, T, dtype=torch.bool)
    mask[1, 0, 20:] = False
    mask[2, 0, 5:] = False
    gates, cp = torch.randn(B, 4 * H), torch.randn(B, H)
    f = torch.tanh if act == 'tanh' else torch.relu
    leaves = [t.clone().requires_grad_(True) for t in (K, Q, v, gates, cp)] + ([C.clone().requires_grad_(True)] if with_loc else [])
    Kr, Qr, vr, gr, cr = leaves[:5]
    Cr = leaves[5] if with_loc else None
    e_ref = (f(Kr + Qr + (Cr if with_loc else 0)) * vr.view(1, 1, -1)).sum(-1).unsqueeze(1)
    aw_ref = torch.softmax(e_ref.masked_fill(mask == 0, float(torch.finfo(torch.float32).min)) * 1.7, dim=-1)
    gi, gf, gg, go = gr.chunk(4, dim=1)
    c_ref = torch.sigmoid(gf) * cr + torch.sigmoid(gi) * torch.tanh(gg)
    h_ref = torch.sigmoid(go) * torch.tanh(c_ref)
    w1, w2, w3 = _weights(aw_ref.detach()), _weights(h_ref.detach()), _weights(c_ref.detach()) * 0.5
    ((aw_ref * w1).sum() + (h_ref * w2).sum() + (c_ref * w3).sum()).backward()
    ctx, dev = _env(where)
    with ctx:
        dl = [t.clone().to(dev).requires_grad_(True) for t in ([K, Q, v, gates, cp] + ([C] if with_loc else []))]
        e = ops.add_energy(dl[0], dl[1], dl[2], act, dl[5] if with_loc else None).unsqueeze(1)
        aw = ops.row_softmax(e, mask.to(dev), 1.7)
        h, c = ops.lstm_cell(dl[3], dl[4])
        ((aw * w1.to(dev)).sum() + (h * w2.to(dev)).sum() + (c * w3.to(dev)).sum()).backward()
        outs = [t.detach().cpu() for t in (e, aw, h, c)]
        grads = [t.grad.detach().cpu() for t in dl]
    for got, ref, what in zip(outs, (e_ref, aw_ref, h_ref, c_ref), ('e', 'aw', 'h', 'c')):
        torch.testing.assert_close(got, ref.detach(), rtol=2e-5, atol=2e-6, msg=lambda m, w=what: w + ': ' + m)
    for got, leaf, what in zip(grads, leaves, ('dK', 'dQ', 'dv', 'dgates', 'dc_prev', 'dC')):
        scale = leaf.grad.abs().max().item()
        assert ((got - leaf.grad).abs().max().item() / scale) < 2e-5, what
    assert aw[1, 0, 20:].abs().sum() == 0 and grads[0][2, 5:].abs().sum() == 0     # masked frames: no weight, no gradient


# ---- one-launch refresh of all bf16 weight shadows (ops.refresh_weight_shadows / nsp_shadow_refresh)
def check_shadow_refresh(where):
    from neural_sp_amd import ops
    torch.manual_seed(3)
    ctx, dev = _env(where)
    with ctx, ops.compute_mode('bf16'):
        P = lambda *s: nn.Parameter(torch.randn(*s, device=dev))
        w1, w2, w3 = P(70, 24), P(130, 24), P(33, 24)           # stacked projections (ragged row counts)
        wq = P(100, 72)                                           # plain + transposed (N % 64 != 0: zero padded)
        wo = P(43, 40)                                            # rows padded to 64
        a, b, a2 = P(96, 20), P(96, 24), P(96, 24)               # LSTM-style concatenations (20 % 8 != 0: padded gap)

        def get():
            return dict(
                plain=ops.weight_bf16(wq), t=ops._weight_t_shadow(wq, True), stack=ops._stacked_weight_bf16([w1, w2, w3]),
                stackt=ops._stacked_weight_t_bf16([w1, w2, w3]), rowpad=ops._rows_padded_bf16(wo, 64),
                cat=ops._cat_cached(b, '_nsp_lstm_cat', (a, b), lambda: torch.cat([ops.weight_bf16(a), ops.weight_bf16(b)], dim=1).contiguous(),
                                    lambda t: [(a, 0, 0, a.shape[0], a.shape[1], False), (b, 0, ops._r8(a.shape[1]), b.shape[0], b.shape[1], False)]),
                catT=ops._cat_cached(b, '_nsp_lstm_catT', (a2, b), lambda: torch.cat(
                    [ops._weight_t_shadow(a2, True)[:, :96], ops._weight_t_shadow(b, True)[:, :96]], dim=1).contiguous(),
                    lambda t: [(a2, 0, 0, a2.shape[1], a2.shape[0], True), (b, 0, 96, b.shape[1], b.shape[0], True)]))
        first = get()
        ref_first = {k: v.clone() for k, v in first.items()}
        with torch.no_grad():
            for p in (w1, w2, w3, wq, wo, a, b, a2):
                p.add_(torch.randn_like(p))                       # bumps ._version like an optimizer step
        ops.refresh_weight_shadows()
        second = get()
        for k in first:
            assert second[k] is first[k], k + ': the getter rebuilt a shadow the refresh should have updated in place'
            assert not torch.equal(second[k], ref_first[k]), k
        # against shadows built from scratch
        ops.invalidate_weight_shadows(nn.ParameterList([w1, w2, w3, wq, wo, a, b, a2]))
        third = get()
        for k in first:
            assert third[k] is not first[k]
            assert torch.equal(third[k].cpu(), second[k].cpu()), k


def check_weights_changed_without_version_bump(where):
    """A fused optimizer (torch.optim.Adam(fused=True) on the device) rewrites the parameters WITHOUT touching
    Parameter._version; every bf16 / transposed weight copy must follow all the same.  Emulated here with .data writes:
    the second forward must see the new weights (= a freshly built model holding them)."""
    import argparse
    from neural_sp_amd import ops
    from neural_sp_amd.configs import conformer_rnnt_args, synthetic_batch
    from neural_sp_amd.speech2text import Speech2Text
    args = conformer_rnnt_args('XS', n_layers=2, vocab=40, ctc_weight=0.3, ctc_fc_list='32', dropout=0.0)
    batch = synthetic_batch(B=2, t_range=(40, 48), u_range=(2, 4), vocab=40, seed=0)
    ctx, dev = _env(where)
    torch.manual_seed(0)
    model = Speech2Text(args)
    with ctx, ops.compute_mode('bf16'):
        if where == 'gpu':
            model = model.to(dev)
        l0 = model(batch, task='all')[0].item()
        versions = [p._version for p in model.parameters()]
        with torch.no_grad():
            for p in model.parameters():
                p.data.mul_(1.25)                     # values change, version counters do not
        assert versions == [p._version for p in model.parameters()]
        l1 = model(batch, task='all')[0].item()
        fresh = Speech2Text(args)
        fresh.load_state_dict(model.state_dict())
        if where == 'gpu':
            fresh = fresh.to(dev)
        l2 = fresh(batch, task='all')[0].item()
        model.eval()
        with torch.no_grad():
            for p in model.parameters():
                p.data.mul_(0.8)                      # back to the original weights, evaluation mode
        l3 = model(batch, task='all', is_eval=True)[0].item()
    assert abs(l1 - l0) / abs(l0) > 1e-3, 'the modified weights did not reach the kernels'
    assert abs(l1 - l2) / abs(l2) < 1e-5, (l1, l2)
    assert abs(l3 - l0) / abs(l0) < 2e-3, (l3, l0)    # (0.8 * 1.25 = 1 up to rounding)
