"""torch-CPU stand-ins for neural_sp_amd.ops -- TEST INFRASTRUCTURE ONLY (`-m "not gpu"` host-logic tests).

`with host_logic_on_cpu():` lets the *host side* of the product (neural_sp_amd.encoders / modules / decoders /
speech2text: module wiring, length arithmetic, reshapes, parameter names, state_dict loading) run on CPU
tensors so that it can be checked against the reference-generated fixtures without a GPU:
  * ops whose kernels the host emulator can build (csrc/norm_subsample.hip) run THROUGH the real ctypes glue and
    autograd Functions on the emulated kernels (tests/hipemu);
  * every other op used by the encoder + CTC path is replaced by the plain-torch expression of what its C-ABI
    entry point is documented to compute (include/nsp_hip.h).  These stand-ins say nothing about the HIP kernels
    themselves -- tests/test_*_gpu.py do that -- they only make the surrounding Python executable here.
Nothing under neural_sp_amd/ imports this module; the product has no CPU path.
"""
import contextlib
import math

import torch
import torch.nn.functional as F

NEG_INF32 = float(torch.finfo(torch.float32).min)


def _act(x, act):
    from neural_sp_amd import ops
    if not isinstance(act, int):
        act = ops.ACT[act]
    if act == 0:
        return x
    if act == 1:
        return torch.relu(x)
    if act == 2:
        return x * torch.sigmoid(x)
    if act == 3:
        return torch.tanh(x)
    if act == 4:
        return F.gelu(x)
    if act == 5:
        return F.gelu(x, approximate='tanh')
    raise NotImplementedError(act)


def _linear(x, weight, bias=None, act='none', res=None, alpha=1.0, dropout_p=0.0):
    assert dropout_p == 0.0, 'the CPU stand-ins are for dropout-free parity runs'
    y = _act(F.linear(x, weight.reshape(weight.shape[0], -1), bias), act) * alpha
    return y if res is None else res + y


def _ffn(x, w1, b1, w2, b2, act, p_h=0.0, res=None, alpha=1.0, p_o=0.0):
    assert p_h == 0.0 and p_o == 0.0
    return _linear(_linear(x, w1, b1, act), w2, b2, res=res, alpha=alpha)


def _layer_norm(x, gamma, beta, eps=1e-12, act='none', gemm_only=False):
    return _act(F.layer_norm(x, (x.shape[-1],), gamma, beta, eps), act)


def _layer_norm_split(x, gamma, beta, eps=1e-12):
    return F.layer_norm(x, (x.shape[-1],), gamma, beta, eps), x


class _AttentionFn(object):
    """ops.AttentionFn: softmax((q_ac k^T + shift(q_bd pos^T)) / sqrt(d_k)) v with the in-kernel mask predicate
    (nsp_attn_mask_params: klens, causal + lookahead, chunk_nl / chunk_nc)."""

    @staticmethod
    def apply(q_ac, q_bd, k, v, pos, klens, cfg):
        B, Tq, H, dk = q_ac.shape
        Tk = k.shape[1]
        e = torch.einsum('bihd,bjhd->bijh', q_ac, k)
        i = torch.arange(Tq)[:, None]
        j = torch.arange(Tk)[None, :]
        if pos is not None:
            R = pos.shape[0]
            bd = torch.einsum('bihd,rhd->birh', q_bd if q_bd is not None else q_ac, pos)
            idx = (j - i).abs().clamp(max=R - 1)
            e = e + torch.gather(bd, 2, idx[None, :, :, None].expand(B, Tq, Tk, H))
        e = e / math.sqrt(dk)
        vis = torch.ones(B, Tq, Tk, dtype=torch.bool)
        if klens is not None:
            vis &= j[None] < klens.long()[:, None, None]
        if cfg.get('causal'):
            vis &= (j <= i + int(cfg.get('lookahead', 0)))[None]
        nc, nl = int(cfg.get('chunk_nc', 0)), int(cfg.get('chunk_nl', 0))
        if nc > 0:
            c0 = (i // nc) * nc
            vis &= ((j >= (c0 - nl).clamp(min=0)) & (j < c0 + nc))[None]
        e = e.masked_fill(~vis[:, :, :, None], NEG_INF32)
        aw = torch.softmax(e, dim=2)
        cv = torch.einsum('bijh,bjhd->bihd', aw, v).reshape(B, Tq, H * dk)
        return cv, aw.permute(0, 3, 1, 2)


def _depthwise_conv1d(x, weight, bias, causal=False):
    C, _, k = weight.shape
    pad = (k - 1) if causal else (k - 1) // 2
    h = F.conv1d(x.transpose(2, 1), weight, bias, padding=pad, groups=C)
    if causal:
        h = h[:, :, :-pad]
    return h.transpose(2, 1)


def _conv3x3_relu(x_cl, weight, bias):
    return torch.relu(F.conv2d(x_cl.permute(0, 3, 1, 2), weight, bias, padding=1)).permute(0, 2, 3, 1)


def _maxpool2d(x_cl, pt, pf, to_btcf=False):
    y = F.max_pool2d(x_cl.permute(0, 3, 1, 2), (pt, pf), (pt, pf), ceil_mode=True)   # [B,C,T',F']
    return y.permute(0, 2, 1, 3) if to_btcf else y.permute(0, 2, 3, 1)


def _scale_add_bcast(x, z, alpha):
    return (alpha * x.reshape(-1, z.numel()) + z.reshape(1, -1)).view_as(x)


def _ctc_loss(logits, labels, elens, ylens, lsm_prob=0.0, sum_elens=1, blank=0):
    """nsp_ctc_loss_fwd_bwd (+ nsp_ctc_kldiv_fwd_bwd): ctc.py:139-150, criterion.py:110-127."""
    B, T, V = logits.shape
    lp = torch.log_softmax(logits, dim=-1)
    tgt = torch.cat([labels[b, :int(ylens[b])] for b in range(B)]).long()
    nll = F.ctc_loss(lp.transpose(0, 1), tgt, elens.long(), ylens.long(), blank=blank, reduction='none',
                     zero_infinity=True)
    loss = nll.sum() / B
    if lsm_prob > 0:
        kl = logits.new_zeros(())
        for b in range(B):
            n = int(elens[b])
            p = torch.softmax(logits[b, :n], -1)
            kl = kl + (p * (lp[b, :n] - math.log(1.0 / (V - 1)))).sum()
        loss = loss * (1 - lsm_prob) + kl / sum_elens * lsm_prob
    return loss.view(1), nll


def _lstm_state(x, w_ih, w_hh, b_ih, b_hh, h0, c0):
    """ops.lstm_state: the same layer started from (h0, c0) -> (y, h_n, c_n)"""
    y, h, c = _lstm(x, w_ih, w_hh, b_ih, b_hh, h0, c0, True)
    return y, h, c


def _lstm(x, w_ih, w_hh, b_ih, b_hh, h0=None, c0=None, return_state=False):
    """ops.lstm: one nn.LSTM layer, batch_first, zero initial state, gate order i,f,g,o (nsp_lstm_fwd)."""
    B, L, _ = x.shape
    H = w_hh.shape[1]
    h = x.new_zeros(B, H) if h0 is None else h0
    c = x.new_zeros(B, H) if c0 is None else c0
    gi = F.linear(x, w_ih, b_ih + b_hh)
    outs = []
    for t in range(L):
        i, f, g, o = (gi[:, t] + F.linear(h, w_hh)).chunk(4, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h = torch.sigmoid(o) * torch.tanh(c)
        outs.append(h)
    y = torch.stack(outs, dim=1)
    return (y, h, c) if return_state else y


def _xe_lsm_loss(logits, ys_int32, lsm_prob, ignore_index, bs):
    """nsp_xe_lsm_fwd_bwd (criterion.py:45-86, torch_utils.py:129-145) -> (loss [1], loss_rows, correct)"""
    V = logits.shape[-1]
    lg, yo = logits.reshape(-1, V), ys_int32.reshape(-1).long()
    mask = yo == ignore_index
    lp = torch.log_softmax(lg, dim=-1)
    tgt = torch.full_like(lp, lsm_prob / (V - 1))
    tgt.scatter_(1, yo.masked_fill(mask, 0).unsqueeze(1), 1 - lsm_prob)
    rows = -(tgt * lp).sum(1).masked_fill(mask, 0)
    correct = ((lg.argmax(1) == yo) & ~mask).int()
    return (rows.sum() / bs).view(1), rows.detach(), correct


def _h2d_packed(arrays, device):
    import numpy as np
    return torch.from_numpy(np.concatenate([np.asarray(a, dtype=np.float32).reshape(-1) for a in arrays]))


def _pad_batch(packed, offsets, lens, B, Tmax, Fdim, pad_value=0.0):
    out = torch.full((B, Tmax, Fdim), pad_value, dtype=torch.float32)
    for b in range(B):
        n = int(lens[b])
        out[b, :n] = packed[int(offsets[b]):int(offsets[b]) + n * Fdim].view(n, Fdim)
    return out


def _xl_pos_table(inv_freq, L):
    pos = torch.arange(-1, -L - 1, -1.0, dtype=torch.float32)
    s = torch.einsum('i,j->ij', pos, inv_freq.float())
    return torch.cat([s.sin(), s.cos()], dim=-1)


FORCED_ALIGN = {'result': None}


def _ctc_forced_align(logits, labels, elens, ylens, blank=0):
    """nsp_ctc_forced_align is pinned bit-exact on the device (tests/golden/ctc_align.pt); the host-logic tests hand
    in the trigger points the REFERENCE aligner produced for the fixture (stored in it) instead of restating it"""
    assert FORCED_ALIGN['result'] is not None, 'set cpu_ops_shim.FORCED_ALIGN["result"] to the fixture\'s trigger points'
    return FORCED_ALIGN['result'].to(torch.int32)


@contextlib.contextmanager
def host_logic_on_cpu(real_kernels=False, real_conv=True, mode='f32'):
    """real_kernels=True: every op runs the real .hip kernels on the emulator -- the conv front-end and its 2-D pooling,
    GEMMs and the LSTM step kernels (fp32 MFMA emulated as wave collectives), attention soft-max, LayerNorm, CTC, XE,
    depthwise conv, GLU, pooling, dropout, ... -- only the pinned-memory H2D staging (which needs a device) is replaced."""
    from neural_sp_amd import ops
    from tests.hipemu.shim import emulated_kernels
    fakes = dict(
        linear=_linear, ffn=_ffn, layer_norm=_layer_norm, layer_norm_split=_layer_norm_split,
        AttentionFn=_AttentionFn, glu=lambda x: F.glu(x, dim=-1), depthwise_conv1d=_depthwise_conv1d,
        maxpool1d_time=lambda x, f: F.max_pool1d(x.transpose(2, 1), f, f, ceil_mode=True).transpose(2, 1),
        conv3x3_relu=_conv3x3_relu, maxpool2d=_maxpool2d, scale=lambda x, a: x * a,
        dropout=lambda x, p, training: x if (p == 0 or not training) else (_ for _ in ()).throw(AssertionError('dropout')),
        add=lambda x, z, alpha=1.0, beta=1.0: alpha * x + beta * z, scale_add_bcast=_scale_add_bcast,
        xl_pos_table=_xl_pos_table, ctc_loss=_ctc_loss, h2d_packed=_h2d_packed, pad_batch=_pad_batch, lstm=_lstm, lstm_state=_lstm_state,
        xe_lsm_loss=_xe_lsm_loss, argmax_rows=lambda x2d: x2d.argmax(-1).int(), ctc_forced_align=_ctc_forced_align,
    )
    if real_kernels:
        # pinned staging needs a device; everything else is real.  real_conv=False keeps the (slow to emulate: thousands of
        # fp32 MFMAs per tile) conv front-end on its torch stand-in
        fakes = {k: fakes[k] for k in (('h2d_packed',) if real_conv else ('h2d_packed', 'conv3x3_relu', 'maxpool2d'))}
    saved = {k: getattr(ops, k) for k in fakes}
    saved_mode = ops.get_compute_mode()
    for k, v in fakes.items():
        setattr(ops, k, v)
    assert mode == 'f32' or real_kernels, 'the torch stand-ins are fp32 only'
    ops.set_compute_mode(mode)
    try:
        with emulated_kernels():
            yield
    finally:
        for k, v in saved.items():
            setattr(ops, k, v)
        ops.set_compute_mode(saved_mode)
