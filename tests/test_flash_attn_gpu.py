"""Fused (flash-style) attention kernels vs a plain torch fp32 reference on the same
bf16-rounded inputs: output, LSE and all gradients (q, k, v, position-score table)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device('cuda:0')


def _rel(a, b):
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def _reference(qkv, QP, klens, H, clamp, scale, causal, lookahead, nl, nc):
    B, T, d3 = qkv.shape
    d = d3 // 3
    dk = d // H
    q, k, v = [t.reshape(B, T, H, dk) for t in qkv.split(d, dim=-1)]
    i = torch.arange(T, device=qkv.device)[:, None]
    j = torch.arange(T, device=qkv.device)[None, :]
    e = torch.einsum('bihd,bjhd->bhij', q, k)
    if QP is not None:
        rel = (i - j).abs().clamp(max=clamp)
        e = e + torch.gather(QP.permute(0, 2, 1, 3), 3, rel[None, None].expand(B, H, T, T))
    e = e * scale
    vis = (j[None] < klens[:, None, None])
    if causal:
        vis = vis & (j <= i + lookahead)[None]
    if nc > 0:
        c0 = (i // nc) * nc
        vis = vis & ((j >= (c0 - nl).clamp(min=0)) & (j < c0 + nc))[None]
    e = e.masked_fill(~vis[:, None], torch.finfo(torch.float32).min)
    P = torch.softmax(e, -1)
    O = torch.einsum('bhij,bjhd->bihd', P, v).reshape(B, T, d)
    return O, torch.logsumexp(e, -1)


@pytest.mark.parametrize('T,with_pos,causal,nc', [(130, True, False, 0), (64, False, False, 0),
                                                  (200, True, True, 0), (96, True, False, 16),
                                                  (330, True, False, 0), (330, False, False, 0)])
def test_flash_attention_matches_reference(T, with_pos, causal, nc):
    from neural_sp_amd import ops
    torch.manual_seed(T)
    dev = _dev()
    B, H, dk, clamp = 3, 2, 64, 10
    d = H * dk
    R, Rp = clamp + 1, 16
    qkv = (torch.randn(B, T, 3 * d, device=dev) * 0.5).bfloat16()
    QP = torch.zeros(B, T, H, Rp, device=dev)
    QP[..., :R] = torch.randn(B, T, H, R, device=dev)
    klens = torch.tensor([T, max(1, T - 37), max(1, T // 3)], device=dev, dtype=torch.int32)
    scale = 1.0 / math.sqrt(dk)
    nl = 32 if nc else 0
    q32 = qkv.float().requires_grad_()
    QPr = QP.clone().requires_grad_() if with_pos else None
    Oref, LSEref = _reference(q32, QPr, klens, H, clamp, scale, causal, 1, nl, nc)
    dO = torch.randn_like(Oref).bfloat16()
    grads = torch.autograd.grad(Oref, [q32] + ([QPr] if with_pos else []), dO.float())
    mp = ops._mask_params(B, H, T, T, R if with_pos else 0, clamp if with_pos else -1, scale, klens, causal, 1,
                          nl, nc, r_pitch=Rp if with_pos else 0)
    qkv2 = qkv.view(B * T, 3 * d)
    O, O32, LSE, keep = ops.flash_attn_fwd_raw(qkv2, d, QP if with_pos else None, mp)
    assert _rel(O32.view(B, T, d), Oref.detach()) < 4e-3      # one bf16 probability operand (integer row max: backward reproduces it bit for bit)
    assert _rel(O.float().view(B, T, d), Oref.detach()) < 1.5e-2
    lse = LSE[0] * math.log(2.0) - torch.log(LSE[1])   # LSE[0]: row max in the log2 domain
    ok = LSEref.detach() > -1e30  # fully masked rows: max + log(sum) is not representable in fp32
    assert _rel(lse[ok], LSEref.detach()[ok]) < 1e-3
    dqkv = torch.zeros(B * T, 3 * d, device=dev, dtype=torch.bfloat16)
    dq32, dQP = ops.flash_attn_bwd_raw(qkv2, d, QP if with_pos else None, dO.view(B * T, d), O32, LSE, keep, mp, dqkv)
    g = grads[0].view(B * T, 3 * d)
    assert _rel(dq32, g[:, :d]) < 2e-2, 'dq'
    assert _rel(dqkv[:, d:2 * d].float(), g[:, d:2 * d]) < 2e-2, 'dk'
    assert _rel(dqkv[:, 2 * d:].float(), g[:, 2 * d:]) < 2e-2, 'dv'
    if with_pos:
        assert _rel(dQP[..., :R], grads[1][..., :R]) < 2e-2, 'dQP'


def test_flash_attention_dropout_mask_is_consistent_between_forward_and_backward():
    """Dropout inside the fused kernels: the mask is a pure function of (seed, offset, row, key).
    T = 64 (one tile): with V = I the forward output IS the dropped probability matrix, which
    exposes the mask; output and all gradients must then match a torch reference using that mask.
    T = 150 (several tiles): keep rate, determinism, and the adjoint identity <dO, O(V)> = <dV, V>
    (O is linear in V for a fixed mask), which ties the backward's regenerated mask to the forward's."""
    from neural_sp_amd import ops
    dev = _dev()
    B, H, dk, clamp, pdrop = 2, 2, 64, 10, 0.25
    d = H * dk
    R, Rp = clamp + 1, 16
    scale = 1.0 / math.sqrt(dk)
    # ---- T = 64: full check against an explicit-mask reference
    T = 64
    torch.manual_seed(3)
    qk = (torch.randn(B, T, 2 * d, device=dev) * 0.5).bfloat16()
    QP = torch.zeros(B, T, H, Rp, device=dev)
    QP[..., :R] = torch.randn(B, T, H, R, device=dev)
    klens = torch.tensor([T, T - 9], device=dev, dtype=torch.int32)
    mp = ops._mask_params(B, H, T, T, R, clamp, scale, klens, False, 0, 0, 0, dropout_p=pdrop, seed=1234,
                          offset=5 << 40, r_pitch=Rp)
    eye = torch.eye(T, dk, device=dev).repeat(1, H)[None].expand(B, T, d)            # V[j] = e_j per head
    qkv_eye = torch.cat([qk, eye.bfloat16()], dim=-1).contiguous().view(B * T, 3 * d)
    Pd, _, _, _ = ops.flash_attn_fwd_raw(qkv_eye, d, QP, mp)                               # [B*T, d]: Pdrop[b,i,h,j]
    Pd = Pd.float().view(B, T, H, T).permute(0, 2, 1, 3)                             # [B,H,i,j]
    q32 = qk.float()
    e = torch.einsum('bihd,bjhd->bhij', q32[..., :d].reshape(B, T, H, dk), q32[..., d:].reshape(B, T, H, dk))
    i = torch.arange(T, device=dev)[:, None]
    j = torch.arange(T, device=dev)[None, :]
    rel = (i - j).abs().clamp(max=clamp)
    e = (e + torch.gather(QP.permute(0, 2, 1, 3), 3, rel[None, None].expand(B, H, T, T))) * scale
    vis = (j[None] < klens[:, None, None])[:, None]
    P = torch.softmax(e.masked_fill(~vis, torch.finfo(torch.float32).min), -1)
    mask = (Pd > 0).float()
    live = (P > 1e-4) & vis                     # where the probability cannot have rounded to zero
    rate = 1.0 - mask[live].mean().item()
    assert abs(rate - pdrop) < 0.03, rate
    assert _rel(Pd[live], (P * mask / (1 - pdrop))[live]) < 2e-2
    # gradients with that mask
    v = (torch.randn(B, T, d, device=dev) * 0.5).bfloat16()
    qkv = torch.cat([qk, v], dim=-1).contiguous()
    x32 = qkv.float().requires_grad_()
    QPr = QP.clone().requires_grad_()
    qq, kk, vv = [t.reshape(B, T, H, dk) for t in x32.split(d, dim=-1)]
    e2 = (torch.einsum('bihd,bjhd->bhij', qq, kk)
          + torch.gather(QPr.permute(0, 2, 1, 3), 3, rel[None, None].expand(B, H, T, T))) * scale
    P2 = torch.softmax(e2.masked_fill(~vis, torch.finfo(torch.float32).min), -1) * mask / (1 - pdrop)
    Oref = torch.einsum('bhij,bjhd->bihd', P2, vv).reshape(B, T, d)
    dO = torch.randn_like(Oref).bfloat16()
    gx, gqp = torch.autograd.grad(Oref, [x32, QPr], dO.float())
    O, O32, LSE, keep = ops.flash_attn_fwd_raw(qkv.view(B * T, 3 * d), d, QP, mp)
    assert _rel(O.float().view(B, T, d), Oref.detach()) < 2e-2
    dqkv = torch.zeros(B * T, 3 * d, device=dev, dtype=torch.bfloat16)
    dq32, dQP = ops.flash_attn_bwd_raw(qkv.view(B * T, 3 * d), d, QP, dO.view(B * T, d), O32, LSE, keep, mp, dqkv)
    g = gx.view(B * T, 3 * d)
    assert _rel(dq32, g[:, :d]) < 3e-2, 'dq'
    assert _rel(dqkv[:, d:2 * d].float(), g[:, d:2 * d]) < 3e-2, 'dk'
    assert _rel(dqkv[:, 2 * d:].float(), g[:, 2 * d:]) < 3e-2, 'dv'
    assert _rel(dQP[..., :R], gqp[..., :R]) < 3e-2, 'dQP'
    # ---- T = 150: several key tiles
    T = 150
    qkv = (torch.randn(B, T, 3 * d, device=dev) * 0.5).bfloat16().view(B * T, 3 * d)
    QP = torch.zeros(B, T, H, Rp, device=dev)
    QP[..., :R] = torch.randn(B, T, H, R, device=dev)
    klens = torch.tensor([T, T - 40], device=dev, dtype=torch.int32)
    mp = ops._mask_params(B, H, T, T, R, clamp, scale, klens, False, 0, 0, 0, dropout_p=pdrop, seed=99,
                          offset=7 << 40, r_pitch=Rp)
    mp0 = ops._mask_params(B, H, T, T, R, clamp, scale, klens, False, 0, 0, 0, r_pitch=Rp)
    O1, O1_32, LSE1, keep1 = ops.flash_attn_fwd_raw(qkv, d, QP, mp)
    O2, _, _, keep2 = ops.flash_attn_fwd_raw(qkv, d, QP, mp)
    O0, _, _, keep0 = ops.flash_attn_fwd_raw(qkv, d, QP, mp0)
    assert torch.equal(O1, O2) and keep0 is None      # (keep1 / keep2: words of skipped key tiles are never written)
    assert _rel(O1.float(), O0.float()) > 0.05          # dropout does something
    dO = torch.randn(B * T, d, device=dev).bfloat16()
    dqkv = torch.zeros(B * T, 3 * d, device=dev, dtype=torch.bfloat16)
    ops.flash_attn_bwd_raw(qkv, d, QP, dO, O1_32, LSE1, keep1, mp, dqkv)
    lhs = (dO.float() * O1.float()).sum().item()
    rhs = (dqkv[:, 2 * d:].float() * qkv[:, 2 * d:].float()).sum().item()
    # <dO, O> is a sum of random-sign terms: normalise by the norms, not by the (possibly tiny) sum
    scale_ = (dO.float().norm() * O1.float().norm()).item()
    print('[flash dropout adjoint] <dO,O> %.4f  <dV,V> %.4f  |diff| / (|dO||O|) %.2e' % (lhs, rhs, abs(lhs - rhs) / scale_))
    assert abs(lhs - rhs) / scale_ < 2e-3, (lhs, rhs, scale_)


@pytest.mark.parametrize('T,H,B', [(200, 2, 3), (800, 8, 2)])
def test_flash_attention_backward_keeps_softmax_shift_invariance(T, H, B):
    """q and k with a LARGE component common to all positions (what non-zero LayerNorm / projection
    biases produce).  softmax(q.(k+c)) = softmax(q.k): the true dq is orthogonal to the common key,
    sum_j dk_j = 0 -- any inconsistency between D_i = dO_i.O_i and sum_j P_ij dP_ij leaks D-error x
    common component into dq / dk.  With a single bf16 P in the forward this test sees cosines of
    ~0.5 against the fp32 reference; with the hi+lo pair (fp32-grade O) they are >= 0.999.
    (T=800, H=8 is the first-stage shape of Conformer-L.)"""
    from neural_sp_amd import ops
    torch.manual_seed(T + H)
    dev = _dev()
    dk, clamp = 64, 10
    d = H * dk
    R, Rp = clamp + 1, 16
    qkv = torch.randn(B, T, 3 * d, device=dev) * 0.4
    qkv[..., :d] += torch.randn(1, 1, d, device=dev) * 1.5          # common query component
    qkv[..., d:2 * d] += torch.randn(1, 1, d, device=dev) * 1.5     # common key component
    qkv = qkv.bfloat16()
    QP = torch.zeros(B, T, H, Rp, device=dev)
    QP[..., :R] = torch.randn(B, T, H, R, device=dev)
    klens = torch.tensor([T, max(1, T - 37), max(1, T // 3)][:B], device=dev, dtype=torch.int32)
    scale = 1.0 / math.sqrt(dk)
    q32 = qkv.float().requires_grad_()
    QPr = QP.clone().requires_grad_()
    Oref, _ = _reference(q32, QPr, klens, H, clamp, scale, False, 0, 0, 0)
    dO = torch.randn_like(Oref).bfloat16()
    g, gqp = torch.autograd.grad(Oref, [q32, QPr], dO.float())
    g = g.view(B * T, 3 * d)
    mp = ops._mask_params(B, H, T, T, R, clamp, scale, klens, False, 0, 0, 0, r_pitch=Rp)
    qkv2 = qkv.view(B * T, 3 * d)
    O, O32, LSE, keep = ops.flash_attn_fwd_raw(qkv2, d, QP, mp)
    dqkv = torch.zeros(B * T, 3 * d, device=dev, dtype=torch.bfloat16)
    dq32, dQP = ops.flash_attn_bwd_raw(qkv2, d, QP, dO.view(B * T, d), O32, LSE, keep, mp, dqkv)

    def cos(a, b):
        return torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0).item()
    c = {'dq': cos(dq32, g[:, :d]), 'dk': cos(dqkv[:, d:2 * d].float(), g[:, d:2 * d]),
         'dv': cos(dqkv[:, 2 * d:].float(), g[:, 2 * d:]), 'dQP': cos(dQP[..., :R], gqp[..., :R])}
    # what the weight gradients see: dq / dk contracted with an input that ALSO has a common component
    x = torch.randn(B * T, 96, device=dev) + 2.0
    c['x^T dq'] = cos(x.t() @ dq32, x.t() @ g[:, :d])
    c['x^T dk'] = cos(x.t() @ dqkv[:, d:2 * d].float(), x.t() @ g[:, d:2 * d])
    print('[flash shift-invariance T=%d H=%d] cosines %s' % (T, H, {k: round(v, 5) for k, v in c.items()}))
    assert min(c.values()) > 0.999, c


@pytest.mark.parametrize('T,with_pos,pdrop', [(130, True, 0.0), (200, True, 0.1), (96, False, 0.0)])
def test_flash_backward_finishes_the_query_gradient(T, with_pos, pdrop):
    """Round 6: with dq32 == NULL the dQ kernel writes the FINISHED query gradient -- dS k plus the position term's share
    dQP . pos (relative_multihead_attention.py:188-193) -- as bf16 into column block 0 of dqkv.  It must equal what the
    fp32 hand-over path gives (dq32 from the same kernel + the position product in torch), rounded once."""
    from neural_sp_amd import ops
    torch.manual_seed(T + 7)
    dev = _dev()
    B, H, dk, clamp = 2, 2, 64, 10
    d = H * dk
    R, Rp = clamp + 1, 16
    qkv = (torch.randn(B * T, 3 * d, device=dev) * 0.5).bfloat16()
    QP = None
    pos16 = None
    if with_pos:
        QP = torch.zeros(B, T, H, Rp, device=dev)
        QP[..., :R] = torch.randn(B, T, H, R, device=dev)
        pos16 = torch.zeros(Rp, d, device=dev)
        pos16[:R] = torch.randn(R, d, device=dev) * 0.3
        pos16 = pos16.bfloat16()
    klens = torch.tensor([T, max(1, T - 29)], device=dev, dtype=torch.int32)
    mp = ops._mask_params(B, H, T, T, R if with_pos else 0, clamp if with_pos else -1, 1.0 / math.sqrt(dk), klens, False, 0, 0, 0,
                          dropout_p=pdrop, seed=11, offset=3 << 40, r_pitch=Rp if with_pos else 0)
    O, O32, LSE, keep = ops.flash_attn_fwd_raw(qkv, d, QP, mp)
    dO = torch.randn(B * T, d, device=dev).bfloat16()
    dqkv_a = torch.zeros(B * T, 3 * d, device=dev, dtype=torch.bfloat16)
    dq32, dQP = ops.flash_attn_bwd_raw(qkv, d, QP, dO, O32, LSE, keep, mp, dqkv_a)
    want = dq32.clone()
    if with_pos:
        want += torch.einsum('mhr,rhc->mhc', dQP.view(B * T, H, Rp)[..., :R], pos16[:R].float().view(R, H, dk)).reshape(B * T, d)
    dqkv_b = torch.zeros(B * T, 3 * d, device=dev, dtype=torch.bfloat16)
    none32, dQP_b = ops.flash_attn_bwd_raw(qkv, d, QP, dO, O32, LSE, keep, mp, dqkv_b, pos16=pos16, dq_in_dqkv=True)
    assert none32 is None
    got = dqkv_b[:, :d].float()
    assert _rel(got, want) < 6e-3, _rel(got, want)                 # one bf16 rounding of the fp32 result
    assert torch.equal(dqkv_a[:, d:], dqkv_b[:, d:])               # dK / dV untouched by the output form
    if with_pos:
        assert torch.equal(dQP, dQP_b)
