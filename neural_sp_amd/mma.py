"""Monotonic multi-head attention (MMA) -- the source attention of the reference's streaming Transformer decoders
(`transformer_dec_attn_type: mocha`, 14 recipes: modules/mocha/mocha.py:20-311 with atype='scaled_dot',
monotonic_energy.py, chunk_energy.py, hma_train.py:12-67, mocha_train.py:13-83, headdrop.py), training ('parallel')
mode over all target positions at once.

H_ma monotonic heads each select frames with the expected-alignment recurrence of hard monotonic attention
(alpha_i from alpha_{i-1}: a loop over the L target positions, vectorised over batch, heads and frames); every
monotonic head feeds H_ca chunkwise heads that soft-max over the w frames ending at the selected one (moving sums as
1-d convolutions with ones); the H_ma * H_ca context vectors are concatenated and projected (w_value / w_out).
The four projections are the MFMA GEMMs of ops.linear, the energies and the context batched nsp_gemm calls
(ops.head_scores / ops.head_context); the alpha recurrence and the chunkwise beta are the scan kernels of csrc/mocha.hip
(ops.mono_alpha once per target position, ops.chunk_beta once for all positions).  Parameter names follow the reference
(`src_attn.monotonic_energy.{w_key,w_query,r}`, `src_attn.chunk_energy.{w_key,w_query}`, `src_attn.{w_value,w_out}`).

Not built: test-time (hard) attention -- decoding an MMA model raises; additive energies, the 1-d conv on the keys,
DeCoT for MMA.
"""
import math
import random

import torch
import torch.nn as nn

from neural_sp_amd import ops

NEG_INF = float(torch.finfo(torch.float32).min)


class _ScaledDotEnergy(nn.Module):
    """monotonic_energy.py:19-157 / chunk_energy.py:17-123 with atype='scaled_dot': e = q k^T / sqrt(adim) (+ r)"""

    def __init__(self, kdim, qdim, adim, n_heads, bias, param_init, init_r=None):
        super().__init__()
        assert adim % n_heads == 0
        self.d_k = adim // n_heads
        self.n_heads = n_heads
        self.scale = math.sqrt(adim)
        self.w_key = nn.Linear(kdim, adim, bias=bias)
        self.w_query = nn.Linear(qdim, adim, bias=bias)
        self.r = nn.Parameter(torch.Tensor([init_r])) if init_r is not None else None
        if param_init == 'xavier_uniform':
            nn.init.xavier_uniform_(self.w_key.weight, gain=1 / math.sqrt(2))
            nn.init.xavier_uniform_(self.w_query.weight, gain=1 / math.sqrt(2))
            if bias:
                nn.init.constant_(self.w_key.bias, 0.)
                nn.init.constant_(self.w_query.bias, 0.)

    def forward(self, key, query, mask):
        """key [B,T,kdim], query [B,L,qdim], mask [B,L,T] bool (True = valid) -> e [B,H,L,T]"""
        bs, klen = key.shape[:2]
        qlen = query.shape[1]
        k = ops.linear(key, self.w_key.weight, self.w_key.bias).view(bs, klen, self.n_heads, self.d_k)
        q = ops.linear(query, self.w_query.weight, self.w_query.bias).view(bs, qlen, self.n_heads, self.d_k)
        e = ops.head_scores(q, k, 1.0 / self.scale)                                  # batched MFMA GEMM (was torch.einsum)
        if self.r is not None:
            e = e + self.r
        if mask is not None:
            e = e.masked_fill(~mask.unsqueeze(1), NEG_INF)
        return e


class MMA(nn.Module):
    """mocha.py:20-311 (scaled-dot energies, multi-head), training mode."""

    def __init__(self, kdim, qdim, adim, odim, chunk_size, n_heads_mono=1, n_heads_chunk=1, init_r=-4, eps=1e-6,
                 noise_std=1.0, no_denominator=False, sharpening_factor=1.0, dropout=0., dropout_head=0., bias=True,
                 param_init='', conv1d=False, share_chunkwise_attention=False, stableemit_weight=0.0):
        super().__init__()
        if conv1d:
            raise NotImplementedError('MMA: 1-d convolution on the keys')
        if n_heads_mono < 1:
            raise NotImplementedError('MMA without monotonic heads')
        assert adim % (n_heads_mono * n_heads_chunk) == 0
        self.d_k = adim // (n_heads_mono * n_heads_chunk)
        self.w = chunk_size
        self.milk = chunk_size == -1
        self.n_heads = n_heads_mono
        self.H_ma, self.H_ca = n_heads_mono, n_heads_chunk
        self.H_total = self.H_ma * self.H_ca
        self.eps, self.noise_std, self.no_denom = eps, noise_std, no_denominator
        self.sharpening_factor = sharpening_factor
        self.share_ca = share_chunkwise_attention
        self.stableemit_weight, self._stableemit_weight = stableemit_weight, 0
        self.monotonic_energy = _ScaledDotEnergy(kdim, qdim, adim, n_heads_mono, bias, param_init, init_r=init_r)
        self.chunk_energy = None
        if chunk_size > 1 or self.milk:
            self.chunk_energy = _ScaledDotEnergy(kdim, qdim, adim,
                                                 n_heads_chunk if self.share_ca else self.H_ma * n_heads_chunk,
                                                 bias, param_init)
        if self.H_total > 1:
            self.w_value = nn.Linear(kdim, adim, bias=bias)
            self.w_out = nn.Linear(adim, odim, bias=bias)
            if param_init == 'xavier_uniform':
                nn.init.xavier_uniform_(self.w_value.weight, gain=1 / math.sqrt(2))
                nn.init.xavier_uniform_(self.w_out.weight)
                if bias:
                    nn.init.constant_(self.w_value.bias, 0.)
                    nn.init.constant_(self.w_out.bias, 0.)
        self.dropout_attn = nn.Dropout(p=dropout)
        self.dropout_head = dropout_head

    def reset(self):
        pass

    def trigger_stableemit(self):
        self._stableemit_weight = self.stableemit_weight

    def forward(self, key, value, query, mask, residual=None, out_dropout=0.0, mode='parallel'):
        """key = value [B,T,d], query [B,L,d], mask [B,L,T] bool -> (cv [B,L,odim] (+ residual), alpha [B,H_ma,L,T],
        {'beta', 'p_choose'})"""
        if mode != 'parallel':
            raise NotImplementedError('MMA: test-time (hard) attention / decoding is not built')
        bs, klen = key.shape[:2]
        qlen = query.shape[1]
        e_ma = self.monotonic_energy(key, query, mask)                               # [B,H_ma,L,T]
        # parallel_monotonic_attention (hma_train.py:12-67)
        if self.noise_std > 0:
            e_ma = e_ma + torch.zeros_like(e_ma).normal_(std=self.noise_std)
        # p_choose -> exclusive cumprod -> alpha_i from alpha_{i-1}: one scan kernel per target position and direction
        # (csrc/mocha.hip, rows = batch x monotonic heads); the recurrence over positions stays a host loop
        aw_prev = key.new_zeros(bs, self.H_ma, 1, klen)
        aw_prev[:, :, :, 0] = 1.0
        alphas, pcs = [], []
        for i in range(qlen):
            aw_prev, pc = ops.mono_alpha(e_ma[:, :, i:i + 1], aw_prev, self.eps, self.no_denom, self._stableemit_weight)
            alphas.append(aw_prev)
            pcs.append(pc)
        alpha = torch.cat(alphas, dim=2)                                             # [B,H_ma,L,T]
        p_choose = torch.cat(pcs, dim=2)
        alpha_masked = alpha
        if self.dropout_head > 0 and self.training:                                  # HeadDrop (headdrop.py:10-32)
            keep = [0.0 if random.random() < self.dropout_head else 1.0 for _ in range(self.H_ma)]
            n_eff = int(sum(keep))
            head_mask = alpha.new_tensor(keep).view(1, self.H_ma, 1, 1)
            alpha_masked = alpha * head_mask
            if n_eff > 0:
                alpha_masked = alpha_masked * (self.H_ma / n_eff)
        beta = None
        if self.chunk_energy is not None:
            # soft_chunkwise_attention (mocha_train.py:13-58)
            u = self.chunk_energy(key, query, mask)                                  # [B,(H_ma*)H_ca,L,T]
            a = alpha_masked.unsqueeze(2)                                            # [B,H_ma,1,L,T]
            if self.H_ca > 1:
                a = a.repeat(1, 1, self.H_ca, 1, 1)
            u = u.unsqueeze(1)                                                       # [B,1,(H_ma*)H_ca,L,T]
            if self.H_ma > 1 and not self.share_ca:
                u = u.view(bs, self.H_ma, self.H_ca, qlen, klen)
            full = (bs, self.H_ma, self.H_ca, qlen, klen)
            beta = ops.chunk_beta(u.expand(full), a.expand(full), self.w, self.sharpening_factor)
            beta = self.dropout_attn(beta.reshape(bs, -1, qlen, klen))               # [B,H_ma*H_ca,L,T]
        po = out_dropout if self.training else 0.0
        if self.H_total > 1:
            v = ops.linear(value, self.w_value.weight, self.w_value.bias).view(bs, klen, self.H_total, self.d_k)
            aw = alpha_masked if self.w == 1 else beta                               # [B,H_total,L,T]
            cv = ops.head_context(aw, v).reshape(bs, qlen, self.H_total * self.d_k)     # batched MFMA GEMM (was torch.einsum)
            cv = ops.linear(cv, self.w_out.weight, self.w_out.bias, res=residual, dropout_p=po)
        else:
            cv = ops.head_context(alpha_masked if self.w == 1 else beta, value.unsqueeze(2)).view(bs, qlen, value.shape[-1])
            if residual is not None:
                cv = residual + ops.dropout(cv, po, self.training)
        return cv, alpha, {'beta': beta, 'p_choose': p_choose}
