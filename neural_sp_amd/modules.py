"""L1 modules of the Speech2Text hot path, mirroring the reference's classes, constructor
arguments and parameter names (so reference checkpoints load with load_state_dict), with
every tensor op routed to the HIP kernels in neural_sp_amd.ops.

Reference files (relative to the reference root):
  neural_sp/models/modules/positionwise_feed_forward.py
  neural_sp/models/modules/relative_multihead_attention.py
  neural_sp/models/modules/multihead_attention.py
  neural_sp/models/modules/conformer_convolution.py
  neural_sp/models/modules/positional_embedding.py
  neural_sp/models/modules/initialization.py
"""
import logging
import copy
import math
import random

import torch
import torch.nn as nn

from neural_sp_amd import ops

logger = logging.getLogger(__name__)


# ---------------------------------------------------------------- initialisation
def init_with_xavier_uniform(n, p):
    """initialization.py:37-52"""
    if p.dim() == 1:
        nn.init.constant_(p, 0.)
    elif p.dim() in [2, 3, 4]:
        nn.init.xavier_uniform_(p)
    else:
        raise ValueError(n)


def init_with_lecun_normal(n, p, param_init):
    """initialization.py:55-80"""
    if p.dim() == 1:
        nn.init.constant_(p, 0.)
    elif p.dim() == 2:
        nn.init.normal_(p, mean=0., std=1. / math.sqrt(p.size(1)))
    elif p.dim() in (3, 4):
        fan_in = p.size(1) * p[0][0].numel()
        nn.init.normal_(p, mean=0., std=1. / math.sqrt(fan_in))
    else:
        raise ValueError(n)


def init_with_uniform(n, p, param_init):
    """initialization.py:83-99"""
    if p.dim() == 1:
        nn.init.constant_(p, 0.)
    elif p.dim() in [2, 3, 4]:
        nn.init.uniform_(p, a=-param_init, b=param_init)
    else:
        raise ValueError(n)


class AttnMask(object):
    """What the reference passes as a `[B,T,T]` byte mask (transformer.py:633-686), kept
    symbolic: key j is visible to query i iff j < klens[b] (make_san_mask), and
    j <= i + lookahead if causal (causal()), and, when chunk_nc > 0,
    max(0, c0 - chunk_nl) <= j < c0 + chunk_nc with c0 the start of i's chunk
    (make_chunkwise_san_mask).  The kernels evaluate the predicate in place."""

    def __init__(self, klens_dev, causal=False, lookahead=0, chunk_nl=0, chunk_nc=0):
        self.klens = klens_dev
        self.causal, self.lookahead = bool(causal), int(lookahead)
        self.chunk_nl, self.chunk_nc = int(chunk_nl), int(chunk_nc)

    def cfg(self):
        return {'causal': self.causal, 'lookahead': self.lookahead,
                'chunk_nl': self.chunk_nl, 'chunk_nc': self.chunk_nc}

    def dense(self, qlen, klen):
        """the bool `[B, qlen, klen]` tensor of a plain padding mask (key j < klens[b]), for the host-level
        monotonic attention of neural_sp_amd.mma"""
        assert not self.causal and self.chunk_nc == 0
        j = torch.arange(klen, device=self.klens.device).view(1, 1, klen)
        return (j < self.klens.view(-1, 1, 1)).expand(-1, qlen, -1)


# ---------------------------------------------------------------- FFN
class LinearGLUBlock(nn.Module):
    """modules/glu.py:11-26: F.glu(fc(xs), dim=-1) -- one GEMM + the GLU kernel."""

    def __init__(self, idim):
        super().__init__()
        self.fc = nn.Linear(idim, idim * 2)

    def forward(self, xs):
        return ops.glu(ops.linear(xs, self.fc.weight, self.fc.bias))


class PositionwiseFeedForward(nn.Module):
    """positionwise_feed_forward.py:22-89: w_2(dropout(act(w_1 x))) -- two MFMA GEMMs with
    bias+activation+dropout fused in the first epilogue.  `forward(xs, residual, alpha,
    out_dropout)` additionally folds the caller's `alpha*dropout(.) + residual` into the
    second epilogue (conformer_block.py:131-134)."""

    def __init__(self, d_model, d_ff, dropout, activation, param_init, bottleneck_dim=0):
        super().__init__()
        self.bottleneck_dim = bottleneck_dim
        if bottleneck_dim > 0:
            self.w_1_e = nn.Linear(d_model, bottleneck_dim)
            self.w_1_d = nn.Linear(bottleneck_dim, d_ff)
            self.w_2_e = nn.Linear(d_ff, bottleneck_dim)
            self.w_2_d = nn.Linear(bottleneck_dim, d_model)
        else:
            self.w_1 = nn.Linear(d_model, d_ff)
            self.w_2 = nn.Linear(d_ff, d_model)
        self.dropout_p = dropout
        if activation not in ('relu', 'gelu', 'gelu_accurate', 'swish', 'glu'):
            raise NotImplementedError(activation)
        self.activation_name = activation
        if activation == 'glu':
            # positionwise_feed_forward.py:58-59 / modules/glu.py:11-26: the "activation" is a module with
            # its own Linear(d_ff, 2 d_ff) followed by F.glu (parameters `activation.fc.*`)
            self.activation = LinearGLUBlock(d_ff)
        else:
            self.activation = activation
        if param_init == 'xavier_uniform':
            for n, p in self.named_parameters():
                init_with_xavier_uniform(n, p)

    def forward(self, xs, residual=None, alpha=1.0, out_dropout=0.0):
        p = self.dropout_p if self.training else 0.0
        po = out_dropout if self.training else 0.0
        if self.activation_name == 'glu':
            # w_1 -> fc (d_ff -> 2 d_ff) -> GLU -> dropout -> w_2: the GEMMs and the GLU kernel of the conv module
            if self.bottleneck_dim > 0:
                h = ops.linear(ops.linear(xs, self.w_1_e.weight, self.w_1_e.bias), self.w_1_d.weight, self.w_1_d.bias)
            else:
                h = ops.linear(xs, self.w_1.weight, self.w_1.bias)
            h = ops.dropout(self.activation(h), p, self.training)
            if self.bottleneck_dim > 0:
                h = ops.linear(h, self.w_2_e.weight, self.w_2_e.bias)
                return ops.linear(h, self.w_2_d.weight, self.w_2_d.bias, res=residual, alpha=alpha, dropout_p=po)
            return ops.linear(h, self.w_2.weight, self.w_2.bias, res=residual, alpha=alpha, dropout_p=po)
        if self.bottleneck_dim > 0:
            h = ops.linear(xs, self.w_1_e.weight, self.w_1_e.bias)
            h = ops.linear(h, self.w_1_d.weight, self.w_1_d.bias, act=self.activation, dropout_p=p)
            h = ops.linear(h, self.w_2_e.weight, self.w_2_e.bias)
            return ops.linear(h, self.w_2_d.weight, self.w_2_d.bias, res=residual, alpha=alpha, dropout_p=po)
        return ops.ffn(xs, self.w_1.weight, self.w_1.bias, self.w_2.weight, self.w_2.bias,
                       self.activation, p, residual, alpha, po)


# ---------------------------------------------------------------- attention
class RelativeMultiheadAttentionMechanism(nn.Module):
    """relative_multihead_attention.py:21-220.  Quirks kept:
      * non-XL mode projects the position table with w_value (:176), not a separate matrix;
      * q is computed from `key` (:171); all linears are bias-free (:41);
      * _rel_shift is the symmetric gather BD[i, min(|i-j|, clamp)] (:112-144);
      * HeadDrop does not affect the context (:207-215)."""

    def __init__(self, kdim, qdim, adim, odim, n_heads, dropout, dropout_head=0.,
                 bias=False, param_init='', xl_like=False, clamp_len=-1):
        super().__init__()
        assert adim % n_heads == 0 and kdim == qdim
        self.d_k = adim // n_heads
        self.n_heads = n_heads
        self.scale = math.sqrt(self.d_k)
        self.xl_like = xl_like
        self.clamp_len = clamp_len
        self.dropout_attn_p = dropout
        self.dropout_head = dropout_head
        self.w_key = nn.Linear(kdim, adim, bias=bias)
        self.w_value = nn.Linear(kdim, adim, bias=bias)
        self.w_query = nn.Linear(qdim, adim, bias=bias)
        self.w_out = nn.Linear(adim, odim, bias=bias)
        if xl_like:
            self.w_pos = nn.Linear(qdim, adim, bias=bias)
        if param_init == 'xavier_uniform':
            self.reset_parameters(bias)

    def reset_parameters(self, bias):
        nn.init.xavier_uniform_(self.w_key.weight, gain=1 / math.sqrt(2))
        nn.init.xavier_uniform_(self.w_value.weight, gain=1 / math.sqrt(2))
        nn.init.xavier_uniform_(self.w_query.weight, gain=1 / math.sqrt(2))
        if bias:
            nn.init.constant_(self.w_key.bias, 0.)
            nn.init.constant_(self.w_value.bias, 0.)
            nn.init.constant_(self.w_query.bias, 0.)
        nn.init.xavier_uniform_(self.w_out.weight)
        if bias:
            nn.init.constant_(self.w_out.bias, 0.)
        if self.xl_like:
            nn.init.xavier_uniform_(self.w_pos.weight)
            if bias:
                nn.init.constant_(self.w_pos.bias, 0.)

    def forward(self, key, query, pos_embs, mask, u_bias=None, v_bias=None,
                residual=None, out_dropout=0.0):
        """key `[B,T,d]`; pos_embs `[T,1,d]` or `[T,d]`; mask: AttnMask.
        Returns (cv `[B,T,odim]` = residual + dropout(w_out(context)), aw `[B,H,T,T]`)."""
        bs, klen = key.shape[:2]
        qlen = query.shape[1]
        if klen != qlen:
            raise NotImplementedError('streaming cache (klen != qlen) is inference-only and out of scope')
        H, dk = self.n_heads, self.d_k
        d = H * dk
        if (ops.bf16_mode() and u_bias is None and v_bias is None and d % 8 == 0 and dk % 8 == 0
                and self.w_out.weight.shape[0] == d):
            pe = pos_embs.reshape(-1, pos_embs.shape[-1])
            R = min(self.clamp_len + 1, klen) if self.clamp_len > 0 else klen
            w_pos = self.w_pos if self.xl_like else self.w_value
            cfg = mask.cfg() if mask is not None else {}
            cfg.update(H=H, clamp=self.clamp_len, dropout=self.dropout_attn_p, training=self.training)
            cache = getattr(pos_embs, '_nsp_pe16', None)        # (one dict per forward, on the tensor every layer receives)
            if cache is None:
                cache = {}
                try:
                    pos_embs._nsp_pe16 = cache
                except Exception:
                    pass
            cfg['pe16_cache'] = cache
            cv, aw = ops.SelfAttnFn.apply(key, self.w_query.weight, self.w_key.weight, self.w_value.weight,
                                          self.w_query.bias, self.w_key.bias, self.w_value.bias,
                                          self.w_out.weight, self.w_out.bias, pe[:R], w_pos.weight,
                                          mask.klens if mask is not None else None, cfg, residual,
                                          out_dropout if self.training else 0.0)
            return ops.tag_prep(cv), aw
        k = ops.linear(key, self.w_key.weight, self.w_key.bias).view(bs, klen, H, dk)
        v = ops.linear(key, self.w_value.weight, self.w_value.bias).view(bs, klen, H, dk)
        q = ops.linear(key, self.w_query.weight, self.w_query.bias).view(bs, qlen, H, dk)
        pe = pos_embs.reshape(-1, pos_embs.shape[-1])
        R = min(self.clamp_len + 1, klen) if self.clamp_len > 0 else klen
        pe = pe[:R]  # rows beyond clamp_len are never gathered (:139-141)
        w_pos = self.w_pos if self.xl_like else self.w_value
        pos = ops.linear(pe, w_pos.weight, w_pos.bias).view(R, H, dk)
        q_ac, q_bd = q, None
        if u_bias is not None:
            assert self.xl_like
            q_ac = q + u_bias[None, None]
        if v_bias is not None:
            assert self.xl_like
            q_bd = q + v_bias[None, None]
        cfg = mask.cfg() if mask is not None else {}
        cfg.update(clamp=self.clamp_len, dropout=self.dropout_attn_p, training=self.training)
        cv, aw = ops.AttentionFn.apply(q_ac, q_bd, k, v, pos, mask.klens if mask is not None else None, cfg)
        po = out_dropout if self.training else 0.0
        cv = ops.linear(cv, self.w_out.weight, self.w_out.bias, res=residual, dropout_p=po)
        return cv, aw


class MultiheadAttentionMechanism(nn.Module):
    """multihead_attention.py:21-157, atype='scaled_dot' (the encoder self-attention
    path; 'add' belongs to the LAS decoder and is out of scope).  bias=True."""

    def __init__(self, kdim, qdim, adim, odim, n_heads, dropout, dropout_head=0.,
                 atype='scaled_dot', bias=True, param_init='', xl_like=False, clamp_len=-1):
        super().__init__()
        if atype != 'scaled_dot':
            raise NotImplementedError(atype)
        assert adim % n_heads == 0
        self.d_k = adim // n_heads
        self.n_heads = n_heads
        self.scale = math.sqrt(self.d_k)
        self.dropout_attn_p = dropout
        self.dropout_head = dropout_head
        self.w_key = nn.Linear(kdim, adim, bias=bias)
        self.w_value = nn.Linear(kdim, adim, bias=bias)
        self.w_query = nn.Linear(qdim, adim, bias=bias)
        self.w_out = nn.Linear(adim, odim, bias=bias)
        if param_init == 'xavier_uniform':
            nn.init.xavier_uniform_(self.w_key.weight, gain=1 / math.sqrt(2))
            nn.init.xavier_uniform_(self.w_value.weight, gain=1 / math.sqrt(2))
            nn.init.xavier_uniform_(self.w_query.weight, gain=1 / math.sqrt(2))
            nn.init.xavier_uniform_(self.w_out.weight)
            if bias:
                for m in (self.w_key, self.w_value, self.w_query, self.w_out):
                    nn.init.constant_(m.bias, 0.)

    def reset(self):
        pass

    def forward(self, key, value, query, mask, residual=None, out_dropout=0.0, **unused):
        bs, klen = key.shape[:2]
        qlen = query.shape[1]
        H, dk = self.n_heads, self.d_k
        headdrop = self.dropout_head > 0 and self.training
        d = H * dk
        if (ops.bf16_mode() and key is value and key is query and d % 8 == 0 and dk % 8 == 0
                and self.w_out.weight.shape[0] == d and key.shape[-1] == d and not headdrop):
            cfg = mask.cfg() if mask is not None else {}
            cfg.update(H=H, clamp=-1, dropout=self.dropout_attn_p, training=self.training)
            cv, aw = ops.SelfAttnFn.apply(key, self.w_query.weight, self.w_key.weight, self.w_value.weight,
                                          self.w_query.bias, self.w_key.bias, self.w_value.bias,
                                          self.w_out.weight, self.w_out.bias, None, None,
                                          mask.klens if mask is not None else None, cfg, residual,
                                          out_dropout if self.training else 0.0)
            return ops.tag_prep(cv), aw, {}
        k = ops.linear(key, self.w_key.weight, self.w_key.bias).view(bs, klen, H, dk)
        v = ops.linear(value, self.w_value.weight, self.w_value.bias).view(bs, klen, H, dk)
        q = ops.linear(query, self.w_query.weight, self.w_query.bias).view(bs, qlen, H, dk)
        cfg = mask.cfg() if mask is not None else {}
        cfg.update(clamp=-1, dropout=self.dropout_attn_p, training=self.training)
        cv, aw = ops.AttentionFn.apply(q, None, k, v, None, mask.klens if mask is not None else None, cfg)
        if headdrop:
            # HeadDrop (modules/headdrop.py:10-32, multihead_attention.py:147-148): whole heads are zeroed with
            # probability dropout_head (Python's `random`, one draw per head) and the survivors rescaled; the context
            # is linear in the attention weights, so the per-head factor is applied to the context vectors
            keep = [0.0 if random.random() < self.dropout_head else 1.0 for _ in range(H)]
            n_eff = sum(keep)
            scale = cv.new_tensor([kk * (H / n_eff if n_eff > 0 else 1.0) for kk in keep]).view(1, 1, H, 1)
            cv = (cv.view(bs, qlen, H, dk) * scale).view(bs, qlen, H * dk)
        po = out_dropout if self.training else 0.0
        cv = ops.linear(cv, self.w_out.weight, self.w_out.bias, res=residual, dropout_p=po)
        return cv, aw, {}


# ---------------------------------------------------------------- Conformer conv module
class ConformerConvBlock(nn.Module):
    """conformer_convolution.py:17-129 on channels-last `[B,T,C]` (no transposes):
    pointwise(d->2d) GEMM -> GLU -> depthwise k-tap conv -> LayerNorm+Swish (one kernel)
    -> pointwise GEMM with the block's dropout+residual in its epilogue."""

    def __init__(self, d_model, kernel_size, param_init, normalization='batch_norm', causal=False):
        super().__init__()
        assert (kernel_size - 1) % 2 == 0, 'kernel_size must be the odd number.'
        assert kernel_size >= 3, 'kernel_size must be larger than 3.'
        self.kernel_size = kernel_size
        self.causal = causal
        self.padding = (kernel_size - 1) if causal else (kernel_size - 1) // 2
        self.pointwise_conv1 = nn.Conv1d(d_model, d_model * 2, kernel_size=1, stride=1, padding=0)
        self.depthwise_conv = nn.Conv1d(d_model, d_model, kernel_size=kernel_size, stride=1,
                                        padding=self.padding, groups=d_model, bias=True)
        if normalization == 'batch_norm':
            # couples the utterances of a batch (and, under DDP, only those of one rank: the reference does
            # not use SyncBatchNorm either); statistics include the padded frames, as in the reference
            self.norm = nn.BatchNorm1d(d_model)
        elif normalization == 'group_norm':
            num_groups = 2
            self.norm = nn.GroupNorm(num_groups=max(1, d_model // num_groups), num_channels=d_model)
            if d_model // self.norm.num_groups != 2 or d_model % 4:
                raise NotImplementedError('group_norm is built for pairs of channels (even d_model, multiple of 4)')
        elif normalization == 'layer_norm':
            self.norm = nn.LayerNorm(d_model, eps=1e-12)
        else:
            raise NotImplementedError(normalization)
        self.pointwise_conv2 = nn.Conv1d(d_model, d_model, kernel_size=1, stride=1, padding=0)
        convs = [self.pointwise_conv1, self.pointwise_conv2, self.depthwise_conv]
        if param_init == 'xavier_uniform':
            for conv_layer in convs:
                for n, p in conv_layer.named_parameters():
                    init_with_xavier_uniform(n, p)
        elif param_init == 'lecun':
            for conv_layer in convs:
                for n, p in conv_layer.named_parameters():
                    init_with_lecun_normal(n, p, 0.1)

    def forward(self, xs, residual=None, out_dropout=0.0):
        C = xs.shape[-1]
        h = ops.linear_glu_dwconv(xs, self.pointwise_conv1.weight, self.pointwise_conv1.bias,   # [2C,C,1] == [2C,C]
                                  self.depthwise_conv.weight, self.depthwise_conv.bias, self.causal)
        if isinstance(self.norm, nn.LayerNorm):
            h = ops.layer_norm(h, self.norm.weight, self.norm.bias, self.norm.eps, act='swish', gemm_only=True)
        elif isinstance(self.norm, nn.BatchNorm1d):
            # "time-independent normalization" on the [B*T, C, 1] view (:119-122): one row per frame
            h = ops.batch_norm_act(h, self.norm, self.training, act='swish')
        else:
            h = ops.group_norm2_act(h, self.norm.weight, self.norm.bias, self.norm.eps, act='swish')
        po = out_dropout if self.training else 0.0
        return ops.linear(h, self.pointwise_conv2.weight, self.pointwise_conv2.bias,
                          res=residual, dropout_p=po)


# ---------------------------------------------------------------- positional encodings
class XLPositionalEmbedding(nn.Module):
    """positional_embedding.py:98-140 (dropout is applied to the TABLE, :139)."""

    def __init__(self, d_model, dropout):
        super().__init__()
        self.d_model = d_model
        self.scale = math.sqrt(d_model)
        inv_freq = 1 / (10000 ** (torch.arange(0.0, d_model, 2.0) / d_model))
        self.register_buffer("inv_freq", inv_freq)
        self.dropout_p = dropout

    def forward(self, xs, scale=False, n_cache=0):
        if scale:
            xs = ops.scale(xs, self.scale)
        pos_emb = ops.xl_pos_table(self.inv_freq, xs.shape[1] + n_cache)
        pos_emb = ops.dropout(pos_emb, self.dropout_p, self.training)
        return xs, pos_emb.unsqueeze(1)


class CausalConv1d(nn.Module):
    """causal_conv.py:15-71 (dilation 1, groups 1): Conv1d(padding = k-1) with the last k-1 outputs dropped, i.e.
    y[t] = b + sum_j W[:,:,j] x[t + j - (k-1)].  Channels-last: window gather over time (left-padded) + one MFMA GEMM
    against the `[C_out, k*C_in]` view of the weight; `act` goes into the GEMM's caller."""

    def __init__(self, in_channels, out_channels, kernel_size, param_init=''):
        super().__init__()
        self.padding = kernel_size - 1
        self.conv1d = nn.Conv1d(in_channels, out_channels, kernel_size, padding=self.padding)
        if param_init == 'xavier_uniform':
            for n, p in self.named_parameters():
                init_with_xavier_uniform(n, p)
        elif param_init == 'lecun':
            for n, p in self.named_parameters():
                init_with_lecun_normal(n, p, 0.1)

    def forward(self, xs):
        co, ci, k = self.conv1d.weight.shape
        g = ops.time_window_gather(xs, k, 1, self.padding, xs.size(1))
        w2 = self.conv1d.weight.permute(0, 2, 1).contiguous().view(co, k * ci)
        return ops.linear(g, w2, self.conv1d.bias)


class PositionalEncoding(nn.Module):
    """positional_embedding.py:18-95: pe_type 'none', 'add' (sinusoidal table) or '1dconv<N>L' (N causal Conv1d ->
    LayerNorm -> ReLU -> Dropout stages over the token embeddings: the decoder-side encoding of the reference's
    Transformer recipes; the N stages start from ONE deep-copied initialisation, as there)."""

    def __init__(self, d_model, dropout, pe_type, param_init, max_len=5000,
                 conv_kernel_size=3, layer_norm_eps=1e-12):
        super().__init__()
        self.d_model = d_model
        self.pe_type = pe_type
        self.scale = math.sqrt(d_model)
        if '1dconv' in pe_type:
            causal_conv1d = CausalConv1d(d_model, d_model, conv_kernel_size, param_init=param_init)
            layers = []
            for _ in range(int(pe_type.replace('1dconv', '')[0])):
                layers += [copy.deepcopy(causal_conv1d), nn.LayerNorm(d_model, eps=layer_norm_eps), nn.ReLU(),
                           nn.Dropout(p=dropout)]
            self.pe = nn.Sequential(*layers)        # parameter container: `pe.{0,4,8}.conv1d.*`, `pe.{1,5,9}.*`
        elif pe_type != 'none':
            pe = torch.zeros(max_len, d_model, dtype=torch.float32)
            position = torch.arange(0, max_len, dtype=torch.float32).unsqueeze(1)
            div_term = torch.exp(torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model))
            pe[:, 0::2] = torch.sin(position * div_term)
            pe[:, 1::2] = torch.cos(position * div_term)
            self.register_buffer('pe', pe.unsqueeze(0))
        self.dropout_p = dropout

    def forward(self, xs, scale=True, offset=0):
        alpha = self.scale if scale else 1.0
        if self.pe_type == 'none':
            xs = ops.scale(xs, alpha) if alpha != 1.0 else xs
        elif self.pe_type == 'add':
            xs = ops.scale_add_bcast(xs, self.pe[0, offset:xs.size(1) + offset].contiguous(), alpha)
        elif '1dconv' in self.pe_type:
            xs = ops.scale(xs, alpha) if alpha != 1.0 else xs
            for n in range(len(self.pe) // 4):
                conv, norm = self.pe[4 * n], self.pe[4 * n + 1]
                xs = ops.layer_norm(conv(xs), norm.weight, norm.bias, norm.eps, act='relu')
                xs = ops.dropout(xs, self.dropout_p, self.training)
            return xs                                # no further dropout on this branch (positional_embedding.py:92-93)
        else:
            raise NotImplementedError(self.pe_type)
        return ops.dropout(xs, self.dropout_p, self.training)
