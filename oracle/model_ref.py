"""CPU restatement of the reference Speech2Text training hot path (TEST ORACLE ONLY).

Plain torch-CPU ops in any float dtype (fp64 for a tight oracle), functional over a
reference-named state_dict, so the same weights drive the reference, this oracle and the
HIP path.  Each function cites the reference lines it follows.  Pinned against fixtures
produced by the reference itself (tests/golden/*.pt via oracle/gen_golden.py); see
tests/test_oracle_cpu.py.  Nothing under neural_sp_amd/ imports this file.
"""
import math

import torch
import torch.nn.functional as F

from oracle.rnnt_ref import rnnt_loss_ref

NEG_INF32 = float(torch.finfo(torch.float32).min)


def _lin(x, sd, name, bias=True):
    w = sd[name + '.weight']
    b = sd.get(name + '.bias') if bias else None
    return F.linear(x, w, b)


def _ln(x, sd, name, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[name + '.weight'], sd[name + '.bias'], eps)


def _act(x, name):
    if name == 'swish':
        return x * torch.sigmoid(x)
    if name == 'relu':
        return torch.relu(x)
    if name == 'gelu_accurate':
        return F.gelu(x)
    if name == 'gelu':
        return F.gelu(x, approximate='tanh')
    raise NotImplementedError(name)


def _pool_len(n, k):  # conv.py:471-474 ceil-mode max-pool length
    return int(math.ceil((n + 1 - (k - 1) - 1) // k + 1))


# ----------------------------------------------------------------------------- conv frontend
def conv_frontend(xs, xlens, sd, args, prefix='enc.conv'):
    """conv.py:167-195 + Conv2dBlock.forward :347-396 (3x3, pad 1, ReLU, MaxPool2d ceil)."""
    B, T, Fq = xs.shape
    x = xs.view(B, T, args.conv_in_channel, Fq // args.conv_in_channel).transpose(2, 1)
    pools = [[int(v) for v in p.strip('()').split(',')] for p in args.conv_poolings.split('_')]
    xlens = list(xlens)
    for i, pool in enumerate(pools):
        p = '%s.layers.%d' % (prefix, i)
        x = torch.relu(F.conv2d(x, sd[p + '.conv1.weight'], sd[p + '.conv1.bias'], padding=1))
        x = torch.relu(F.conv2d(x, sd[p + '.conv2.weight'], sd[p + '.conv2.bias'], padding=1))
        if pool[0] * pool[1] > 1:
            x = F.max_pool2d(x, tuple(pool), tuple(pool), ceil_mode=True)
            xlens = [_pool_len(n, pool[0]) for n in xlens]
    B, C, T2, F2 = x.shape
    x = x.transpose(2, 1).contiguous().view(B, T2, C * F2)
    if (prefix + '.bridge.weight') in sd:
        x = _lin(x, sd, prefix + '.bridge')
    return x, xlens


# ----------------------------------------------------------------------------- attention
def visible_mask(xlens, T, unidir=False, lookahead=0, N_l=0, N_c=0):
    """transformer.py:633-686 -> bool `[B,T,T]` (True = visible)."""
    j = torch.arange(T)[None, None, :]
    i = torch.arange(T)[None, :, None]
    vis = j < torch.tensor(xlens)[:, None, None]
    vis = vis.expand(len(xlens), T, T).clone()
    if unidir:
        vis &= (j <= i + lookahead)
    if N_c > 0:
        c0 = (i // N_c) * N_c
        vis &= (j >= (c0 - N_l).clamp(min=0)) & (j < c0 + N_c)
    return vis


def rel_mha(x, pos_embs, vis, sd, p, H, clamp_len, xl_like=False, u_bias=None, v_bias=None):
    """relative_multihead_attention.py:146-220 (incl. _rel_shift :112-144)."""
    B, T, d = x.shape
    dk = d // H
    k = F.linear(x, sd[p + '.w_key.weight']).view(B, T, H, dk)
    v = F.linear(x, sd[p + '.w_value.weight']).view(B, T, H, dk)
    q = F.linear(x, sd[p + '.w_query.weight']).view(B, T, H, dk)
    wp = sd[p + ('.w_pos.weight' if xl_like else '.w_value.weight')]
    pe = F.linear(pos_embs, wp).view(-1, H, dk)
    qa = q + u_bias[None, None] if u_bias is not None else q
    qb = q + v_bias[None, None] if v_bias is not None else q
    AC = torch.einsum('bihd,bjhd->bijh', qa, k)
    BD = torch.einsum('bihd,jhd->bijh', qb, pe)
    idx = (torch.arange(T)[None, :] - torch.arange(T)[:, None]).abs()
    if clamp_len > 0:
        idx = idx.clamp(max=clamp_len)
    BD = torch.gather(BD, 2, idx[None, :, :, None].expand(B, T, T, H))
    e = (AC + BD) / math.sqrt(dk)
    if vis is not None:
        e = e.masked_fill(~vis[:, :, :, None], NEG_INF32)
    aw = torch.softmax(e, dim=2)
    cv = torch.einsum('bijh,bjhd->bihd', aw, v).reshape(B, T, d)
    return F.linear(cv, sd[p + '.w_out.weight'])


def mha(x, vis, sd, p, H):
    """multihead_attention.py:93-157 (scaled_dot, bias=True)."""
    B, T, d = x.shape
    dk = d // H
    k = _lin(x, sd, p + '.w_key').view(B, T, H, dk)
    v = _lin(x, sd, p + '.w_value').view(B, T, H, dk)
    q = _lin(x, sd, p + '.w_query').view(B, T, H, dk)
    e = torch.einsum('bihd,bjhd->bijh', q, k) / math.sqrt(dk)
    if vis is not None:
        e = e.masked_fill(~vis[:, :, :, None], NEG_INF32)
    aw = torch.softmax(e, dim=2)
    cv = torch.einsum('bijh,bjhd->bihd', aw, v).reshape(B, T, d)
    return _lin(cv, sd, p + '.w_out')


def ffn(x, sd, p, act):
    """positionwise_feed_forward.py:77-89 ('glu' = LinearGLUBlock, modules/glu.py:11-26)"""
    h = _lin(x, sd, p + '.w_1')
    h = F.glu(_lin(h, sd, p + '.activation.fc'), dim=-1) if act == 'glu' else _act(h, act)
    return _lin(h, sd, p + '.w_2')


def subsample(xs, xlens, sd, stype, f, p):
    """encoders/subsampling.py (one layer, factor f > 1) -> (xs, xlens)."""
    B, T, C = xs.shape
    if stype == 'max_pool':      # :188-209
        return (F.max_pool1d(xs.transpose(2, 1), f, f, ceil_mode=True).transpose(2, 1),
                [_pool_len(n, f) for n in xlens])
    if stype == 'mean_pool':     # :226-246; lengths through update_lens_1d's generic branch (conv.py:446-448)
        return (F.avg_pool1d(xs.transpose(2, 1), f, f, 0, ceil_mode=True).transpose(2, 1),
                [int(math.floor((n - (f - 1) - 1) // f + 1)) for n in xlens])
    if stype == 'drop':          # :111-128
        return xs[:, ::f], [max(1, math.ceil(n / f)) for n in xlens]
    if stype == 'add':           # :146-172
        assert f <= 2
        xp = xs if T % 2 == 0 else torch.cat([xs, xs.new_zeros(B, 1, C)], dim=1)
        return xp[:, ::2] + xp[:, 1::2], [max(1, math.ceil(n / f)) for n in xlens]
    if stype == 'concat':        # :24-52: frames t-f+1..t side by side for every t with (t+1) % f == 0
        To = T // f
        h = xs[:, :To * f].reshape(B, To, f * C)
        return torch.relu(_lin(h, sd, p + '.proj')), [max(1, n // f) for n in xlens]
    if stype == 'conv1d':        # :70-94 (kernel 3, stride f, padding 1)
        h = F.conv1d(xs.transpose(2, 1), sd[p + '.conv1d.weight'], sd[p + '.conv1d.bias'], stride=f, padding=1)
        return torch.relu(h.transpose(2, 1)), [int(math.floor((n + 2 - 2 - 1) // f + 1)) for n in xlens]
    raise NotImplementedError(stype)


def conformer_conv(x, sd, p, k, causal, normalization='layer_norm', training=True, bn_out=None):
    """conformer_convolution.py:98-129.  batch_norm / group_norm act on the `[B*T, C, 1]` view (:119-122);
    in training mode BatchNorm1d uses the batch statistics and the updated running statistics
    (momentum 0.1, unbiased variance) are handed back through `bn_out`."""
    B, T, C = x.shape
    h = x.transpose(2, 1)
    h = F.conv1d(h, sd[p + '.pointwise_conv1.weight'], sd[p + '.pointwise_conv1.bias'])
    h = F.glu(h, dim=1)
    pad = (k - 1) if causal else (k - 1) // 2
    h = F.conv1d(h, sd[p + '.depthwise_conv.weight'], sd[p + '.depthwise_conv.bias'], padding=pad, groups=C)
    if causal:
        h = h[:, :, :-pad]
    h = h.transpose(2, 1)
    if normalization == 'layer_norm':
        h = F.layer_norm(h, (C,), sd[p + '.norm.weight'], sd[p + '.norm.bias'], 1e-12)
    elif normalization == 'batch_norm':
        h2 = h.reshape(B * T, C)
        if training:
            mean, var = h2.mean(0), h2.var(0, unbiased=False)
            if bn_out is not None:
                M = B * T
                bn_out[p + '.norm.running_mean'] = 0.9 * sd[p + '.norm.running_mean'] + 0.1 * mean.detach()
                bn_out[p + '.norm.running_var'] = 0.9 * sd[p + '.norm.running_var'] + 0.1 * var.detach() * M / (M - 1)
                bn_out[p + '.norm.num_batches_tracked'] = sd[p + '.norm.num_batches_tracked'] + 1
        else:
            mean, var = sd[p + '.norm.running_mean'], sd[p + '.norm.running_var']
        h = ((h2 - mean) / torch.sqrt(var + 1e-5) * sd[p + '.norm.weight'] + sd[p + '.norm.bias']).view(B, T, C)
    elif normalization == 'group_norm':
        h = F.group_norm(h.reshape(B * T, C, 1), max(1, C // 2), sd[p + '.norm.weight'], sd[p + '.norm.bias'],
                         1e-5).view(B, T, C)
    else:
        raise NotImplementedError(normalization)
    h = (h * torch.sigmoid(h)).transpose(2, 1)
    h = F.conv1d(h, sd[p + '.pointwise_conv2.weight'], sd[p + '.pointwise_conv2.bias'])
    return h.transpose(2, 1)


def xl_pos_emb(T, inv_freq, dtype):
    """positional_embedding.py:131-138"""
    pos = torch.arange(-1, -T - 1, -1.0, dtype=torch.float32)
    s = torch.einsum('i,j->ij', pos, inv_freq.float())
    return torch.cat([s.sin(), s.cos()], dim=-1).to(dtype)


# ----------------------------------------------------------------------------- encoder
def encoder_forward(xs, xlens, sd, args, training=True, bn_out=None, sub_out=None):
    """transformer.py:419-617 / conformer_block.py:95-182 / transformer_block.py:79-141,
    eval-mode semantics (no dropout, LayerDrop scaling only if dropout_enc_layer > 0); `training` only
    selects batch vs running statistics of a batch_norm convolution module."""
    cnorm = getattr(args, 'conformer_normalization', 'layer_norm')
    n_sub = {'sub1': getattr(args, 'enc_n_layers_sub1', 0), 'sub2': getattr(args, 'enc_n_layers_sub2', 0)}
    dtype = xs.dtype
    enc_type = args.enc_type
    is_conf = 'conformer' in enc_type
    v2 = 'conformer_v2' in enc_type
    unidir = 'uni' in enc_type
    d, H = args.transformer_enc_d_model, args.transformer_enc_n_heads
    eps = args.transformer_layer_norm_eps
    n_layers = args.enc_n_layers
    sub = [1] * n_layers
    for i, s in enumerate(map(int, args.subsample.split('_')[:n_layers])):
        sub[i] = s
    las = [0] * n_layers
    for i, s in enumerate(map(int, args.transformer_enc_lookaheads.split('_')[:n_layers])):
        las[i] = s
    N_l = int(str(args.lc_chunk_size_left).split('_')[-1])
    N_c = int(str(args.lc_chunk_size_current).split('_')[-1])
    N_r = int(str(args.lc_chunk_size_right).split('_')[-1])
    lc = N_c > 0 and not unidir
    stype = args.lc_type if lc else ''
    causal_conv = unidir or stype == 'mask'
    B = xs.shape[0]
    if lc:
        from neural_sp_amd.encoders import chunkwise  # pure data movement, same as utils.py:13-45
        xs = chunkwise(xs, 0, N_c, 0) if stype == 'mask' else chunkwise(xs, N_l, N_c, N_r)
    if 'conv' in enc_type:
        xs, xlens = conv_frontend(xs, xlens, sd, args)
        fac = 1
        for p in args.conv_poolings.split('_'):
            fac *= int(p.strip('()').split(',')[0])
        N_l, N_c, N_r = max(0, N_l // fac), N_c // fac, N_r // fac
    else:
        xs = _lin(xs, sd, 'enc.embed')
    if stype == 'mask':
        xs = xs.contiguous().view(B, -1, xs.size(2))[:, :max(xlens)]
    pe_type = args.transformer_enc_pe_type
    rel = 'relative' in pe_type
    xs = xs * math.sqrt(d)
    pos = None
    if rel:
        pos = xl_pos_emb(xs.shape[1], sd['enc.pos_emb.inv_freq'], dtype)
    elif pe_type == 'add':
        xs = xs + sd['enc.pos_enc.pe'][:, :xs.shape[1]].to(dtype)
    u_bias = sd.get('enc.u_bias')
    v_bias = sd.get('enc.v_bias')
    ld = args.dropout_enc_layer

    def mask(lth):
        if stype == 'reshape':
            return None
        if stype == 'mask':
            return visible_mask(xlens, xs.shape[1], False, 0, N_l, N_c)
        return visible_mask(xlens, xs.shape[1], unidir, las[lth])

    vis = mask(0)
    def block(xs, p, scale_ld, vis, pos, ub, vb):
        """one encoder block with parameter prefix p"""
        if ld > 0:
            xs = xs / (1 - scale_ld)
        if is_conf and not v2:
            xs = xs + 0.5 * ffn(_ln(xs, sd, p + '.norm1', eps), sd, p + '.feed_forward_macaron', 'swish')
            xs = xs + rel_mha(_ln(xs, sd, p + '.norm2', eps), pos, vis, sd, p + '.self_attn', H,
                              args.transformer_enc_clamp_len, pe_type == 'relative_xl', ub, vb)
            xs = xs + conformer_conv(_ln(xs, sd, p + '.norm3', eps), sd, p + '.conv',
                                     args.conformer_kernel_size, causal_conv, cnorm, training, bn_out)
            xs = xs + 0.5 * ffn(_ln(xs, sd, p + '.norm4', eps), sd, p + '.feed_forward', 'swish')
            xs = _ln(xs, sd, p + '.norm5', eps)
        elif is_conf:
            xs = xs + 0.5 * ffn(_ln(xs, sd, p + '.norm1', eps), sd, p + '.feed_forward_macaron', 'swish')
            xs = xs + conformer_conv(_ln(xs, sd, p + '.norm2', eps), sd, p + '.conv',
                                     args.conformer_kernel_size, causal_conv, cnorm, training, bn_out)
            xs = xs + mha(_ln(xs, sd, p + '.norm3', eps), vis, sd, p + '.self_attn', H)
            xs = xs + 0.5 * ffn(_ln(xs, sd, p + '.norm4', eps), sd, p + '.feed_forward', 'swish')
            xs = _ln(xs, sd, p + '.norm5', eps)
        else:
            xn = _ln(xs, sd, p + '.norm1', eps)
            if pe_type == 'relative_xl':  # the 'relaive' typo: only relative_xl is RelMHA here
                xs = xs + rel_mha(xn, pos, vis, sd, p + '.self_attn', H, args.transformer_enc_clamp_len,
                                  True, ub, vb)
            else:
                xs = xs + mha(xn, vis, sd, p + '.self_attn', H)
            xs = xs + ffn(_ln(xs, sd, p + '.norm2', eps), sd, p + '.feed_forward',
                          args.transformer_ffn_activation)
        return xs

    for l in range(n_layers):
        xs = block(xs, 'enc.layers.%d' % l, ld * (l + 1) / n_layers, vis, pos, u_bias, v_bias)
        for name, n in n_sub.items():
            # transformer.py:568-580,619-630: picked up after layer n, task-specific block WITHOUT the global u/v
            # biases, then bridge / LayerNorm
            if n > 0 and l == n - 1 and sub_out is not None:
                xsub = xs
                if getattr(args, 'task_specific_layer', False):
                    xsub = block(xs, 'enc.layer_' + name, ld * n / n_layers, vis, pos, None, None)
                if ('enc.bridge_%s.weight' % name) in sd:
                    xsub = _lin(xsub, sd, 'enc.bridge_' + name)
                if ('enc.norm_out_%s.weight' % name) in sd:
                    xsub = _ln(xsub, sd, 'enc.norm_out_' + name, eps)
                sub_out[name] = (xsub, list(xlens))
        if l < n_layers - 1 and sub[l] > 1:
            xs, xlens = subsample(xs, xlens, sd, args.subsample_type, sub[l], 'enc.subsample_layers.%d' % l)
            N_l, N_c, N_r = max(0, N_l // sub[l]), N_c // sub[l], N_r // sub[l]
            if rel:
                pos = xl_pos_emb(xs.shape[1], sd['enc.pos_emb.inv_freq'], dtype)
            vis = mask(l + 1)
        elif l < n_layers - 1 and las[l] != las[l + 1]:
            vis = mask(l + 1)
    if stype == 'reshape':
        xs = xs[:, N_l:N_l + N_c].contiguous().view(B, -1, xs.size(2))[:, :max(xlens)]
    xs = _ln(xs, sd, 'enc.norm_out', eps)
    if 'enc.bridge.weight' in sd:
        xs = _lin(xs, sd, 'enc.bridge')
    return xs, xlens


# ----------------------------------------------------------------------------- losses
def ctc_head(eouts, sd, p):
    """ctc.py:81-91"""
    if (p + '.output.weight') in sd:
        return _lin(eouts, sd, p + '.output')
    x, i = eouts, 0
    while (p + '.output.fc%d.weight' % i) in sd:
        x = _lin(x, sd, p + '.output.fc%d' % i)
        i += 1
    return x


def ctc_loss_ref(logits, elens, ys, lsm_prob):
    """ctc.py:105-150 + criterion.py:110-127"""
    B, T, V = logits.shape
    ylens = torch.tensor([len(y) for y in ys], dtype=torch.int32)
    ys_cat = torch.cat([torch.tensor(y, dtype=torch.int32) for y in ys])
    el = torch.tensor(elens, dtype=torch.int32)
    loss = F.ctc_loss(logits.transpose(1, 0).log_softmax(2), ys_cat, el, ylens,
                      reduction='sum', zero_infinity=True) / B
    if lsm_prob > 0:
        probs = torch.softmax(logits, dim=-1)
        lp = torch.log_softmax(logits, dim=-1)
        kl = probs * (lp - math.log(1 / (V - 1)))
        kl = sum(kl[b, :elens[b]].sum() for b in range(B)) / sum(elens)
        loss = loss * (1 - lsm_prob) + kl * lsm_prob
    return loss


def lstm_ref(x, sd, p, sfx='', state=None, return_state=False):
    """nn.LSTM(1 layer, batch_first) with zero initial state (rnn_transducer.py:278-311); sfx='_reverse' selects
    the parameters of the backward direction of a bidirectional layer."""
    B, L, _ = x.shape
    w_ih, w_hh = sd[p + '.weight_ih_l0' + sfx], sd[p + '.weight_hh_l0' + sfx]
    b = sd[p + '.bias_ih_l0' + sfx] + sd[p + '.bias_hh_l0' + sfx]
    n = w_hh.shape[1]
    h, c = state if state is not None else (x.new_zeros(B, n), x.new_zeros(B, n))
    outs = []
    gi = F.linear(x, w_ih, b)
    for t in range(L):
        g = gi[:, t] + F.linear(h, w_hh)
        i, f, gg, o = g.chunk(4, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        outs.append(h)
    y = torch.stack(outs, dim=1)
    return (y, (h, c)) if return_state else y


def rnn_encoder_forward(xs, xlens, sd, args, sub_out=None):
    """encoders/rnn.py:268-383, full-context (B)LSTM encoder, eval-mode semantics (no dropout).  A packed
    (bidirectional) LSTM (rnn.py:534-541) = every utterance run on its own frames, the backward direction from its
    own last frame, zeros in the padded output beyond its length."""
    xlens = list(xlens)
    if 'conv' in args.enc_type:
        xs, xlens = conv_frontend(xs, xlens, sd, args)
    bidir = 'blstm' in args.enc_type
    n_layers, H = args.enc_n_layers, args.enc_n_units
    sub = [1] * n_layers
    for i, f in enumerate(map(int, args.subsample.split('_')[:n_layers])):
        sub[i] = f
    n_sub = {'sub1': getattr(args, 'enc_n_layers_sub1', 0), 'sub2': getattr(args, 'enc_n_layers_sub2', 0)}
    B = xs.shape[0]
    # rnn.py:104: N_c / N_r come from lc_chunk_size_left (sic, build.py:146) / lc_chunk_size_right
    N_c = int(str(args.lc_chunk_size_left).split('_')[0]) // args.n_stacks
    N_r = int(str(args.lc_chunk_size_right).split('_')[0]) // args.n_stacks
    lc_bidir = (N_c > 0 or N_r > 0) and bidir
    if lc_bidir and N_c > 0:
        # _forward_latency_controlled (rnn.py:427-510): chunks of N_c frames with N_r frames of right context; the forward
        # LSTM carries its state over the chunk centres, the backward LSTM is local to chunk + context
        fac = 1
        if 'conv' in args.enc_type:
            for q in args.conv_poolings.split('_'):
                fac *= int(q.strip('()').split(',')[0])
        N_c, N_r = N_c // fac, N_r // fac
        T = xs.shape[1]
        states = [None] * n_layers
        outs = []
        import math as _m
        xl = list(xlens)
        for ci in range(_m.ceil(T / N_c)):
            ch = xs[:, ci * N_c:ci * N_c + N_c + N_r]
            nc = N_c
            for l in range(n_layers):
                yb = lstm_ref(ch.flip(1), sd, 'enc.rnn_bwd.%d' % l).flip(1)
                if ch.shape[1] <= nc:
                    yf, states[l] = lstm_ref(ch, sd, 'enc.rnn.%d' % l, state=states[l], return_state=True)
                else:
                    y1, states[l] = lstm_ref(ch[:, :nc], sd, 'enc.rnn.%d' % l, state=states[l], return_state=True)
                    y2 = lstm_ref(ch[:, nc:], sd, 'enc.rnn.%d' % l, state=states[l])
                    yf = torch.cat([y1, y2], dim=1)
                ch = yf + yb if args.bidirectional_sum_fwd_bwd else torch.cat([yf, yb], dim=-1)
                if args.enc_n_projs > 0 and l != n_layers - 1:
                    ch = torch.relu(_lin(ch, sd, 'enc.proj.%d' % l))
                if sub[l] > 1:
                    ch, xl_new = subsample(ch, xl, sd, args.subsample_type, sub[l], 'enc.subsample.%d' % l)
                    if ci == 0:
                        xl = xl_new
                    nc = nc // sub[l]
            outs.append(ch[:, :nc])
        xs = torch.cat(outs, dim=1)
        xlens = xl
        if 'enc.bridge.weight' in sd:
            xs = _lin(xs, sd, 'enc.bridge')
        return xs[:, :max(xlens)], xlens
    for l in range(n_layers):
        p = 'enc.rnn.%d' % l
        if lc_bidir:
            # _forward_full_context (rnn.py:404-411): no packing -- both directions see the padded frames
            yf = lstm_ref(xs, sd, p)
            yb = lstm_ref(xs.flip(1), sd, 'enc.rnn_bwd.%d' % l).flip(1)
            xs = yf + yb if args.bidirectional_sum_fwd_bwd else torch.cat([yf, yb], dim=-1)
        else:
            T = max(xlens)
            out = xs.new_zeros(B, T, H * (2 if bidir else 1))
            for b in range(B):
                n = xlens[b]
                out[b, :n, :H] = lstm_ref(xs[b:b + 1, :n], sd, p)[0]
                if bidir:
                    out[b, :n, H:] = lstm_ref(xs[b:b + 1, :n].flip(1), sd, p, '_reverse')[0].flip(0)
            if bidir and args.bidirectional_sum_fwd_bwd:
                out = out[..., :H] + out[..., H:]
            xs = out
        for name, n in n_sub.items():          # rnn.py:512-524
            if n > 0 and l == n - 1 and sub_out is not None:
                xsub = xs
                if getattr(args, 'task_specific_layer', False):
                    xsub = torch.relu(_lin(xs, sd, 'enc.layer_' + name))
                if ('enc.bridge_%s.weight' % name) in sd:
                    xsub = _lin(xsub, sd, 'enc.bridge_' + name)
                sub_out[name] = (xsub, list(xlens))
        if args.enc_n_projs > 0 and l != n_layers - 1:
            xs = torch.relu(_lin(xs, sd, 'enc.proj.%d' % l))
        if sub[l] > 1:
            xs, xlens = subsample(xs, xlens, sd, args.subsample_type, sub[l], 'enc.subsample.%d' % l)
    if 'enc.bridge.weight' in sd:
        xs = _lin(xs, sd, 'enc.bridge')
    return xs[:, :max(xlens)], xlens


def rnnt_branch(eouts, elens, ys, sd, args, p='dec_fwd'):
    """rnn_transducer.py:217-276 with the lattice of oracle/rnnt_ref.py."""
    B = eouts.shape[0]
    U = max(len(y) for y in ys)
    ys_in = torch.full((B, U + 1), 3, dtype=torch.long)
    lab = torch.zeros((B, max(U, 1)), dtype=torch.long)
    for b, y in enumerate(ys):
        ys_in[b, 0] = 2
        ys_in[b, 1:len(y) + 1] = torch.tensor(y)
        lab[b, :len(y)] = torch.tensor(y)
    emb = F.embedding(ys_in, sd[p + '.embed.weight'], padding_idx=3)
    d = emb
    for l in range(args.dec_n_layers):
        d = lstm_ref(d, sd, p + '.rnn.%d' % l)
        if args.dec_n_projs > 0:
            d = torch.relu(_lin(d, sd, p + '.proj.%d' % l))
    h = torch.tanh(_lin(eouts, sd, p + '.w_enc')[:, :, None] + F.linear(d, sd[p + '.w_dec.weight'])[:, None])
    logits = _lin(h, sd, p + '.output')
    nll = rnnt_loss_ref(torch.log_softmax(logits, -1), lab, torch.tensor(elens), torch.tensor([len(y) for y in ys]))
    return nll.mean()


def transformer_decoder_att(eouts, elens, ys, sd, args, training, p='dec_fwd'):
    """decoders/transformer.py:373-458 + modules/transformer.py:171-260 (scaled-dot, no cache) +
    criterion.py:45-86 -> (loss_att, acc %, ppl).  dropout / LayerDrop = 0."""
    B, T, _ = eouts.shape
    d, H = args.transformer_dec_d_model, args.transformer_dec_n_heads
    eps = args.transformer_layer_norm_eps
    ylens = [len(y) + 1 for y in ys]
    L = max(ylens)
    ys_in = torch.full((B, L), 3, dtype=torch.long)
    ys_out = torch.full((B, L), 3, dtype=torch.long)
    for b, y in enumerate(ys):
        ys_in[b, 0] = 2
        ys_in[b, 1:len(y) + 1] = torch.tensor(y)
        ys_out[b, :len(y)] = torch.tensor(y)
        ys_out[b, len(y)] = 2
    j = torch.arange(L)
    tgt_vis = (ys_out != 3)[:, None, :] & (j[None, :] <= j[:, None])[None]          # [B,L,L]
    src_vis = (torch.arange(T)[None, :] < torch.tensor(elens)[:, None])[:, None, :].expand(B, L, T)
    out = F.embedding(ys_in, sd[p + '.embed.weight'], padding_idx=3) * math.sqrt(d)
    if args.transformer_dec_pe_type == 'add':
        out = out + sd[p + '.pos_enc.pe'][:, :L].to(out.dtype)
    elif '1dconv' in args.transformer_dec_pe_type:
        # positional_embedding.py:41-53,92-93: N x (causal Conv1d k=3 -> LayerNorm -> ReLU [-> Dropout])
        for n in range(int(args.transformer_dec_pe_type.replace('1dconv', '')[0])):
            q = '%s.pos_enc.pe.%d' % (p, 4 * n)
            w = sd[q + '.conv1d.weight']
            pad = w.shape[-1] - 1
            out = F.conv1d(out.transpose(2, 1), w, sd[q + '.conv1d.bias'], padding=pad)[:, :, :-pad].transpose(2, 1)
            out = torch.relu(_ln(out, sd, '%s.pos_enc.pe.%d' % (p, 4 * n + 1), args.transformer_layer_norm_eps))

    hd = getattr(args, 'dropout_head', 0.0) if (training and getattr(args, 'transformer_dec_attn_type', '') == 'mocha') else 0.0

    def headdrop(aw, nh, head_dim):
        """modules/headdrop.py:10-32: Python's `random`, one draw per head; survivors rescaled"""
        import random
        keep = [0.0 if random.random() < hd else 1.0 for _ in range(nh)]
        n_eff = sum(keep)
        shape = [1] * aw.dim()
        shape[head_dim] = nh
        return aw * aw.new_tensor([k * (nh / n_eff if n_eff > 0 else 1.0) for k in keep]).view(shape)

    def mha2(q_in, kv_in, vis, pp, drop_heads=False):
        Bq, Lq, _ = q_in.shape
        Tk = kv_in.shape[1]
        dk = d // H
        k = _lin(kv_in, sd, pp + '.w_key').view(Bq, Tk, H, dk)
        v = _lin(kv_in, sd, pp + '.w_value').view(Bq, Tk, H, dk)
        q = _lin(q_in, sd, pp + '.w_query').view(Bq, Lq, H, dk)
        e = torch.einsum('bihd,bjhd->bijh', q, k) / math.sqrt(dk)
        e = e.masked_fill(~vis[:, :, :, None], NEG_INF32)
        aw = torch.softmax(e, dim=2)
        if drop_heads and hd > 0:
            aw = headdrop(aw, H, 3)
        cv = torch.einsum('bijh,bjhd->bihd', aw, v).reshape(Bq, Lq, d)
        return _lin(cv, sd, pp + '.w_out')

    def mma(q_in, pp):
        """monotonic multi-head attention, training mode (mocha.py:164-311 with scaled-dot energies, hma_train.py:12-67,
        mocha_train.py:13-58), no noise / HeadDrop -> (context, alpha [B,H_ma,L,T])"""
        Hm, Hc, w = args.mocha_n_heads_mono, args.mocha_n_heads_chunk, args.mocha_chunk_size
        share = getattr(args, 'share_chunkwise_attention', False)

        def energy(pe, nh, r=None):
            k = _lin(eouts, sd, pe + '.w_key').view(B, T, nh, d // nh)
            qq = _lin(q_in, sd, pe + '.w_query').view(B, L, nh, d // nh)
            e = torch.einsum('bihd,bjhd->bhij', qq, k) / math.sqrt(d)
            if r is not None:
                e = e + r
            return e.masked_fill(~src_vis[:, None], NEG_INF32)
        e_ma = energy(pp + '.monotonic_energy', Hm, sd[pp + '.monotonic_energy.r'])
        pc = torch.sigmoid(e_ma)
        x = torch.log(torch.clamp(1 - pc, min=args.mocha_eps, max=1.0))
        cp = torch.exp(torch.cumsum(torch.cat([x.new_zeros(x.shape[:-1] + (1,)), x[..., :-1]], dim=-1), dim=-1))
        aw = eouts.new_zeros(B, Hm, 1, T)
        aw[..., 0] = 1.0
        alphas = []
        for i in range(L):
            den = 1 if args.mocha_no_denominator else torch.clamp(cp[:, :, i:i + 1], min=args.mocha_eps, max=1.0)
            aw = pc[:, :, i:i + 1] * cp[:, :, i:i + 1] * torch.cumsum(aw / den, dim=-1)
            alphas.append(aw)
        alpha = torch.cat(alphas, dim=2)
        alpha_m = headdrop(alpha, Hm, 1) if hd > 0 else alpha
        att = alpha_m
        if w > 1 or w == -1:
            def msum(z, back, fwd):
                shp = z.shape
                y = F.conv1d(F.pad(z.reshape(-1, 1, shp[-1]), [back, fwd]), z.new_ones(1, 1, back + fwd + 1))
                return y.view(shp[:-1] + (y.shape[-1],))
            u = energy(pp + '.chunk_energy', Hc if share else Hm * Hc).unsqueeze(1)       # [B,1,(Hm*)Hc,L,T]
            a = alpha_m.unsqueeze(2).repeat(1, 1, Hc, 1, 1)
            if Hm > 1 and not share:
                u = u.view(B, Hm, Hc, L, T)
            u = u - u.max(dim=-1, keepdim=True)[0]
            se = torch.clamp(torch.exp(u), min=1e-5)
            sf = args.attn_sharpening_factor
            if w == -1:
                att = se * msum(a * sf / torch.cumsum(se, dim=-1), 0, T - 1)
            else:
                att = se * msum(a * sf / msum(se, w - 1, 0), 0, w - 1)
            att = att.reshape(B, -1, L, T)
        Ht = Hm * Hc
        v = _lin(eouts, sd, pp + '.w_value').view(B, T, Ht, d // Ht)
        cv = torch.einsum('bhlt,bthd->blhd', att, v).reshape(B, L, d)
        return _lin(cv, sd, pp + '.w_out'), alpha

    is_mma = getattr(args, 'transformer_dec_attn_type', 'scaled_dot') == 'mocha'
    first = getattr(args, 'mocha_first_layer', 1) if is_mma else 1
    alphas_all = []
    for l in range(args.dec_n_layers):
        q = '%s.layers.%d' % (p, l)
        yn = _ln(out, sd, q + '.norm1', eps)
        out = out + mha2(yn, yn, tgt_vis, q + '.self_attn', drop_heads=True)
        if l >= first - 1:                      # transformer.py:168: the layers below mocha_first_layer have no source attention
            on = _ln(out, sd, q + '.norm2', eps)
            if is_mma:
                cvm, al = mma(on, q + '.src_attn')
                out = out + cvm
                alphas_all.append(al.masked_fill((ys_out == 3)[:, None, :, None], 0))
            else:
                out = out + mha2(on, eouts, src_vis, q + '.src_attn')
        out = out + ffn(_ln(out, sd, q + '.norm3', eps), sd, q + '.feed_forward', args.transformer_ffn_activation)
    transformer_decoder_att.last_quantity = None
    if is_mma:                                  # transformer.py:444-452
        n_ref = (ys_out != 3).sum(1).to(out.dtype)
        n_pred = sum(torch.abs(a.sum(3).sum(2).sum(1) / a.shape[1]) for a in alphas_all) / len(alphas_all)
        transformer_decoder_att.last_quantity = torch.mean(torch.abs(n_pred - n_ref))
    logits = _lin(_ln(out, sd, p + '.norm_out', eps), sd, p + '.output')
    V = logits.shape[-1]
    lg, yo = logits.view(-1, V), ys_out.view(-1)
    mask = yo == 3
    lsm = args.lsm_prob if training else 0.0
    lp = torch.log_softmax(lg, dim=-1)
    tgt = torch.full_like(lp, lsm / (V - 1))
    tgt.scatter_(1, yo.masked_fill(mask, 0).unsqueeze(1), 1 - lsm)
    rows = -(tgt * lp).sum(1).masked_fill(mask, 0)
    n_tokens = float((~mask).sum())
    loss = rows.sum() / B
    ppl = math.exp(rows.sum().item() / n_tokens)
    acc = float(((lg.argmax(1) == yo) & ~mask).sum()) * 100 / n_tokens
    return loss, acc, ppl


def _xe_lsm(logits, ys_out, lsm, B):
    """criterion.py:45-86 + torch_utils.py:128-145 -> (loss, acc %, ppl); pad id 3."""
    V = logits.shape[-1]
    lg, yo = logits.reshape(-1, V), ys_out.reshape(-1)
    mask = yo == 3
    lp = torch.log_softmax(lg, dim=-1)
    tgt = torch.full_like(lp, lsm / (V - 1))
    tgt.scatter_(1, yo.masked_fill(mask, 0).unsqueeze(1), 1 - lsm)
    rows = -(tgt * lp).sum(1).masked_fill(mask, 0)
    n_tokens = float((~mask).sum())
    return rows.sum() / B, float(((lg.argmax(1) == yo) & ~mask).sum()) * 100 / n_tokens, math.exp(rows.sum().item() / n_tokens)


def rnn_decoder_att(eouts, elens, ys, sd, args, training, quantity_weight, p='dec_fwd', ss_prob=0.0, stableemit=0.0,
                    ctc_trigger_points=None, forced_trigger_points=None):
    """decoders/las.py:618-776 (teacher forcing; ss_prob > 0 = scheduled sampling after it has been triggered,
    :668,675-676: Python's global `random` stream decides per step, the arg-max of the previous step's own output
    distribution is fed back; no LM) with the single-head attentions
    of modules/attention.py:96-181 ('location', 'add') and the training-time MoChA of
    modules/mocha/mocha.py:164-311, hma_train.py:12-67, mocha_train.py:13-58 (one head, additive energies,
    no noise).  -> (loss_att, acc %, ppl, quantity loss or None); quantity_weight is applied by the caller (las.py:486-489)."""
    B, T, D = eouts.shape
    ylens = [len(y) + 1 for y in ys]
    L = max(ylens)
    ys_in = torch.full((B, L), 3, dtype=torch.long)
    ys_out = torch.full((B, L), 3, dtype=torch.long)
    for b, y in enumerate(ys):
        ys_in[b, 0] = 2
        ys_in[b, 1:len(y) + 1] = torch.tensor(y)
        ys_out[b, :len(y)] = torch.tensor(y)
        ys_out[b, len(y)] = 2
    vis = (torch.arange(T)[None, :] < torch.tensor(elens)[:, None])               # [B,T]
    nl, H = args.dec_n_layers, args.dec_n_units
    hx = [eouts.new_zeros(B, H) for _ in range(nl)]
    cx = [eouts.new_zeros(B, H) for _ in range(nl)]
    cv = eouts.new_zeros(B, D)
    emb = F.embedding(ys_in, sd[p + '.embed.weight'], padding_idx=3)
    sc = p + '.score'
    mocha = args.attn_type == 'mocha'
    if mocha:
        key_ma = _lin(eouts, sd, sc + '.monotonic_energy.w_key')
        wv = sd[sc + '.monotonic_energy.v.weight_v']
        v_ma = wv * (sd[sc + '.monotonic_energy.v.weight_g'].view(-1, 1) / wv.norm(dim=1, keepdim=True))
        w = args.mocha_chunk_size
        chunk = w > 1 or w == -1
        if chunk:
            key_ca = _lin(eouts, sd, sc + '.chunk_energy.w_key')
        aw = eouts.new_zeros(B, T)
        aw[:, 0] = 1.0
    else:
        key = _lin(eouts, sd, sc + '.w_key')
        aw = eouts.new_zeros(B, T)

    def excl_cumsum(x):
        return torch.cumsum(torch.cat([x.new_zeros(x.shape[0], 1), x[:, :-1]], dim=-1), dim=-1)

    def moving_sum(x, back, forward):
        return F.conv1d(F.pad(x, [back, forward]).unsqueeze(1), x.new_ones(1, 1, back + forward + 1)).squeeze(1)

    import random
    metric = getattr(args, 'mocha_latency_metric', '')
    if forced_trigger_points is not None:      # las.py:647-649: the boundary of <eos> is the last frame
        forced_trigger_points = forced_trigger_points.clone()
        for b in range(B):
            forced_trigger_points[b, ylens[b] - 1] = elens[b] - 1
    douts, cvs, aws = [], [], []
    for i in range(L):
        y_emb = emb[:, i]
        if i > 0 and ss_prob > 0 and random.random() < ss_prob:
            with torch.no_grad():
                prev = torch.tanh(_lin(torch.cat([douts[-1], cvs[-1]], dim=-1), sd, p + '.output_bn'))
                y_prev = _lin(prev, sd, p + '.output').argmax(-1)
            y_emb = F.embedding(y_prev, sd[p + '.embed.weight'], padding_idx=3)
        dout = torch.cat([y_emb, cv], dim=-1)
        for l in range(nl):
            q = '%s.rnn.%d' % (p, l)
            gates = F.linear(dout, sd[q + '.weight_ih'], sd[q + '.bias_ih']) + F.linear(hx[l], sd[q + '.weight_hh'], sd[q + '.bias_hh'])
            gi, gf, gg, go = gates.chunk(4, dim=1)
            cx[l] = torch.sigmoid(gf) * cx[l] + torch.sigmoid(gi) * torch.tanh(gg)
            hx[l] = torch.sigmoid(go) * torch.tanh(cx[l])
            dout = hx[l]
            if args.dec_n_projs > 0:
                dout = torch.relu(_lin(dout, sd, '%s.proj.%d' % (p, l)))
            if l == 0:
                dscore = dout
        if mocha:
            e = (torch.relu(key_ma + F.linear(dscore, sd[sc + '.monotonic_energy.w_query.weight'])[:, None]) * v_ma.view(1, 1, -1)).sum(-1)
            e = (e + sd[sc + '.monotonic_energy.r']).masked_fill(~vis, NEG_INF32)
            pc = torch.sigmoid(e)
            if stableemit > 0:          # StableEmit once triggered (hma_train.py:43-44)
                pc = (1 - stableemit) * pc
            cp = torch.exp(excl_cumsum(torch.log(torch.clamp(1 - pc, min=args.mocha_eps, max=1.0))))
            den = 1 if args.mocha_no_denominator else torch.clamp(cp, min=args.mocha_eps, max=1.0)
            aw = pc * cp * torch.cumsum(aw / den, dim=-1)
            if 'decot' in metric:              # hma_train.py:59-63
                limit = forced_trigger_points[:, i:i + 1].long() + args.mocha_decot_lookahead
                aw = aw.masked_fill(torch.arange(T)[None, :] > limit, 0)
            att = aw
            if chunk:
                u = (torch.relu(key_ca + F.linear(dscore, sd[sc + '.chunk_energy.w_query.weight'])[:, None])
                     * sd[sc + '.chunk_energy.v.weight'].view(1, 1, -1)).sum(-1).masked_fill(~vis, NEG_INF32)
                u = u - u.max(dim=-1, keepdim=True)[0]
                se = torch.clamp(torch.exp(u), min=1e-5)
                if w == -1:
                    att = se * moving_sum(aw * args.attn_sharpening_factor / torch.cumsum(se, dim=-1), 0, T - 1)
                else:
                    att = se * moving_sum(aw * args.attn_sharpening_factor / moving_sum(se, w - 1, 0), 0, w - 1)
        else:
            tmp = key + F.linear(dscore, sd[sc + '.w_query.weight'])[:, None]
            if args.attn_type == 'location':
                cw = sd[sc + '.conv.weight']
                cf = F.conv2d(aw[:, None, None, :], cw, padding=(0, (cw.shape[-1] - 1) // 2)).squeeze(2).transpose(2, 1)
                tmp = tmp + F.linear(cf, sd[sc + '.w_conv.weight'])
            e = F.linear(torch.tanh(tmp), sd[sc + '.v.weight']).squeeze(-1)
            if args.attn_type == 'triggered_attention':       # attention.py:165-169 (lookahead 2, las.py:233)
                e = e.masked_fill(torch.arange(T)[None, :] > forced_trigger_points[:, i:i + 1].long() + 2, NEG_INF32)
            e = e.masked_fill(~vis, NEG_INF32)
            aw = torch.softmax(e * args.attn_sharpening_factor, dim=-1)
            att = aw
        cv = torch.bmm(att[:, None], eouts).squeeze(1)
        douts.append(dout)
        cvs.append(cv)
        aws.append(aw)
    feats = torch.cat([torch.stack(douts, 1), torch.stack(cvs, 1)], dim=-1)
    logits = _lin(torch.tanh(_lin(feats, sd, p + '.output_bn')), sd, p + '.output')
    loss, acc, ppl = _xe_lsm(logits, ys_out, args.lsm_prob if training else 0.0, B)
    lq = None
    a = torch.stack(aws, 1).masked_fill((ys_out == 3)[:, :, None], 0)              # [B,L,T] (attention padding, :724-726)
    if mocha:
        lq = torch.mean(torch.abs(a.sum(2).sum(1) - (ys_out != 3).sum(1).to(a.dtype)))
    if ctc_trigger_points is not None or ('ctc_sync' not in metric and forced_trigger_points is not None):
        # las.py:757-769: |expected attended frame - reference boundary| per token, over the number of tokens
        pts = ctc_trigger_points if 'ctc_sync' in metric else forced_trigger_points
        exp_pts = (torch.arange(T, dtype=a.dtype)[None, None, :] * a).sum(2)
        rnn_decoder_att.last_latency = torch.abs(exp_pts - pts[:, :L].to(a.dtype)).sum() / float(sum(ylens))
    else:
        rnn_decoder_att.last_latency = None
    return loss, acc, ppl, lq


def _sub_args(args, sub):
    """speech2text.py:172-177: the auxiliary decoder is configured like the main one except for dec_config_sub*"""
    import argparse
    a = argparse.Namespace(**vars(args))
    for k, v in (getattr(args, 'dec_config_' + sub, None) or {}).items():
        setattr(a, k, v)
    return a


def speech2text_loss(sd, args, batch, dtype=torch.float64, training=True, quantity_weight=0.0, bn_out=None,
                     scheduled_sampling=False, stableemit=False, ctc_trigger_points=None, latency_weight=0.0):
    """speech2text.py:271-345 -> (loss, {'loss.ctc', 'loss.transducer'}, eouts, elens).
    bn_out (dict, optional) receives the running statistics a training-mode BatchNorm would leave behind."""
    sd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
    xlens = [len(x) for x in batch['xs']]
    T = max(xlens)
    xs = torch.zeros(len(xlens), T, args.input_dim, dtype=dtype)
    for b, x in enumerate(batch['xs']):
        xs[b, :len(x)] = torch.as_tensor(x, dtype=dtype)
    sub_out = {}
    if 'former' in args.enc_type:
        eouts, elens = encoder_forward(xs, xlens, sd, args, training, bn_out, sub_out)
    else:
        eouts, elens = rnn_encoder_forward(xs, xlens, sd, args, sub_out)
    main_w = args.total_weight - args.sub1_weight - args.sub2_weight
    ctc_w = min(args.ctc_weight, main_w)
    loss = eouts.new_zeros(())
    obs = {'loss.ctc': None, 'loss.transducer': None}
    if ctc_w > 0:
        lc = ctc_loss_ref(ctc_head(eouts, sd, 'dec_fwd.ctc'), elens, batch['ys'], args.ctc_lsm_prob)
        obs['loss.ctc'] = lc.item()
        loss = loss + lc * ctc_w
    if args.dec_type == 'lstm_transducer' and main_w - ctc_w > 0:
        lt = rnnt_branch(eouts, elens, batch['ys'], sd, args)
        obs['loss.transducer'] = lt.item()
        loss = loss + lt * (main_w - ctc_w)
    if args.dec_type == 'transformer' and main_w - ctc_w > 0:
        la, acc, ppl = transformer_decoder_att(eouts, elens, batch['ys'], sd, args, training)
        obs.pop('loss.transducer')
        obs.update({'loss.att': la.item(), 'acc.att': acc, 'ppl.att': ppl})
        if transformer_decoder_att.last_quantity is not None:       # transformer.py:362-366
            obs['loss.quantity'] = transformer_decoder_att.last_quantity.item()
            la = la + transformer_decoder_att.last_quantity * quantity_weight
        loss = loss + la * (main_w - ctc_w)
    if args.dec_type in ('lstm', 'gru') and main_w - ctc_w > 0:
        la, acc, ppl, lq = rnn_decoder_att(eouts, elens, batch['ys'], sd, args, training, quantity_weight,
                                           ss_prob=args.ss_prob if scheduled_sampling else 0.0,
                                           stableemit=args.mocha_stableemit_weight if stableemit else 0.0,
                                           ctc_trigger_points=ctc_trigger_points if training else None,
                                           forced_trigger_points=(torch.as_tensor(batch['trigger_points']).long()
                                                                  if batch.get('trigger_points') is not None and (getattr(args, 'mocha_latency_metric', '') in ('minlt', 'decot', 'decot_ctc_sync')
                                                                       or args.attn_type == 'triggered_attention') else None))
        obs.pop('loss.transducer')
        obs.update({'loss.att': la.item(), 'acc.att': acc, 'ppl.att': ppl})   # (recorded before the quantity loss is added)
        if lq is not None:
            obs['loss.quantity'] = lq.item()
            la = la + lq * quantity_weight
        if getattr(args, 'mocha_latency_metric', '') and rnn_decoder_att.last_latency is not None:
            obs['loss.latency'] = rnn_decoder_att.last_latency.item() if training else 0
            la = la + rnn_decoder_att.last_latency * latency_weight     # las.py:488-491, after trigger_latency_loss()
        loss = loss + la * (main_w - ctc_w)
    # auxiliary tasks (speech2text.py:326-343): forward decoders only
    for sub in ('sub1', 'sub2'):
        w = getattr(args, sub + '_weight', 0.0)
        ys_sub = batch.get('ys_' + sub) or []
        if w <= 0 or len(ys_sub) == 0:
            continue
        a = _sub_args(args, sub)
        es, el = sub_out[sub]
        p = 'dec_fwd_' + sub
        cw = min(getattr(args, 'ctc_weight_' + sub), w)
        if cw > 0:
            lc = ctc_loss_ref(ctc_head(es, sd, p + '.ctc'), el, ys_sub, a.ctc_lsm_prob)
            obs['loss.ctc-' + sub] = lc.item()
            loss = loss + lc * cw
        if w - cw > 0:
            if a.dec_type == 'transformer':
                la, acc, ppl = transformer_decoder_att(es, el, ys_sub, sd, a, training, p=p)
            elif a.dec_type in ('lstm', 'gru'):
                la, acc, ppl, _ = rnn_decoder_att(es, el, ys_sub, sd, a, training, 0.0, p=p)
            else:
                raise NotImplementedError(a.dec_type)
            obs.update({'loss.att-' + sub: la.item(), 'acc.att-' + sub: acc, 'ppl.att-' + sub: ppl})
            loss = loss + la * (w - cw)
    return loss, obs, eouts, elens
