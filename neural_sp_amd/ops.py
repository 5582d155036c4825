"""Host-side plumbing over the C ABI of libnsp_hip.so.

Raw wrappers (``*_raw``) take torch CUDA tensors, pass their device pointers +
shapes + the current HIP stream through ctypes, and return nothing that was not
allocated here by torch.  ``torch.autograd.Function`` subclasses stitch the raw
kernels into autograd.  No op has a CPU/eager fallback: a CPU tensor raises.
"""
import ctypes
import math
import os

import numpy as np
import torch

from neural_sp_amd import _lib
from neural_sp_amd._lib import AttnMaskParams

ACT = {'none': 0, None: 0, '': 0, 'relu': 1, 'swish': 2, 'tanh': 3, 'gelu_accurate': 4, 'gelu': 5}
# NOTE: reference modules/gelu.py: gelu() is the tanh approximation, gelu_accurate() the erf form.

_COMPUTE_MODE = {'mode': 0}  # 0 = bf16 MFMA, 1 = exact fp32 MFMA


def set_compute_mode(mode):
    """'bf16' (bf16 MFMA operands, fp32 accumulate) or 'f32' (exact fp32 MFMA; parity mode)."""
    _COMPUTE_MODE['mode'] = {'bf16': 0, 'f32': 1, 'fp32': 1}[mode]


def get_compute_mode():
    return 'f32' if _COMPUTE_MODE['mode'] == 1 else 'bf16'


class compute_mode(object):
    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = get_compute_mode()
        set_compute_mode(self.mode)

    def __exit__(self, *a):
        set_compute_mode(self.prev)


def _check(rc, what):
    if rc != 0:
        raise RuntimeError('%s failed with code %d' % (what, rc))


def _p(t):
    if t is None:
        return None
    assert t.is_cuda, 'neural_sp_amd ops are HIP-only: got a CPU tensor (no CPU fallback exists)'
    return t.data_ptr()


def on_kernel_device(t):
    """is `t` where the kernels run?  (the HIP device; tests/hipemu points this at the host for the emulated library)"""
    return t.is_cuda


def _require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise AssertionError('neural_sp_amd ops are HIP-only: got a CPU tensor (no CPU fallback exists)')


def _stream():
    # raw C accessors: torch.cuda.current_stream() walks ~10 python frames (8 us per call, 2800
    # calls per step = a quarter of the host time of a step)
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


# Small zero-initialised scratch (atomic accumulators for bias / LayerNorm / depthwise-conv
# gradients: ~230 of them per Conformer-L step) comes out of chunks that are zeroed ONCE with a
# single fill and handed out as slices.  A chunk is never reused or re-zeroed: it dies with its
# last slice, so a slice that autograd keeps as a .grad stays valid.  One pool per stream (the
# fill is only ordered with work on the stream it was enqueued on).  Slices share the chunk's
# autograd version counter, so they are for backward-side scratch / gradients only -- never for
# a tensor that is saved for backward.
_ZERO_POOL = {}
_ZERO_CHUNK = 1 << 20  # bytes


def zeros_small(shape, device, dtype=torch.float32):
    n = 1
    for v in shape:
        n *= int(v)
    nbytes = (n * (4 if dtype == torch.float32 else 2) + 255) // 256 * 256
    if nbytes > _ZERO_CHUNK // 4 or dtype not in (torch.float32, torch.bfloat16):
        return torch.zeros(shape, device=device, dtype=dtype)
    idx = device.index if device.type == 'cuda' and device.index is not None else torch._C._cuda_getDevice()
    key = (idx, torch._C._cuda_getCurrentRawStream(idx))
    ent = _ZERO_POOL.get(key)
    if ent is None or ent[1] + nbytes > _ZERO_CHUNK:
        ent = [torch.zeros((_ZERO_CHUNK,), device=device, dtype=torch.uint8), 0]
        _ZERO_POOL[key] = ent
    off = ent[1]
    ent[1] = off + nbytes
    return ent[0][off:off + n * (4 if dtype == torch.float32 else 2)].view(dtype).view(shape)


# ---- residual-gradient hand-over (round 4).  In a pre-norm block  y = res + dropout(alpha * Linear(h)),  z = LN(y), ...  the
# gradient of y is produced by the LayerNorm backward kernel of the NEXT sub-block, and the first thing the Linear's
# backward does with it is a pass that scales it, re-applies the dropout mask and rounds it to the bf16 operand of its
# gradient GEMMs (grad_prep_colsum: 52 launches, 2.6 ms per step).  The Linear's forward therefore OFFERS what that pass
# needs (alpha, p, seed, offset) on its output tensor; a LayerNorm whose input carries an offer makes its backward
# kernel write the prepared image as well (nsp_layernorm_bwd_prep) and leaves it here under the offer's token; the
# Linear's backward takes it if the gradient it receives is that very tensor (same storage: nothing was accumulated
# into it on the way) and falls back to grad_prep otherwise.
_PREP = {}             # token -> (dx, its version counter, g16 [rows, N] bf16, gsum [N] fp32)
_PREP_TOKEN = [0]
_LAST_PREP = [None]
_PREP_STATS = {'made': 0, 'taken': 0}     # (tests: how many hand-overs the LayerNorm kernels made / the Linears took)


def _prep_offer(ok, alpha, p, seed, offset, N):
    if not (ok and bf16_mode() and N % 8 == 0 and os.environ.get('NSP_LN_PREP', '1') != '0'):
        _LAST_PREP[0] = None
        return None
    _PREP_TOKEN[0] += 1
    _LAST_PREP[0] = (_PREP_TOKEN[0], float(alpha), float(p), int(seed), int(offset), int(N))
    return _PREP_TOKEN[0]


def tag_prep(y):
    """put the offer of the Function that has just produced y on y (call right after .apply)"""
    lp, _LAST_PREP[0] = _LAST_PREP[0], None
    if lp is not None and torch.is_tensor(y) and y.shape[-1] == lp[5]:
        try:
            y._nsp_prep = lp
        except Exception:
            pass
    return y


def _prep_take(token, dy2d):
    """the prepared image filed under `token`, if the gradient that arrived IS the dx it was made from: same memory (the
    entry keeps dx alive, so the address cannot have been recycled, and a tensor with a second owner is never accumulated
    into in place by the engine) and unmodified since (version counter: an in-place `+=` of a second consumer's gradient
    bumps it)"""
    ent = _PREP.pop(token, None) if token is not None else None
    if ent is None:
        return None
    dx, ver, g16, gsum = ent
    if dx.data_ptr() == dy2d.data_ptr() and dx._version == ver and dy2d._version == ver and tuple(g16.shape) == tuple(dy2d.shape):
        _PREP_STATS['taken'] += 1
        return g16, gsum
    return None


def h2d(x, device, dtype=None):
    """Host data (numpy array / list / CPU tensor) -> device tensor through a PINNED staging block,
    asynchronously on the current stream.  A pageable-memory copy blocks the host until everything
    queued on the stream before it has run (measured: 11 such copies = 7 ms of a 20 ms forward and
    a drained HIP queue each time); pinned blocks come from torch's caching host allocator, which
    also keeps a block from being reused before its copy has completed."""
    t = x if torch.is_tensor(x) else torch.as_tensor(x)
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    device = torch.device(device)
    if t.device.type != 'cpu' or device.type == 'cpu':
        return t.to(device)
    p = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    p.copy_(t)
    return p.to(device, non_blocking=True)


def h2d_packed(arrays, device):
    """List of float arrays -> one flat fp32 device tensor (concatenated), assembled directly in
    a pinned block: one host pass and one asynchronous copy."""
    sizes = [int(a.size) for a in arrays]
    p = torch.empty((sum(sizes),), dtype=torch.float32, pin_memory=True)
    pn = p.numpy()
    off = 0
    for a, n in zip(arrays, sizes):
        pn[off:off + n] = np.asarray(a, dtype=np.float32).reshape(-1)
        off += n
    return p.to(device, non_blocking=True)


def _f32c(t):
    if getattr(t, '_nsp_placeholder', False):
        # LayerNorm*Fn in throughput mode hands autograd a stride-0 NaN where only the bf16 image `_nsp16` exists: a consumer
        # that arrives here is about to read it as data (e.g. a GEMM whose width is not a multiple of 8 takes the fp32 path)
        raise RuntimeError('neural_sp_amd: the fp32 image of this LayerNorm output was not written (bf16 image only); its '
                           'consumer cannot use the bf16 image -- set NSP_LN_SKIP32=0 or make the width a multiple of 8')
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


# --------------------------------------------------------------------------
# GEMM
# --------------------------------------------------------------------------
def _dt(t):
    return 1 if t is not None and t.dtype == torch.bfloat16 else 0


import struct as _struct  # noqa: E402

_GEMM_PACK = _struct.Struct('<25qd2qd2Q7q').pack       # nsp_gemm_packed's 38 slots


def gemm_raw(M, N, K, A, a_rs, a_cs, B, b_ks, b_ns, C, ldc,
             batch=(1, 1), a_b=(0, 0), b_b=(0, 0), c_b=(0, 0),
             bias=None, act=0, pre_out=None, dact_src=None, dact=0, res=None,
             alpha=1.0, splitk=1, mode=None, a_off=0, b_off=0, c_off=0,
             dropout_p=0.0, seed=0, offset=0, c_ss=0, colsum_slabs=None):
    """C = epi(A @ B) with arbitrary strides (element offsets *_off into the tensors).
    A/B are both fp32 or both bf16; C / pre_out / dact_src may be fp32 or bf16 with bf16 operands."""
    _require_device(A, B, C, pre_out, dact_src, bias, res)
    esz = A.element_size()
    # one struct.pack + a two-argument C call (nsp_gemm_packed) instead of 39 ctypes argument conversions (~8 us per call,
    # ~3 ms of the host-bound 16-utterance step)
    _check(_lib.lib().nsp_gemm_packed(_GEMM_PACK(
        M, N, K, A.data_ptr() + esz * a_off, a_rs, a_cs, B.data_ptr() + B.element_size() * b_off, b_ks, b_ns,
        C.data_ptr() + C.element_size() * c_off, ldc, batch[0], batch[1], a_b[0], a_b[1], b_b[0], b_b[1],
        c_b[0], c_b[1], bias.data_ptr() if bias is not None else 0, act,
        pre_out.data_ptr() + pre_out.element_size() * c_off if pre_out is not None else 0,
        dact_src.data_ptr() + dact_src.element_size() * c_off if dact_src is not None else 0, dact,
        res.data_ptr() + 4 * c_off if res is not None else 0, alpha, splitk,
        _COMPUTE_MODE['mode'] if mode is None else mode, dropout_p, seed, offset,
        _dt(A), _dt(B), _dt(C), _dt(pre_out), _dt(dact_src), c_ss,
        colsum_slabs.data_ptr() if colsum_slabs is not None else 0), _stream()), 'nsp_gemm')


def bf16_mode():
    return _COMPUTE_MODE['mode'] == 0


def to_bf16(x2d, pad_to=8):
    """bf16 shadow copy [rows, roundup(cols, pad_to)] (zero padded) of a 2-D fp32 tensor."""
    if x2d.dtype == torch.bfloat16:
        return x2d
    rows, cols = x2d.shape
    ld = (cols + pad_to - 1) // pad_to * pad_to
    if ld > (cols + 7) // 8 * 8:
        out = torch.zeros((rows, ld), device=x2d.device, dtype=torch.bfloat16)
    else:
        out = torch.empty((rows, ld), device=x2d.device, dtype=torch.bfloat16)
    _check(_lib.lib().nsp_cast_bf16(_p(x2d), _p(out), (rows), (cols),
                                    (x2d.stride(0)), (ld), _stream()),
           'nsp_cast_bf16')
    return out


# ---- registry of bf16 weight shadows: all stale shadows are rebuilt by ONE kernel launch per optimizer step
# (nsp_shadow_refresh) instead of one cast / transpose / cat per shadow at first use.  Every shadow getter below
# registers what it built: the destination tensor and its parts (source parameter, sub-block, transposed or not).
import weakref  # noqa: E402

_SHADOWS = {}          # (id(owner), attr) -> record
_SHADOW_TABLE = {}       # device -> {'key', 'table', 'n', 'tiles'}: the device-side table of the last refresh
_WEIGHT_EPOCH = [0]


def _wkey(w):
    """Cache key of everything derived from parameter `w`: (Parameter._version, weight epoch).  The version counter alone
    is NOT enough: torch.optim.Adam(fused=True) (and any other fused / multi-tensor update that writes through raw
    pointers) leaves _version untouched while the values change -- measured on the MI355X box in round 3, where it meant
    that bench.py's steps 2.. had been multiplying with the bf16 weights of step 1.  The epoch is bumped by
    refresh_weight_shadows(force=True), which every TRAINING forward calls: all derived copies are then either
    refreshed by its one launch (registered bf16 shadows) or rebuilt at first use (everything else)."""
    return (w._version, _WEIGHT_EPOCH[0])



def _shadow_register(owner, attr, dst, parts, key_params, single=True, extra=()):
    """parts: [(src_param, dst_row0, dst_col0, rows, cols, transpose)] with src viewed as [N, K] row-major.
    The getter caches `(key, dst) + extra` on `owner.attr`, key = _wkey(key_params[0]) (single) or the tuple of the
    key_params' _wkey.  The registry holds WEAK references only -- to the owner, the source parameters AND the shadow
    (which lives on as long as the owner's attribute does): a model that is deleted takes its records with it
    (round-3 advisor finding: the old closures captured parameters and shadows strongly, so every model ever built in
    the process stayed alive and was re-cast on every step)."""
    try:
        rec = {'owner': weakref.ref(owner), 'attr': attr, 'dst': weakref.ref(dst), 'single': bool(single), 'extra': tuple(extra),
               'keyp': [weakref.ref(kp) for kp in key_params],
               'parts': [(weakref.ref(sp), int(r0), int(c0), int(rows), int(cols), bool(tr)) for sp, r0, c0, rows, cols, tr in parts]}
    except TypeError:
        return
    _SHADOWS[(id(owner), attr)] = rec
    _SHADOW_TABLE.clear()


def _shadow_entry(rec):
    """the tuple the getter would cache now, or None when anything it refers to is gone"""
    dst = rec['dst']()
    ps = [r() for r in rec['keyp']]
    if dst is None or any(q is None for q in ps):
        return None
    key = _wkey(ps[0]) if rec['single'] else tuple(_wkey(q) for q in ps)
    return (key, dst) + rec['extra']


def optimizer_stepped():
    """Start a new weight epoch: every copy derived from a parameter (bf16 casts, transposes, stacked / padded images)
    counts as stale from now on.  Speech2Text calls refresh_weight_shadows(force=True) in every training forward, which
    does the same; code that drives ops.linear / ops.LSTMStackFn directly with a FUSED optimizer (which does not move
    Parameter._version) calls this after optimizer.step() -- or lets `track_optimizer(optimizer)` do it."""
    _WEIGHT_EPOCH[0] += 1
    _PREP.clear()


def track_optimizer(optimizer):
    """register a step post-hook on `optimizer` that starts a new weight epoch after every step; returns the handle"""
    return optimizer.register_step_post_hook(lambda *a, **k: optimizer_stepped())


def refresh_weight_shadows(force=False):
    """Bring every registered bf16 shadow whose source parameters have moved on (optimizer step, load_state_dict, weight
    noise) up to date with ONE launch per device on the current stream.  Called at the top of a training / evaluation
    step; the per-weight getters then find their caches fresh.  Shadows not yet registered are built at first use.
    force=True (training steps): start a new weight epoch first -- every derived copy counts as stale whatever the
    parameters' version counters say (see _wkey)."""
    if force:
        _WEIGHT_EPOCH[0] += 1
        _PREP.clear()        # (hand-overs nobody took: a backward that was never run)
    if not _SHADOWS or not bf16_mode():
        return
    stale, dead = {}, []
    for k, rec in _SHADOWS.items():
        owner, dst = rec['owner'](), rec['dst']()
        srcs = [sp() for sp, *_ in rec['parts']]
        if owner is None or dst is None or any(w is None for w in srcs):
            dead.append(k)
            continue
        ent = getattr(owner, rec['attr'], None)
        if ent is None or ent[1] is not dst:
            dead.append(k)                      # invalidated or rebuilt elsewhere: the getter re-registers
            continue
        # the kernel reads raw fp32 rows: a parameter that changed dtype / layout / device since it was registered
        # (model.half(), .to(other device), a non-contiguous .data) goes back to its getter, which handles those
        if any(w.dtype != torch.float32 or not w.is_contiguous() or w.device != dst.device for w in srcs):
            dead.append(k)
            try:
                delattr(owner, rec['attr'])
            except Exception:
                pass
            continue
        now = _shadow_entry(rec)
        if now is None:
            dead.append(k)
        elif ent[0] != now[0]:
            stale.setdefault(dst.device, []).append((rec, dst, srcs, owner, now))
    for k in dead:
        del _SHADOWS[k]
    for dev, recs in stale.items():
        key = tuple((id(rec), dst.data_ptr()) + tuple(w.data_ptr() for w in srcs) for rec, dst, srcs, _, _ in recs)
        tab = _SHADOW_TABLE.get(dev)
        if tab is None or tab['key'] != key:
            rows_, tiles = [], 0
            for rec, dst, srcs, _, _ in recs:
                for (sp, r0, c0, rows, cols, tr), w in zip(rec['parts'], srcs):
                    K = w[0].numel()
                    rows_.append([w.data_ptr(), dst.data_ptr() + 2 * (r0 * dst.stride(0) + c0), rows, cols, K, dst.stride(0),
                                  1 if tr else 0, tiles])
                    tiles += ((rows + 63) // 64) * ((cols + 63) // 64)
            tab = _SHADOW_TABLE[dev] = dict(key=key, table=h2d(torch.tensor(rows_, dtype=torch.int64), dev), n=len(rows_), tiles=tiles)
        if dev.type == 'cuda' and dev != torch.device('cuda', torch.cuda.current_device()):
            with torch.cuda.device(dev):
                _check(_lib.lib().nsp_shadow_refresh(_p(tab['table']), tab['n'], tab['tiles'], _stream()), 'nsp_shadow_refresh')
        else:
            _check(_lib.lib().nsp_shadow_refresh(_p(tab['table']), tab['n'], tab['tiles'], _stream()), 'nsp_shadow_refresh')
        for rec, dst, srcs, owner, now in recs:
            try:
                setattr(owner, rec['attr'], now)      # (the entry computed above: versions and epoch cannot have moved since)
            except Exception:
                pass


def weight_bf16(w):
    """bf16 shadow [N, roundup8(K)] of a parameter (any dim >= 2, flattened to 2-D), cached ON
    the parameter object and refreshed when its version counter moves (one cast per optimizer
    step).  Callers must pass the Parameter itself, not a temporary view of it."""
    ent = getattr(w, '_nsp_bf16', None)
    if ent is not None and ent[0] == _wkey(w) and ent[1].device == w.device:
        return ent[1]
    wb = to_bf16(w.detach().reshape(w.shape[0], -1))
    try:
        w._nsp_bf16 = (_wkey(w), wb)
        _shadow_register(w, '_nsp_bf16', wb, [(w, 0, 0, w.shape[0], w[0].numel(), False)], [w])
    except Exception:
        pass
    return wb


def _pick_splitk(M, N, K, batch=1):
    """Split the reduction when the output grid alone cannot fill 256 CUs."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128) * batch
    if tiles >= 256 or K < 1024:
        return 1
    want = max(1, int(os.environ.get('NSP_SPLITK_TARGET', '512')) // tiles)
    sk = int(max(1, min(want, K // 256, 64)))
    # no empty splits: the kernels cut the k-tiles (64 wide) into ceil(nkt / sk) per split
    nkt = (K + 63) // 64
    per = -(-nkt // sk)
    return -(-nkt // per)


def _wgrad_splitk(N, K, rows, bf16):
    """Reduction splits of dW[N, K] = dY[rows, N]^T X[rows, K]: the library's plan for its 256 x 256 weight-gradient
    kernel (tiles x splits ~ one workgroup per CU) when the shape is eligible, else the 128-tile rule above."""
    if bf16:
        sk = _lib.lib().nsp_wgrad_splitk(N, K, rows)
        if sk > 0:
            return sk
    return _pick_splitk(N, K, rows)


def linear_fwd(x2d, weight, bias=None, act=0, res=None, alpha=1.0, pre_out=None, out=None,
               dropout_p=0.0, seed=0, offset=0, out_bf16=False):
    """y[M,N] = res + dropout(alpha*act(x2d[M,K] @ weight[N,K]^T + bias)).  In bf16 mode the
    operands are bf16 shadows (x2d may already be bf16, weight is cached)."""
    M = x2d.shape[0]
    N, K = weight.shape[0], weight[0].numel()
    if bf16_mode() and K % 8 == 0:
        xa, wb = to_bf16(x2d), weight_bf16(weight)
    else:
        assert x2d.dtype == torch.float32
        xa, wb = x2d, weight.reshape(N, K)
    assert xa.stride(1) == 1 and wb.stride(1) == 1
    if out is None:
        out = torch.empty((M, N), device=x2d.device,
                          dtype=torch.bfloat16 if (out_bf16 and xa.dtype == torch.bfloat16) else torch.float32)
    gemm_raw(M, N, K, xa, xa.stride(0), 1, wb, 1, wb.stride(0), out, out.stride(0),
             bias=bias, act=act, res=res, alpha=alpha, pre_out=pre_out,
             dropout_p=dropout_p, seed=seed, offset=offset)
    return out


def linear_dgrad(dy2d, weight, dact_src=None, dact=0, alpha=1.0, res=None, out=None, out_bf16=False):
    """dx[M,K] = res + alpha*(dy2d[M,N] @ weight[N,K]) * act'(dact_src)."""
    M = dy2d.shape[0]
    N, K = weight.shape[0], weight[0].numel()
    # a bf16 dy whose width is not a multiple of 8 arrives zero-padded to roundup8(N) (grad_prep /
    # to_bf16); the W^T shadow is zero-padded to roundup64(N), so reducing over the padded width is exact
    use16 = bf16_mode() and K % 8 == 0 and (N % 8 == 0 or dy2d.dtype == torch.bfloat16)
    if use16:
        ga = to_bf16(dy2d)
    else:
        assert dy2d.dtype == torch.float32
        ga = dy2d
    if out is None:
        out = torch.empty((M, K), device=dy2d.device,
                          dtype=torch.bfloat16 if (out_bf16 and use16) else torch.float32)
    if use16:
        # W^T shadow [K, roundup64(N)] makes the data gradient a KC x KC product
        wt = _weight_t_shadow(weight, True)
        gemm_raw(M, K, _r8(N), ga, ga.stride(0), 1, wt, 1, wt.stride(0), out, out.stride(0),
                 dact_src=dact_src, dact=dact, alpha=alpha, res=res)
    else:
        wb = weight.reshape(N, K)
        gemm_raw(M, K, N, ga, ga.stride(0), 1, wb, wb.stride(0), 1, out, out.stride(0),
                 dact_src=dact_src, dact=dact, alpha=alpha, res=res)
    return out


def linear_wgrad(dy2d, x2d, alpha=1.0):
    """dW[N,K] = alpha * dy2d[M,N]^T @ x2d[M,K] (split over M, atomically reduced)."""
    M, N = dy2d.shape[0], dy2d.shape[1]
    K = x2d.shape[1]
    if bf16_mode() and (dy2d.dtype == torch.bfloat16 or N % 8 == 0) and (x2d.dtype == torch.bfloat16 or K % 8 == 0):
        ga, xa = to_bf16(dy2d), to_bf16(x2d)
        N, K = min(N, ga.shape[1]), min(K, xa.shape[1])
    else:
        ga, xa = dy2d, x2d
    if ga.dtype != xa.dtype:
        ga, xa = ga.float(), xa.float()
    sk = _wgrad_splitk(N, K, M, ga.dtype == torch.bfloat16)
    if sk > 1 and (N * K) % 4 == 0:
        # split slabs + deterministic reduction (no atomics, no zero fill)
        part = torch.empty((sk, N, K), device=dy2d.device, dtype=torch.float32)
        gemm_raw(N, K, M, ga, 1, ga.stride(0), xa, xa.stride(0), 1, part, K, alpha=alpha, splitk=sk,
                 c_ss=N * K)
        dw = torch.empty((N, K), device=dy2d.device, dtype=torch.float32)
        _check(_lib.lib().nsp_splitk_reduce(_p(part), _p(dw), (sk), (N * K),
                                            _stream()), 'nsp_splitk_reduce')
        return dw
    dw = (torch.zeros if sk > 1 else torch.empty)((N, K), device=dy2d.device, dtype=torch.float32)
    gemm_raw(N, K, M, ga, 1, ga.stride(0), xa, xa.stride(0), 1, dw, K, alpha=alpha, splitk=sk)
    return dw


def grad_prep(dy2d, pre, act, alpha, p, seed, offset, out_bf16, want_colsum=False):
    """alpha * dy * dropout_mask * act'(pre) in one pass; bf16 output feeds the MFMA GEMMs.
    want_colsum: also return the column sums of the result (the Linear's bias gradient), accumulated
    inside the same kernel -> (g, db) instead of g."""
    N = dy2d.shape[1]
    if pre is None and p <= 0 and alpha == 1.0 and not out_bf16:
        return (dy2d, colsum(dy2d)) if want_colsum else dy2d
    n = dy2d.numel()
    if n % 4 or (out_bf16 and N % 8):
        # rare odd widths: unfused path
        g = dy2d
        if p > 0:
            g = dropout_raw(g, p, seed, offset, alpha)
            alpha = 1.0
        if pre is not None:
            g = dact_mul(g, pre if pre.dtype == torch.float32 else pre.float(), act, alpha)
        elif alpha != 1.0:
            g = axpby(g, None, alpha, 0.0)
        g = to_bf16(g) if out_bf16 else g
        return (g, colsum(g)[:N]) if want_colsum else g
    out = torch.empty(dy2d.shape, device=dy2d.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    fuse = want_colsum and N % 4 == 0 and N <= 4096 and os.environ.get('NSP_FUSED_COLSUM', '1') != '0'
    slabs = None
    if fuse:
        slabs = torch.empty((_lib.lib().nsp_grad_prep_slabs(n // N), N), device=dy2d.device, dtype=torch.float32)
    _check(_lib.lib().nsp_grad_prep(_p(dy2d), _p(pre), (_dt(pre)), _p(out),
                                    (int(out_bf16)), (act if pre is not None else 0),
                                    (alpha), (p), (seed),
                                    (offset), (n), N, _p(slabs), _stream()),
           'nsp_grad_prep')
    if want_colsum:
        return out, (colsum(slabs) if fuse else colsum(out)[:N])
    return out


def colsum(x2d, alpha=1.0):
    rows, cols = x2d.shape
    out = zeros_small((cols,), x2d.device)
    fn = _lib.lib().nsp_colsum_bf16 if x2d.dtype == torch.bfloat16 else _lib.lib().nsp_colsum
    _check(fn(_p(x2d), _p(out), (rows), (cols),
              (x2d.stride(0)), (1), _stream()), 'nsp_colsum')
    if alpha != 1.0:
        out.mul_(alpha)
    return out


class LinearFn(torch.autograd.Function):
    """y = res + dropout(alpha * act(x W^T + b)); x is [..., K] (nn.Linear semantics).

    Everything after the contraction (bias, activation, scale, dropout mask, residual) is
    the GEMM epilogue; backward regenerates the dropout mask from (seed, offset).  In bf16
    mode the input's bf16 shadow (not the fp32 tensor) is what is saved for backward."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, res, alpha, dropout_p):
        K = x.shape[-1]
        assert weight.dtype == torch.float32 and weight.is_contiguous()
        N = weight.shape[0]
        use16 = bf16_mode() and K % 8 == 0
        sh = getattr(x, '_nsp16', None)
        if use16 and sh is not None and sh.shape[-1] == K:
            x2d = xa = sh.reshape(-1, K)
        else:
            x2d = (x if x.dtype == torch.bfloat16 else _f32c(x)).reshape(-1, K)
            xa = to_bf16(x2d) if use16 else x2d
        res2d = _f32c(res).reshape(-1, N) if res is not None else None
        pre = None
        if act != 0:
            pre = torch.empty((x2d.shape[0], N), device=x.device,
                              dtype=torch.bfloat16 if use16 else torch.float32)
        seed, offset = next_dropout_seed() if dropout_p > 0 else (0, 0)
        y = linear_fwd(xa, weight, bias, act, res2d, alpha, pre_out=pre,
                       dropout_p=dropout_p, seed=seed, offset=offset)
        ctx.save_for_backward(xa, weight, pre)
        ctx.prep_token = _prep_offer(res is not None and use16 and act == 0, alpha, dropout_p, seed, offset, N)
        ctx.act, ctx.alpha = act, alpha
        ctx.mode = get_compute_mode()      # backward runs in the mode of its forward (nested compute_mode blocks)
        ctx.drop = (dropout_p, seed, offset)
        ctx.has_bias, ctx.has_res = bias is not None, res is not None
        ctx.xshape = x.shape
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        xa, weight, pre = ctx.saved_tensors
        N = weight.shape[0]
        dy2d = _f32c(dy).reshape(-1, N)
        dres = dy if ctx.has_res else None
        p, seed, offset = ctx.drop
        # bf16 image of the gradient; for N % 8 != 0 (e.g. a 10001-word CTC head) it is zero-padded to
        # roundup8(N) columns and the padded rows / entries of dW / db are dropped below
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        with compute_mode(ctx.mode):
            got = _prep_take(getattr(ctx, 'prep_token', None), dy2d)
            dx = dw = db = None
            if got is not None:          # the LayerNorm backward that produced dy has prepared it already
                g, db = got
                db = db[:N] if want_db else None
            else:
                g = grad_prep(dy2d, pre, ctx.act, ctx.alpha, p, seed, offset, xa.dtype == torch.bfloat16, want_colsum=want_db)
                if want_db:
                    g, db = g
                    db = db[:N]
            if ctx.needs_input_grad[0]:
                dx = linear_dgrad(g, weight)[:, :ctx.xshape[-1]].reshape(ctx.xshape)
            if ctx.needs_input_grad[1]:
                dw = linear_wgrad(g, xa)[:N].view(weight.shape)
        return dx, dw, db, None, dres, None, None


def linear(x, weight, bias=None, act='none', res=None, alpha=1.0, dropout_p=0.0):
    return tag_prep(LinearFn.apply(x, weight, bias, ACT[act] if not isinstance(act, int) else act, res,
                                   float(alpha), float(dropout_p)))


def dropout_raw(x, p, seed, offset, alpha=1.0):
    x = _f32c(x)
    y = torch.empty_like(x)
    _check(_lib.lib().nsp_dropout(_p(x), _p(y), (p), (alpha),
                                  (seed), (offset),
                                  (x.numel()), _stream()), 'nsp_dropout')
    return y


class DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p):
        seed, offset = next_dropout_seed()
        ctx.drop = (p, seed, offset)
        return dropout_raw(x, p, seed, offset)

    @staticmethod
    def backward(ctx, dy):
        p, seed, offset = ctx.drop
        return dropout_raw(dy, p, seed, offset), None


def dropout(x, p, training):
    """nn.Dropout semantics with the counter-based mask of the HIP kernels."""
    if not training or p <= 0.0:
        return x
    return DropoutFn.apply(x, float(p))


class ScaleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, alpha):
        ctx.alpha = alpha
        return axpby(x, None, alpha, 0.0)

    @staticmethod
    def backward(ctx, dy):
        return axpby(dy, None, ctx.alpha, 0.0), None


def scale(x, alpha):
    return ScaleFn.apply(x, float(alpha))


class AddFn(torch.autograd.Function):
    """alpha*x + beta*z (both same shape)."""

    @staticmethod
    def forward(ctx, x, z, alpha, beta):
        ctx.ab = (alpha, beta)
        return axpby(x, z, alpha, beta)

    @staticmethod
    def backward(ctx, dy):
        a, b = ctx.ab
        return ((dy if a == 1.0 else axpby(dy, None, a, 0.0)),
                (dy if b == 1.0 else axpby(dy, None, b, 0.0)), None, None)


def add(x, z, alpha=1.0, beta=1.0):
    return AddFn.apply(x, z, float(alpha), float(beta))


def dact_mul(dy, pre, act, alpha=1.0):
    """dy * act'(pre) * alpha (elementwise)."""
    out = torch.empty_like(dy)
    _check(_lib.lib().nsp_dact_mul(_p(dy), _p(pre), _p(out), (act), (alpha),
                                   (dy.numel()), _stream()), 'nsp_dact_mul')
    return out


def axpby(x, z=None, alpha=1.0, beta=1.0, out=None):
    """alpha*x + beta*z on contiguous fp32 tensors."""
    x = _f32c(x)
    z = _f32c(z) if z is not None else None
    out = torch.empty_like(x) if out is None else out
    _check(_lib.lib().nsp_axpby(_p(x), _p(z), _p(out), (alpha), (beta),
                                (x.numel()), _stream()), 'nsp_axpby')
    return out


def act_fwd(x, act):
    x = _f32c(x)
    y = torch.empty_like(x)
    _check(_lib.lib().nsp_act_fwd(_p(x), _p(y), (act), (x.numel()),
                                  _stream()), 'nsp_act_fwd')
    return y


# --------------------------------------------------------------------------
# LayerNorm
# --------------------------------------------------------------------------
def layernorm_fwd_raw(x2d, gamma, beta, eps, act=0, want_pre=False, want16=False, want32=True):
    rows, d = x2d.shape
    y = torch.empty_like(x2d) if want32 else None
    y16 = torch.empty((rows, d), device=x2d.device, dtype=torch.bfloat16) if want16 else None
    mean = torch.empty((rows,), device=x2d.device, dtype=torch.float32)
    rstd = torch.empty((rows,), device=x2d.device, dtype=torch.float32)
    y_pre = torch.empty_like(x2d) if (act != 0 and want_pre) else None
    nbytes = rows * d * (4 + (4 if want32 else 0) + (2 if want16 else 0) + (4 if y_pre is not None else 0))
    with _kev_class('layernorm_fwd', nbytes, 'byte'):
        _check(_lib.lib().nsp_layernorm_fwd(_p(x2d), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd),
                                            (rows), (d), (eps),
                                            (act), _p(y_pre), _p(y16), _stream()), 'nsp_layernorm_fwd')
    return y, mean, rstd, y_pre, y16


def layernorm_bwd_raw(dy2d, x2d, gamma, mean, rstd, y_pre, act=0, dres=None, beta_recompute=None, prep=None):
    """beta_recompute (with y_pre None): the activation's pre-image is recomputed inside the kernel as xhat gamma + beta.
    prep = an offer (token, alpha, p, seed, offset, N) carried by the LayerNorm's input (see _PREP): the kernel also
    writes the prepared bf16 image of dx and its column sums, left in _PREP under the token."""
    rows, d = x2d.shape
    dx = torch.empty_like(x2d)
    dgb = zeros_small((2, d), x2d.device)
    use_prep = prep is not None and act == 0 and prep[5] == d and d % 8 == 0 and bf16_mode()
    nbytes = rows * d * (12 + (4 if y_pre is not None else 0) + (4 if dres is not None else 0) + (2 if use_prep else 0))     # dy, x, dx (+ pre-activation, residual gradient, prepared image)
    with _kev_class('layernorm_bwd', nbytes, 'byte'):
        if use_prep:
            g16 = torch.empty((rows, d), device=x2d.device, dtype=torch.bfloat16)
            gsum = zeros_small((d,), x2d.device)
            _check(_lib.lib().nsp_layernorm_bwd_prep(_p(dy2d), _p(x2d), _p(gamma), _p(mean), _p(rstd), _p(dres), _p(dx),
                                                     (dgb.data_ptr()), (dgb.data_ptr() + 4 * d), _p(g16), _p(gsum),
                                                     prep[1], prep[2], prep[3], prep[4], (rows), (d), _stream()),
                   'nsp_layernorm_bwd_prep')
            _PREP[prep[0]] = (dx, dx._version, g16, gsum)
            _PREP_STATS['made'] += 1
        elif beta_recompute is not None and y_pre is None and act != 0:
            _check(_lib.lib().nsp_layernorm_bwd_recompute(_p(dy2d), _p(x2d), _p(gamma), _p(beta_recompute), _p(mean), _p(rstd),
                                                          _p(dres), _p(dx), (dgb.data_ptr()), (dgb.data_ptr() + 4 * d),
                                                          (rows), (d), (act), _stream()), 'nsp_layernorm_bwd_recompute')
        else:
            _check(_lib.lib().nsp_layernorm_bwd(_p(dy2d), _p(x2d), _p(gamma), _p(mean), _p(rstd), _p(y_pre), _p(dres),
                                                _p(dx), (dgb.data_ptr()),
                                                (dgb.data_ptr() + 4 * d),
                                                (rows), (d), (act),
                                                _stream()), 'nsp_layernorm_bwd')
    return dx, dgb[0], dgb[1]


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, act, gemm_only=False):
        x2d = _f32c(x).reshape(-1, x.shape[-1])
        want16 = bf16_mode() and x2d.shape[1] % 8 == 0
        # gemm_only (the caller's promise: the result is read by a GEMM and nothing else -- the Conformer conv module's
        # LayerNorm + Swish in front of its second pointwise conv): in throughput mode only the bf16 image is written,
        # 2 of the kernel's 14 bytes per element; the fp32 output is a stride-0 NaN (see LayerNormSplitFn) and the
        # activation's pre-image is recomputed in backward from xhat, gamma, beta instead of stored
        lean = bool(gemm_only) and want16 and act != 0 and os.environ.get('NSP_LN_SKIP32', '1') != '0'
        y, mean, rstd, y_pre, y16 = layernorm_fwd_raw(x2d, gamma, beta, eps, act, want_pre=not lean, want16=want16,
                                                      want32=not lean)
        ctx.save_for_backward(x2d, gamma, mean, rstd, y_pre, beta if lean else None)
        ctx.act = act
        ctx.prep = getattr(x, '_nsp_prep', None) if act == 0 else None
        out = _nan_scalar(x.device).expand(x.shape).view(x.shape) if lean else y.view(x.shape)
        if lean:
            out._nsp_placeholder = True
        if y16 is not None:
            out._nsp16 = y16  # bf16 shadow [rows, d] for the consuming GEMM (saves its cast pass)
        return out

    @staticmethod
    def backward(ctx, dy):
        x2d, gamma, mean, rstd, y_pre, beta_re = ctx.saved_tensors
        dy2d = _f32c(dy).reshape(x2d.shape)
        dx, dg, db = layernorm_bwd_raw(dy2d, x2d, gamma, mean, rstd, y_pre, ctx.act, beta_recompute=beta_re, prep=ctx.prep)
        return dx.view(dy.shape), dg, db, None, None, None


def layer_norm(x, gamma, beta, eps=1e-12, act='none', gemm_only=False):
    return LayerNormFn.apply(x, gamma, beta, eps, ACT[act] if not isinstance(act, int) else act, gemm_only)


class LayerNormSplitFn(torch.autograd.Function):
    """(LN(x), x) for the pre-norm residual pattern  x + branch(LN(x)):  the second output is x
    itself, to be used as the residual operand.  x then has ONE consumer in the autograd graph, and
    backward receives the branch gradient and the residual gradient together: their sum is formed
    inside the LayerNorm backward kernel instead of by a separate accumulation kernel per sub-block."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x2d = _f32c(x).reshape(-1, x.shape[-1])
        want16 = bf16_mode() and x2d.shape[1] % 8 == 0
        # Throughput mode: the normalised activations of a pre-norm sub-block are read by GEMMs only (FFN, stacked QKV,
        # pointwise conv), which take the bf16 image `_nsp16`; the fp32 image was 4 of the kernel's 10 bytes per element
        # and nobody's input (round 4: LayerNorm forward 4.3 -> ms/step).  The fp32-typed tensor handed to autograd is a
        # stride-0 NaN: a consumer that ignores `_nsp16` fails loudly instead of reading stale memory.
        skip32 = want16 and os.environ.get('NSP_LN_SKIP32', '1') != '0'
        y, mean, rstd, _, y16 = layernorm_fwd_raw(x2d, gamma, beta, eps, 0, want_pre=False, want16=want16, want32=not skip32)
        ctx.save_for_backward(x2d, gamma, mean, rstd)
        ctx.prep = getattr(x, '_nsp_prep', None)
        if skip32:
            out = _nan_scalar(x.device).expand(x.shape)
            out = out.view(x.shape)
            out._nsp_placeholder = True
        else:
            out = y.view(x.shape)
        if y16 is not None:
            out._nsp16 = y16
        return out, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dres):
        x2d, gamma, mean, rstd = ctx.saved_tensors
        if dy is None:
            return dres, None, None, None
        dy2d = _f32c(dy).reshape(x2d.shape)
        r2d = _f32c(dres).reshape(x2d.shape) if dres is not None else None
        dx, dg, db = layernorm_bwd_raw(dy2d, x2d, gamma, mean, rstd, None, 0, dres=r2d, prep=ctx.prep)
        return dx.view(dy.shape), dg, db, None


class LayerNormPairFn(torch.autograd.Function):
    """(LN_b(LN_a(x)), LN_a(x)): the last LayerNorm of a Conformer block and the first one of the next block
    (conformer_block.py:176-180, :132-133) in one kernel per direction (round 6; bf16 mode only).  The first output is the
    pre-norm input of the next block's feed-forward module -- a stride-0 NaN placeholder carrying the bf16 image `_nsp16`,
    as LayerNormSplitFn's -- the second is y = LN_a(x), the residual stream.  Backward receives the feed-forward module's
    data gradient and the residual gradient, recomputes y from x, and leaves dx (plus the prepared image for the linear
    layer in front of x, if x carries an offer): the gradient of y never exists in memory."""

    @staticmethod
    def forward(ctx, x, ga, ba, eps_a, gb, bb, eps_b):
        x2d = _f32c(x).reshape(-1, x.shape[-1])
        rows, d = x2d.shape
        dev = x2d.device
        y = torch.empty_like(x2d)
        z16 = torch.empty((rows, d), device=dev, dtype=torch.bfloat16)
        st = torch.empty((4, rows), device=dev, dtype=torch.float32)          # mean_a, rstd_a, mean_b, rstd_b
        with _kev_class('layernorm_fwd', rows * d * 10, 'byte'):
            _check(_lib.lib().nsp_layernorm_pair_fwd(_p(x2d), _p(ga), _p(ba), eps_a, _p(gb), _p(bb), eps_b, _p(y), _p(z16),
                                                     st[0].data_ptr(), st[1].data_ptr(), st[2].data_ptr(), st[3].data_ptr(),
                                                     rows, d, _stream()), 'nsp_layernorm_pair_fwd')
        ctx.save_for_backward(x2d, ga, ba, gb, st)
        ctx.prep = getattr(x, '_nsp_prep', None)
        ctx.xshape = x.shape
        zn = _nan_scalar(dev).expand(x.shape).view(x.shape)
        zn._nsp_placeholder = True
        zn._nsp16 = z16
        return zn, y.view(x.shape)

    @staticmethod
    def backward(ctx, dz, dres):
        x2d, ga, ba, gb, st = ctx.saved_tensors
        rows, d = x2d.shape
        dev = x2d.device
        dz2d = _f32c(dz).reshape(rows, d) if dz is not None else torch.zeros_like(x2d)
        r2d = _f32c(dres).reshape(rows, d) if dres is not None else torch.zeros_like(x2d)
        dx = torch.empty_like(x2d)
        dgb = zeros_small((4, d), dev)
        prep = ctx.prep
        use_prep = prep is not None and prep[5] == d and d % 8 == 0 and bf16_mode()
        g16 = gsum = None
        if use_prep:
            g16 = torch.empty((rows, d), device=dev, dtype=torch.bfloat16)
            gsum = zeros_small((d,), dev)
        with _kev_class('layernorm_bwd', rows * d * (16 + (2 if use_prep else 0)), 'byte'):
            _check(_lib.lib().nsp_layernorm_pair_bwd(
                _p(dz2d), _p(r2d), _p(x2d), _p(ga), _p(ba), st[0].data_ptr(), st[1].data_ptr(), _p(gb), st[2].data_ptr(),
                st[3].data_ptr(), _p(dx), dgb[0].data_ptr(), dgb[1].data_ptr(), dgb[2].data_ptr(), dgb[3].data_ptr(),
                _p(g16), _p(gsum), prep[1] if use_prep else 1.0, prep[2] if use_prep else 0.0,
                prep[3] if use_prep else 0, prep[4] if use_prep else 0, rows, d, _stream()), 'nsp_layernorm_pair_bwd')
        if use_prep:
            _PREP[prep[0]] = (dx, dx._version, g16, gsum)
            _PREP_STATS['made'] += 1
        return dx.view(ctx.xshape), dgb[0], dgb[1], None, dgb[2], dgb[3], None


def layer_norm_pair_ok(x, d):
    """can ops.layer_norm_pair replace layer_norm + the next block's layer_norm_split?  (throughput mode, lean outputs)"""
    return (bf16_mode() and d % 8 == 0 and d <= 1024 and os.environ.get('NSP_LN_SKIP32', '1') != '0'
            and os.environ.get('NSP_LN_PAIR', '1') != '0')


def layer_norm_pair(x, ga, ba, eps_a, gb, bb, eps_b):
    """-> (LN_b(LN_a(x)) as a bf16-image placeholder, LN_a(x)); see LayerNormPairFn."""
    return LayerNormPairFn.apply(x, ga, ba, float(eps_a), gb, bb, float(eps_b))


_NAN = {}


def _nan_scalar(dev):
    t = _NAN.get(dev)
    if t is None:
        t = _NAN[dev] = torch.full((1,), float('nan'), device=dev, dtype=torch.float32)
    return t


def layer_norm_split(x, gamma, beta, eps=1e-12):
    """-> (LN(x), x_residual); see LayerNormSplitFn."""
    return LayerNormSplitFn.apply(x, gamma, beta, eps)


# --------------------------------------------------------------------------
# Attention (score GEMMs + fused masked/relative softmax)
# --------------------------------------------------------------------------
def _mask_params(B, H, Tq, Tk, R, clamp, scale, klens, causal=False, lookahead=0,
                 chunk_nl=0, chunk_nc=0, dropout_p=0.0, seed=0, offset=0, p_bf16=0, tk_pitch=0, r_pitch=0):
    p = AttnMaskParams()
    p.B, p.H, p.Tq, p.Tk, p.R = B, H, Tq, Tk, R
    p.clamp = int(clamp)
    p.scale = float(scale)
    p.klens = klens.data_ptr() if klens is not None else None
    p.causal, p.lookahead = int(bool(causal)), int(lookahead)
    p.chunk_nl, p.chunk_nc = int(chunk_nl), int(chunk_nc)
    p.dropout_p = float(dropout_p)
    p.seed, p.offset = int(seed), int(offset)
    p.p_bf16, p.tk_pitch, p.r_pitch = int(p_bf16), int(tk_pitch or Tk), int(r_pitch or R)
    return p


def attn_softmax_fwd_raw(S, QP, mp, Pdrop=None, Pout=None):
    """softmax in place over S (fp32) unless Pout (bf16 image) is given."""
    _check(_lib.lib().nsp_attn_softmax_fwd(_p(S), _p(QP), _p(Pout if Pout is not None else S), _p(Pdrop),
                                           ctypes.byref(mp), _stream()), 'nsp_attn_softmax_fwd')


def attn_softmax_bwd_raw(P, dP, dQP, mp, dS=None):
    """dS in place over dP (fp32) unless dS (bf16 image) is given."""
    _check(_lib.lib().nsp_attn_softmax_bwd(_p(P), _p(dP), _p(dS if dS is not None else dP), _p(dQP),
                                           ctypes.byref(mp), _stream()), 'nsp_attn_softmax_bwd')


class AttentionFn(torch.autograd.Function):
    """softmax((q_ac k^T + shift(q_bd pos^T)) / sqrt(dk), masks) v  on [B,T,H,dk] tensors.

    Replaces relative_multihead_attention.py:179-215 and multihead_attention.py:124-153.
    `pos` is the projected position table [R,H,dk] (None for plain MHA); masks are
    evaluated in-kernel from `klens` (int32, device) + (causal, lookahead, chunk_nl, chunk_nc).
    Returns the context [B,Tq,H*dk]; the probabilities are kept on ctx.aw for plotting.
    """

    @staticmethod
    def forward(ctx, q_ac, q_bd, k, v, pos, klens, cfg):
        B, Tq, H, dk = q_ac.shape
        Tk = k.shape[1]
        d = H * dk
        dev = q_ac.device
        q_ac, k, v = _f32c(q_ac), _f32c(k), _f32c(v)
        q_bd = _f32c(q_bd) if q_bd is not None else q_ac
        clamp = cfg.get('clamp', -1)
        S = torch.empty((B, H, Tq, Tk), device=dev, dtype=torch.float32)
        gemm_raw(Tq, Tk, dk, q_ac, d, 1, k, 1, d, S, Tk, batch=(B, H), a_b=(Tq * d, dk),
                 b_b=(Tk * d, dk), c_b=(H * Tq * Tk, Tq * Tk))
        QP, R = None, 0
        if pos is not None:
            pos = _f32c(pos)
            R = pos.shape[0]
            QP = torch.empty((B, Tq, H, R), device=dev, dtype=torch.float32)
            gemm_raw(B * Tq, R, dk, q_bd, d, 1, pos, 1, d, QP, H * R, batch=(H, 1), a_b=(dk, 0),
                     b_b=(dk, 0), c_b=(R, 0))
        p_drop = cfg.get('dropout', 0.0) if cfg.get('training', False) else 0.0
        seed, offset = next_dropout_seed() if p_drop > 0 else (0, 0)
        mp = _mask_params(B, H, Tq, Tk, R, clamp, 1.0 / math.sqrt(dk), klens,
                          cfg.get('causal', False), cfg.get('lookahead', 0),
                          cfg.get('chunk_nl', 0), cfg.get('chunk_nc', 0), p_drop, seed, offset)
        Pd = torch.empty_like(S) if p_drop > 0 else None
        attn_softmax_fwd_raw(S, QP, mp, Pd)
        P = S
        Puse = Pd if Pd is not None else P
        O = torch.empty((B, Tq, H, dk), device=dev, dtype=torch.float32)
        gemm_raw(Tq, dk, Tk, Puse, Tk, 1, v, d, 1, O, d, batch=(B, H), a_b=(H * Tq * Tk, Tq * Tk),
                 b_b=(Tk * d, dk), c_b=(Tq * d, dk))
        ctx.save_for_backward(q_ac, q_bd, k, v, pos, P, Pd, klens)
        ctx.mp_args = (B, H, Tq, Tk, R, clamp, 1.0 / math.sqrt(dk), cfg.get('causal', False),
                       cfg.get('lookahead', 0), cfg.get('chunk_nl', 0), cfg.get('chunk_nc', 0),
                       p_drop, seed, offset)
        ctx.same_q = q_bd is q_ac
        ctx.mark_non_differentiable(P)
        return O.view(B, Tq, d), P

    @staticmethod
    def backward(ctx, dO, _dP_unused):
        q_ac, q_bd, k, v, pos, P, Pd, klens = ctx.saved_tensors
        (B, H, Tq, Tk, R, clamp, scale, causal, lookahead, nl, nc, p_drop, seed, offset) = ctx.mp_args
        dk_ = q_ac.shape[-1]
        d = H * dk_
        dev = dO.device
        dO = _f32c(dO).view(B, Tq, H, dk_)
        Puse = Pd if Pd is not None else P
        # dP = dO v^T
        dP = torch.empty((B, H, Tq, Tk), device=dev, dtype=torch.float32)
        gemm_raw(Tq, Tk, dk_, dO, d, 1, v, 1, d, dP, Tk, batch=(B, H), a_b=(Tq * d, dk_),
                 b_b=(Tk * d, dk_), c_b=(H * Tq * Tk, Tq * Tk))
        # dV = Puse^T dO
        dV = torch.empty((B, Tk, H, dk_), device=dev, dtype=torch.float32)
        gemm_raw(Tk, dk_, Tq, Puse, 1, Tk, dO, d, 1, dV, d, batch=(B, H),
                 a_b=(H * Tq * Tk, Tq * Tk), b_b=(Tq * d, dk_), c_b=(Tk * d, dk_))
        mp = _mask_params(B, H, Tq, Tk, R, clamp, scale, klens, causal, lookahead, nl, nc,
                          p_drop, seed, offset)
        dQP = torch.empty((B, Tq, H, R), device=dev, dtype=torch.float32) if pos is not None else None
        attn_softmax_bwd_raw(P, dP, dQP, mp)
        dS = dP
        # dq_ac = dS k ; dk = dS^T q_ac
        dQ = torch.empty((B, Tq, H, dk_), device=dev, dtype=torch.float32)
        gemm_raw(Tq, dk_, Tk, dS, Tk, 1, k, d, 1, dQ, d, batch=(B, H), a_b=(H * Tq * Tk, Tq * Tk),
                 b_b=(Tk * d, dk_), c_b=(Tq * d, dk_))
        dK = torch.empty((B, Tk, H, dk_), device=dev, dtype=torch.float32)
        gemm_raw(Tk, dk_, Tq, dS, 1, Tk, q_ac, d, 1, dK, d, batch=(B, H),
                 a_b=(H * Tq * Tk, Tq * Tk), b_b=(Tq * d, dk_), c_b=(Tk * d, dk_))
        dQbd = dpos = None
        if pos is not None:
            # d q_bd = dQP pos   (accumulated into dQ when q_bd is q_ac)
            tgt = dQ if ctx.same_q else torch.empty_like(dQ)
            gemm_raw(B * Tq, dk_, R, dQP, H * R, 1, pos, d, 1, tgt, d, batch=(H, 1), a_b=(R, 0),
                     b_b=(dk_, 0), c_b=(dk_, 0), res=tgt if ctx.same_q else None)
            if not ctx.same_q:
                dQbd = tgt
            if ctx.needs_input_grad[4]:
                dpos = torch.zeros((R, H, dk_), device=dev, dtype=torch.float32)
                sk = max(1, min(64, (B * Tq) // 256))
                gemm_raw(R, dk_, B * Tq, dQP, 1, H * R, q_bd, d, 1, dpos, d, batch=(H, 1),
                         a_b=(R, 0), b_b=(dk_, 0), c_b=(dk_, 0), splitk=sk)
                if sk == 1:
                    pass
        return dQ, dQbd, dK, dV, dpos, None, None


_DROPOUT_STATE = {'seed': None, 'counter': 0}
_M64 = (1 << 64) - 1


def _mix64(x):
    """splitmix64 finaliser (host side, once per dropout site)."""
    x = (x + 0x9E3779B97F4A7C15) & _M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M64
    return x ^ (x >> 31)


def manual_dropout_seed(seed):
    """Fix the base seed of the dropout masks explicitly (and restart the site counter)."""
    _DROPOUT_STATE['seed'] = int(seed) & _M64
    _DROPOUT_STATE['counter'] = 0


def _dropout_base_seed():
    """Base seed of this process: taken ONCE, at the first dropout site, from torch's seed (what
    train.py:57-58 sets with torch.manual_seed(args.seed)) and the data-parallel rank, so that runs
    are reproducible from the training seed and replicas draw different masks."""
    s = _DROPOUT_STATE['seed']
    if s is None:
        rank = 0
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                rank = dist.get_rank()
            else:
                rank = int(os.environ.get('RANK', '0'))
        except Exception:
            rank = 0
        s = _mix64((int(torch.initial_seed()) & _M64) ^ _mix64(0x5EED + rank))
        _DROPOUT_STATE['seed'] = s
    return s


def next_dropout_seed():
    """(seed, offset) for one dropout site.  Every site gets its own 64-bit stream seed =
    mix(base seed, site counter); the offset (element index base) stays 0, so no number of sites
    or elements can make two sites' counters collide or overflow the 64-bit index."""
    _DROPOUT_STATE['counter'] += 1
    return _mix64(_dropout_base_seed() ^ ((_DROPOUT_STATE['counter'] * 0xD1342543DE82EF95) & _M64)), 0


def invalidate_weight_shadows(module):
    """Drop every cached bf16 / transposed / stacked weight shadow of `module`'s parameters.  The caches are keyed by
    (Parameter._version, weight epoch) -- see _wkey: a training forward starts a new epoch by itself; in-place writes
    through `.data` OUTSIDE training steps (EMA code, manual checkpoint surgery before an evaluation) change neither,
    call this after such an update."""
    for p in module.parameters():
        for name in ('_nsp_bf16', '_nsp_t16', '_nsp_t32', '_nsp_stack16', '_nsp_stackt16',
                     '_nsp_lstm_cat', '_nsp_lstm_catT', '_nsp_rowpad16'):
            if hasattr(p, name):
                try:
                    delattr(p, name)
                except AttributeError:
                    pass
            _SHADOWS.pop((id(p), name), None)
    _SHADOW_TABLE.clear()


# --------------------------------------------------------------------------
# GLU / depthwise conv / max-pool (Conformer conv module, subsampling)
# --------------------------------------------------------------------------
class GLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _f32c(x)
        C = x.shape[-1] // 2
        rows = x.numel() // (2 * C)
        y = torch.empty(x.shape[:-1] + (C,), device=x.device, dtype=torch.float32)
        _check(_lib.lib().nsp_glu_fwd(_p(x), _p(y), (rows), (C), _stream()),
               'nsp_glu_fwd')
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        dy = _f32c(dy)
        C = x.shape[-1] // 2
        rows = x.numel() // (2 * C)
        dx = torch.empty_like(x)
        _check(_lib.lib().nsp_glu_bwd(_p(x), _p(dy), _p(dx), (rows), (C),
                                      _stream()), 'nsp_glu_bwd')
        return dx


def glu(x):
    return GLUFn.apply(x)


class LinearGLUFn(torch.autograd.Function):
    """glu(x W^T + b) for the Conformer conv module's first pointwise conv (conformer_convolution.py:107-112) in bf16
    mode: the GEMM epilogue writes the [M, 2C] pre-activation as bf16 -- that image is all that forward's GLU reads and
    all that backward keeps -- and backward's GLU kernel emits d(pre) directly as the bf16 operand of the two gradient
    GEMMs, with the bias gradient's column sums accumulated on the way.  (As linear + glu the module moved a fp32
    [M, 2C] tensor four times per direction: 16 KB per frame in backward against 6 KB here.)"""

    @staticmethod
    def forward(ctx, x, weight, bias):
        K = x.shape[-1]
        N = weight.shape[0]
        C = N // 2
        sh = getattr(x, '_nsp16', None)
        if sh is not None and sh.shape[-1] == K:
            xa = sh.reshape(-1, K)
        else:
            xa = to_bf16((x if x.dtype == torch.bfloat16 else _f32c(x)).reshape(-1, K))
        M = xa.shape[0]
        h2 = linear_fwd(xa, weight, bias, out_bf16=True)                        # bf16 [M, 2C] (the Parameter itself: its shadow is cached on it)
        y = torch.empty(x.shape[:-1] + (C,), device=x.device, dtype=torch.float32)
        _check(_lib.lib().nsp_glu_fwd_b16(_p(h2), _p(y), M, C, _stream()), 'nsp_glu_fwd_b16')
        ctx.save_for_backward(xa, weight, h2)
        ctx.mode = get_compute_mode()
        ctx.has_bias = bias is not None
        ctx.xshape = x.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        xa, weight, h2 = ctx.saved_tensors
        N = weight.shape[0]
        C = N // 2
        M = xa.shape[0]
        dy2d = _f32c(dy).reshape(M, C)
        with compute_mode(ctx.mode):
            g = torch.empty((M, N), device=dy.device, dtype=torch.bfloat16)
            slabs = torch.empty((_lib.lib().nsp_grad_prep_slabs(M), N), device=dy.device, dtype=torch.float32)
            _check(_lib.lib().nsp_glu_bwd_b16(_p(h2), _p(dy2d), _p(g), _p(slabs), M, C, _stream()), 'nsp_glu_bwd_b16')
            dx = dw = db = None
            if ctx.has_bias and ctx.needs_input_grad[2]:
                db = colsum(slabs)
            if ctx.needs_input_grad[0]:
                dx = linear_dgrad(g, weight).reshape(ctx.xshape)
            if ctx.needs_input_grad[1]:
                dw = linear_wgrad(g, xa).view(weight.shape)
        return dx, dw, db


def linear_glu(x, weight, bias=None):
    """glu(linear(x)): one autograd node on a bf16 intermediate in throughput mode (LinearGLUFn), linear + glu otherwise"""
    N, K = weight.shape[0], x.shape[-1]
    C = N // 2
    if (bf16_mode() and K % 8 == 0 and C % 8 == 0 and C // 4 <= 256 and 256 % (C // 4) == 0
            and os.environ.get('NSP_LINEAR_GLU', '1') != '0'):
        return LinearGLUFn.apply(x, weight, bias)
    return glu(linear(x, weight, bias))


def _dwconv_fwd(x, wt, bias, k, pad, flip):
    B, T, C = x.shape
    y = torch.empty_like(x)
    _check(_lib.lib().nsp_dwconv1d_fwd(_p(x), _p(wt), _p(bias), _p(y), (B), (T),
                                       (C), (k), (pad),
                                       (flip), _stream()), 'nsp_dwconv1d_fwd')
    return y


class DepthwiseConv1dFn(torch.autograd.Function):
    """x [B,T,C] channels-last; weight [C,1,k] (nn.Conv1d groups=C layout); causal pads left k-1."""

    @staticmethod
    def forward(ctx, x, weight, bias, causal):
        x = _f32c(x)
        C, _, k = weight.shape
        wt = weight.reshape(C, k).t().contiguous()  # tap-major [k, C]
        pad = (k - 1) if causal else (k - 1) // 2
        y = _dwconv_fwd(x, wt, bias, k, pad, 0)
        ctx.save_for_backward(x, wt)
        ctx.k, ctx.pad, ctx.has_bias = k, pad, bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wt = ctx.saved_tensors
        dy = _f32c(dy)
        B, T, C = x.shape
        k, pad = ctx.k, ctx.pad
        dx = _dwconv_fwd(dy, wt, None, k, k - 1 - pad, 1) if ctx.needs_input_grad[0] else None
        if k <= 15 and ((k + 1) * C) % 4 == 0:
            tsplit = max(1, min(16, T // 48))
            part = torch.empty((B * tsplit, (k + 1) * C), device=x.device, dtype=torch.float32)
            _check(_lib.lib().nsp_dwconv1d_wgrad_slabs(_p(x), _p(dy), _p(part), tsplit, B, T, C, k, pad, _stream()),
                   'nsp_dwconv1d_wgrad_slabs')
            buf = torch.empty((k + 1, C), device=x.device, dtype=torch.float32)
            _check(_lib.lib().nsp_splitk_reduce(_p(part), _p(buf), B * tsplit, (k + 1) * C, _stream()),
                   'nsp_splitk_reduce')
            dw = buf[:k].t().contiguous().view(C, 1, k)
            db = buf[k] if ctx.has_bias else None
            return dx, dw, db, None
        buf = zeros_small((k + 1, C), x.device)
        _check(_lib.lib().nsp_dwconv1d_wgrad(_p(x), _p(dy), (buf.data_ptr()),
                                             (buf.data_ptr() + 4 * k * C),
                                             (B), (T), (C),
                                             (k), (pad), _stream()),
               'nsp_dwconv1d_wgrad')
        dw = buf[:k].t().contiguous().view(C, 1, k)
        db = buf[k] if ctx.has_bias else None
        return dx, dw, db, None


def depthwise_conv1d(x, weight, bias, causal=False):
    return DepthwiseConv1dFn.apply(x, weight, bias, causal)


class LinearGLUDwconvFn(torch.autograd.Function):
    """depthwise_conv(glu(x W^T + b)) of the Conformer conv module (conformer_convolution.py:107-113) as ONE autograd
    node in bf16 mode (round 6).  The pointwise conv's GEMM leaves its [M, 2C] output as a bf16 image; the depthwise
    kernels apply the GLU to it as they load it, and the conv's data-gradient kernel finishes with the GLU's backward
    (bf16 operand of the pointwise conv's gradient GEMMs + its column sums = that conv's bias gradient).  No fp32 GLU
    output and no fp32 gradient of it exist: 16 bytes per element of hand-overs and two launches per direction fewer
    than LinearGLUFn + DepthwiseConv1dFn, same arithmetic."""

    @staticmethod
    def forward(ctx, x, weight, bias, dw_weight, dw_bias):
        K = x.shape[-1]
        N = weight.shape[0]
        C = N // 2
        B, T = x.shape[0], x.shape[1]
        sh = getattr(x, '_nsp16', None)
        if sh is not None and sh.shape[-1] == K:
            xa = sh.reshape(-1, K)
        else:
            xa = to_bf16((x if x.dtype == torch.bfloat16 else _f32c(x)).reshape(-1, K))
        h2 = linear_fwd(xa, weight, bias, out_bf16=True)                        # bf16 [M, 2C]
        k = dw_weight.shape[-1]
        wt = dw_weight.reshape(C, k).t().contiguous()                           # tap-major [k, C]
        pad = (k - 1) // 2
        y = torch.empty((B, T, C), device=x.device, dtype=torch.float32)
        _check(_lib.lib().nsp_dwconv1d_glu_fwd(_p(h2), _p(wt), _p(dw_bias), _p(y), B, T, C, k, pad, _stream()),
               'nsp_dwconv1d_glu_fwd')
        ctx.save_for_backward(xa, weight, h2, wt)
        ctx.mode = get_compute_mode()
        ctx.has_bias, ctx.has_dw_bias = bias is not None, dw_bias is not None
        ctx.dims = (B, T, C, k, pad)
        ctx.xshape = x.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        xa, weight, h2, wt = ctx.saved_tensors
        B, T, C, k, pad = ctx.dims
        M, N = B * T, 2 * C
        dev = dy.device
        dy = _f32c(dy)
        L = _lib.lib()
        dx = dw = db = ddw = ddb = None
        with compute_mode(ctx.mode):
            if ctx.needs_input_grad[3] or ctx.needs_input_grad[4]:
                tsplit = max(1, min(16, T // 48))      # (longer ranges = fewer halo rows measured SLOWER: 114 vs 105 us per call)
                part = torch.empty((B * tsplit, (k + 1) * C), device=dev, dtype=torch.float32)
                _check(L.nsp_dwconv1d_glu_wgrad_slabs(_p(h2), _p(dy), _p(part), tsplit, B, T, C, k, pad, _stream()),
                       'nsp_dwconv1d_glu_wgrad_slabs')
                buf = torch.empty((k + 1, C), device=dev, dtype=torch.float32)
                _check(L.nsp_splitk_reduce(_p(part), _p(buf), B * tsplit, (k + 1) * C, _stream()), 'nsp_splitk_reduce')
                ddw = buf[:k].t().contiguous().view(C, 1, k)
                ddb = buf[k] if ctx.has_dw_bias else None
            if any(ctx.needs_input_grad[:3]):
                g = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
                slabs = torch.empty((L.nsp_dwconv1d_glu_bwd_slabs(B, T, C), N), device=dev, dtype=torch.float32)
                _check(L.nsp_dwconv1d_glu_bwd(_p(h2), _p(dy), _p(wt), _p(g), _p(slabs), B, T, C, k, pad, _stream()),
                       'nsp_dwconv1d_glu_bwd')
                if ctx.has_bias and ctx.needs_input_grad[2]:
                    db = colsum(slabs)
                if ctx.needs_input_grad[0]:
                    dx = linear_dgrad(g, weight).reshape(ctx.xshape)
                if ctx.needs_input_grad[1]:
                    dw = linear_wgrad(g, xa).view(weight.shape)
        return dx, dw, db, ddw, ddb


def linear_glu_dwconv(x, weight, bias, dw_weight, dw_bias, causal=False):
    """depthwise_conv1d(glu(linear(x))): fused node in throughput mode where the kernels apply (non-causal, k in {7, 15},
    the LinearGLUFn shapes); the two-node form otherwise (NSP_GLU_DWCONV=0 forces it)."""
    N, K = weight.shape[0], x.shape[-1]
    C = N // 2
    k = dw_weight.shape[-1]
    if (bf16_mode() and x.dim() == 3 and not causal and k in (7, 15) and K % 8 == 0 and C % 8 == 0 and C // 4 <= 256
            and 256 % (C // 4) == 0 and os.environ.get('NSP_LINEAR_GLU', '1') != '0'
            and os.environ.get('NSP_GLU_DWCONV', '1') != '0'):
        return LinearGLUDwconvFn.apply(x, weight, bias, dw_weight, dw_bias)
    return depthwise_conv1d(linear_glu(x, weight, bias), dw_weight, dw_bias, causal)


class MaxPool1dFn(torch.autograd.Function):
    """MaxPool1d(kernel=stride=factor, ceil_mode=True) over time of [B,T,C]."""

    @staticmethod
    def forward(ctx, x, factor):
        x = _f32c(x)
        B, T, C = x.shape
        To = (T + factor - 1) // factor
        y = torch.empty((B, To, C), device=x.device, dtype=torch.float32)
        am = torch.empty((B, To, C), device=x.device, dtype=torch.int32)
        _check(_lib.lib().nsp_maxpool1d_fwd(_p(x), _p(y), _p(am), (B), (T),
                                            (C), (factor), _stream()),
               'nsp_maxpool1d_fwd')
        ctx.save_for_backward(am)
        ctx.dims = (B, T, C, factor)
        return y

    @staticmethod
    def backward(ctx, dy):
        am, = ctx.saved_tensors
        B, T, C, factor = ctx.dims
        dy = _f32c(dy)
        dx = torch.empty((B, T, C), device=dy.device, dtype=torch.float32)
        _check(_lib.lib().nsp_maxpool1d_bwd(_p(dy), _p(am), _p(dx), (B), (T),
                                            (C), (factor), _stream()),
               'nsp_maxpool1d_bwd')
        return dx, None


def maxpool1d_time(x, factor):
    return MaxPool1dFn.apply(x, factor)


class TimeWindowFn(torch.autograd.Function):
    """Strided windows over time of [B,T,C] (the subsamplers of subsampling.py other than max-pool):
    gather=False: y[b,to,:] = s(to) * sum_j x[b, to*stride+j-pad, :]  ([B,To,C]; s = 1/coverage if mean)
    gather=True:  y[b,to,j*C:(j+1)*C] = x[b, to*stride+j-pad, :]        ([B,To,k*C], im2col for a GEMM)"""

    @staticmethod
    def forward(ctx, x, k, stride, pad, To, mean, gather):
        x = _f32c(x)
        B, T, C = x.shape
        y = torch.empty((B, To, k * C if gather else C), device=x.device, dtype=torch.float32)
        L = _lib.lib()
        if gather:
            _check(L.nsp_time_window_gather_fwd(_p(x), _p(y), B, T, To, C, k, stride, pad, _stream()),
                   'nsp_time_window_gather_fwd')
        else:
            _check(L.nsp_time_window_sum_fwd(_p(x), _p(y), B, T, To, C, k, stride, pad, int(mean), _stream()),
                   'nsp_time_window_sum_fwd')
        ctx.cfg = (B, T, To, C, k, stride, pad, int(mean), gather)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, T, To, C, k, stride, pad, mean, gather = ctx.cfg
        dy = _f32c(dy)
        dx = torch.empty((B, T, C), device=dy.device, dtype=torch.float32)
        L = _lib.lib()
        if gather:
            _check(L.nsp_time_window_gather_bwd(_p(dy), _p(dx), B, T, To, C, k, stride, pad, _stream()),
                   'nsp_time_window_gather_bwd')
        else:
            _check(L.nsp_time_window_sum_bwd(_p(dy), _p(dx), B, T, To, C, k, stride, pad, mean, _stream()),
                   'nsp_time_window_sum_bwd')
        return dx, None, None, None, None, None, None


def time_window_sum(x, k, stride, pad, To, mean=False):
    return TimeWindowFn.apply(x, int(k), int(stride), int(pad), int(To), bool(mean), False)


def time_window_gather(x, k, stride, pad, To):
    return TimeWindowFn.apply(x, int(k), int(stride), int(pad), int(To), False, True)


def _flip_mask_raw(x, x_ld, y, y_ld, lens_dev, B, T, C, flip):
    _check(_lib.lib().nsp_time_flip_mask(_p(x), x_ld, _p(y), y_ld, _p(lens_dev), B, T, C, int(flip), _stream()),
           'nsp_time_flip_mask')


class TimeFlipMaskFn(torch.autograd.Function):
    """y[b,t] = x[b, flip ? len_b-1-t : t] for t < len_b, 0 beyond (what packing does around an LSTM,
    encoders/rnn.py:534-541).  Self-adjoint."""

    @staticmethod
    def forward(ctx, x, lens_dev, flip):
        x = _f32c(x)
        B, T, C = x.shape
        y = torch.empty_like(x)
        _flip_mask_raw(x, C, y, C, lens_dev, B, T, C, flip)
        ctx.save_for_backward(lens_dev)
        ctx.flip = flip
        return y

    @staticmethod
    def backward(ctx, dy):
        lens_dev, = ctx.saved_tensors
        dy = _f32c(dy)
        B, T, C = dy.shape
        dx = torch.empty_like(dy)
        _flip_mask_raw(dy, C, dx, C, lens_dev, B, T, C, ctx.flip)
        return dx, None, None


def time_flip_mask(x, lens_dev, flip):
    return TimeFlipMaskFn.apply(x, lens_dev, bool(flip))


class BiDirMergeFn(torch.autograd.Function):
    """[mask(y_fwd) | flip_mask(y_bwd_reversed)] -> `[B,T,2H]`: the padded output of a packed bidirectional LSTM
    layer from its two left-to-right runs, written straight into the halves of one buffer (no concat pass)."""

    @staticmethod
    def forward(ctx, y_f, y_r, lens_dev):
        y_f, y_r = _f32c(y_f), _f32c(y_r)
        B, T, H = y_f.shape
        out = torch.empty((B, T, 2 * H), device=y_f.device, dtype=torch.float32)
        _flip_mask_raw(y_f, H, out, 2 * H, lens_dev, B, T, H, 0)
        _flip_mask_raw(y_r, H, out[:, :, H:], 2 * H, lens_dev, B, T, H, 1)
        ctx.save_for_backward(lens_dev)
        return out

    @staticmethod
    def backward(ctx, dout):
        lens_dev, = ctx.saved_tensors
        dout = _f32c(dout)
        B, T, H2 = dout.shape
        H = H2 // 2
        d_f = torch.empty((B, T, H), device=dout.device, dtype=torch.float32)
        d_r = torch.empty((B, T, H), device=dout.device, dtype=torch.float32)
        _flip_mask_raw(dout, H2, d_f, H, lens_dev, B, T, H, 0)
        _flip_mask_raw(dout[:, :, H:], H2, d_r, H, lens_dev, B, T, H, 1)
        return d_f, d_r, None


def bidir_merge(y_f, y_r, lens_dev):
    return BiDirMergeFn.apply(y_f, y_r, lens_dev)


def _col_part(M, C, device):
    """workspace of the two-level column reductions: [slabs, 2, C]"""
    return torch.empty((_lib.lib().nsp_col_reduce_slabs(M), 2, C), device=device, dtype=torch.float32)


class BatchNormActFn(torch.autograd.Function):
    """act(BatchNorm1d(x)) over the rows of x [..., C] (conformer_convolution.py:119-122 with
    normalization='batch_norm': statistics over all B*T frames, padded ones included, like the
    reference).  training: batch statistics + in-place running-statistics update (buffers are passed as
    plain tensors); eval: running statistics."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, num_batches_tracked, training, momentum,
                eps, act):
        x = _f32c(x)
        C = x.shape[-1]
        M = x.numel() // C
        L = _lib.lib()
        y = torch.empty_like(x)
        if training:
            part = _col_part(M, C, x.device)
            mean = torch.empty((C,), device=x.device, dtype=torch.float32)
            scale = torch.empty((C,), device=x.device, dtype=torch.float32)
            _check(L.nsp_bn_stats(_p(x), M, C, eps, momentum, _p(part), _p(mean), _p(scale),
                                  _p(running_mean), _p(running_var), _p(num_batches_tracked), _stream()),
                   'nsp_bn_stats')
            is_var = 0
        else:
            mean, scale, is_var = running_mean, running_var, 1
        _check(L.nsp_bn_act_fwd(_p(x), _p(mean), _p(scale), is_var, eps, _p(gamma), _p(beta), act, _p(y),
                                M, C, _stream()), 'nsp_bn_act_fwd')
        if training:
            ctx.save_for_backward(x, gamma, beta, mean, scale)
        else:
            # the running buffers are updated in place by later training steps: keep this call's values
            ctx.save_for_backward(x, gamma, beta, mean.clone(), scale.clone())
        ctx.cfg = (M, C, is_var, eps, act, int(bool(training)))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, scale = ctx.saved_tensors
        M, C, is_var, eps, act, training = ctx.cfg
        dy = _f32c(dy)
        part = _col_part(M, C, x.device)
        dgamma = torch.empty((C,), device=x.device, dtype=torch.float32)
        dbeta = torch.empty((C,), device=x.device, dtype=torch.float32)
        dx = torch.empty_like(x)
        _check(_lib.lib().nsp_bn_act_bwd(_p(x), _p(dy), _p(mean), _p(scale), is_var, eps, _p(gamma), _p(beta),
                                         act, training, _p(part), _p(dgamma), _p(dbeta), _p(dx), M, C,
                                         _stream()), 'nsp_bn_act_bwd')
        return dx, dgamma, dbeta, None, None, None, None, None, None, None


def batch_norm_act(x, bn, training, act='none'):
    """`bn`: an nn.BatchNorm1d used as the parameter / buffer container (affine, track_running_stats)."""
    if not (bn.affine and bn.track_running_stats) or bn.momentum is None:
        raise NotImplementedError('BatchNorm1d without affine / running statistics / momentum')
    return BatchNormActFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                bn.num_batches_tracked, bool(training), float(bn.momentum), float(bn.eps),
                                ACT[act])


class GroupNorm2ActFn(torch.autograd.Function):
    """act(GroupNorm(C/2 groups)(x)) over the rows of x [..., C]: every group is a pair of adjacent
    channels (conformer_convolution.py:61-63 for any even d_model); nothing but x is saved."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, act):
        x = _f32c(x)
        C = x.shape[-1]
        M = x.numel() // C
        y = torch.empty_like(x)
        _check(_lib.lib().nsp_gn2_act_fwd(_p(x), _p(gamma), _p(beta), eps, act, _p(y), M, C, _stream()),
               'nsp_gn2_act_fwd')
        ctx.save_for_backward(x, gamma, beta)
        ctx.cfg = (M, C, eps, act)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta = ctx.saved_tensors
        M, C, eps, act = ctx.cfg
        dy = _f32c(dy)
        part = _col_part(M, C, x.device)
        dgamma = torch.empty((C,), device=x.device, dtype=torch.float32)
        dbeta = torch.empty((C,), device=x.device, dtype=torch.float32)
        dx = torch.empty_like(x)
        _check(_lib.lib().nsp_gn2_act_bwd(_p(x), _p(dy), _p(gamma), _p(beta), eps, act, _p(part), _p(dgamma),
                                          _p(dbeta), _p(dx), M, C, _stream()), 'nsp_gn2_act_bwd')
        return dx, dgamma, dbeta, None, None


def group_norm2_act(x, gamma, beta, eps, act='none'):
    return GroupNorm2ActFn.apply(x, gamma, beta, float(eps), ACT[act])


# --------------------------------------------------------------------------
# Conv2d frontend (channels-last)
# --------------------------------------------------------------------------
def _maps16():
    """bf16 storage of the [B,T,F,32] feature maps of the conv front-end (throughput mode)."""
    return bf16_mode() and os.environ.get('NSP_CONV_BF16_MAPS', '1') != '0'


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _conv3x3_fwd(x, w_cl, bias, relu, mask_src=None, out16=False):
    B, T, F, Ci = x.shape
    Co = w_cl.shape[0]
    io16 = out16 or (Ci != 1 and x.dtype == torch.bfloat16)
    if io16:
        if Ci != 1 and x.dtype != torch.bfloat16:
            x = x.to(torch.bfloat16)
        if mask_src is not None and mask_src.dtype != torch.bfloat16:
            mask_src = mask_src.to(torch.bfloat16)
    else:
        x = _f32c(x)
        if mask_src is not None:
            mask_src = _f32c(mask_src)
    if Ci == 1:
        x = _f32c(x)
    y = torch.empty((B, T, F, Co), device=x.device, dtype=torch.bfloat16 if io16 else torch.float32)
    _check(_lib.lib().nsp_conv2d3x3_fwd(_p(x), _p(w_cl), _p(bias), _p(y), B, T, F, Ci, Co, int(relu),
                                        _p(mask_src), _COMPUTE_MODE['mode'], int(io16), _stream()),
           'nsp_conv2d3x3_fwd (only 3x3, pad 1, stride 1, C_in in {1,32}, C_out=32 are built)')
    return y


class Conv3x3ReLUFn(torch.autograd.Function):
    """relu(conv2d(x, w, b, padding=1)) on channels-last x [B,T,F,Ci]; weight in the
    reference's nn.Conv2d layout [Co,Ci,3,3] (conv.py:303-307,317-321).  In bf16 mode the
    32-channel maps (outputs, their gradients) are stored as bf16.

    ReLU backward is fused into whoever produces this node's incoming gradient: the output
    carries a flag object; a consumer that knows its input is a ReLU output (next conv's data
    gradient epilogue, max-pool backward) masks with it and sets flag['masked']."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x_in = x
        x = _c(x)
        w_cl = weight.permute(0, 2, 3, 1).contiguous()  # [Co,3,3,Ci]
        with _kev_class('conv_frontend_fwd', x.numel() * x.element_size() + x.numel() // x.shape[-1] * w_cl.shape[0] * (2 if _maps16() else 4), 'byte'):
            y = _conv3x3_fwd(x, w_cl, bias, True, out16=_maps16())
        if x.shape[-1] != 1 and y.dtype == torch.bfloat16 and x.dtype != torch.bfloat16:
            x = x.to(torch.bfloat16)
        ctx.save_for_backward(x, w_cl, y)
        ctx.flag = {'masked': False, 'consumers': 0}
        y._nsp_relu = ctx.flag
        ctx.in_flag = getattr(x_in, '_nsp_relu', None)
        if ctx.in_flag is not None:
            ctx.in_flag['consumers'] += 1
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w_cl, y = ctx.saved_tensors
        B, T, F, Ci = x.shape
        Co = w_cl.shape[0]
        io16 = y.dtype == torch.bfloat16
        dy = _c(dy)
        if io16 and dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        elif not io16:
            dy = _f32c(dy)
        if ctx.flag['masked'] and ctx.flag['consumers'] == 1:
            dz = dy  # the single consumer already applied (y > 0)
        else:
            y32, dy32 = _f32c(y), _f32c(dy)      # (named: a temporary would be freed before the launch)
            dz32 = torch.empty(dy.shape, device=dy.device, dtype=torch.float32)
            _check(_lib.lib().nsp_relu_bwd(_p(y32), _p(dy32), _p(dz32), dy.numel(), _stream()), 'nsp_relu_bwd')
            dz = dz32.to(torch.bfloat16) if io16 else dz32
        dx = None
        if ctx.needs_input_grad[0]:
            # data gradient = conv of dz with the tap-flipped, channel-transposed bank; if x is
            # itself a ReLU output its backward mask (x > 0) rides in this kernel's epilogue
            w_t = w_cl.flip(1, 2).permute(3, 1, 2, 0).contiguous()  # [Ci,3,3,Co]
            fuse = ctx.in_flag is not None and ctx.in_flag['consumers'] == 1
            dx = _conv3x3_fwd(dz, w_t, None, False, mask_src=x if fuse else None, out16=io16)
            if fuse:
                ctx.in_flag['masked'] = True
        buf = zeros_small((Co * 9 * Ci + Co,), x.device)
        _check(_lib.lib().nsp_conv2d3x3_wgrad(_p(x), _p(dz), buf.data_ptr(), buf.data_ptr() + 4 * Co * 9 * Ci,
                                              B, T, F, Ci, Co, _COMPUTE_MODE['mode'], int(io16), _stream()),
               'nsp_conv2d3x3_wgrad')
        # canonical strides (C_in = 1 would otherwise keep the permuted ones and make DDP's bucket
        # views mismatch)
        dw = buf[:Co * 9 * Ci].view(Co, 3, 3, Ci).permute(0, 3, 1, 2).contiguous().view(Co, Ci, 3, 3)
        db = buf[Co * 9 * Ci:]
        return dx, dw, db


def _conv3x3_relu_im2col(x_cl, weight, bias):
    """relu(conv2d(x, w, b, padding=1)) for input-channel counts the MFMA conv kernels do not take (conv_in_channel =
    3: the first layer of the TIMIT / WSJ Transformer recipes): im2col + the GEMM with bias / ReLU in its epilogue.
    First-layer use only: the input carries no gradient."""
    assert not x_cl.requires_grad, 'the im2col convolution is built for the input layer (no data gradient)'
    x = _f32c(x_cl)
    B, T, F, Ci = x.shape
    Co = weight.shape[0]
    K = 9 * Ci
    Kp = (K + 7) // 8 * 8
    cols = torch.empty((B * T * F, Kp), device=x.device, dtype=torch.float32)
    _check(_lib.lib().nsp_im2col3x3(_p(x), _p(cols), B, T, F, Ci, Kp, _stream()), 'nsp_im2col3x3')
    w2 = weight.reshape(Co, K)
    if Kp > K:
        w2 = torch.cat([w2, w2.new_zeros(Co, Kp - K)], dim=1)
    return linear(cols, w2.contiguous(), bias, act='relu').view(B, T, F, Co)


def conv3x3_relu(x_cl, weight, bias):
    if x_cl.shape[-1] not in (1, 32):
        return _conv3x3_relu_im2col(x_cl, weight, bias)
    return Conv3x3ReLUFn.apply(x_cl, weight, bias)


class MaxPool2dFn(torch.autograd.Function):
    """MaxPool2d(kernel=stride=(pt,pf), ceil_mode=True) on [B,T,F,C]; optionally emits
    [B,T',C,F'] (flattened = the reference's [B,T',C*F'] feature order).  bf16 in -> bf16 out."""

    @staticmethod
    def forward(ctx, x, pt, pf, to_btcf):
        in_flag = getattr(x, '_nsp_relu', None)
        x = _c(x)
        if x.dtype not in (torch.float32, torch.bfloat16):
            x = x.float()
        B, T, F, C = x.shape
        To, Fo = (T + pt - 1) // pt, (F + pf - 1) // pf
        shape = (B, To, C, Fo) if to_btcf else (B, To, Fo, C)
        y = torch.empty(shape, device=x.device, dtype=x.dtype)
        am = torch.empty(shape, device=x.device, dtype=torch.uint8)   # window-relative arg-max, 1 byte per element
        d16 = int(x.dtype == torch.bfloat16)
        _check(_lib.lib().nsp_maxpool2d_fwd(_p(x), _p(y), _p(am), B, T, F, C, pt, pf, int(to_btcf), d16, d16,
                                            _stream()), 'nsp_maxpool2d_fwd')
        ctx.in_flag = in_flag
        if in_flag is not None:
            in_flag['consumers'] += 1
            ctx.save_for_backward(am, x)
        else:
            ctx.save_for_backward(am)
        ctx.dims = (B, T, F, C, pt, pf, to_btcf, x.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        am = ctx.saved_tensors[0]
        B, T, F, C, pt, pf, to_btcf, xdt = ctx.dims
        dy = _c(dy)
        if dy.dtype not in (torch.float32, torch.bfloat16):
            dy = dy.float()
        dx = torch.empty((B, T, F, C), device=dy.device, dtype=xdt)
        fuse = ctx.in_flag is not None and ctx.in_flag['consumers'] == 1
        relu_src = ctx.saved_tensors[1] if fuse else None
        _check(_lib.lib().nsp_maxpool2d_bwd(_p(dy), _p(am), _p(dx), B, T, F, C, pt, pf, int(to_btcf),
                                            _p(relu_src), int(dy.dtype == torch.bfloat16),
                                            int(xdt == torch.bfloat16), _stream()), 'nsp_maxpool2d_bwd')
        if fuse:
            ctx.in_flag['masked'] = True
        return dx, None, None, None


def maxpool2d(x_cl, pt, pf, to_btcf=False):
    return MaxPool2dFn.apply(x_cl, pt, pf, to_btcf)


# --------------------------------------------------------------------------
# CTC loss (+ label smoothing) and RNN-T loss
# --------------------------------------------------------------------------
class CTCLossFn(torch.autograd.Function):
    """loss = (1-lsm) * sum_b nll_b / B + lsm * KL(p || uniform_{V-1}) / sum(elens)
    (ctc.py:124-129,139-150; criterion.py:110-127).  The gradient w.r.t. the logits
    is produced in the same pass and scaled by the incoming gradient in backward."""

    @staticmethod
    def forward(ctx, logits, labels, elens, ylens, lsm_prob, sum_elens, blank):
        logits = _f32c(logits)
        B, T, V = logits.shape
        Lmax = max(1, labels.shape[1])
        dev = logits.device
        L = _lib.lib()
        L.nsp_ctc_workspace_bytes.restype = ctypes.c_longlong
        ws_bytes = L.nsp_ctc_workspace_bytes((B), (T), (Lmax))
        ws = torch.empty((ws_bytes // 4,), device=dev, dtype=torch.float32)
        nll = torch.empty((B,), device=dev, dtype=torch.float32)
        grad = torch.empty_like(logits)
        _check(L.nsp_ctc_loss_fwd_bwd(_p(logits), _p(labels), _p(elens), _p(ylens), _p(nll), _p(grad),
                                      ((1.0 - lsm_prob) / B), _p(ws), (B),
                                      (T), (V), (Lmax),
                                      (blank), _stream()), 'nsp_ctc_loss_fwd_bwd')
        nll = torch.where(torch.isfinite(nll), nll, torch.zeros_like(nll))  # zero_infinity
        loss = nll.sum() / B
        if lsm_prob > 0:
            kl = torch.zeros((1,), device=dev, dtype=torch.float32)
            _check(L.nsp_ctc_kldiv_fwd_bwd(_p(logits), _p(elens), _p(kl), _p(grad),
                                           (lsm_prob / sum_elens), (1),
                                           (B), (T), (V), _stream()),
                   'nsp_ctc_kldiv_fwd_bwd')
            loss = loss * (1 - lsm_prob) + kl[0] / sum_elens * lsm_prob
        ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(nll)
        return loss.view(1), nll

    @staticmethod
    def backward(ctx, dloss, _dnll):
        grad, = ctx.saved_tensors
        return grad * dloss.view(1, 1, 1), None, None, None, None, None, None


def ctc_loss(logits, labels, elens, ylens, lsm_prob=0.0, sum_elens=1, blank=0):
    return CTCLossFn.apply(logits, labels, elens, ylens, float(lsm_prob), int(sum_elens), int(blank))


class XELossFn(torch.autograd.Function):
    """criterion.py:45-86 cross_entropy_lsm (normalize_length=False): sum over the non-pad target
    positions of the label-smoothed XE, divided by the batch size; also the per-position 'correct'
    flags (torch_utils.py:129-145).  Loss, gradient and arg-max come out of ONE kernel."""

    @staticmethod
    def forward(ctx, logits, ys, lsm_prob, ignore_index, bs):
        logits = _f32c(logits)
        V = logits.shape[-1]
        rows = logits.numel() // V
        dev = logits.device
        loss_rows = torch.empty((rows,), device=dev, dtype=torch.float32)
        correct = torch.empty((rows,), device=dev, dtype=torch.int32)
        need_grad = ctx.needs_input_grad[0]
        grad = torch.empty_like(logits) if need_grad else None
        _check(_lib.lib().nsp_xe_lsm_fwd_bwd(_p(logits), _p(ys), _p(loss_rows), _p(correct), _p(grad), rows, V,
                                             ignore_index, lsm_prob, 1.0 / bs, _stream()), 'nsp_xe_lsm_fwd_bwd')
        if need_grad:
            ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(loss_rows, correct)
        return (loss_rows.sum() / bs).view(1), loss_rows, correct

    @staticmethod
    def backward(ctx, dloss, _a, _b):
        grad, = ctx.saved_tensors
        return grad * dloss.view(*([1] * grad.dim())), None, None, None, None


def xe_lsm_loss(logits, ys_int32, lsm_prob, ignore_index, bs):
    """-> (loss [1], loss_rows [rows], correct int32 [rows]); ys_int32: int32 device tensor [rows]."""
    return XELossFn.apply(logits, ys_int32, float(lsm_prob), int(ignore_index), int(bs))


class RNNTJointLossFn(torch.autograd.Function):
    """mean_b -log P(y_b | x_b) of the RNN-Transducer from the joint inputs
    (rnn_transducer.py:239-256,262-276):
        h = tanh(enc_proj[B,T,1,J] + dec_proj[B,1,U+1,J]); logits = h W_out^T + b_out
    The lattice works on (lse, lp_blank, lp_label) only; backward turns the saved logits into
    d loss/d logits (a bf16 image in bf16 mode) and contracts them with the MFMA GEMMs."""

    @staticmethod
    def forward(ctx, enc_proj, dec_proj, w_out, b_out, labels, elens, ylens, blank):
        enc_proj, dec_proj = _f32c(enc_proj), _f32c(dec_proj)
        B, T, J = enc_proj.shape
        U1 = dec_proj.shape[1]
        V = w_out.shape[0]
        dev = enc_proj.device
        L = _lib.lib()
        use16 = bf16_mode() and J % 8 == 0
        h = torch.empty((B, T, U1, J), device=dev, dtype=torch.bfloat16 if use16 else torch.float32)
        _check(L.nsp_rnnt_joint_tanh_fwd(_p(enc_proj), _p(dec_proj), _p(None if use16 else h),
                                         _p(h if use16 else None), (B), (T),
                                         (U1), (J), _stream()), 'nsp_rnnt_joint_tanh_fwd')
        logits = linear_fwd(h.view(-1, J), w_out, b_out)  # [B*T*U1, V] fp32
        n = B * T * U1
        aux = torch.empty((7, n), device=dev, dtype=torch.float32)  # lse, lpb, lpl, alpha, beta, gb, gl
        nll = torch.empty((B,), device=dev, dtype=torch.float32)
        _check(L.nsp_rnnt_logsoftmax_gather(_p(logits), _p(labels), _p(elens), _p(ylens), _p(aux[0]),
                                            _p(aux[1]), _p(aux[2]), (B), (T),
                                            (U1), (V), (blank), _stream()),
               'nsp_rnnt_logsoftmax_gather')
        _check(L.nsp_rnnt_lattice(_p(aux[1]), _p(aux[2]), _p(elens), _p(ylens), _p(aux[3]), _p(aux[4]),
                                  _p(nll), _p(aux[5]), _p(aux[6]), (B), (T),
                                  (U1), _stream()), 'nsp_rnnt_lattice')
        ctx.save_for_backward(h, logits, aux, w_out, labels, elens, ylens)
        ctx.dims = (B, T, U1, J, V, blank, use16)
        ctx.has_bias = b_out is not None
        ctx.mark_non_differentiable(nll)
        return nll.mean().view(1), nll

    @staticmethod
    def backward(ctx, dloss, _dnll):
        h, logits, aux, w_out, labels, elens, ylens = ctx.saved_tensors
        B, T, U1, J, V, blank, use16 = ctx.dims
        L = _lib.lib()
        n = B * T * U1
        wscale = 1.0 / B
        dl = _f32c(dloss).reshape(-1)  # upstream gradient stays on the device (no host sync)
        Vp = (V + 63) // 64 * 64
        d16 = torch.empty((n, Vp), device=h.device, dtype=torch.bfloat16) if use16 else None
        db_fused = zeros_small((V,), h.device) if ctx.has_bias else None
        _check(L.nsp_rnnt_grad_logits(_p(logits), _p(aux[0]), _p(labels), _p(aux[5]), _p(aux[6]),
                                      _p(elens), _p(ylens), (wscale), _p(dl), (B),
                                      (T), (U1), (V),
                                      (blank), _p(d16), (Vp), _p(db_fused), _stream()),
               'nsp_rnnt_grad_logits')
        dlogits = d16[:, :V] if use16 else logits   # bf16 image view (pitch Vp) or in-place fp32
        h2d = h.view(-1, J)
        if use16:
            # operands are already bf16 images: call the GEMMs directly (pitch Vp on dlogits)
            sk = _wgrad_splitk(V, J, n, True)
            part = torch.empty((sk, V, J), device=h.device, dtype=torch.float32)
            gemm_raw(V, J, n, d16, 1, Vp, h2d, J, 1, part, J, splitk=sk, c_ss=V * J)
            dw = torch.empty((V, J), device=h.device, dtype=torch.float32)
            _check(L.nsp_splitk_reduce(_p(part), _p(dw), (sk), (V * J), _stream()),
                   'nsp_splitk_reduce')
            wt = _weight_t_shadow(w_out, True)                      # [J, roundup64(V)]
            db = db_fused if Vp <= 1024 else (colsum(d16[:, :V]) if ctx.has_bias else None)
            de = torch.empty((B, T, J), device=h.device, dtype=torch.float32)
            dg = torch.empty((B, U1, J), device=h.device, dtype=torch.float32)
            if J % 32 == 0 and U1 <= 512:
                # dz = (dlogits W_out) * (1 - h^2) straight out of the GEMM epilogue as a bf16 image,
                # then ONE pass over it for both reductions (sum_u -> de, sum_t -> dg)
                dz = torch.empty((n, J), device=h.device, dtype=torch.bfloat16)
                gemm_raw(n, J, Vp, d16, Vp, 1, wt, 1, wt.stride(0), dz, J, dact_src=h2d, dact=6)
                nslab = max(1, min(16, T // 16))
                slabs = torch.empty((nslab, B * U1 * J), device=h.device, dtype=torch.float32)
                _check(L.nsp_rnnt_joint_dz_reduce(_p(dz), _p(de), _p(slabs), nslab, B, T, U1, J, _stream()),
                       'nsp_rnnt_joint_dz_reduce')
                _check(L.nsp_splitk_reduce(_p(slabs), _p(dg), nslab, B * U1 * J, _stream()), 'nsp_splitk_reduce')
                return de, dg, dw.view(w_out.shape), db, None, None, None, None
            dh = torch.empty((n, J), device=h.device, dtype=torch.float32)
            gemm_raw(n, J, Vp, d16, Vp, 1, wt, 1, wt.stride(0), dh, J)
        else:
            dw = linear_wgrad(dlogits, h2d)
            db = db_fused if V <= 1024 else (colsum(dlogits) if ctx.has_bias else None)
            dh = linear_dgrad(dlogits, w_out)  # [B*T*U1, J]
        de = torch.empty((B, T, J), device=h.device, dtype=torch.float32)
        dg = torch.empty((B, U1, J), device=h.device, dtype=torch.float32)
        _check(L.nsp_rnnt_joint_tanh_bwd(_p(None if use16 else h), _p(h if use16 else None), _p(dh), _p(de),
                                         _p(dg), (B), (T), (U1),
                                         (J), _stream()), 'nsp_rnnt_joint_tanh_bwd')
        return de, dg, dw.view(w_out.shape), db, None, None, None, None


def _rows_padded_bf16(w, mult):
    """bf16 shadow of a [N,K] parameter with N zero-padded to a multiple of `mult` (cached like weight_bf16)."""
    wb = weight_bf16(w)
    N = wb.shape[0]
    Np = (N + mult - 1) // mult * mult
    if Np == N:
        return wb
    ent = getattr(w, '_nsp_rowpad16', None)
    if ent is not None and ent[0] == _wkey(w) and ent[1].device == w.device and ent[1].shape[0] == Np:
        return ent[1]
    out = torch.zeros((Np, wb.shape[1]), device=wb.device, dtype=torch.bfloat16)
    out[:N] = wb
    try:
        w._nsp_rowpad16 = (_wkey(w), out)
        _shadow_register(w, '_nsp_rowpad16', out, [(w, 0, 0, N, w[0].numel(), False)], [w])
    except Exception:
        pass
    return out


def _vec_padded(b, n, device):
    """fp32 [n] copy of a 1-D parameter (zeros beyond its length / if None), cached on the parameter."""
    if b is None:
        return torch.zeros((n,), device=device, dtype=torch.float32)
    ent = getattr(b, '_nsp_vecpad', None)
    if ent is not None and ent[0] == _wkey(b) and ent[1].device == b.device and ent[1].numel() == n:
        return ent[1]
    out = torch.zeros((n,), device=b.device, dtype=torch.float32)
    out[:b.numel()] = b.detach()
    try:
        b._nsp_vecpad = (_wkey(b), out)
    except Exception:
        pass
    return out


class RNNTJointLossFusedFn(torch.autograd.Function):
    """mean_b -log P(y_b | x_b) of the RNN-Transducer WITHOUT the [B,T,U+1,V] logit tensor
    (rnn_transducer.py:239-256,262-276), bf16 throughput mode (csrc/rnnt_fused.hip):

      h = tanh(enc_proj[b,t] + dec_proj[b,u]) for the VALID lattice nodes only, stored compacted as
      bf16 [M, J] (utterance b: rows roff[b].., dense [T_b][U_b+1]);  the logit GEMM h W_out^T + b
      keeps per-row (max, sum exp) partials + the blank / label logits in its epilogue; lattice;
      backward runs the same GEMM again and its epilogue emits d loss/d logits as the bf16 operand of
      the weight-gradient / data-gradient GEMMs (tanh' fused in the latter's epilogue), then one pass
      over dz gives both joint-input gradients."""

    @staticmethod
    def forward(ctx, enc_proj, dec_proj, w_out, b_out, labels, elens, ylens, blank, elens_host, ylens_host):
        enc_proj, dec_proj = _f32c(enc_proj), _f32c(dec_proj)
        B, T, J = enc_proj.shape
        U1 = dec_proj.shape[1]
        V = w_out.shape[0]
        Vp = (V + 63) // 64 * 64
        dev = enc_proj.device
        L = _lib.lib()
        Tb = [min(int(t), T) for t in elens_host]
        Ub = [min(int(u), U1 - 1) for u in ylens_host]
        roff_h = np.zeros((B + 1,), dtype=np.int64)
        np.cumsum([t * (u + 1) for t, u in zip(Tb, Ub)], out=roff_h[1:])
        M = int(roff_h[-1])
        roff = h2d(roff_h, dev)
        if labels.shape[1] != U1 - 1:
            # labels arrive as [B, max(1, Umax)]; the kernels index them with pitch U1 - 1
            lab2 = torch.zeros((B, max(1, U1 - 1)), device=dev, dtype=torch.int32)
            w = min(lab2.shape[1], labels.shape[1])
            lab2[:, :w] = labels[:, :w]
            labels = lab2
        h16 = torch.empty((max(M, 1), J), device=dev, dtype=torch.bfloat16)
        lab = torch.empty((max(M, 1),), device=dev, dtype=torch.int32)
        _check(L.nsp_rnnt_joint_tanh_compact(_p(enc_proj), _p(dec_proj), _p(labels), _p(elens), _p(ylens), _p(roff),
                                             _p(h16), _p(lab), B, T, U1, J, _stream()), 'nsp_rnnt_joint_tanh_compact')
        w16 = _rows_padded_bf16(w_out, 64)                                   # [Vp, J]
        bias = _vec_padded(b_out, Vp, dev)
        npart = Vp // 64
        aux = torch.empty((7, max(M, 1)), device=dev, dtype=torch.float32)   # lse, lpb, lpl, alpha, beta, gb, gl
        nll = torch.empty((B,), device=dev, dtype=torch.float32)
        rows = _rnnt_rows_kernel(J, Vp)
        if rows:
            # node-stationary kernel: a workgroup sweeps the whole vocabulary for its 256 nodes -- lse and the two
            # log-probabilities come out directly (no per-block partials, no merge pass)
            _check(rnnt_joint_gemm_timed(L.nsp_rnnt_joint_rows, M, Vp, J,
                                         1, _p(h16), _p(w16), _p(bias), M, V, Vp, J, blank, _p(lab), _p(aux[0]),
                                         _p(aux[1]), _p(aux[2]), None, None, 1.0, None, _stream()),
                   'nsp_rnnt_joint_rows(lse)')
        else:
            part = torch.empty((max(M, 1), npart, 2), device=dev, dtype=torch.float32)
            _check(rnnt_joint_gemm_timed(L.nsp_rnnt_joint_gemm, M, Vp, J,
                                         1, _p(h16), _p(w16), _p(bias), M, V, Vp, J, blank, _p(lab), _p(part),
                                         _p(aux[1]), _p(aux[2]), None, None, 1.0, None, None, _stream()),
                   'nsp_rnnt_joint_gemm(lse)')
            _check(L.nsp_rnnt_lse_merge(_p(part), npart, _p(aux[0]), _p(aux[1]), _p(aux[2]), _p(lab), M, _stream()),
                   'nsp_rnnt_lse_merge')
            del part
        _check(L.nsp_rnnt_lattice_compact(_p(aux[1]), _p(aux[2]), _p(elens), _p(ylens), _p(roff), _p(aux[3]),
                                          _p(aux[4]), _p(nll), _p(aux[5]), _p(aux[6]), B, U1, _stream()),
               'nsp_rnnt_lattice_compact')
        ctx.save_for_backward(h16, lab, aux, w_out, bias, roff, elens, ylens)
        ctx.dims = (B, T, U1, J, V, Vp, blank, M)
        ctx.has_bias = b_out is not None
        ctx.mark_non_differentiable(nll)
        return nll.mean().view(1), nll

    @staticmethod
    def backward(ctx, dloss, _dnll):
        h16, lab, aux, w_out, bias, roff, elens, ylens = ctx.saved_tensors
        B, T, U1, J, V, Vp, blank, M = ctx.dims
        L = _lib.lib()
        dev = h16.device
        dl = _f32c(dloss).reshape(-1)                    # upstream gradient stays on the device
        w16 = _rows_padded_bf16(w_out, 64)
        d16 = torch.empty((max(M, 1), Vp), device=dev, dtype=torch.bfloat16)
        if _rnnt_rows_kernel(J, Vp):
            dbpart = torch.empty(((M + 255) // 256, Vp), device=dev, dtype=torch.float32) if ctx.has_bias else None
            _check(rnnt_joint_gemm_timed(L.nsp_rnnt_joint_rows, M, Vp, J,
                                         2, _p(h16), _p(w16), _p(bias), M, V, Vp, J, blank, _p(lab), _p(aux[0]),
                                         _p(aux[5]), _p(aux[6]), _p(dbpart), _p(d16), 1.0 / B, _p(dl), _stream()),
                   'nsp_rnnt_joint_rows(dlogits)')
        else:
            nslab = (M + 127) // 128 * 2
            dbpart = torch.zeros((max(nslab, 1), Vp), device=dev, dtype=torch.float32) if ctx.has_bias else None
            rec = torch.empty((max(M, 1), 4), device=dev, dtype=torch.float32)   # per-node records of the DLOGITS epilogue
            _check(rnnt_joint_gemm_timed(L.nsp_rnnt_joint_gemm, M, Vp, J,
                                         2, _p(h16), _p(w16), _p(bias), M, V, Vp, J, blank, _p(lab), _p(aux[0]),
                                         _p(aux[5]), _p(aux[6]), _p(dbpart), _p(d16), 1.0 / B, _p(dl), _p(rec), _stream()),
                   'nsp_rnnt_joint_gemm(dlogits)')
        db = colsum(dbpart)[:V] if ctx.has_bias else None
        # dW = dlogits^T h  (split over the M rows, deterministic slab reduction)
        sk = _wgrad_splitk(V, J, M, True)
        part = torch.empty((sk, V, J), device=dev, dtype=torch.float32)
        gemm_raw(V, J, M, d16, 1, Vp, h16, J, 1, part, J, splitk=sk, c_ss=V * J)
        dw = torch.empty((V, J), device=dev, dtype=torch.float32)
        _check(L.nsp_splitk_reduce(_p(part), _p(dw), sk, V * J, _stream()), 'nsp_splitk_reduce')
        # dz = (dlogits W_out) * (1 - h^2): tanh' is the epilogue of the data-gradient GEMM
        wt = _weight_t_shadow(w_out, True)                                   # [J, roundup64(V)]
        dz = torch.empty((max(M, 1), J), device=dev, dtype=torch.bfloat16)
        gemm_raw(M, J, Vp, d16, Vp, 1, wt, 1, wt.stride(0), dz, J, dact_src=h16, dact=6)
        del d16
        de = torch.empty((B, T, J), device=dev, dtype=torch.float32)
        nsl = max(1, min(16, T // 16))
        slabs = torch.empty((nsl, B * U1 * J), device=dev, dtype=torch.float32)
        _check(L.nsp_rnnt_joint_dz_reduce_compact(_p(dz), _p(elens), _p(ylens), _p(roff), _p(de), _p(slabs), nsl,
                                                  B, T, U1, J, _stream()), 'nsp_rnnt_joint_dz_reduce_compact')
        dg = torch.empty((B, U1, J), device=dev, dtype=torch.float32)
        _check(L.nsp_splitk_reduce(_p(slabs), _p(dg), nsl, B * U1 * J, _stream()), 'nsp_splitk_reduce')
        return de, dg, dw.view(w_out.shape), db, None, None, None, None, None, None


def _rnnt_rows_kernel(J, Vp):
    """the node-stationary joint kernel (csrc/rnnt_fused.hip, nsp_rnnt_joint_rows) takes joint widths whose operand
    fragments fit a wave's registers and vocabularies whose bias fits the LDS left beside the two-slice ring (J = 512:
    3072 words); NSP_RNNT_ROWS=0 keeps the tiled GEMM epilogues (read on every call: A/B, tests)"""
    return (J in (128, 256, 512) and 2 * 64 * J * 2 + 8 * 2048 + 2 * 8 * 64 * 4 + Vp * 4 <= 163840
            and os.environ.get('NSP_RNNT_ROWS', '1') != '0')


def rnnt_joint_fused_supported(J, U1, M):
    return bf16_mode() and J % 32 == 0 and U1 <= 512 and M > 0 and os.environ.get('NSP_RNNT_FUSED', '1') != '0'


def rnnt_joint_loss(enc_proj, dec_proj, w_out, b_out, labels, elens, ylens, blank=0, elens_host=None,
                    ylens_host=None):
    """elens_host / ylens_host (python ints) select the fused, compacted path in bf16 mode (the row
    offsets of the compact lattice are computed on the host); without them, or in fp32 parity mode, the
    materialising path runs."""
    if elens_host is not None and ylens_host is not None:
        T, J, U1 = enc_proj.shape[1], enc_proj.shape[2], dec_proj.shape[1]
        M = sum(min(int(t), T) * (min(int(u), U1 - 1) + 1) for t, u in zip(elens_host, ylens_host))
        if rnnt_joint_fused_supported(J, U1, M):
            return RNNTJointLossFusedFn.apply(enc_proj, dec_proj, w_out, b_out, labels, elens, ylens, int(blank),
                                              [int(t) for t in elens_host], [int(u) for u in ylens_host])
    return RNNTJointLossFn.apply(enc_proj, dec_proj, w_out, b_out, labels, elens, ylens, int(blank))


# --------------------------------------------------------------------------
# positional tables / input glue
# --------------------------------------------------------------------------
def xl_pos_table(inv_freq, L):
    d = inv_freq.numel() * 2
    out = torch.empty((L, d), device=inv_freq.device, dtype=torch.float32)
    _check(_lib.lib().nsp_xl_pos_table(_p(inv_freq), _p(out), (L), (d), _stream()),
           'nsp_xl_pos_table')
    return out


class ScaleAddBcastFn(torch.autograd.Function):
    """alpha*x + z broadcast over the leading (batch) dim; z carries no gradient."""

    @staticmethod
    def forward(ctx, x, z, alpha):
        x = _f32c(x)
        z = _f32c(z)
        y = torch.empty_like(x)
        _check(_lib.lib().nsp_scale_add_bcast(_p(x), _p(z), _p(y), (alpha),
                                              (x.numel()), (z.numel()),
                                              _stream()), 'nsp_scale_add_bcast')
        ctx.alpha = alpha
        return y

    @staticmethod
    def backward(ctx, dy):
        return axpby(dy, None, ctx.alpha, 0.0), None, None


def scale_add_bcast(x, z, alpha):
    return ScaleAddBcastFn.apply(x, z, float(alpha))


def specaug_apply_(xs, freq_bands, time_bands):
    """Zero [start,end) bands in place on [B,T,F] (spec_augment.py:112-140)."""
    B, T, F = xs.shape
    fb = h2d(np.asarray(freq_bands, dtype=np.int32).reshape(-1), xs.device) if len(freq_bands) else None
    tb = h2d(np.asarray(time_bands, dtype=np.int32).reshape(-1), xs.device) if len(time_bands) else None
    _check(_lib.lib().nsp_specaug_apply(_p(xs), (B), (T), (F),
                                        _p(fb), (len(freq_bands)), _p(tb),
                                        (len(time_bands)), _stream()), 'nsp_specaug_apply')
    return xs


def pad_batch(packed, offsets, lens, B, Tmax, F, pad_value=0.0):
    """Ragged -> padded [B,Tmax,F] on device from ONE packed H2D copy (pad_list, torch_utils.py:56)."""
    out = torch.empty((B, Tmax, F), device=packed.device, dtype=torch.float32)
    _check(_lib.lib().nsp_pad_batch(_p(packed), _p(offsets), _p(lens), _p(out), (B),
                                    (Tmax), (F), (pad_value),
                                    _stream()), 'nsp_pad_batch')
    return out


def ctc_forced_align(logits, labels, elens, ylens, blank=0):
    """CTC forced alignment (ctc.py:628-753) -> trigger points IntTensor `[B, Lmax+1]` (device).
    NOTE: trigger_points has Lmax+1 columns where Lmax = labels.shape[1] (>= 1)."""
    logits = _f32c(logits)
    B, T, V = logits.shape
    Lmax = max(1, labels.shape[1])
    L = _lib.lib()
    L.nsp_ctc_align_workspace_bytes.restype = ctypes.c_longlong
    nbytes = L.nsp_ctc_align_workspace_bytes((B), (T), (Lmax))
    ws = torch.empty((nbytes // 4 + 1,), device=logits.device, dtype=torch.float32)
    tp = torch.empty((B, Lmax + 1), device=logits.device, dtype=torch.int32)
    _check(L.nsp_ctc_forced_align(_p(logits), _p(labels), _p(elens), _p(ylens), _p(tp), _p(ws),
                                  (B), (T), (V), (Lmax),
                                  (blank), _stream()), 'nsp_ctc_forced_align')
    return tp


def argmax_rows(x2d):
    """int32 [rows] arg-max over the last dim of a contiguous fp32 [rows, V] matrix (first index on ties)."""
    x2d = _f32c(x2d)
    rows, V = x2d.shape
    out = torch.empty((rows,), device=x2d.device, dtype=torch.int32)
    _check(_lib.lib().nsp_argmax_rows(_p(x2d), _p(out), rows, V, x2d.stride(0), _stream()), 'nsp_argmax_rows')
    return out


def lstm_cell_step(gates, h_prev, c_prev, update=None):
    """One LSTM cell update (gate order i,f,g,o) from pre-activations [B,4H]; rows whose int32
    `update` flag is 0 keep (h_prev, c_prev).  Inference-side (greedy decoding), no autograd."""
    gates, h_prev, c_prev = _f32c(gates), _f32c(h_prev), _f32c(c_prev)
    B, H = h_prev.shape
    h, c = torch.empty_like(h_prev), torch.empty_like(c_prev)
    _check(_lib.lib().nsp_lstm_cell_step(_p(gates), _p(h_prev), _p(c_prev), _p(update), _p(h), _p(c), B, H,
                                         _stream()), 'nsp_lstm_cell_step')
    return h, c


# --------------------------------------------------------------------------
# monotonic (chunkwise) attention training: the scans of MoChA / MMA (csrc/mocha.hip)
# --------------------------------------------------------------------------
class MonoAlphaFn(torch.autograd.Function):
    """alpha [..., klen] of parallel_monotonic_attention (hma_train.py:12-67) for one target position:
    p_choose = (1 - stableemit) sigmoid(e), cumprod in log space, alpha = p c cumsum(aw_prev / den).
    -> (alpha, p_choose); p_choose is returned for the caller's bookkeeping only (not differentiable here)."""

    @staticmethod
    def forward(ctx, e, aw_prev, eps, no_denom, stableemit):
        klen = e.shape[-1]
        e2, a2 = _f32c(e).reshape(-1, klen), _f32c(aw_prev.expand_as(e)).reshape(-1, klen)
        rows = e2.shape[0]
        alpha, pch, cp = torch.empty_like(e2), torch.empty_like(e2), torch.empty_like(e2)
        _check(_lib.lib().nsp_mono_alpha_fwd(_p(e2), _p(a2), _p(alpha), _p(pch), _p(cp), rows, klen, float(eps),
                                             int(bool(no_denom)), float(stableemit), _stream()), 'nsp_mono_alpha_fwd')
        ctx.save_for_backward(pch, cp, a2)
        ctx.cfg = (rows, klen, float(eps), int(bool(no_denom)), float(stableemit), e.shape, aw_prev.shape)
        pch_out = pch.view(e.shape)
        ctx.mark_non_differentiable(pch_out)
        return alpha.view(e.shape), pch_out

    @staticmethod
    def backward(ctx, dalpha, _dp):
        pch, cp, a2 = ctx.saved_tensors
        rows, klen, eps, no_denom, lam, eshape, ashape = ctx.cfg
        da_ = _f32c(dalpha).reshape(rows, klen)
        de, daw = torch.empty_like(da_), torch.empty_like(da_)
        _check(_lib.lib().nsp_mono_alpha_bwd(_p(da_), _p(pch), _p(cp), _p(a2), _p(de), _p(daw), rows, klen, eps, no_denom,
                                             lam, _stream()), 'nsp_mono_alpha_bwd')
        daw = daw.view(eshape)
        if tuple(ashape) != tuple(eshape):
            daw = daw.sum_to_size(ashape)
        return de.view(eshape), daw, None, None, None


def mono_alpha(e, aw_prev, eps, no_denom=False, stableemit=0.0):
    return MonoAlphaFn.apply(e, aw_prev, eps, no_denom, stableemit)


class ChunkBetaFn(torch.autograd.Function):
    """beta [..., klen] of soft_chunkwise_attention (mocha_train.py:13-58) from chunk energies u and alpha (same
    shape); w = chunk size (1 < w <= 64) or -1 (MILk)."""

    @staticmethod
    def forward(ctx, u, alpha, w, sf):
        klen = u.shape[-1]
        u2, a2 = _f32c(u).reshape(-1, klen), _f32c(alpha).reshape(-1, klen)
        beta = torch.empty_like(u2)
        _check(_lib.lib().nsp_chunk_beta_fwd(_p(u2), _p(a2), _p(beta), u2.shape[0], klen, int(w), float(sf), _stream()),
               'nsp_chunk_beta_fwd')
        ctx.save_for_backward(u2, a2)
        ctx.cfg = (int(w), float(sf), u.shape)
        return beta.view(u.shape)

    @staticmethod
    def backward(ctx, dbeta):
        u2, a2 = ctx.saved_tensors
        w, sf, shape = ctx.cfg
        db = _f32c(dbeta).reshape(u2.shape)
        du, da_ = torch.empty_like(u2), torch.empty_like(u2)
        _check(_lib.lib().nsp_chunk_beta_bwd(_p(db), _p(u2), _p(a2), _p(du), _p(da_), u2.shape[0], u2.shape[1], w, sf,
                                             _stream()), 'nsp_chunk_beta_bwd')
        return du.view(shape), da_.view(shape), None, None


def chunk_beta(u, alpha, w, sf=1.0):
    if not chunk_beta_supported(w, u.shape[-1]):
        return _chunk_beta_tensor_ops(u, alpha, w, sf)
    return ChunkBetaFn.apply(u, alpha, w, sf)


def chunk_beta_supported(w, klen=0):
    """what the scan kernels of csrc/mocha.hip take: chunk sizes up to 64 frames (or MILk, w = -1), rows up to 8192 keys"""
    return (w == -1 or 1 < w <= 64) and klen <= 8192


def _window_sums(x2d, back, fwd):
    """sum over [j - back, j + fwd] of every row of x2d (zeros outside): a ones-filter correlation"""
    import torch.nn.functional as F
    ones = x2d.new_ones(1, 1, back + fwd + 1)
    return F.conv1d(F.pad(x2d, [back, fwd]).unsqueeze(1), ones).squeeze(1)


def _chunk_beta_tensor_ops(u, alpha, w, sf):
    """Fallback for chunk sizes / row lengths the scan kernels do not take (any w, any klen): the same chunkwise soft-max
    (mocha_train.py:36-57: shift by the row maximum, exp clamped at 1e-5, window sums of the denominators and of
    alpha / denominator) as differentiable tensor ops.  No recipe of the reference reaches it (chunk sizes 4..16)."""
    klen = u.shape[-1]
    u2, a2 = u.reshape(-1, klen).float(), alpha.reshape(-1, klen).float()
    ex = torch.clamp(torch.exp(u2 - u2.max(dim=-1, keepdim=True)[0]), min=1e-5)
    if w == -1:
        den = torch.cumsum(ex, dim=-1)
        beta = ex * _window_sums(a2 * sf / den, 0, klen - 1)
    else:
        den = _window_sums(ex, w - 1, 0)
        beta = ex * _window_sums(a2 * sf / den, 0, w - 1)
    return beta.view(u.shape)


class AddEnergyFn(torch.autograd.Function):
    """e [B,T] = sum_a v_a act(K[b,t,a] + Q[b,a] (+ C[b,t,a])): additive attention energies (attention.py:148-156 with
    tanh, monotonic_energy.py / chunk_energy.py with relu) in one pass over the key projection."""

    @staticmethod
    def forward(ctx, K, Q, C, v, act):
        B, T, A = K.shape
        K, Q, v = _f32c(K), _f32c(Q).reshape(B, A), _f32c(v).reshape(A)
        C = _f32c(C) if C is not None else None
        e = torch.empty((B, T), device=K.device, dtype=torch.float32)
        _check(_lib.lib().nsp_add_energy_fwd(_p(K), _p(Q), _p(C), _p(v), _p(e), B, T, A, ACT[act], _stream()), 'nsp_add_energy_fwd')
        ctx.save_for_backward(K, Q, C, v)
        ctx.act = ACT[act]
        ctx.qshape, ctx.vshape = None, None
        return e

    @staticmethod
    def backward(ctx, de):
        K, Q, C, v = ctx.saved_tensors
        B, T, A = K.shape
        de = _f32c(de)
        dtmp = torch.empty_like(K)
        dQ = torch.empty((B, A), device=K.device, dtype=torch.float32)
        dvp = torch.empty((B, A), device=K.device, dtype=torch.float32)
        _check(_lib.lib().nsp_add_energy_bwd(_p(de), _p(K), _p(Q), _p(C), _p(v), _p(dtmp), _p(dQ), _p(dvp), B, T, A, ctx.act,
                                             _stream()), 'nsp_add_energy_bwd')
        return dtmp, dQ, (dtmp if C is not None else None), colsum(dvp), None


def add_energy(K, Q, v, act, C=None):
    """K [B,T,A], Q [B,1,A] or [B,A], v [A] or [1,A] -> e [B,T]"""
    B, T, A = K.shape
    return AddEnergyFn.apply(K, Q.reshape(B, A), C, v.reshape(A), act)


class RowSoftmaxFn(torch.autograd.Function):
    """softmax(sharp * e) over the last dim with masked_fill(mask == 0, NEG_INF) semantics (attention.py:170-176)"""

    @staticmethod
    def forward(ctx, e, mask, sharp):
        T = e.shape[-1]
        e2 = _f32c(e).reshape(-1, T)
        m2 = None
        if mask is not None:
            m2 = mask.expand_as(e).reshape(-1, T).to(torch.uint8).contiguous()
        aw = torch.empty_like(e2)
        _check(_lib.lib().nsp_row_softmax_fwd(_p(e2), _p(m2), _p(aw), e2.shape[0], T, float(sharp), _stream()), 'nsp_row_softmax_fwd')
        ctx.save_for_backward(aw, m2)
        ctx.cfg = (float(sharp), e.shape)
        return aw.view(e.shape)

    @staticmethod
    def backward(ctx, daw):
        aw, m2 = ctx.saved_tensors
        sharp, shape = ctx.cfg
        d2 = _f32c(daw).reshape(aw.shape)
        de = torch.empty_like(aw)
        _check(_lib.lib().nsp_row_softmax_bwd(_p(aw), _p(d2), _p(m2), _p(de), aw.shape[0], aw.shape[1], sharp, _stream()),
               'nsp_row_softmax_bwd')
        return de.view(shape), None, None


def row_softmax(e, mask=None, sharp=1.0):
    return RowSoftmaxFn.apply(e, mask, sharp)


class LSTMCellFn(torch.autograd.Function):
    """(h, c) = LSTMCell non-linearity on pre-activation gates [B,4H] (i, f, g, o) and c_prev [B,H] (las.py:860-866)"""

    @staticmethod
    def forward(ctx, gates, c_prev):
        B, H4 = gates.shape
        H = H4 // 4
        gates, c_prev = _f32c(gates), _f32c(c_prev)
        h, c = torch.empty_like(c_prev), torch.empty_like(c_prev)
        _check(_lib.lib().nsp_lstm_cell_fwd(_p(gates), _p(c_prev), _p(h), _p(c), B, H, _stream()), 'nsp_lstm_cell_fwd')
        ctx.save_for_backward(gates, c_prev, c)
        return h, c

    @staticmethod
    def backward(ctx, dh, dc):
        gates, c_prev, c = ctx.saved_tensors
        B, H = c.shape
        dh = _f32c(dh) if dh is not None else None
        dc = _f32c(dc) if dc is not None else None
        dg, dcp = torch.empty_like(gates), torch.empty_like(c)
        _check(_lib.lib().nsp_lstm_cell_bwd(_p(dh), _p(dc), _p(gates), _p(c_prev), _p(c), _p(dg), _p(dcp), B, H, _stream()),
               'nsp_lstm_cell_bwd')
        return dg, dcp


def lstm_cell(gates, c_prev):
    return LSTMCellFn.apply(gates, c_prev)


class HeadScoresFn(torch.autograd.Function):
    """S[b,h,i,j] = alpha * q[b,i,h,:] . k[b,j,h,:] on [B,L,H,dk] x [B,T,H,dk] (monotonic_energy.py / chunk_energy.py
    'scaled_dot'): the batched nsp_gemm of AttentionFn without its softmax."""

    @staticmethod
    def forward(ctx, q, k, alpha):
        B, L, H, dk = q.shape
        T = k.shape[1]
        d = H * dk
        q, k = _f32c(q), _f32c(k)
        S = torch.empty((B, H, L, T), device=q.device, dtype=torch.float32)
        gemm_raw(L, T, dk, q, d, 1, k, 1, d, S, T, batch=(B, H), a_b=(L * d, dk), b_b=(T * d, dk),
                 c_b=(H * L * T, L * T), alpha=alpha)
        ctx.save_for_backward(q, k)
        ctx.alpha = alpha
        return S

    @staticmethod
    def backward(ctx, dS):
        q, k = ctx.saved_tensors
        B, L, H, dk = q.shape
        T = k.shape[1]
        d = H * dk
        dS = _f32c(dS)
        dQ = torch.empty_like(q)
        gemm_raw(L, dk, T, dS, T, 1, k, d, 1, dQ, d, batch=(B, H), a_b=(H * L * T, L * T), b_b=(T * d, dk),
                 c_b=(L * d, dk), alpha=ctx.alpha)
        dK = torch.empty_like(k)
        gemm_raw(T, dk, L, dS, 1, T, q, d, 1, dK, d, batch=(B, H), a_b=(H * L * T, L * T), b_b=(L * d, dk),
                 c_b=(T * d, dk), alpha=ctx.alpha)
        return dQ, dK, None


def head_scores(q, k, alpha=1.0):
    return HeadScoresFn.apply(q, k, alpha)


class HeadContextFn(torch.autograd.Function):
    """cv[b,i,h,:] = sum_j aw[b,h,i,j] v[b,j,h,:]  ([B,H,L,T] x [B,T,H,dk] -> [B,L,H,dk])"""

    @staticmethod
    def forward(ctx, aw, v):
        B, H, L, T = aw.shape
        dk = v.shape[-1]
        d = H * dk
        aw, v = _f32c(aw), _f32c(v)
        O = torch.empty((B, L, H, dk), device=aw.device, dtype=torch.float32)
        gemm_raw(L, dk, T, aw, T, 1, v, d, 1, O, d, batch=(B, H), a_b=(H * L * T, L * T), b_b=(T * d, dk), c_b=(L * d, dk))
        ctx.save_for_backward(aw, v)
        return O

    @staticmethod
    def backward(ctx, dO):
        aw, v = ctx.saved_tensors
        B, H, L, T = aw.shape
        dk = v.shape[-1]
        d = H * dk
        dO = _f32c(dO)
        dA = torch.empty_like(aw)
        gemm_raw(L, T, dk, dO, d, 1, v, 1, d, dA, T, batch=(B, H), a_b=(L * d, dk), b_b=(T * d, dk), c_b=(H * L * T, L * T))
        dV = torch.empty_like(v)
        gemm_raw(T, dk, L, aw, 1, T, dO, d, 1, dV, d, batch=(B, H), a_b=(H * L * T, L * T), b_b=(L * d, dk), c_b=(T * d, dk))
        return dA, dV


def head_context(aw, v):
    return HeadContextFn.apply(aw, v)


# --------------------------------------------------------------------------
# per-launch timing of the GEMM kernel with HIP events (bench.py roofline)
# --------------------------------------------------------------------------
_KEV = {'on': False, 'events': [], 'flops': 0.0, 'side_events': [], 'side_flops': 0.0}
_gemm_raw_untimed = gemm_raw


def _kev_record(e0, e1, flops, algo=None):
    # launches on a side stream (CTC head, prediction-network projections) are kept apart: they run
    # beside main-stream kernels, so their event pairs measure co-scheduling, not the kernel
    # `algo`: the ALGORITHMIC flops of the launch where they differ from the executed 2 M N K (the
    # recomputed logits of the RNN-T joint backward count zero, the vocabulary padding 1000 -> 1024 is not work)
    if torch.cuda.current_stream() == torch.cuda.default_stream():
        _KEV['events'].append((e0, e1))
        _KEV['flops'] += flops
        _KEV['algo_flops'] = _KEV.get('algo_flops', 0.0) + (flops if algo is None else algo)
    else:
        _KEV['side_events'].append((e0, e1))
        _KEV['side_flops'] += flops


def _gemm_raw_timed(M, N, K, *a, **k):
    if not _KEV['on']:
        return _gemm_raw_untimed(M, N, K, *a, **k)
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    _gemm_raw_untimed(M, N, K, *a, **k)
    e1.record()
    batch = k.get('batch', (1, 1))
    _kev_record(e0, e1, 2.0 * M * N * K * batch[0] * batch[1])


gemm_raw = _gemm_raw_timed


def rnnt_joint_gemm_timed(fn, M, Vp, J, *args):
    """Call nsp_rnnt_joint_gemm and, when bench.py's per-launch events are on, record it like any other GEMM.
    args[0] = 1 (forward logits: algorithmic work 2 M V J) or 2 (backward: the logits are RECOMPUTED -- executed,
    not algorithmic), args[5] = V."""
    if not _KEV['on']:
        return fn(*args)
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn(*args)
    e1.record()
    _kev_record(e0, e1, 2.0 * M * Vp * J, algo=(2.0 * M * args[5] * J if args[0] == 1 else 0.0))
    return rc


import contextlib as _contextlib  # noqa: E402


class _NullCtx(object):
    __slots__ = ()

    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NULL_CTX = _NullCtx()


class _KevClassCtx(object):
    __slots__ = ('name', 'work', 'unit', 'e0', 'nbytes')

    def __init__(self, name, work, unit, nbytes=0.0):
        self.name, self.work, self.unit, self.nbytes = name, work, unit, nbytes

    def __enter__(self):
        if torch.cuda.current_stream() != torch.cuda.default_stream():
            self.name = self.name + '@side'
        self.e0 = torch.cuda.Event(enable_timing=True)
        self.e0.record()
        return None

    def __exit__(self, *exc):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        c = _KEV.setdefault('classes', {}).setdefault(self.name, {'events': [], 'work': 0.0, 'unit': self.unit, 'bytes': 0.0})
        c['events'].append((self.e0, e1))
        c['work'] += float(self.work)
        c['bytes'] += float(self.nbytes)
        return False


def _kev_class(name, work, unit, nbytes=0.0):
    """bench.py's `roofline.classes`: HIP events around one launch (group) of a NON-GEMM kernel class on sampled steps,
    with its algorithmic work (`unit` = 'flop' or 'byte').  Side-stream launches are recorded under `<name>@side`:
    their event pairs measure co-scheduling with main-stream kernels.  (Off -- every step but bench.py's sampled ones --
    this returns one shared no-op object: a generator-based context manager cost ~2 us at ~200 call sites per step.)"""
    return _KevClassCtx(name, work, unit, nbytes) if _KEV['on'] else _NULL_CTX


def kernel_events_start():
    _KEV['on'], _KEV['events'], _KEV['flops'] = True, [], 0.0
    _KEV['side_events'], _KEV['side_flops'] = [], 0.0
    _KEV['algo_flops'], _KEV['classes'] = 0.0, {}


def kernel_events_enable(on):
    """Pause / resume event recording without dropping what has been collected (bench.py
    instruments a subset of the timed steps: two event records per GEMM launch cost ~3 ms/step)."""
    _KEV['on'] = bool(on)


def kernel_events_stop():
    """-> {'launches', 'ms' (sum of HIP-event durations on the launch stream), 'flops'} of the main-stream
    launches, and the same three numbers of the side-stream launches under 'side'."""
    _KEV['on'] = False
    torch.cuda.synchronize()
    ms = sum(e0.elapsed_time(e1) for e0, e1 in _KEV['events'])
    sms = sum(e0.elapsed_time(e1) for e0, e1 in _KEV['side_events'])
    out = {'launches': len(_KEV['events']), 'ms': ms, 'flops': _KEV['flops'], 'algo_flops': _KEV.get('algo_flops', 0.0),
           'side': {'launches': len(_KEV['side_events']), 'ms': sms, 'flops': _KEV['side_flops']},
           'classes': {n: {'launches': len(c['events']), 'ms': sum(a.elapsed_time(b) for a, b in c['events']),
                           'work': c['work'], 'unit': c['unit'], 'bytes': c.get('bytes', 0.0)}
                       for n, c in _KEV.get('classes', {}).items()}}
    _KEV['events'], _KEV['side_events'], _KEV['classes'] = [], [], {}
    return out


# --------------------------------------------------------------------------
# LSTM (RNN-T prediction network)
# --------------------------------------------------------------------------
def _weight_t_shadow(w, bf16):
    """W^T [K, N] (bf16 or fp32) cached on the parameter like weight_bf16."""
    name = '_nsp_t16' if bf16 else '_nsp_t32'
    ent = getattr(w, name, None)
    if ent is not None and ent[0] == _wkey(w) and ent[1].device == w.device:
        return ent[1]
    wt = w.detach().reshape(w.shape[0], -1).t().contiguous()   # [K, N]
    if bf16:
        wt = to_bf16(wt, pad_to=64)                              # [K, roundup64(N)], zero padded
    try:
        setattr(w, name, (_wkey(w), wt))
        if bf16:
            _shadow_register(w, name, wt, [(w, 0, 0, w[0].numel(), w.shape[0], True)], [w])
    except Exception:
        pass
    return wt


class LSTMFn(torch.autograd.Function):
    """y = LSTM(x) for one layer, batch_first, zero initial state (PyTorch gate order i,f,g,o)."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh):
        B, L, I = x.shape
        H = w_hh.shape[1]
        dev = x.device
        use16 = bf16_mode()
        bias = axpby(b_ih, b_hh, 1.0, 1.0)
        x2d = _f32c(x).reshape(B * L, I)
        xa = to_bf16(x2d) if (use16 and I % 8 == 0) else x2d
        gi = linear_fwd(xa, w_ih, bias)                                  # [B*L, 4H]
        y = torch.empty((B, L, H), device=dev, dtype=torch.float32)
        ysh = torch.empty((B, L, H), device=dev, dtype=torch.bfloat16) if use16 else y
        c_all = torch.empty((B, L, H), device=dev, dtype=torch.float32)
        gates = torch.empty((B, L, 4 * H), device=dev, dtype=torch.float32)
        whh = weight_bf16(w_hh) if use16 else w_hh
        _check(_lib.lib().nsp_lstm_fwd(_p(gi), _p(whh), _p(y), _p(ysh), _p(c_all), _p(gates),
                                       (B), (L), (H),
                                       (_COMPUTE_MODE['mode']), _stream()), 'nsp_lstm_fwd')
        ctx.save_for_backward(xa, w_ih, w_hh, ysh, c_all, gates)
        ctx.dims = (B, L, I, H)
        ctx.use16 = use16
        return y

    @staticmethod
    def backward(ctx, dy):
        xa, w_ih, w_hh, ysh, c_all, gates = ctx.saved_tensors
        B, L, I, H = ctx.dims
        dev = dy.device
        use16 = ctx.use16
        dy = _f32c(dy)
        dgates = torch.empty((B, L, 4 * H), device=dev, dtype=torch.float32)
        dgsh = torch.empty((B, L, 4 * H), device=dev, dtype=torch.bfloat16) if use16 else dgates
        dc = torch.empty((B, H), device=dev, dtype=torch.float32)
        whh_t = _weight_t_shadow(w_hh, use16)
        mode = 0 if use16 else 1
        _check(_lib.lib().nsp_lstm_bwd(_p(dy), _p(whh_t), _p(c_all), _p(gates), _p(dgates), _p(dgsh),
                                       _p(dc), (B), (L), (H),
                                       (mode), _stream()), 'nsp_lstm_bwd')
        g2d = dgsh.view(B * L, 4 * H)
        # h_{t-1} for every (b,t): the outputs shifted by one step (zeros at t=0)
        hprev = torch.zeros_like(ysh)
        hprev[:, 1:] = ysh[:, :-1]
        dx = linear_dgrad(g2d, w_ih)[:, :I].reshape(B, L, I) if ctx.needs_input_grad[0] else None
        dw_ih = linear_wgrad(g2d, xa).view(w_ih.shape)
        dw_hh = linear_wgrad(g2d, hprev.view(B * L, H)).view(w_hh.shape)
        db = colsum(g2d)
        return dx, dw_ih, dw_hh, db, db


class LSTMStateFn(torch.autograd.Function):
    """(y, h_n, c_n) = LSTM(x | h0, c0) for one layer, batch_first: an LSTM that starts from a GIVEN state and hands
    its final state on, differentiable in both (the forward direction of the latency-controlled BLSTM carries its
    state from chunk to chunk and is trained through it, encoders/rnn.py:466-475).  The step kernels are those of
    LSTMFn: the n steps live in rows 1..n of buffers with n + 2 rows, row 0 holds the initial state, row n + 1 is the
    zero "next step" through which the gradient w.r.t. the final state enters (nsp_lstm_*_range)."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, h0, c0):
        B, n, I = x.shape
        H = w_hh.shape[1]
        L = n + 2
        dev = x.device
        use16 = bf16_mode()
        bias = axpby(b_ih, b_hh, 1.0, 1.0)
        x_ext = torch.zeros((B, L, I), device=dev, dtype=torch.float32)
        x_ext[:, 1:n + 1] = _f32c(x)
        x2d = x_ext.view(B * L, I)
        xa = to_bf16(x2d) if (use16 and I % 8 == 0) else x2d
        gi = linear_fwd(xa, w_ih, bias)                                  # [B*L, 4H] (rows 0 and n+1 unused)
        y = torch.zeros((B, L, H), device=dev, dtype=torch.float32)
        c_all = torch.zeros((B, L, H), device=dev, dtype=torch.float32)
        y[:, 0] = h0
        c_all[:, 0] = c0
        if use16:
            ysh = torch.zeros((B, L, H), device=dev, dtype=torch.bfloat16)
            ysh[:, 0] = h0.to(torch.bfloat16)
        else:
            ysh = y
        gates = torch.zeros((B, L, 4 * H), device=dev, dtype=torch.float32)
        whh = weight_bf16(w_hh) if use16 else w_hh
        _check(_lib.lib().nsp_lstm_fwd_range(_p(gi), _p(whh), _p(y), _p(ysh), _p(c_all), _p(gates), B, L, H,
                                             _COMPUTE_MODE['mode'], 1, n + 1, _stream()), 'nsp_lstm_fwd_range')
        ctx.save_for_backward(xa, w_ih, w_hh, ysh, c_all, gates)
        ctx.dims = (B, n, I, H)
        ctx.use16 = use16
        return y[:, 1:n + 1], y[:, n].clone(), c_all[:, n].clone()

    @staticmethod
    def backward(ctx, dy, dh_n, dc_n):
        xa, w_ih, w_hh, ysh, c_all, gates = ctx.saved_tensors
        B, n, I, H = ctx.dims
        L = n + 2
        dev = xa.device
        use16 = ctx.use16
        dy_ext = torch.zeros((B, L, H), device=dev, dtype=torch.float32)
        if dy is not None:
            dy_ext[:, 1:n + 1] = dy
        if dh_n is not None:
            dy_ext[:, n] += dh_n
        dgates = torch.zeros((B, L, 4 * H), device=dev, dtype=torch.float32)
        dgsh = torch.zeros((B, L, 4 * H), device=dev, dtype=torch.bfloat16) if use16 else dgates
        dc = _f32c(dc_n).clone() if dc_n is not None else torch.zeros((B, H), device=dev, dtype=torch.float32)
        whh_t = _weight_t_shadow(w_hh, use16)
        _check(_lib.lib().nsp_lstm_bwd_range(_p(dy_ext), _p(whh_t), _p(c_all), _p(gates), _p(dgates), _p(dgsh), _p(dc),
                                             B, L, H, 0 if use16 else 1, 1, n + 1, _stream()), 'nsp_lstm_bwd_range')
        g2d = dgsh.view(B * L, 4 * H)
        hprev = torch.zeros_like(ysh)
        hprev[:, 1:] = ysh[:, :-1]                   # h_{t-1}; row 1 sees the initial state
        dx = None
        if ctx.needs_input_grad[0]:
            dx = linear_dgrad(g2d, w_ih)[:, :I].reshape(B, L, I)[:, 1:n + 1]
        dw_ih = linear_wgrad(g2d, xa).view(w_ih.shape)
        dw_hh = linear_wgrad(g2d, hprev.view(B * L, H)).view(w_hh.shape)
        db = colsum(g2d)
        # d/dh0 = dgates_1 W_hh (the recurrent term a step 0 would have received); d/dc0 = what the kernel left in dc
        dh0 = linear_dgrad((dgsh if use16 else dgates)[:, 1].contiguous(), w_hh)[:, :H] if ctx.needs_input_grad[5] else None
        dc0 = dc if ctx.needs_input_grad[6] else None
        return dx, dw_ih, dw_hh, db, db, dh0, dc0


def lstm_state(x, w_ih, w_hh, b_ih, b_hh, h0, c0):
    return LSTMStateFn.apply(x, w_ih, w_hh, b_ih, b_hh, h0, c0)


def _cat_cached(owner, name, parts, build, layout=None):
    """torch.cat of derived weight shadows, cached on `owner` and keyed by the versions of `parts`.
    layout(t) -> the shadow-registry parts of the result (see _shadow_register), when it is a bf16 shadow."""
    key = tuple(_wkey(p) for p in parts)
    ent = getattr(owner, name, None)
    if ent is not None and ent[0] == key and ent[1].device == owner.device:
        return ent[1]
    t = build()
    try:
        setattr(owner, name, (key, t))
        if layout is not None and t.dtype == torch.bfloat16:
            _shadow_register(owner, name, t, layout(t), list(parts), single=False)
    except Exception:
        pass
    return t


_LSTM_DEAD_CHECKS = []   # (pinned flag, event, which) of persistent launches whose `dead` word is still in flight


def _lstm_poll_dead(block=False):
    """Raise if a finished persistent launch reported a grid-barrier timeout (sync[1] != 0): its
    outputs are garbage (forward additionally poisons y_top[0], backward dg16[0] with NaN)."""
    keep = []
    for flag, ev, which in _LSTM_DEAD_CHECKS:
        if block:
            ev.synchronize()
        if ev.query():
            if int(flag[0]) != 0:
                del _LSTM_DEAD_CHECKS[:]
                raise RuntimeError('nsp_lstm_stack_%s_persistent: grid barrier timed out (workgroups were not '
                                   'co-resident); set NSP_LSTM_PERSISTENT=0 to use one launch per stage' % which)
        else:
            keep.append((flag, ev, which))
    _LSTM_DEAD_CHECKS[:] = keep


def _lstm_layerwise(B, H):
    """Run a multi-layer stack layer by layer (see LSTMStackFn._forward_layerwise)?  OPT-IN (NSP_LSTM_LAYERWISE=1), only
    where the persistent kernels apply (H % 256 == 0, H <= 1024).  Measured at 2 x 1024, B = 128, L = 200
    (profiles/r03ag_lstm_layerwise.log): 20.9 ms on 64 CUs against the wavefront's 13.8 ms on 128 -- 25 % fewer
    CU-milliseconds, but the bench step does not move (126.2 vs 125.4 ms), so the wavefront stays the default."""
    return (os.environ.get('NSP_LSTM_LAYERWISE', '0') == '1' and os.environ.get('NSP_LSTM_PERSISTENT', '1') != '0'
            and H % 256 == 0 and H <= 1024)


def _lstm_xchg(P, cols, dev):
    """Scratch for the persistent kernels' fragment-major hand-over (csrc/lstm.hip): L x 64 x cols bf16 per layer
    (cols = 2H forward: h and dropout(h); 4H backward); slabs of a larger batch reuse it one after the other."""
    bufs = [torch.empty((P.L * 64 * cols,), device=dev, dtype=torch.bfloat16) for _ in range(P.nl)]
    for l, b in enumerate(bufs):
        P.xchg[l] = b.data_ptr()
    return bufs


def _lstm_stack_launch(which, P, dev):
    # (roofline.classes: recurrent + inter-layer products of the stack, 2 flops per MAC; backward = data + weight-side products)
    work = 2.0 * P.B * P.L * 4 * P.H * P.H * (2 * P.nl - 1) * (1 if which == 'fwd' else 2)
    with _kev_class('lstm_' + which, work, 'flop'):
        return _lstm_stack_launch_impl(which, P, dev)


def _lstm_stack_launch_impl(which, P, dev):
    """Persistent single-launch recurrence when the shape qualifies (H % 256 == 0, H <= 1024;
    NSP_LSTM_PERSISTENT=0 disables) AND the device can hold the whole grid (the C side checks
    occupancy x CU count and answers NSP_EUNSUPPORTED otherwise), else one launch per wavefront
    stage.  The persistent kernels take at most 64 utterances (4 batch blocks); a larger batch is
    cut into slabs of 64 -- utterances are independent and [B, L, .] tensors are contiguous per
    utterance -- launched one after the other (the dropout counters are offset so that the masks do
    not depend on the cut).  Each launch's `dead` word is copied to pinned memory behind the kernel
    and checked (without a sync) at the next launch / by ops.lstm_check()."""
    lib = _lib.lib()
    if os.environ.get('NSP_LSTM_PERSISTENT', '1') != '0' and P.H % 256 == 0 and P.H <= 1024 \
            and P.nl * (P.H // 16) <= 256:
        _lstm_poll_dead()
        fn = lib.nsp_lstm_stack_fwd_persistent if which == 'fwd' else lib.nsp_lstm_stack_bwd_persistent
        B, L, H, nl = P.B, P.L, P.H, P.nl
        fell_back = False
        checks = []
        for b0 in range(0, B, 64):
            Q = P
            if B > 64:
                Q = _lib.LstmStackParams()
                ctypes.memmove(ctypes.byref(Q), ctypes.byref(P), ctypes.sizeof(P))
                Q.B = min(64, B - b0)
                rows = b0 * L

                def off(ptr, per_row_bytes):
                    return ptr + rows * per_row_bytes if ptr else ptr
                Q.gi0 = off(P.gi0, 4 * H * 4)
                Q.dy_top = off(P.dy_top, H * 4)
                Q.y_top = off(P.y_top, H * 4)
                for l in range(nl):
                    Q.hp16[l] = off(P.hp16[l], H * 2)
                    Q.yd16[l] = off(P.yd16[l], H * 2)
                    Q.c_all[l] = off(P.c_all[l], H * 4)
                    Q.gates[l] = off(P.gates[l], 4 * H * 4)
                    Q.dg16[l] = off(P.dg16[l], 4 * H * 2)
                    Q.dc[l] = (P.dc[l] + b0 * H * 4) if P.dc[l] else P.dc[l]
                    Q.offset[l] = P.offset[l] + rows * H
            sync = torch.empty((2,), device=dev, dtype=torch.int32)   # zeroed by the call itself
            rc = fn(ctypes.byref(Q), sync.data_ptr(), _stream())
            if rc == -2 and b0 == 0:        # NSP_EUNSUPPORTED: the grid does not fit this device
                fell_back = True
                break
            _check(rc, 'nsp_lstm_stack_%s_persistent' % which)
            flag = torch.empty((1,), dtype=torch.int32, pin_memory=True)
            flag.copy_(sync[1:2], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            _LSTM_DEAD_CHECKS.append((flag, ev, which))
            checks.append((flag, ev))
        if not fell_back:
            return checks
    fn = lib.nsp_lstm_stack_fwd if which == 'fwd' else lib.nsp_lstm_stack_bwd
    _check(fn(ctypes.byref(P), _stream()), 'nsp_lstm_stack_' + which)
    return []


# ---- in-step rescue of a FORWARD whose persistent launch timed out at its grid barrier (its workgroups never became
# co-resident).  The forward registers what a re-run needs; the decoder calls lstm_forward_resolve() where the prediction
# network's output joins the step, BEFORE anything has consumed it (decoders.RNNT defers the network's tail -- dropout and
# the output projection -- to that point): the host waits for the launch's event (free in practice: the launch started a
# whole encoder forward earlier), reads its `dead` word, and on a time-out re-runs the recurrence with one launch per stage
# on the same stream into the same tensors, switches the process to per-stage launches and says so.  A BACKWARD time-out
# still raises at the next poll (its results have been consumed by the weight-gradient GEMMs by the time it is known).
_LSTM_FWD_RESCUE = []
_LSTM_RESCUES = [0]          # (how many forwards have been re-run in this process: tests, logs)


def lstm_forward_resolve(block=True):
    """-> number of forward launches that had to be re-run.  block=False (the prediction network ran INLINE on the step's only
    stream -- stock DistributedDataParallel): waiting for the launch would drain the whole queue the host has built up, so a
    launch that has not finished yet is left to the one-step-late check (ops.lstm_check / the next launch), as before"""
    if not _LSTM_FWD_RESCUE:
        return 0
    pending = list(_LSTM_FWD_RESCUE)
    del _LSTM_FWD_RESCUE[:]
    n = 0
    for P, keep, checks, stream in pending:
        dead = False
        if not block and any(not ev.query() for _, ev in checks):
            continue
        for flag, ev in checks:
            ev.synchronize()
            dead = dead or int(flag[0]) != 0
        if os.environ.get('NSP_LSTM_TEST_FAKE_TIMEOUT', '0') == '1':      # test hook: treat the next launch as timed out, once
            os.environ['NSP_LSTM_TEST_FAKE_TIMEOUT'] = '0'
            dead = True
        if not dead:
            continue
        n += 1
        _LSTM_RESCUES[0] += 1
        mine = set(id(f) for f, _ in checks)
        _LSTM_DEAD_CHECKS[:] = [c for c in _LSTM_DEAD_CHECKS if id(c[0]) not in mine]
        os.environ['NSP_LSTM_PERSISTENT'] = '0'
        import logging
        logging.getLogger(__name__).warning(
            'neural_sp_amd: a persistent LSTM forward timed out at its grid barrier (its %d workgroups did not become '
            'co-resident); re-running the recurrence with one launch per stage and keeping per-stage launches for the rest '
            'of this process (NSP_LSTM_PERSISTENT=0)', P.nl * (P.H // 16))
        _check(_lib.lib().nsp_lstm_stack_fwd(ctypes.byref(P), stream), 'nsp_lstm_stack_fwd (rescue)')
    return n


def lstm_check():
    """Wait for the outstanding persistent-LSTM launches and raise if one of them timed out."""
    _lstm_poll_dead(block=True)


class LSTMStackFn(torch.autograd.Function):
    """All LSTM layers of the prediction network as one wavefront over (layer, time)
    (nsp_lstm_stack_*; bf16 mode).  y = top layer's output, before the decoder's final dropout;
    the dropout BETWEEN layers (rnn_transducer.py:303) is fused into the recurrence."""

    @staticmethod
    def forward(ctx, x, p_drop, *ws):
        nl = len(ws) // 4
        B, L, I = x.shape
        H = ws[1].shape[1]
        dev = x.device
        x2d = _f32c(x).reshape(B * L, I)
        xa = to_bf16(x2d) if I % 8 == 0 else x2d
        gi0 = linear_fwd(xa, ws[0], axpby(ws[2], ws[3], 1.0, 1.0))                # [B*L, 4H] fp32
        if nl > 1 and _lstm_layerwise(B, H):
            return LSTMStackFn._forward_layerwise(ctx, xa, gi0, p_drop, ws, (nl, B, L, I, H), dev)
        P = _lib.LstmStackParams()
        P.nl, P.B, P.L, P.H, P.dropout_p = nl, B, L, H, float(p_drop)
        P.gi0 = gi0.data_ptr()
        y_top = torch.empty((B, L, H), device=dev, dtype=torch.float32)
        P.y_top = y_top.data_ptr()
        keep = [gi0]
        hp16, yd16, c_all, gates, seeds = [], [], [], [], []
        for l in range(nl):
            w_ih, w_hh, b_ih, b_hh = ws[4 * l:4 * l + 4]
            if l == 0:
                wl = weight_bf16(w_hh)
            else:
                wl = _cat_cached(w_hh, '_nsp_lstm_cat', (w_ih, w_hh),
                                 lambda a=w_ih, b=w_hh: torch.cat([weight_bf16(a), weight_bf16(b)], dim=1).contiguous(),
                                 lambda t, a=w_ih, b=w_hh: [(a, 0, 0, a.shape[0], a.shape[1], False),
                                                            (b, 0, _r8(a.shape[1]), b.shape[0], b.shape[1], False)])
                bl = axpby(b_ih, b_hh, 1.0, 1.0)
                P.bias[l] = bl.data_ptr()
                keep.append(bl)
            P.w[l] = wl.data_ptr()
            keep.append(wl)
            hp16.append(torch.empty((B, L, H), device=dev, dtype=torch.bfloat16))
            c_all.append(torch.empty((B, L, H), device=dev, dtype=torch.float32))
            gates.append(torch.empty((B, L, 4 * H), device=dev, dtype=torch.float32))
            P.hp16[l], P.c_all[l], P.gates[l] = hp16[l].data_ptr(), c_all[l].data_ptr(), gates[l].data_ptr()
            if l < nl - 1:
                yd16.append(torch.empty((B, L, H), device=dev, dtype=torch.bfloat16))
                P.yd16[l] = yd16[l].data_ptr()
                sd = next_dropout_seed() if p_drop > 0 else (0, 0)
                seeds.append(sd)
                P.seed[l], P.offset[l] = sd
        xchg = _lstm_xchg(P, 2 * H, dev)
        checks = _lstm_stack_launch('fwd', P, dev)
        del xchg
        if checks or os.environ.get('NSP_LSTM_TEST_FAKE_TIMEOUT', '0') == '1':
            del _LSTM_FWD_RESCUE[:-1]          # (launches nobody resolved -- ops.lstm_stack driven directly -- do not pile up: at most
                                               #  two forwards' activations are ever pinned here: main + one auxiliary decoder)
            _LSTM_FWD_RESCUE.append((P, keep + [y_top] + hp16 + yd16 + c_all + gates, checks or [], _stream()))
        ctx.save_for_backward(xa, *ws, *hp16, *yd16, *c_all, *gates)
        ctx.cfg = (nl, B, L, I, H, float(p_drop), seeds)
        return y_top

    @staticmethod
    def _forward_layerwise(ctx, xa, gi0, p_drop, ws, dims, dev):
        """One persistent launch PER LAYER (H/16 workgroups each) with the next layer's input projection as one GEMM
        over all steps in between, instead of the (layer, time) wavefront on nl * H/16 workgroups.  The wavefront
        minimises latency; but the recurrence runs on a side stream beside the encoder, where what it costs the step
        is the CUs it holds times how long it holds them ("The persistent LSTM in round 3"): a stage of the
        single-layer kernel reads only its own h (the neighbour's half is a GEMM at full speed).  See _lstm_layerwise
        for what it measured."""
        nl, B, L, I, H = dims
        hp16, yd16, c_all, gates, seeds = [], [], [], [], []
        gi, y = gi0, None
        for l in range(nl):
            w_ih, w_hh, b_ih, b_hh = ws[4 * l:4 * l + 4]
            if l > 0:
                gi = linear_fwd(yd16[l - 1].view(B * L, H), w_ih, axpby(b_ih, b_hh, 1.0, 1.0))
            P = _lib.LstmStackParams()
            P.nl, P.B, P.L, P.H, P.dropout_p = 1, B, L, H, 0.0
            P.gi0 = gi.data_ptr()
            y = torch.empty((B, L, H), device=dev, dtype=torch.float32)
            P.y_top = y.data_ptr()
            wl = weight_bf16(w_hh)
            P.w[0] = wl.data_ptr()
            hp16.append(torch.empty((B, L, H), device=dev, dtype=torch.bfloat16))
            c_all.append(torch.empty((B, L, H), device=dev, dtype=torch.float32))
            gates.append(torch.empty((B, L, 4 * H), device=dev, dtype=torch.float32))
            P.hp16[0], P.c_all[0], P.gates[0] = hp16[l].data_ptr(), c_all[l].data_ptr(), gates[l].data_ptr()
            xchg = _lstm_xchg(P, 2 * H, dev)
            _lstm_stack_launch('fwd', P, dev)
            del xchg, wl
            if l < nl - 1:
                sd = next_dropout_seed() if p_drop > 0 else (0, 0)
                seeds.append(sd)
                yd = dropout_raw(y, p_drop, sd[0], sd[1]) if p_drop > 0 else y
                yd16.append(to_bf16(yd.view(B * L, H)).view(B, L, H))
        ctx.save_for_backward(xa, *ws, *hp16, *yd16, *c_all, *gates)
        ctx.cfg = (nl, B, L, I, H, float(p_drop), seeds)
        return y

    @staticmethod
    def backward(ctx, dy):
        nl, B, L, I, H, p_drop, seeds = ctx.cfg
        sv = ctx.saved_tensors
        xa = sv[0]
        ws = sv[1:1 + 4 * nl]
        o = 1 + 4 * nl
        hp16 = sv[o:o + nl]; o += nl
        yd16 = sv[o:o + nl - 1]; o += nl - 1
        c_all = sv[o:o + nl]; o += nl
        gates = sv[o:o + nl]
        dev = dy.device
        dy = _f32c(dy)
        keep, dg16 = [], []
        layerwise = nl > 1 and _lstm_layerwise(B, H)
        if layerwise:
            dg16 = [None] * nl
            dyl = dy
            for l in range(nl - 1, -1, -1):
                P = _lib.LstmStackParams()
                P.nl, P.B, P.L, P.H, P.dropout_p = 1, B, L, H, 0.0
                P.dy_top = dyl.data_ptr()
                wt = _weight_t_shadow(ws[4 * l + 1], True)                        # [H, 4H]
                P.w[0] = wt.data_ptr()
                dg16[l] = torch.empty((B, L, 4 * H), device=dev, dtype=torch.bfloat16)
                dc = torch.empty((B, H), device=dev, dtype=torch.float32)
                P.dg16[0], P.dc[0] = dg16[l].data_ptr(), dc.data_ptr()
                P.c_all[0], P.gates[0] = c_all[l].data_ptr(), gates[l].data_ptr()
                xchg = _lstm_xchg(P, 4 * H, dev)
                _lstm_stack_launch('bwd', P, dev)
                del xchg
                if l > 0:   # gradient of layer l's input = dropout(h_{l-1}): one GEMM over all steps, then the mask
                    dyl = linear_dgrad(dg16[l].view(B * L, 4 * H), ws[4 * l])[:, :H]
                    dyl = dyl if dyl.is_contiguous() else dyl.contiguous()
                    if p_drop > 0:
                        dyl = dropout_raw(dyl, p_drop, seeds[l - 1][0], seeds[l - 1][1])
        P = _lib.LstmStackParams()
        P.nl, P.B, P.L, P.H, P.dropout_p = nl, B, L, H, p_drop
        P.dy_top = dy.data_ptr()
        for l in range(0 if layerwise else nl):
            w_hh = ws[4 * l + 1]
            if l == nl - 1:
                wt = _weight_t_shadow(w_hh, True)                                 # [H, 4H]
            else:
                w_ih_up = ws[4 * (l + 1)]
                wt = _cat_cached(w_hh, '_nsp_lstm_catT', (w_ih_up, w_hh),
                                 lambda a=w_ih_up, b=w_hh: torch.cat(
                                     [_weight_t_shadow(a, True)[:, :4 * H], _weight_t_shadow(b, True)[:, :4 * H]],
                                     dim=1).contiguous(),                         # [H, 8H]
                                 lambda t, a=w_ih_up, b=w_hh: [(a, 0, 0, a.shape[1], a.shape[0], True),
                                                               (b, 0, 4 * H, b.shape[1], b.shape[0], True)])
                P.seed[l], P.offset[l] = seeds[l]
            keep.append(wt)
            P.w[l] = wt.data_ptr()
            dg16.append(torch.empty((B, L, 4 * H), device=dev, dtype=torch.bfloat16))
            dc = torch.empty((B, H), device=dev, dtype=torch.float32)
            keep.append(dc)
            P.dg16[l], P.dc[l] = dg16[l].data_ptr(), dc.data_ptr()
            P.c_all[l], P.gates[l] = c_all[l].data_ptr(), gates[l].data_ptr()
        if not layerwise:
            xchg = _lstm_xchg(P, 4 * H, dev)
            _lstm_stack_launch('bwd', P, dev)
            del xchg
        grads = []
        dx = None
        for l in range(nl):
            w_ih, w_hh = ws[4 * l], ws[4 * l + 1]
            g2d = dg16[l].view(B * L, 4 * H)
            inp = xa if l == 0 else yd16[l - 1].view(B * L, H)
            dw_ih = linear_wgrad(g2d, inp).view(w_ih.shape)
            dw_hh = linear_wgrad(g2d, hp16[l].view(B * L, H)).view(w_hh.shape)
            db = colsum(g2d)
            grads += [dw_ih, dw_hh, db, db]
            if l == 0 and ctx.needs_input_grad[0]:
                dx = linear_dgrad(g2d, w_ih)[:, :I].reshape(B, L, I)
        return (dx, None, *grads)


def lstm_stack(x, layers, p_drop):
    """layers: [(w_ih, w_hh, b_ih, b_hh), ...] (nn.LSTM parameters of consecutive 1-layer LSTMs)."""
    flat = [t for lay in layers for t in lay]
    return LSTMStackFn.apply(x, float(p_drop), *flat)


def lstm_stack_supported(layers, x):
    H = layers[0][1].shape[1]
    return (bf16_mode() and len(layers) <= _lib.LSTM_MAX_LAYERS and H % 64 == 0
            and all(lay[1].shape[1] == H for lay in layers)
            and all(lay[0].shape[1] == H for lay in layers[1:]))


class ReplayGraphFirstFn(torch.autograd.Function):
    """Identity on a detached copy of `y` whose backward replays y's own graph from inside this
    node.  The autograd engine orders ready nodes by creation time (latest first); a sub-graph
    that was built EARLY to overlap with other forward work (the prediction network, started on a
    side stream before the encoder) would therefore run its backward LAST, after the whole
    encoder backward has been enqueued -- serialising the LSTM backward at the end of the step.
    This node is created late (at the joint), so the replay starts first.

    The replay is a re-entrant backward, and the engine ends every backward by making the CALLER's
    current stream wait for the streams the gradients were accumulated on.  Called from the main
    stream that wait would stall the encoder backward behind the whole LSTM backward (measured:
    8.6 ms of idle main stream per step at batch 64).  The replay is therefore issued with the
    side stream current; the main stream joins it once, from a callback queued on the OUTER
    backward's graph task (it runs when that backward has finished enqueuing)."""

    @staticmethod
    def forward(ctx, leaf, holder, side):
        ctx.holder = holder
        ctx.side = side
        return leaf.view_as(leaf)

    @staticmethod
    def backward(ctx, dy):
        y = ctx.holder.pop()
        side = ctx.side
        if side is None or not dy.is_cuda:
            torch.autograd.backward(y, dy)
            return None, None, None
        main = torch.cuda.current_stream(dy.device)
        done = torch.cuda.Event()
        torch.autograd.Variable._execution_engine.queue_callback(lambda: main.wait_event(done))
        side.wait_stream(main)          # dy was produced on the main stream
        dy.record_stream(side)
        with torch.cuda.stream(side):
            torch.autograd.backward(y, dy)
            done.record(side)
        return None, None, None


def replay_graph_first(y, side_stream=None):
    leaf = y.detach().requires_grad_(True)
    return ReplayGraphFirstFn.apply(leaf, [y], side_stream)


def lstm(x, w_ih, w_hh, b_ih, b_hh):
    return LSTMFn.apply(x, w_ih, w_hh, b_ih, b_hh)


# --------------------------------------------------------------------------
# fused position-wise feed-forward (two GEMMs, everything else in their epilogues)
# --------------------------------------------------------------------------
class FFNFn(torch.autograd.Function):
    """out = res + dropout_o(alpha * (dropout_h(act(x W1^T + b1)) W2^T + b2))
    (positionwise_feed_forward.py:89 + conformer_block.py:131-134).  In bf16 mode the hidden
    activation and its pre-activation live only as bf16 tensors; backward gets d(pre) straight
    out of the W2 data-gradient GEMM epilogue (x act'(pre) x dropout mask), so no elementwise
    pass ever touches a [M, d_ff] tensor."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, act, p_h, res, alpha, p_o):
        K = x.shape[-1]
        dff, N = w1.shape[0], w2.shape[0]
        use16 = bf16_mode() and K % 8 == 0 and dff % 8 == 0
        sh = getattr(x, '_nsp16', None)
        if use16 and sh is not None:
            xa = sh.reshape(-1, K)
        else:
            x2d = _f32c(x).reshape(-1, K)
            xa = to_bf16(x2d) if use16 else x2d
        M = xa.shape[0]
        dt = torch.bfloat16 if use16 else torch.float32
        pre = torch.empty((M, dff), device=x.device, dtype=dt)
        h = torch.empty((M, dff), device=x.device, dtype=dt)
        s1 = next_dropout_seed() if p_h > 0 else (0, 0)
        s2 = next_dropout_seed() if p_o > 0 else (0, 0)
        linear_fwd(xa, w1, b1, act, None, 1.0, pre_out=pre, out=h, dropout_p=p_h, seed=s1[0], offset=s1[1])
        res2d = _f32c(res).reshape(-1, N) if res is not None else None
        y = linear_fwd(h, w2, b2, 0, res2d, alpha, dropout_p=p_o, seed=s2[0], offset=s2[1])
        ctx.save_for_backward(xa, w1, w2, pre, h)
        ctx.prep_token = _prep_offer(res is not None and use16, alpha, p_o, s2[0], s2[1], N)
        ctx.cfg = (act, p_h, s1, alpha, p_o, s2, res is not None, x.shape, use16)
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        xa, w1, w2, pre, h = ctx.saved_tensors
        act, p_h, s1, alpha, p_o, s2, has_res, xshape, use16 = ctx.cfg
        N = w2.shape[0]
        dy2d = _f32c(dy).reshape(-1, N)
        # both bias gradients ride in the kernels that produce their operands: db2 inside grad_prep,
        # db1 as column-sum slabs of the data-gradient GEMM's epilogue (no pass over g2 / d(pre))
        got = _prep_take(getattr(ctx, 'prep_token', None), dy2d)
        if got is not None:
            g2, db2 = got
        else:
            g2, db2 = grad_prep(dy2d, None, 0, alpha, p_o, s2[0], s2[1], use16, want_colsum=True)
        dw2 = linear_wgrad(g2, h).view(w2.shape)
        # d(pre) = (g2 W2) * dropout_h mask * act'(pre): all in the data-gradient epilogue
        M, dff = pre.shape
        dpre = torch.empty((M, dff), device=dy.device, dtype=pre.dtype)
        if use16 and os.environ.get('NSP_FUSED_COLSUM', '1') != '0':
            wt = _weight_t_shadow(w2, True)   # [dff, roundup64(N)]
            # one slab row per 32-row block of the tile grid (tiles may overhang M: up to the next multiple of 128)
            slabs = torch.zeros(((M + 127) // 128 * 4, dff), device=dy.device, dtype=torch.float32)
            gemm_raw(M, dff, N, g2, g2.stride(0), 1, wt, 1, wt.stride(0), dpre, dff,
                     dact_src=pre, dact=act, dropout_p=p_h, seed=s1[0], offset=s1[1], colsum_slabs=slabs)
            db1 = colsum(slabs)
        elif use16:
            wt = _weight_t_shadow(w2, True)
            gemm_raw(M, dff, N, g2, g2.stride(0), 1, wt, 1, wt.stride(0), dpre, dff,
                     dact_src=pre, dact=act, dropout_p=p_h, seed=s1[0], offset=s1[1])
            db1 = colsum(dpre)
        else:
            gemm_raw(M, dff, N, g2, g2.stride(0), 1, w2, w2.stride(0), 1, dpre, dff,
                     dact_src=pre, dact=act, dropout_p=p_h, seed=s1[0], offset=s1[1])
            db1 = colsum(dpre)
        dw1 = linear_wgrad(dpre, xa).view(w1.shape)
        dx = linear_dgrad(dpre, w1)[:, :xshape[-1]].reshape(xshape) if ctx.needs_input_grad[0] else None
        return dx, dw1, db1, dw2, db2, None, None, (dy if has_res else None), None, None


def ffn(x, w1, b1, w2, b2, act, p_h=0.0, res=None, alpha=1.0, p_o=0.0):
    return tag_prep(FFNFn.apply(x, w1, b1, w2, b2, ACT[act] if not isinstance(act, int) else act,
                                float(p_h), res, float(alpha), float(p_o)))


# --------------------------------------------------------------------------
# fused self-attention block (bf16 throughput mode)
# --------------------------------------------------------------------------
def _stacked_weight_bf16(ws):
    """bf16 [sum N_i, K] stack of several [N_i, K] parameters (e.g. W_q;W_k;W_v), cached on the first."""
    ver = tuple(_wkey(w) for w in ws)
    ent = getattr(ws[0], '_nsp_stack16', None)
    if ent is not None and ent[0] == ver and ent[1].device == ws[0].device and ent[2] == len(ws):
        return ent[1]
    wb = torch.cat([weight_bf16(w) for w in ws], dim=0).contiguous()
    try:
        ws[0]._nsp_stack16 = (ver, wb, len(ws))
        parts, r0 = [], 0
        for w in ws:
            parts.append((w, r0, 0, w.shape[0], w[0].numel(), False))
            r0 += w.shape[0]
        _shadow_register(ws[0], '_nsp_stack16', wb, parts, list(ws), single=False, extra=(len(ws),))
    except Exception:
        pass
    return wb


def _stacked_weight_t_bf16(ws):
    """bf16 [K, sum N_i] = [W_1^T | W_2^T | ...] (the data-gradient operand of a stacked projection)."""
    ver = tuple(_wkey(w) for w in ws)
    ent = getattr(ws[0], '_nsp_stackt16', None)
    if ent is not None and ent[0] == ver and ent[1].device == ws[0].device and ent[2] == len(ws):
        return ent[1]
    wt = to_bf16(torch.cat([w.detach().reshape(w.shape[0], -1).t() for w in ws], dim=1).contiguous())
    try:
        ws[0]._nsp_stackt16 = (ver, wt, len(ws))
        parts, c0 = [], 0
        for w in ws:
            parts.append((w, 0, c0, w[0].numel(), w.shape[0], True))
            c0 += w.shape[0]
        _shadow_register(ws[0], '_nsp_stackt16', wt, parts, list(ws), single=False, extra=(len(ws),))
    except Exception:
        pass
    return wt


def _r8(n):
    return (n + 7) // 8 * 8


class SelfAttnFn(torch.autograd.Function):
    """out = res + dropout_o( softmax((q k^T + shift(q pos^T))/sqrt(dk), masks) v  W_o^T + b_o )
    with q,k,v = x [W_q;W_k;W_v]^T (+biases), for RelMHA (pos_in given) and plain MHA (pos_in None).

    bf16-mode implementation of relative_multihead_attention.py:146-220 /
    multihead_attention.py:93-157 as ONE autograd node: one stacked QKV GEMM (bf16 out), score /
    context / gradient GEMMs on the bf16 MFMA kernel, probabilities kept only as a bf16 image
    [B,H,Tq,roundup8(Tk)], relative term as a [., roundup8(R)] table.  x may carry a bf16 shadow
    (from LayerNorm).  w_pos is the matrix that projects the position table (w_value itself in
    the reference's non-XL mode, :176): its gradient from the table path is returned separately
    and autograd sums it with the value-projection gradient."""

    @staticmethod
    def forward(ctx, x, wq, wk, wv, bq, bk, bv, wo, bo, pos_in, w_pos, klens, cfg, res, p_o):
        B, T, d = x.shape
        H = cfg['H']
        dk = d // H
        dev = x.device
        M = B * T
        sh = getattr(x, '_nsp16', None)
        x16 = sh.reshape(M, d) if sh is not None else to_bf16(_f32c(x).reshape(M, d))
        wqkv = _stacked_weight_bf16([wq, wk, wv])
        bqkv = torch.cat([bq, bk, bv]) if bq is not None else None
        d3 = 3 * d
        qkv = torch.empty((M, d3), device=dev, dtype=torch.bfloat16)
        gemm_raw(M, d3, d, x16, d, 1, wqkv, 1, d, qkv, d3, bias=bqkv)
        clamp = cfg.get('clamp', -1)
        Tkp = _r8(T)
        fused = (dk == 64 and os.environ.get('NSP_FLASH_ATTN', '1') != '0'
                 and (pos_in is None or (clamp > 0 and pos_in.shape[0] <= 16)))
        QP = pos16 = pe16 = None
        R = Rp = 0
        if pos_in is not None:
            R = pos_in.shape[0]
            Rp = _r8(R)
            # the zero-padded bf16 image of the position table is shared by every layer of this forward: the cache is a dict
            # owned by the position-embedding tensor the encoder made once for the step (modules.py hands it down in cfg;
            # `pos_in` itself is a fresh slice object per layer, an attribute on it would never be found again)
            cache = cfg.get('pe16_cache')
            key = (pos_in.data_ptr(), pos_in._version, R, Rp, str(dev))
            pe16 = cache.get(key) if cache is not None else None
            if pe16 is None:
                pe16 = torch.zeros((Rp, d), device=dev, dtype=torch.bfloat16)  # rows >= R stay zero
                pe16[:R] = to_bf16(_f32c(pos_in))
                if cache is not None:
                    cache[key] = pe16
            pos16 = torch.empty((Rp, d), device=dev, dtype=torch.bfloat16)
            gemm_raw(Rp, d, d, pe16, d, 1, weight_bf16(w_pos), 1, d, pos16, d)
            QP = torch.empty((B, T, H, Rp), device=dev, dtype=torch.float32)
            gemm_raw(M, Rp, dk, qkv, d3, 1, pos16, 1, d, QP, H * Rp, batch=(H, 1), a_b=(dk, 0),
                     b_b=(dk, 0), c_b=(Rp, 0))
        p_att = cfg.get('dropout', 0.0) if cfg.get('training', False) else 0.0
        seed, offset = next_dropout_seed() if p_att > 0 else (0, 0)
        mask_args = (cfg.get('causal', False), cfg.get('lookahead', 0), cfg.get('chunk_nl', 0),
                     cfg.get('chunk_nc', 0), p_att, seed, offset)
        scale = 1.0 / math.sqrt(dk)
        mp = _mask_params(B, H, T, T, R, clamp, scale, klens, *mask_args, p_bf16=1, tk_pitch=Tkp, r_pitch=Rp)
        P16 = Pd16 = LSE = cv32 = keepbits = None
        if fused:
            # flash-style kernel: scores / probabilities never leave the CU
            cv16, cv32, LSE, keepbits = flash_attn_fwd_raw(qkv, d, QP, mp, want_o32=any(ctx.needs_input_grad))
            aw = None
        else:
            S = torch.empty((B, H, T, T), device=dev, dtype=torch.float32)
            gemm_raw(T, T, dk, qkv, d3, 1, qkv, 1, d3, S, T, batch=(B, H), a_b=(T * d3, dk),
                     b_b=(T * d3, dk), c_b=(H * T * T, T * T), b_off=d)
            P16 = torch.empty((B, H, T, Tkp), device=dev, dtype=torch.bfloat16)
            Pd16 = torch.empty_like(P16) if p_att > 0 else None
            attn_softmax_fwd_raw(S, QP, mp, Pd16, Pout=P16)
            del S
            Puse = Pd16 if Pd16 is not None else P16
            cv16 = torch.empty((M, d), device=dev, dtype=torch.bfloat16)
            gemm_raw(T, dk, T, Puse, Tkp, 1, qkv, d3, 1, cv16, d, batch=(B, H), a_b=(H * T * Tkp, T * Tkp),
                     b_b=(T * d3, dk), c_b=(T * d, dk), b_off=2 * d)
            aw = P16
        s_o = next_dropout_seed() if p_o > 0 else (0, 0)
        res2d = _f32c(res).reshape(M, d) if res is not None else None
        out = linear_fwd(cv16, wo, bo, 0, res2d, 1.0, dropout_p=p_o, seed=s_o[0], offset=s_o[1])
        ctx.save_for_backward(x16, wq, wk, wv, wo, w_pos, qkv, pos16, pe16, P16, Pd16, cv16, klens, QP, LSE, cv32, keepbits)
        ctx.cfg = (B, T, d, H, dk, R, Rp, Tkp, clamp, scale, mask_args, p_o, s_o, res is not None,
                   bq is not None, bo is not None)
        ctx.prep_token = _prep_offer(res is not None, 1.0, p_o, s_o[0], s_o[1], d)
        if aw is None:
            aw = _no_aw(cv16.device).view(1)  # the fused path has no probability tensor to hand out (one cached placeholder per device)
        ctx.mark_non_differentiable(aw)
        return out.view(B, T, d), aw

    @staticmethod
    def backward(ctx, dy, _unused):
        x16, wq, wk, wv, wo, w_pos, qkv, pos16, pe16, P16, Pd16, cv16, klens, QP, LSE, cv32, keepbits = ctx.saved_tensors
        (B, T, d, H, dk, R, Rp, Tkp, clamp, scale, mask_args, p_o, s_o, has_res, has_qkv_bias,
         has_o_bias) = ctx.cfg
        has_pos = pos16 is not None
        fused = LSE is not None
        dev = dy.device
        M, d3 = B * T, 3 * d
        dy2d = _f32c(dy).reshape(M, d)
        got = _prep_take(getattr(ctx, 'prep_token', None), dy2d)
        dbo = None
        if got is not None:
            g, dbo = got
            if not has_o_bias:
                dbo = None
        else:
            g = grad_prep(dy2d, None, 0, 1.0, p_o, s_o[0], s_o[1], True, want_colsum=has_o_bias)
            if has_o_bias:
                g, dbo = g
        dwo = linear_wgrad(g, cv16).view(wo.shape)
        dO = linear_dgrad(g, wo, out_bf16=True)                                   # [M, d] bf16
        dqkv = torch.empty((M, d3), device=dev, dtype=torch.bfloat16)
        mp = _mask_params(B, H, T, T, R, clamp, scale, klens, *mask_args, p_bf16=1, tk_pitch=Tkp, r_pitch=Rp)
        if fused:
            # the kernel leaves the finished query gradient (dS k + dQP . pos) as bf16 in dqkv[:, :d]
            dq_acc, dQP = flash_attn_bwd_raw(qkv, d, QP, dO, cv32, LSE, keepbits, mp, dqkv, pos16=pos16, dq_in_dqkv=True)
        else:
            dP = torch.empty((B, H, T, T), device=dev, dtype=torch.float32)       # dP = dO v^T
            gemm_raw(T, T, dk, dO, d, 1, qkv, 1, d3, dP, T, batch=(B, H), a_b=(T * d, dk),
                     b_b=(T * d3, dk), c_b=(H * T * T, T * T), b_off=2 * d)
            Puse = Pd16 if Pd16 is not None else P16
            gemm_raw(T, dk, T, Puse, 1, Tkp, dO, d, 1, dqkv, d3, batch=(B, H),       # dV = P^T dO
                     a_b=(H * T * Tkp, T * Tkp), b_b=(T * d, dk), c_b=(T * d3, dk), c_off=2 * d)
            dS16 = torch.empty((B, H, T, Tkp), device=dev, dtype=torch.bfloat16)
            dQP = torch.empty((B, T, H, Rp), device=dev, dtype=torch.float32) if has_pos else None
            attn_softmax_bwd_raw(P16, dP, dQP, mp, dS=dS16)
            del dP
            dq_acc = None
        dq_pos = dw_pos = None
        if has_pos:
            dQP16 = to_bf16(dQP.view(M * H, Rp)).view(M, H * Rp)
            if not fused:
                # dq (position term) = dQP pos (the fused kernel adds it itself)
                dq_pos = torch.empty((M, d), device=dev, dtype=torch.float32)
                gemm_raw(M, dk, Rp, dQP16, H * Rp, 1, pos16, d, 1, dq_pos, d, batch=(H, 1), a_b=(Rp, 0),
                         b_b=(dk, 0), c_b=(dk, 0))
            if ctx.needs_input_grad[10]:
                dpos = zeros_small((Rp, d), dev)                                      # dQP^T q
                gemm_raw(Rp, dk, M, dQP16, 1, H * Rp, qkv, d3, 1, dpos, d, batch=(H, 1), a_b=(Rp, 0),
                         b_b=(dk, 0), c_b=(dk, 0), splitk=max(1, min(64, M // 256)))
                dpos16 = to_bf16(dpos)
                dw_pos = torch.empty((d, d), device=dev, dtype=torch.float32)        # dpos^T pe
                gemm_raw(d, d, Rp, dpos16, 1, d, pe16, d, 1, dw_pos, d)
                dw_pos = dw_pos.view(w_pos.shape)
        if not fused:
            if has_pos:
                # dq = dS k accumulated in fp32 on top of the position-term gradient, then cast
                gemm_raw(T, dk, T, dS16, Tkp, 1, qkv, d3, 1, dq_pos, d, batch=(B, H),
                         a_b=(H * T * Tkp, T * Tkp), b_b=(T * d3, dk), c_b=(T * d, dk), b_off=d, res=dq_pos)
                _check(_lib.lib().nsp_cast_bf16(_p(dq_pos), _p(dqkv), M, d, d, d3, _stream()), 'nsp_cast_bf16')
            else:
                gemm_raw(T, dk, T, dS16, Tkp, 1, qkv, d3, 1, dqkv, d3, batch=(B, H),
                         a_b=(H * T * Tkp, T * Tkp), b_b=(T * d3, dk), c_b=(T * d3, dk), b_off=d)
            gemm_raw(T, dk, T, dS16, 1, Tkp, qkv, d3, 1, dqkv, d3, batch=(B, H),       # dk = dS^T q
                     a_b=(H * T * Tkp, T * Tkp), b_b=(T * d3, dk), c_b=(T * d3, dk), c_off=d)
        dwqkv = linear_wgrad(dqkv, x16)                                            # [3d, d]
        dbqkv = colsum(dqkv) if has_qkv_bias else None
        wqkv_t = _stacked_weight_t_bf16([wq, wk, wv])                              # [d, 3d]
        dx = torch.empty((M, d), device=dev, dtype=torch.float32)
        gemm_raw(M, d, d3, dqkv, d3, 1, wqkv_t, 1, d3, dx, d)
        dwq, dwk, dwv = dwqkv[:d].view(wq.shape), dwqkv[d:2 * d].view(wk.shape), dwqkv[2 * d:].view(wv.shape)
        dbq = dbk = dbv = None
        if has_qkv_bias:
            dbq, dbk, dbv = dbqkv[:d], dbqkv[d:2 * d], dbqkv[2 * d:]
        return (dx.view(B, T, d), dwq, dwk, dwv, dbq, dbk, dbv, dwo, dbo, None, dw_pos, None, None,
                (dy if has_res else None), None)


# --------------------------------------------------------------------------
# fused (flash-style) attention core, d_k = 64
# --------------------------------------------------------------------------
_NO_AW = {}


def _no_aw(dev):
    t = _NO_AW.get(dev)
    if t is None:
        t = _NO_AW[dev] = torch.zeros(1, device=dev, dtype=torch.bfloat16)
    return t


def flash_attn_fwd_raw(qkv16, d, QP, mp, want_o32=True):
    """-> (O bf16, O32 fp32 or None, LSE, keepbits or None).  O32 is what backward's D = dO . O is formed from; keepbits
    (only with dropout) = the dropout decisions the forward drew, which backward reads instead of drawing them again."""
    M = qkv16.shape[0]
    O = torch.empty((M, d), device=qkv16.device, dtype=torch.bfloat16)
    O32 = torch.empty((M, d), device=qkv16.device, dtype=torch.float32) if want_o32 else None
    LSE = torch.empty((2, mp.B, mp.H, mp.Tq), device=qkv16.device, dtype=torch.float32)  # max, 1/sum
    keep = None
    if mp.dropout_p > 0:
        keep = torch.empty((_lib.lib().nsp_flash_attn_keepbits_bytes(mp.B, mp.H, mp.Tq),), device=qkv16.device,
                           dtype=torch.uint8)
    # algorithmic bytes: q, k, v read (bf16), O written as bf16 and fp32, the two statistics, the position scores, the dropout words
    nb = M * d * (6 + 2 + (4 if want_o32 else 0)) + 8 * mp.B * mp.H * mp.Tq + (QP.numel() * 4 if QP is not None else 0) + (keep.numel() if keep is not None else 0)
    with _kev_class('flash_fwd', 4.0 * mp.B * mp.H * mp.Tq * mp.Tk * 64, 'flop', nb):
        _check(_lib.lib().nsp_flash_attn_fwd(_p(qkv16), d, _p(QP), _p(O), _p(O32), _p(LSE), _p(keep), ctypes.byref(mp),
                                             _stream()), 'nsp_flash_attn_fwd')
    return O, O32, LSE, keep


def flash_attn_bwd_raw(qkv16, d, QP, dO16, O32, LSE, keep, mp, dqkv16, pos16=None, dq_in_dqkv=False):
    """-> (dq32 [M,d] fp32 or None, dQP or None); dK / dV are written into dqkv16 column blocks d / 2d.
    dq_in_dqkv: the query gradient is written FINISHED (incl. the position term's share dQP . pos16 when pos16 is given)
    as bf16 into column block 0 of dqkv16 and no fp32 dq exists; otherwise dq32 = dS k only (pos16 must be None)."""
    M = qkv16.shape[0]
    dev = qkv16.device
    dq32 = None if dq_in_dqkv else torch.empty((M, d), device=dev, dtype=torch.float32)
    dQP = torch.empty_like(QP) if QP is not None else None
    D = torch.empty((mp.B, mp.H, mp.Tq, 4), device=dev, dtype=torch.float32)      # per-query records dQ kernel -> dK/dV kernel
    # algorithmic bytes: q, k, v, dO (bf16), O (fp32), statistics, position scores and their gradient, dropout words read ONCE;
    # dq, dk, dv written (bf16, or dq as fp32)
    nb = (M * d * (6 + 2 + 4 + 4 + (2 if dq_in_dqkv else 4)) + 8 * mp.B * mp.H * mp.Tq + (QP.numel() * 8 if QP is not None else 0)
          + (keep.numel() if keep is not None else 0))
    with _kev_class('flash_bwd', 10.0 * mp.B * mp.H * mp.Tq * mp.Tk * 64, 'flop', nb):
        _check(_lib.lib().nsp_flash_attn_bwd(_p(qkv16), d, _p(QP), _p(dO16), _p(O32), _p(LSE), _p(keep), _p(D), _p(dqkv16),
                                             _p(dq32), _p(dQP), _p(pos16 if dq_in_dqkv else None), ctypes.byref(mp),
                                             _stream()), 'nsp_flash_attn_bwd')
    return dq32, dQP
