"""Host-side plumbing over the C ABI of libnsp_hip.so.

Raw wrappers (``*_raw``) take torch CUDA tensors, pass their device pointers +
shapes + the current HIP stream through ctypes, and return nothing that was not
allocated here by torch.  ``torch.autograd.Function`` subclasses stitch the raw
kernels into autograd.  No op has a CPU/eager fallback: a CPU tensor raises.
"""
import ctypes
import math

import torch

from neural_sp_amd import _lib
from neural_sp_amd._lib import GemmParams, AttnMaskParams

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
    assert t.dtype in (torch.float32, torch.int32, torch.int64), t.dtype
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32c(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


# --------------------------------------------------------------------------
# GEMM
# --------------------------------------------------------------------------
def gemm_raw(M, N, K, A, a_rs, a_cs, B, b_ks, b_ns, C, ldc,
             batch=(1, 1), a_b=(0, 0), b_b=(0, 0), c_b=(0, 0),
             bias=None, act=0, pre_out=None, dact_src=None, dact=0, res=None,
             alpha=1.0, splitk=1, mode=None, a_off=0, b_off=0, c_off=0,
             dropout_p=0.0, seed=0, offset=0):
    """C = epi(A @ B) with arbitrary strides (element offsets *_off into the tensors)."""
    p = GemmParams()
    p.M, p.N, p.K = int(M), int(N), int(K)
    p.A = A.data_ptr() + 4 * a_off
    p.a_rs, p.a_cs = int(a_rs), int(a_cs)
    p.B = B.data_ptr() + 4 * b_off
    p.b_ks, p.b_ns = int(b_ks), int(b_ns)
    p.C = C.data_ptr() + 4 * c_off
    p.ldc = int(ldc)
    p.batch1, p.batch2 = int(batch[0]), int(batch[1])
    p.a_b1, p.a_b2 = int(a_b[0]), int(a_b[1])
    p.b_b1, p.b_b2 = int(b_b[0]), int(b_b[1])
    p.c_b1, p.c_b2 = int(c_b[0]), int(c_b[1])
    p.bias = bias.data_ptr() if bias is not None else None
    p.act = int(act)
    p.pre_out = pre_out.data_ptr() + 4 * c_off if pre_out is not None else None
    p.dact_src = dact_src.data_ptr() + 4 * c_off if dact_src is not None else None
    p.dact = int(dact)
    p.res = res.data_ptr() + 4 * c_off if res is not None else None
    p.alpha = float(alpha)
    p.splitk = int(splitk)
    p.mode = _COMPUTE_MODE['mode'] if mode is None else int(mode)
    p.dropout_p = float(dropout_p)
    p.seed, p.offset = int(seed), int(offset)
    for t in (A, B, C, bias, pre_out, dact_src, res):
        if t is not None:
            assert t.is_cuda and t.dtype == torch.float32
    _check(_lib.lib().nsp_gemm(ctypes.byref(p), _stream()), 'nsp_gemm')


def _pick_splitk(M, N, K, batch=1):
    """Split the reduction when the output grid alone cannot fill 256 CUs."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128) * batch
    if tiles >= 256 or K < 1024:
        return 1
    want = max(1, 512 // tiles)
    return int(max(1, min(want, K // 256, 64)))


def linear_fwd(x2d, weight, bias=None, act=0, res=None, alpha=1.0, pre_out=None, out=None):
    """y[M,N] = res + alpha*act(x2d[M,K] @ weight[N,K]^T + bias)."""
    M, K = x2d.shape
    N = weight.shape[0]
    assert weight.shape[1] == K and x2d.stride(1) == 1 and weight.stride(1) == 1
    y = out if out is not None else torch.empty((M, N), device=x2d.device, dtype=torch.float32)
    gemm_raw(M, N, K, x2d, x2d.stride(0), 1, weight, 1, weight.stride(0), y, y.stride(0),
             bias=bias, act=act, res=res, alpha=alpha, pre_out=pre_out)
    return y


def linear_dgrad(dy2d, weight, dact_src=None, dact=0, alpha=1.0, res=None, out=None):
    """dx[M,K] = res + alpha*(dy2d[M,N] @ weight[N,K]) * act'(dact_src)."""
    M, N = dy2d.shape
    K = weight.shape[1]
    dx = out if out is not None else torch.empty((M, K), device=dy2d.device, dtype=torch.float32)
    gemm_raw(M, K, N, dy2d, dy2d.stride(0), 1, weight, weight.stride(0), 1, dx, dx.stride(0),
             dact_src=dact_src, dact=dact, alpha=alpha, res=res)
    return dx


def linear_wgrad(dy2d, x2d, alpha=1.0):
    """dW[N,K] = alpha * dy2d[M,N]^T @ x2d[M,K] (split over M, atomically reduced)."""
    M, N = dy2d.shape
    K = x2d.shape[1]
    sk = _pick_splitk(N, K, M)
    dw = (torch.zeros if sk > 1 else torch.empty)((N, K), device=dy2d.device, dtype=torch.float32)
    gemm_raw(N, K, M, dy2d, 1, dy2d.stride(0), x2d, x2d.stride(0), 1, dw, K, alpha=alpha, splitk=sk)
    return dw


def colsum(x2d, alpha=1.0):
    rows, cols = x2d.shape
    out = torch.zeros((cols,), device=x2d.device, dtype=torch.float32)
    _check(_lib.lib().nsp_colsum(_p(x2d), _p(out), ctypes.c_int(rows), ctypes.c_int(cols),
                                 ctypes.c_longlong(x2d.stride(0)), ctypes.c_int(1), _stream()),
           'nsp_colsum')
    if alpha != 1.0:
        out.mul_(alpha)
    return out


class LinearFn(torch.autograd.Function):
    """y = res + alpha * act(x W^T + b); x is [..., K] (nn.Linear semantics)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, res, alpha):
        x2d = _f32c(x).reshape(-1, x.shape[-1])
        weight = _f32c(weight)
        res2d = _f32c(res).reshape(-1, weight.shape[0]) if res is not None else None
        pre = None
        if act != 0 and (x.requires_grad or weight.requires_grad):
            pre = torch.empty((x2d.shape[0], weight.shape[0]), device=x.device, dtype=torch.float32)
        y = linear_fwd(x2d, weight, bias, act, res2d, alpha, pre_out=pre)
        ctx.save_for_backward(x2d, weight, pre)
        ctx.act, ctx.alpha = act, alpha
        ctx.has_bias, ctx.has_res = bias is not None, res is not None
        ctx.xshape = x.shape
        return y.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2d, weight, pre = ctx.saved_tensors
        dy2d = _f32c(dy).reshape(-1, weight.shape[0])
        dres = dy if ctx.has_res else None
        if ctx.act != 0:
            dpre = dact_mul(dy2d, pre, ctx.act, ctx.alpha)
        elif ctx.alpha != 1.0:
            dpre = dy2d * ctx.alpha
        else:
            dpre = dy2d
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = linear_dgrad(dpre, weight).view(ctx.xshape)
        if ctx.needs_input_grad[1]:
            dw = linear_wgrad(dpre, x2d)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = colsum(dpre)
        return dx, dw, db, None, dres, None


def linear(x, weight, bias=None, act='none', res=None, alpha=1.0):
    return LinearFn.apply(x, weight, bias, ACT[act] if not isinstance(act, int) else act, res, alpha)


def dact_mul(dy, pre, act, alpha=1.0):
    """dy * act'(pre) * alpha (elementwise)."""
    out = torch.empty_like(dy)
    _check(_lib.lib().nsp_dact_mul(_p(dy), _p(pre), _p(out), ctypes.c_int(act), ctypes.c_float(alpha),
                                   ctypes.c_longlong(dy.numel()), _stream()), 'nsp_dact_mul')
    return out


def axpby(x, z=None, alpha=1.0, beta=1.0, out=None):
    """alpha*x + beta*z on contiguous fp32 tensors."""
    x = _f32c(x)
    z = _f32c(z) if z is not None else None
    out = torch.empty_like(x) if out is None else out
    _check(_lib.lib().nsp_axpby(_p(x), _p(z), _p(out), ctypes.c_float(alpha), ctypes.c_float(beta),
                                ctypes.c_longlong(x.numel()), _stream()), 'nsp_axpby')
    return out


def act_fwd(x, act):
    x = _f32c(x)
    y = torch.empty_like(x)
    _check(_lib.lib().nsp_act_fwd(_p(x), _p(y), ctypes.c_int(act), ctypes.c_longlong(x.numel()),
                                  _stream()), 'nsp_act_fwd')
    return y


# --------------------------------------------------------------------------
# LayerNorm
# --------------------------------------------------------------------------
def layernorm_fwd_raw(x2d, gamma, beta, eps, act=0, want_pre=False):
    rows, d = x2d.shape
    y = torch.empty_like(x2d)
    mean = torch.empty((rows,), device=x2d.device, dtype=torch.float32)
    rstd = torch.empty((rows,), device=x2d.device, dtype=torch.float32)
    y_pre = torch.empty_like(x2d) if (act != 0 and want_pre) else None
    _check(_lib.lib().nsp_layernorm_fwd(_p(x2d), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd),
                                        ctypes.c_int(rows), ctypes.c_int(d), ctypes.c_float(eps),
                                        ctypes.c_int(act), _p(y_pre), _stream()), 'nsp_layernorm_fwd')
    return y, mean, rstd, y_pre


def layernorm_bwd_raw(dy2d, x2d, gamma, mean, rstd, y_pre, act=0):
    rows, d = x2d.shape
    dx = torch.empty_like(x2d)
    dgb = torch.zeros((2, d), device=x2d.device, dtype=torch.float32)
    _check(_lib.lib().nsp_layernorm_bwd(_p(dy2d), _p(x2d), _p(gamma), _p(mean), _p(rstd), _p(y_pre),
                                        _p(dx), ctypes.c_void_p(dgb.data_ptr()),
                                        ctypes.c_void_p(dgb.data_ptr() + 4 * d),
                                        ctypes.c_int(rows), ctypes.c_int(d), ctypes.c_int(act),
                                        _stream()), 'nsp_layernorm_bwd')
    return dx, dgb[0], dgb[1]


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, act):
        x2d = _f32c(x).reshape(-1, x.shape[-1])
        y, mean, rstd, y_pre = layernorm_fwd_raw(x2d, gamma, beta, eps, act, want_pre=True)
        ctx.save_for_backward(x2d, gamma, mean, rstd, y_pre)
        ctx.act = act
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2d, gamma, mean, rstd, y_pre = ctx.saved_tensors
        dy2d = _f32c(dy).reshape(x2d.shape)
        dx, dg, db = layernorm_bwd_raw(dy2d, x2d, gamma, mean, rstd, y_pre, ctx.act)
        return dx.view(dy.shape), dg, db, None, None


def layer_norm(x, gamma, beta, eps=1e-12, act='none'):
    return LayerNormFn.apply(x, gamma, beta, eps, ACT[act] if not isinstance(act, int) else act)


# --------------------------------------------------------------------------
# Attention (score GEMMs + fused masked/relative softmax)
# --------------------------------------------------------------------------
def _mask_params(B, H, Tq, Tk, R, clamp, scale, klens, causal=False, lookahead=0,
                 chunk_nl=0, chunk_nc=0, dropout_p=0.0, seed=0, offset=0):
    p = AttnMaskParams()
    p.B, p.H, p.Tq, p.Tk, p.R = B, H, Tq, Tk, R
    p.clamp = int(clamp)
    p.scale = float(scale)
    p.klens = klens.data_ptr() if klens is not None else None
    p.causal, p.lookahead = int(bool(causal)), int(lookahead)
    p.chunk_nl, p.chunk_nc = int(chunk_nl), int(chunk_nc)
    p.dropout_p = float(dropout_p)
    p.seed, p.offset = int(seed), int(offset)
    return p


def attn_softmax_fwd_raw(S, QP, mp, Pdrop=None):
    _check(_lib.lib().nsp_attn_softmax_fwd(_p(S), _p(QP), _p(Pdrop), ctypes.byref(mp), _stream()),
           'nsp_attn_softmax_fwd')


def attn_softmax_bwd_raw(P, dP, dQP, mp):
    _check(_lib.lib().nsp_attn_softmax_bwd(_p(P), _p(dP), _p(dQP), ctypes.byref(mp), _stream()),
           'nsp_attn_softmax_bwd')
