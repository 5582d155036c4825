"""Poison mode -- TEST INFRASTRUCTURE ONLY.  `NSP_POISON=1` makes every `torch.empty` / `torch.empty_like` /
`Tensor.new_empty` hand out memory filled with NaN (floating point) or a large bit pattern (integers) instead of whatever
the caching allocator recycled.  A kernel that reads a workspace / output element it was supposed to have written first then
produces NaN (or an index far out of range) on EVERY run, instead of the right answer on fresh (zero) hipMalloc blocks and
a slightly wrong one on recycled blocks (the round-4 stock-DDP failure: rank 1's second iteration).  Both tiers: the device
suite (`NSP_POISON=1 pytest -m gpu`) and the host emulator (`NSP_POISON=1 pytest -m "not gpu"`).  The spawned DDP workers
call `enable_from_env()` themselves (a spawned interpreter does not run conftest)."""
import os

_STATE = {'on': False, 'count': 0}


def enable():
    if _STATE['on']:
        return
    import torch
    _STATE['on'] = True
    real_empty, real_empty_like, real_new_empty = torch.empty, torch.empty_like, torch.Tensor.new_empty

    def _fill(t):
        if t.numel() == 0:
            return t
        _STATE['count'] += 1
        with torch.no_grad():
            if t.dtype.is_floating_point or t.dtype.is_complex:
                t.fill_(float('nan'))
            elif t.dtype == torch.bool:
                t.fill_(True)
            elif t.dtype == torch.uint8:
                t.fill_(0xA5)
            elif t.dtype == torch.int8:
                t.fill_(0x5A)
            elif t.dtype == torch.int16:
                t.fill_(0x5A5A)
            elif t.dtype == torch.int32:
                t.fill_(0x5A5A5A5A)
            else:
                t.fill_(0x5A5A5A5A5A5A5A5A)
        return t

    def empty(*a, **k):
        return _fill(real_empty(*a, **k))

    def empty_like(*a, **k):
        return _fill(real_empty_like(*a, **k))

    def new_empty(self, *a, **k):
        return _fill(real_new_empty(self, *a, **k))

    torch.empty = empty
    torch.empty_like = empty_like
    torch.Tensor.new_empty = new_empty


def enable_from_env():
    if os.environ.get('NSP_POISON', '0') not in ('', '0'):
        enable()
        return True
    return False


def count():
    return _STATE['count']


# ---- guard bands: every device tensor made by torch.empty / empty_like / zeros / zeros_like / new_empty / new_zeros (and
# ops.zeros_small) sits between two 4-KB bands of a byte pattern; check_guards() names the allocation whose band a kernel
# wrote into.  The host emulator's AddressSanitizer build does this for every kernel it can run -- the persistent (grid
# barrier) LSTM kernels are not among them; this is their out-of-bounds check, on the device.
_GUARD_BYTES = 4096
_PATTERN = 0xA5
_GUARDED = []


class guards(object):
    """`with guards() as g: step(); g.check()`"""

    def __enter__(self):
        import traceback
        import torch
        from neural_sp_amd import ops
        self.saved = (torch.empty, torch.empty_like, torch.zeros, torch.zeros_like, torch.Tensor.new_empty, torch.Tensor.new_zeros,
                      ops.zeros_small)
        real_empty = torch.empty
        del _GUARDED[:]

        def make(shape, dtype, device, zero):
            dtype = dtype or torch.get_default_dtype()
            n = 1
            for v in shape:
                n *= int(v)
            es = torch.empty((), dtype=dtype).element_size()
            g = _GUARD_BYTES // es
            base = real_empty((n + 2 * g,), dtype=dtype, device=device)
            u8 = base.view(torch.uint8)
            u8[:_GUARD_BYTES] = _PATTERN
            u8[_GUARD_BYTES + n * es:] = _PATTERN
            if zero:
                base[g:g + n].zero_()
            where = ' <- '.join('%s:%d' % (f.filename.rsplit('/', 1)[-1], f.lineno) for f in traceback.extract_stack(limit=7)[:-2][::-1][:4])
            _GUARDED.append((u8, n * es, where, tuple(shape), dtype))
            return base[g:g + n].view(tuple(shape))

        def norm(size):
            if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
                return tuple(size[0])
            return tuple(size)

        def want(device, k):
            dev = torch.device(device) if device is not None else None
            return dev is not None and dev.type == 'cuda' and not k.get('pin_memory') and k.get('memory_format') in (None, torch.contiguous_format) \
                and k.get('layout') in (None, torch.strided) and not k.get('requires_grad')

        def wrap(real, zero):
            def f(*size, dtype=None, device=None, **k):
                if want(device, k) and size and not torch.is_tensor(size[0]):
                    return make(norm(size), dtype, device, zero)
                return real(*size, dtype=dtype, device=device, **k)
            return f

        def wrap_like(real, zero):
            def f(x, *a, dtype=None, device=None, **k):
                dev = device if device is not None else x.device
                if not a and want(dev, k) and x.is_contiguous():
                    return make(tuple(x.shape), dtype or x.dtype, dev, zero)
                return real(x, *a, dtype=dtype, device=device, **k)
            return f

        def wrap_new(real, zero):
            def f(self_, *size, dtype=None, device=None, **k):
                dev = device if device is not None else self_.device
                if want(dev, k):
                    return make(norm(size), dtype or self_.dtype, dev, zero)
                return real(self_, *size, dtype=dtype, device=device, **k)
            return f
        torch.empty, torch.zeros = wrap(self.saved[0], False), wrap(self.saved[2], True)
        torch.empty_like, torch.zeros_like = wrap_like(self.saved[1], False), wrap_like(self.saved[3], True)
        torch.Tensor.new_empty, torch.Tensor.new_zeros = wrap_new(self.saved[4], False), wrap_new(self.saved[5], True)
        ops.zeros_small = lambda shape, device, dtype=torch.float32: torch.zeros(tuple(shape), device=device, dtype=dtype)
        return self

    def __exit__(self, *a):
        import torch
        from neural_sp_amd import ops
        (torch.empty, torch.empty_like, torch.zeros, torch.zeros_like, torch.Tensor.new_empty, torch.Tensor.new_zeros,
         ops.zeros_small) = self.saved
        del _GUARDED[:]

    @staticmethod
    def check():
        """-> list of (where allocated, shape, dtype, 'before' / 'after', first damaged byte offset in the band, damaged bytes)"""
        import torch
        torch.cuda.synchronize()
        bad = []
        for u8, nbytes, where, shape, dtype in _GUARDED:
            lo = (u8[:_GUARD_BYTES] != _PATTERN)
            hi = (u8[_GUARD_BYTES + nbytes:] != _PATTERN)
            if lo.any().item():
                idx = lo.nonzero()
                bad.append((where, shape, str(dtype), 'before', int(idx[-1]) - _GUARD_BYTES, int(lo.sum())))
            if hi.any().item():
                idx = hi.nonzero()
                bad.append((where, shape, str(dtype), 'after', int(idx[0]), int(hi.sum())))
        return bad, len(_GUARDED)
