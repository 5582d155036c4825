"""Route neural_sp_amd.ops' C-ABI calls to the host-emulated kernels (tests/hipemu) -- TEST ONLY.

`with emulated_kernels():` swaps the loader's `lib()` for the emulator library, lets `_p()` pass CPU
addresses and makes `_stream()` a null stream, so the *real* ctypes glue and autograd Functions of
ops.py run unchanged on CPU tensors (every kernel file is in EMULATED_SOURCES).  A symbol absent from the
emulator library raises AttributeError: nothing silently falls back.
"""
import contextlib
import ctypes

from tests.hipemu import build_emu

_emu = None


def emu_lib():
    global _emu
    if _emu is None:
        from neural_sp_amd import _lib
        lib = ctypes.CDLL(build_emu.build())
        for name, (ret, argtypes) in _lib.prototypes().items():
            try:
                fn = getattr(lib, name)
            except AttributeError:
                continue
            fn.restype, fn.argtypes = ret, argtypes
        _emu = lib
    return _emu


@contextlib.contextmanager
def emulated_kernels():
    from neural_sp_amd import _lib, ops
    import torch
    saved = (_lib.lib, ops._p, ops._stream, ops.zeros_small, ops._require_device, ops.on_kernel_device)
    lib = emu_lib()
    _lib.lib = lambda: lib
    ops._p = lambda t: None if t is None else t.data_ptr()
    ops._stream = lambda: 0
    ops.zeros_small = lambda shape, device, dtype=torch.float32: torch.zeros(shape, device=device, dtype=dtype)
    ops._require_device = lambda *tensors: None
    ops.on_kernel_device = lambda t: True
    try:
        yield lib
    finally:
        _lib.lib, ops._p, ops._stream, ops.zeros_small, ops._require_device, ops.on_kernel_device = saved
