"""The C-ABI shared library builds (hipcc cross-compiles gfx950 without a GPU), loads, and
exports every symbol declared in include/nsp_hip.h.  No compute calls here."""
import ctypes
import os

import pytest


def test_library_builds_and_exports_all_symbols():
    from neural_sp_amd import _lib
    if not os.path.exists(_lib.HIPCC) and not os.path.exists(_lib.LIBPATH):
        pytest.skip('no hipcc and no prebuilt library')
    path = _lib.build() if os.path.exists(_lib.HIPCC) else _lib.LIBPATH
    import torch  # noqa: F401  (loads the HIP runtime the library links against)
    lib = ctypes.CDLL(path)
    declared = _lib.exported_symbols()
    assert len(declared) >= 30
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    lib.nsp_version.restype = ctypes.c_int
    assert lib.nsp_version() >= 100


def test_struct_mirrors_match_header_field_order():
    """ctypes mirrors must list the same fields, in order, as the C structs."""
    import re
    from neural_sp_amd import _lib
    hdr = open(os.path.join(os.path.dirname(_lib.__file__), '..', 'include', 'nsp_hip.h')).read()

    def fields(struct_name):
        body = dict((n, b) for b, n in re.findall(r'typedef struct \{(.*?)\} (\w+);', hdr, re.S))[struct_name]
        body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
        out = []
        for decl in body.split(';'):
            decl = decl.strip()
            if not decl:
                continue
            names = decl.split(',')
            first = names[0].split()[-1].lstrip('*')
            out.append(first)
            out += [n.strip().lstrip('*') for n in names[1:]]
        return [re.sub(r'\[.*?\]$', '', n) for n in out]   # array members: name[N] -> name
    assert fields('nsp_gemm_params') == [f[0] for f in _lib.GemmParams._fields_]
    assert fields('nsp_attn_mask_params') == [f[0] for f in _lib.AttnMaskParams._fields_]
    assert fields('nsp_lstm_stack_params') == [f[0] for f in _lib.LstmStackParams._fields_]
    assert _lib.LSTM_MAX_LAYERS == int(re.search(r'#define NSP_LSTM_MAX_LAYERS (\d+)', hdr).group(1))


def test_cpu_tensor_is_rejected_loudly():
    """There is no CPU fallback: feeding host tensors to an op must raise."""
    import torch
    from neural_sp_amd import ops
    with pytest.raises((AssertionError, RuntimeError)):
        ops.linear(torch.randn(4, 8), torch.randn(3, 8))


def test_smoke_body_runs_on_the_emulated_kernels():
    """__graft_entry__.smoke() -- the check the driver runs on cuda:0 before the bench -- with its device pointed at the
    CPU and the C-ABI calls routed to the host emulator of the kernels: loss and every gradient against the oracle in
    fp32 mode, the loss again in bf16 mode (the smoke's own gates)."""
    import pytest
    from tests.hipemu import build_emu
    if not build_emu.available():
        pytest.skip('no host clang++ for the HIP emulator')
    import __graft_entry__ as entry
    from tests.cpu_ops_shim import host_logic_on_cpu
    with host_logic_on_cpu(real_kernels=True, real_conv=False):
        entry.smoke(device='cpu')
