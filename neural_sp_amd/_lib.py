"""Build + load libnsp_hip.so (the gfx950 kernels) and expose its C ABI via ctypes.

The library is built in-tree (neural_sp_amd/lib/libnsp_hip.so) with
``hipcc --offload-arch=gfx950``; hipcc cross-compiles without a GPU.  There is
no CPU fallback: if the library is missing and cannot be built, importing the
ops fails loudly.
"""
import ctypes
import glob
import hashlib
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
LIBDIR = os.path.join(_HERE, 'lib')
LIBPATH = os.path.join(LIBDIR, 'libnsp_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
ARCH = 'gfx950'
# '-packed-fp32-ops' (device target feature OFF): no v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 / v_pk_mov_b32.  Round 5 traced
# the round-4 stock-DDP failure to them: beside a second process on the same device the -O3 build of the first conv layer
# (348 packed fp32 ops) stored a wrong LOW half of a packed accumulator in lanes 48..63 of some waves -- 1 launch in 10^4
# beside a light neighbour, 1 in 3 beside a process running library GEMMs, never alone; the same source without packed fp32
# ops (or at -O1): 0 in 10^4 (tools/conv_first_kernel_stress.py, profiles/r05_packed_fp32_*.log).  The
# host half of the compilation does not know the feature and says so once per file ("not a recognized feature"): harmless.
CFLAGS = ['-O3', '-std=c++17', '-fPIC', '--offload-arch=' + ARCH, '-munsafe-fp-atomics',
          '-Wno-unused-result', '-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def _src_digest():
    h = hashlib.sha1()
    for f in _sources() + sorted(glob.glob(os.path.join(CSRC, '*.h'))) + \
            [os.path.join(_HERE, '..', 'include', 'nsp_hip.h')]:
        with open(f, 'rb') as fh:
            h.update(fh.read())
    h.update(' '.join(CFLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=False):
    """Compile every csrc/*.hip for gfx950 and link libnsp_hip.so (incremental)."""
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, 'build.stamp')
    digest = _src_digest()
    if not force and os.path.exists(LIBPATH) and os.path.exists(stamp):
        with open(stamp) as fh:
            if fh.read().strip() == digest:
                return LIBPATH
    if not os.path.exists(HIPCC):
        raise RuntimeError('hipcc not found at %s and %s is stale/missing' % (HIPCC, LIBPATH))
    # objects compiled with other flags are stale whatever their timestamps say
    flags_stamp = os.path.join(LIBDIR, 'flags.stamp')
    flags = ' '.join(CFLAGS)
    if not (os.path.exists(flags_stamp) and open(flags_stamp).read() == flags):
        force = True
    objs, procs = [], []
    hdr_mtime = max(os.path.getmtime(f) for f in
                    glob.glob(os.path.join(CSRC, '*.h')) + [os.path.join(_HERE, '..', 'include', 'nsp_hip.h')])
    for src in _sources():
        obj = os.path.join(LIBDIR, os.path.basename(src).replace('.hip', '.o'))
        objs.append(obj)
        if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(src)
                and os.path.getmtime(obj) > hdr_mtime):
            continue
        cmd = [HIPCC] + CFLAGS + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError('hipcc failed on %s:\n%s' % (src, out.decode(errors='replace')))
    cmd = [HIPCC, '-shared', '-fPIC', '--offload-arch=' + ARCH, '-o', LIBPATH] + objs
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if res.returncode != 0:
        raise RuntimeError('link failed:\n%s' % res.stdout.decode(errors='replace'))
    with open(stamp, 'w') as fh:
        fh.write(digest)
    with open(flags_stamp, 'w') as fh:
        fh.write(flags)
    return LIBPATH


class GemmParams(ctypes.Structure):
    """Mirror of nsp_gemm_params (include/nsp_hip.h)."""
    _fields_ = [
        ('M', ctypes.c_int), ('N', ctypes.c_int), ('K', ctypes.c_int),
        ('A', ctypes.c_void_p), ('a_rs', ctypes.c_longlong), ('a_cs', ctypes.c_longlong),
        ('B', ctypes.c_void_p), ('b_ks', ctypes.c_longlong), ('b_ns', ctypes.c_longlong),
        ('C', ctypes.c_void_p), ('ldc', ctypes.c_longlong),
        ('batch1', ctypes.c_int), ('batch2', ctypes.c_int),
        ('a_b1', ctypes.c_longlong), ('a_b2', ctypes.c_longlong),
        ('b_b1', ctypes.c_longlong), ('b_b2', ctypes.c_longlong),
        ('c_b1', ctypes.c_longlong), ('c_b2', ctypes.c_longlong),
        ('bias', ctypes.c_void_p), ('act', ctypes.c_int),
        ('pre_out', ctypes.c_void_p), ('dact_src', ctypes.c_void_p), ('dact', ctypes.c_int),
        ('res', ctypes.c_void_p), ('alpha', ctypes.c_float),
        ('splitk', ctypes.c_int), ('mode', ctypes.c_int),
        ('dropout_p', ctypes.c_float),
        ('seed', ctypes.c_ulonglong), ('offset', ctypes.c_ulonglong),
        ('a_dtype', ctypes.c_int), ('b_dtype', ctypes.c_int), ('c_dtype', ctypes.c_int),
        ('pre_dtype', ctypes.c_int), ('dact_dtype', ctypes.c_int),
        ('c_ss', ctypes.c_longlong),
        ('epi_mode', ctypes.c_int), ('epi_ncols', ctypes.c_int), ('epi_blank', ctypes.c_int),
        ('epi_lab', ctypes.c_void_p),
        ('epi_f0', ctypes.c_void_p), ('epi_f1', ctypes.c_void_p), ('epi_f2', ctypes.c_void_p),
        ('epi_f3', ctypes.c_void_p),
        ('epi_scale_dev', ctypes.c_void_p), ('epi_scale', ctypes.c_float),
    ]


class AttnMaskParams(ctypes.Structure):
    """Mirror of nsp_attn_mask_params."""
    _fields_ = [
        ('B', ctypes.c_int), ('H', ctypes.c_int), ('Tq', ctypes.c_int), ('Tk', ctypes.c_int),
        ('R', ctypes.c_int), ('clamp', ctypes.c_int), ('scale', ctypes.c_float),
        ('klens', ctypes.c_void_p), ('causal', ctypes.c_int), ('lookahead', ctypes.c_int),
        ('chunk_nl', ctypes.c_int), ('chunk_nc', ctypes.c_int),
        ('dropout_p', ctypes.c_float), ('seed', ctypes.c_ulonglong), ('offset', ctypes.c_ulonglong),
        ('p_bf16', ctypes.c_int), ('tk_pitch', ctypes.c_int), ('r_pitch', ctypes.c_int),
    ]


LSTM_MAX_LAYERS = 4


class LstmStackParams(ctypes.Structure):
    """Mirror of nsp_lstm_stack_params."""
    _fields_ = [
        ('nl', ctypes.c_int), ('B', ctypes.c_int), ('L', ctypes.c_int), ('H', ctypes.c_int),
        ('dropout_p', ctypes.c_float), ('reserved', ctypes.c_int),
        ('gi0', ctypes.c_void_p), ('dy_top', ctypes.c_void_p), ('y_top', ctypes.c_void_p),
        ('w', ctypes.c_void_p * LSTM_MAX_LAYERS), ('bias', ctypes.c_void_p * LSTM_MAX_LAYERS),
        ('hp16', ctypes.c_void_p * LSTM_MAX_LAYERS), ('yd16', ctypes.c_void_p * LSTM_MAX_LAYERS),
        ('c_all', ctypes.c_void_p * LSTM_MAX_LAYERS), ('gates', ctypes.c_void_p * LSTM_MAX_LAYERS),
        ('dg16', ctypes.c_void_p * LSTM_MAX_LAYERS), ('dc', ctypes.c_void_p * LSTM_MAX_LAYERS),
        ('seed', ctypes.c_ulonglong * LSTM_MAX_LAYERS), ('offset', ctypes.c_ulonglong * LSTM_MAX_LAYERS),
        ('xchg', ctypes.c_void_p * LSTM_MAX_LAYERS),
    ]


_lib = None


def lib():
    """Return the loaded shared library (builds it first if sources changed)."""
    global _lib
    if _lib is None:
        import torch  # noqa: F401  -- loads the process-wide HIP runtime (libamdhip64.so.7) first
        path = LIBPATH
        if os.environ.get('NSP_LIB_OVERRIDE'):
            # development only (tools/*_ab.py): A/B a previously built library against the tree's in one gpurun call
            path = os.environ['NSP_LIB_OVERRIDE']
        elif os.path.exists(HIPCC):
            path = build()
        elif not os.path.exists(path):
            raise RuntimeError('libnsp_hip.so missing and no hipcc to build it: the HIP path is '
                               'mandatory, there is no CPU fallback')
        _lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        _declare_prototypes(_lib)
    return _lib


_CTYPES = {'int': ctypes.c_int, 'float': ctypes.c_float, 'long long': ctypes.c_longlong,
           'unsigned long long': ctypes.c_ulonglong, 'unsigned int': ctypes.c_uint}


def prototypes():
    """{name: (restype, [argtypes])} parsed from include/nsp_hip.h, so that calls can pass plain
    Python ints / floats / addresses (ctypes converts them in C, no per-argument objects)."""
    import re
    hdr = open(os.path.join(_HERE, '..', 'include', 'nsp_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    out = {}
    for m in re.finditer(r'\b(int|long long)\s+(nsp_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;', hdr, re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != 'void':
            for a in args.split(','):
                a = ' '.join(a.split())
                if '*' in a:
                    argtypes.append(ctypes.c_void_p)
                    continue
                base = a.rsplit(' ', 1)[0] if ' ' in a else a
                base = base.replace('const ', '').strip()
                argtypes.append(_CTYPES[base])
        out[name] = (_CTYPES[ret], argtypes)
    return out


def _declare_prototypes(lib):
    for name, (ret, argtypes) in prototypes().items():
        fn = getattr(lib, name)
        fn.restype = ret
        fn.argtypes = argtypes


def exported_symbols():
    """Symbols declared in include/nsp_hip.h (for the ABI-completeness test)."""
    import re
    hdr = open(os.path.join(_HERE, '..', 'include', 'nsp_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    return sorted(set(re.findall(r'\b(nsp_[a-z0-9_]+)\s*\(', hdr)))
