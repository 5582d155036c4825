"""Build tests/hipemu/_build/libnsp_emu.so: selected csrc/*.hip files compiled by the HOST clang++
against the emulation header (tests/hipemu/include/hip/hip_runtime.h) -- TEST INFRASTRUCTURE ONLY.

All sixteen kernel files build (the header emulates the gfx950 builtins they use; `s_waitcnt` asm is stripped here).
Nothing under neural_sp_amd/ imports this module or loads the library it builds.
"""
import hashlib
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'neural_sp_amd', 'csrc')
# NSP_EMU_ASAN=1: the same build with AddressSanitizer (every global / LDS access of every kernel is bounds-checked
# against the real allocation: torch's CPU tensors, the static __shared__ arrays, the dynamic shared buffer).  Run as
#   NSP_EMU_ASAN=1 LD_PRELOAD=$(python tests/hipemu/build_emu.py --asan-runtime) ASAN_OPTIONS=detect_leaks=0 python -m pytest ...
ASAN = os.environ.get('NSP_EMU_ASAN', '0') == '1'
OUT = os.path.join(HERE, '_build_asan' if ASAN else '_build')
LIB = os.path.join(OUT, 'libnsp_emu.so')
EMULATED_SOURCES = ['norm_subsample.hip', 'elementwise.hip', 'xent.hip', 'decode.hip', 'layernorm.hip', 'ctc.hip',
                    'dwconv.hip', 'rnnt.hip', 'rnnt_fused.hip', 'gemm.hip', 'attention.hip', 'lstm.hip', 'conv2d.hip', 'gemm_bf16.hip', 'flash_attn.hip', 'mocha.hip', 'decoder_step.hip']
CXX_CANDIDATES = ['/opt/rocm/lib/llvm/bin/clang++', 'clang++']


def _cxx():
    for c in CXX_CANDIDATES:
        try:
            subprocess.run([c, '--version'], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            return c
        except (OSError, subprocess.CalledProcessError):
            continue
    return None


def available():
    return _cxx() is not None


_DYN = re.compile(r'extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?((?:unsigned\s+)?[A-Za-z_][\w:]*)\s+(\w+)\[\];')


def _rewrite(src, dst):
    """copy a .hip source for the host build: `extern __shared__ T name[];` -> pointer to the launch's buffer;
    the include of common.h is redirected to the real csrc directory"""
    text = open(src).read()
    text = _DYN.sub(r'\1* \2 = reinterpret_cast<\1*>(HIPEMU_DYN_SHARED);', text)
    text = re.sub(r'asm volatile\("s_waitcnt vmcnt\(%0\)"\s*::\s*"n"\((.*?)\)\s*:\s*"memory"\);', r'hipemu_waitcnt_vm(\1);', text)
    text = re.sub(r'asm volatile\("s_waitcnt vmcnt\((\d+)\)"\s*:::\s*"memory"\);', r'hipemu_waitcnt_vm(\1);', text)
    text = re.sub(r'asm volatile\("s_waitcnt[^;]*;', '/* s_waitcnt (lgkmcnt): no-op on the host */;', text)
    text = text.replace('__attribute__((address_space(1)))', '').replace('__attribute__((address_space(3)))', '')
    text = text.replace('#include "common.h"', '#include "%s"' % os.path.join(CSRC, 'common.h'))
    with open(dst, 'w') as fh:
        fh.write(text)


def build():
    cxx = _cxx()
    if cxx is None:
        raise RuntimeError('no host clang++ (ext_vector_type / __bf16 support is needed) for the HIP emulator')
    os.makedirs(OUT, exist_ok=True)
    # one builder at a time (pytest-xdist workers all arrive here after a source change)
    import fcntl
    with open(os.path.join(OUT, 'build.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(cxx)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(cxx):
    srcs = [os.path.join(CSRC, s) for s in EMULATED_SOURCES] + [os.path.join(HERE, 'emu_runtime.cpp')]
    deps = srcs + [os.path.join(HERE, 'include', 'hip', 'hip_runtime.h'), os.path.join(CSRC, 'common.h'),
                   os.path.join(ROOT, 'include', 'nsp_hip.h')]
    h = hashlib.sha1()
    for f in deps:
        with open(f, 'rb') as fh:
            h.update(fh.read())
    stamp = os.path.join(OUT, 'stamp')
    if os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == h.hexdigest():
        return LIB
    gen = []
    for src in srcs[:-1]:
        dst = os.path.join(OUT, os.path.basename(src).replace('.hip', '_emu.cpp'))
        _rewrite(src, dst)
        gen.append(dst)
    cmd = [cxx, '-O1', '-std=c++17', '-fPIC', '-shared', '-pthread', '-Wno-unused-value', '-Wno-unused-result'] + (
        ['-g', '-fsanitize=address', '-shared-libasan', '-fno-omit-frame-pointer', '-DHIPEMU_UCONTEXT=1'] if ASAN else []) + [
           '-Wl,-Bsymbolic',   # libnsp_hip.so (RTLD_GLOBAL) may already be loaded: bind our own nsp_* references locally
           '-I', os.path.join(HERE, 'include'), '-x', 'c++'] + gen + [srcs[-1]] + ['-o', LIB]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if res.returncode != 0:
        raise RuntimeError('emulator build failed:\n' + res.stdout.decode(errors='replace'))
    with open(stamp, 'w') as fh:
        fh.write(h.hexdigest())
    return LIB


def asan_runtime():
    """path of the shared ASan runtime of the host clang (to LD_PRELOAD into python)"""
    out = subprocess.run([_cxx(), '-print-file-name=libclang_rt.asan-x86_64.so'], stdout=subprocess.PIPE).stdout.decode().strip()
    if not os.path.isabs(out) or not os.path.exists(out):      # (clang prints the bare name when it does not find the file)
        import glob
        hits = sorted(glob.glob('/opt/rocm*/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so'))
        out = hits[-1] if hits else out
    return out


if __name__ == '__main__':
    import sys
    print(asan_runtime() if '--asan-runtime' in sys.argv else build())
