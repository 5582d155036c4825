"""Static screen of the built gfx950 code for a hazard hipcc does not guard (round 6).

On gfx950 a buffer store of more than 64 bits still reads its data registers in the cycle after issue.  hipcc's hazard
recogniser inserts the wait state only when the store's scalar-offset field is NOT a register (LLVM's rule for the
pre-gfx10 hazard), so `buffer_store_dwordx4 v[a:b], v_off, s[desc], s_off offen` followed directly by a VALU write of
v[a] stores the new value -- seen as wrong 4 x 4 blocks of a GEMM output in builds whose register allocation happened to
put the two instructions back to back.  The library therefore never puts a register into soffset of a wide buffer store
(the row offset is added to the lane offset instead); this test disassembles every object of the in-tree build and
fails if one appears again.  No GPU needed: llvm-objdump reads the offload bundle."""
import glob
import os
import re
import shutil
import subprocess
import tempfile

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
LIBDIR = os.path.join(HERE, '..', 'neural_sp_amd', 'lib')
OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
# buffer_store_dwordx3 / x4 (and the typed 128-bit forms) whose soffset operand is a register
WIDE_STORE_REG_SOFFSET = re.compile(
    r'\b(?:buffer_store_dwordx[34]|tbuffer_store_format_xyzw?)\s+v\[\d+:\d+\],\s*(?:v\d+|v\[\d+:\d+\]|off),\s*s\[\d+:\d+\],\s*(s\d+|m0|ttmp\d+)\b')


def _objects():
    return sorted(glob.glob(os.path.join(LIBDIR, '*.o')))


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason='llvm-objdump of the ROCm toolchain not found')
def test_no_wide_buffer_store_with_a_register_soffset():
    objs = _objects()
    if not objs:
        pytest.skip('no in-tree objects (run __graft_entry__.build() first)')
    offenders, stores = [], 0
    with tempfile.TemporaryDirectory() as tmp:
        for obj in objs:
            local = os.path.join(tmp, os.path.basename(obj))
            shutil.copy(obj, local)                       # --offloading extracts the bundles next to its input
            subprocess.run([OBJDUMP, '--offloading', local], check=True, capture_output=True, cwd=tmp)
            devs = glob.glob(local + '.*gfx950*')
            assert devs, 'no gfx950 bundle in %s' % obj
            dis = subprocess.run([OBJDUMP, '-d', devs[0]], check=True, capture_output=True, text=True).stdout
            for line in dis.splitlines():
                if 'buffer_store_dwordx' in line or 'tbuffer_store' in line:
                    stores += 1
                    if WIDE_STORE_REG_SOFFSET.search(line):
                        offenders.append((os.path.basename(obj), line.strip()[:110]))
    assert stores > 100, 'the screen saw only %d buffer stores: the disassembly is not what it expects' % stores
    assert not offenders, 'wide buffer stores with a register soffset (gfx950 store-data hazard):\n' + '\n'.join('%s: %s' % o for o in offenders[:10])


def test_the_pattern_recognises_the_hazardous_form():
    assert WIDE_STORE_REG_SOFFSET.search('buffer_store_dwordx4 v[136:139], v1, s[20:23], s0 offen nt')
    assert WIDE_STORE_REG_SOFFSET.search('buffer_store_dwordx3 v[1:3], off, s[4:7], s12')
    assert not WIDE_STORE_REG_SOFFSET.search('buffer_store_dwordx4 v[136:139], v1, s[20:23], 0 offen nt')
    assert not WIDE_STORE_REG_SOFFSET.search('buffer_store_dwordx2 v[52:53], v138, s[24:27], s92 offen nt')       # 64 bits: no hazard
    assert not WIDE_STORE_REG_SOFFSET.search('buffer_load_dwordx4 v[6:9], v10, s[4:7], s3 offen nt')
