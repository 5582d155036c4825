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


# ---- second screen: a VALU write of an SGPR (v_readlane restoring a spilled scalar, v_readfirstlane, a carry-out ...) needs
# five wait states before a VMEM instruction reads that SGPR as its base / offset / descriptor.  hipcc inserts them for the
# instructions it generates; it cannot for a VMEM instruction inside inline asm.  The stream-K twins of the 8-phase GEMM
# exchange partial accumulators through asm loads / stores with a scalar base: without `s_nop 4` at the head of those asm
# blocks the build has 126 such pairs (the restore of a spilled base one to four states in front of the access) and took
# memory access faults on the full grid.
_SREG = re.compile(r'\bs(\d+)\b|\bs\[(\d+):(\d+)\]')
_TWO_DST = ('v_add_co', 'v_sub_co', 'v_subrev_co', 'v_addc_co', 'v_subb_co', 'v_subbrev_co', 'v_mad_u64_u32', 'v_mad_i64_i32', 'v_div_scale')
_VMEM = ('global_', 'buffer_', 'tbuffer_', 'flat_', 'scratch_')


def _sregs(text):
    out = set()
    for m in _SREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def valu_sgpr_to_vmem_pairs(disassembly):
    """(writer, reader, wait states between) for every VALU write of an SGPR followed by a VMEM read of it after fewer than 5 states"""
    hist, found = [], []
    for line in disassembly.splitlines():
        if '//' not in line or not line.startswith('\t'):
            if line.endswith(':'):
                hist = []                                   # a label: a new kernel or block
            continue
        body = line.split('//')[0].strip()
        if not body:
            continue
        mn, _, ops = body.partition(' ')
        if mn.startswith(_VMEM):
            rd = _sregs(ops)
            found += [(txt, body, states) for states, w, txt in hist if states < 5 and (rd & w)]
        cost = 1
        if mn == 's_nop':
            cost = int(ops.strip(), 0) + 1
        hist = [(st + cost, w, t) for st, w, t in hist if st + cost < 6]
        if mn.startswith('v_'):
            parts = [o.strip() for o in ops.split(',')]
            w = set()
            for dst in (parts[:2] if mn.startswith(_TWO_DST) else parts[:1]):
                w |= _sregs(dst)
            if w:
                hist.append((0, w, body))
    return found


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason='llvm-objdump of the ROCm toolchain not found')
def test_no_vmem_read_of_an_sgpr_within_five_states_of_a_valu_write():
    objs = _objects()
    if not objs:
        pytest.skip('no in-tree objects (run __graft_entry__.build() first)')
    offenders = []
    with tempfile.TemporaryDirectory() as tmp:
        for obj in objs:
            local = os.path.join(tmp, os.path.basename(obj))
            shutil.copy(obj, local)
            subprocess.run([OBJDUMP, '--offloading', local], check=True, capture_output=True, cwd=tmp)
            devs = glob.glob(local + '.*gfx950*')
            assert devs, 'no gfx950 bundle in %s' % obj
            dis = subprocess.run([OBJDUMP, '-d', devs[0]], check=True, capture_output=True, text=True).stdout
            offenders += [(os.path.basename(obj),) + o for o in valu_sgpr_to_vmem_pairs(dis)]
    assert not offenders, 'VALU write of an SGPR too close in front of a VMEM read of it:\n' + '\n'.join('%s: %s -> %s (%d states)' % o for o in offenders[:10])


def test_the_second_screen_recognises_the_pair():
    bad = '\tv_readlane_b32 s17, v223, 8    // 0: 0\n\tglobal_load_dwordx4 v[70:73], v0, s[16:17] nt    // 8: 0\n'
    ok = '\tv_readlane_b32 s17, v223, 8    // 0: 0\n\ts_nop 4    // 4: 0\n\tglobal_load_dwordx4 v[70:73], v0, s[16:17] nt    // 8: 0\n'
    other = '\tv_readlane_b32 s20, v223, 8    // 0: 0\n\tglobal_load_dwordx4 v[70:73], v0, s[16:17] nt    // 8: 0\n'
    assert len(valu_sgpr_to_vmem_pairs(bad)) == 1 and not valu_sgpr_to_vmem_pairs(ok) and not valu_sgpr_to_vmem_pairs(other)


def test_the_pattern_recognises_the_hazardous_form():
    assert WIDE_STORE_REG_SOFFSET.search('buffer_store_dwordx4 v[136:139], v1, s[20:23], s0 offen nt')
    assert WIDE_STORE_REG_SOFFSET.search('buffer_store_dwordx3 v[1:3], off, s[4:7], s12')
    assert not WIDE_STORE_REG_SOFFSET.search('buffer_store_dwordx4 v[136:139], v1, s[20:23], 0 offen nt')
    assert not WIDE_STORE_REG_SOFFSET.search('buffer_store_dwordx2 v[52:53], v138, s[24:27], s92 offen nt')       # 64 bits: no hazard
    assert not WIDE_STORE_REG_SOFFSET.search('buffer_load_dwordx4 v[6:9], v10, s[4:7], s3 offen nt')
