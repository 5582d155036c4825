"""SURVEY 8 row f3 on the CPU tier: this package's forced-alignment CLI (bin/ctc_forced_align.py: main()) executed with
injected parser / loader on the host-emulated kernels, its files compared byte for byte with what the REFERENCE's CLI wrote
(tests/golden/ctc_align_files.pt, produced by oracle/gen_align_files.py from the reference's main(), model, aligner and
Idx2char); and, when /root/reference is present, the reference re-run live against that fixture."""
import pytest

from tests import alignment_common as ac


def test_cli_main_writes_the_reference_files(tmp_path):
    from tests.cpu_ops_shim import host_logic_on_cpu
    ref, _ = ac.load()
    # the REAL kernels on the host emulator (encoder, CTC head, nsp_ctc_forced_align); only the conv front-end (slow to
    # emulate) and the pinned H2D staging are torch stand-ins
    with host_logic_on_cpu(real_kernels=True, real_conv=False):
        got, _ = ac.run_cli(tmp_path, device='cpu')
    assert sorted(got) == sorted(ref['files'])
    for k in ref['files']:
        assert got[k] == ref['files'][k], (k, got[k], ref['files'][k])


def test_reference_cli_live_matches_the_fixture():
    from oracle import ref_import
    if not ref_import.available():
        pytest.skip('reference not present (GPU box)')
    from oracle.gen_align_files import reference_files
    live = reference_files()
    ref, _ = ac.load()
    assert live['files'] == ref['files'] and live['dict'] == ref['dict']
