import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu)')
    from tests import poison
    if poison.enable_from_env():
        print('NSP_POISON: torch.empty / empty_like / new_empty return NaN-filled memory')


# Device suite order under `pytest -x`: kernels first, then the reference fixtures, then whole-model / drop-in tests, the
# multi-process tests (two ranks on ONE device, a second tenant) LAST -- a failure in the most environment-sensitive file
# must not hide the kernel and parity results (round 4: the suite stopped in the 2nd of 11 files, 274 tests never ran).
GPU_FILE_ORDER = ['test_kernels_basic_gpu.py', 'test_kernels_conv_loss_gpu.py', 'test_flash_attn_gpu.py', 'test_golden_gpu.py',
                  'test_fullsize_parity_gpu.py', 'test_variants_gpu.py', 'test_dropin_gpu.py', 'test_alignment_e2e_gpu.py',
                  'test_memory_hygiene_gpu.py', 'test_hostile_neighbour_gpu.py', 'test_ddp_hip_gpu.py']


def pytest_collection_modifyitems(config, items):
    rank = {f: i for i, f in enumerate(GPU_FILE_ORDER)}
    items.sort(key=lambda it: rank.get(os.path.basename(str(it.fspath)), -1))      # (stable: CPU files keep their order, in front)
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
