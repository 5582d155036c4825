"""A second tenant of cuda:0 -- TEST INFRASTRUCTURE (tests/test_hostile_neighbour_gpu.py): a process that runs no kernel of
this repository, only library GEMMs, a softmax and an elementwise chain in a loop, until its time is up or its stdin closes.
Beside it the round-4 build of the first conv layer was wrong in one launch out of three (lanes 48..63 of some waves stored
a wrong low half of a packed fp32 accumulator; never when the process had the device to itself).

  python tests/gpu_neighbour.py SECONDS      # prints READY once the loop is running
"""
import sys
import time

import torch


def main(seconds, kind='all'):
    """kind: 'all' (the tests), or one of 'gemm' / 'elementwise' / 'softmax' (tools/r05_neighbour_kinds.sh: which tenant
    triggers the packed-fp32 failure)"""
    dev = torch.device('cuda', 0)
    a = torch.randn(2048, 2048, device=dev, dtype=torch.bfloat16)
    b = torch.randn(2048, 2048, device=dev, dtype=torch.bfloat16)
    c = torch.randn(4 << 20, device=dev)
    t0, said = time.time(), False
    while time.time() - t0 < seconds:
        for _ in range(20):
            if kind in ('all', 'gemm'):
                d = a @ b
            if kind in ('all', 'elementwise'):
                e = torch.relu(c * 1.0001 + 0.5)
            if kind in ('all', 'softmax'):
                f = torch.softmax((d if kind == 'all' else a).float(), dim=-1)
        torch.cuda.synchronize()
        if not said:
            print('READY', flush=True)
            said = True


if __name__ == '__main__':
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 30.0, sys.argv[2] if len(sys.argv) > 2 else 'all')
