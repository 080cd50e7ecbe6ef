"""NCC_vxm forward + gradient on its own at the train-step shapes (HIP events around the C-ABI call).
    python tools/bench_ncc.py            # 160x192x160 B=1 and 160x192x224 B=2
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import ops  # noqa: E402


def main():
    for B, shape in ((1, (160, 192, 160)), (2, (160, 192, 224))):
        g = torch.Generator(device="cuda").manual_seed(3)
        a = torch.rand((B, 1) + shape, device="cuda", generator=g)
        b = torch.rand((B, 1) + shape, device="cuda", generator=g)
        for win in (9, 5):
            for _ in range(3):
                ops._ncc_launch(a, b, True, win)
            ts = []
            for _ in range(20):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); ops._ncc_launch(a, b, True, win); e1.record(); e1.synchronize()
                ts.append(e0.elapsed_time(e1))
            ts.sort()
            n = a.numel()
            print("B=%d %s win %d: %.1f us  (%.0f GB/s of the 12 B/voxel I, J, d_J)" % (B, "x".join(map(str, shape)), win,
                                                                                      ts[len(ts) // 2] * 1e3, 12.0 * n / ts[len(ts) // 2] / 1e6))


if __name__ == "__main__":
    main()
