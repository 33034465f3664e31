"""Developer tool: thousands of LBA-sized one-shot solves (create / run / download / destroy) in one process, as the mapper issues
them once per registered frame: device memory and time per call must stay flat (device blocks, streams, pinned buffers and the
Cholesky set-up are recycled / rebuilt per call).  Needs a GPU:  python tools/soak_lba.py [n_calls]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from tests import helpers as H
from xrsfm_amd import capi


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    probs = [H.make(7, 1200 + 100 * k, 4, seed=50 + k) for k in range(5)]
    probs.append(H.make(9, 1500, 6, seed=60, dropout=0.3))
    opt = capi.default_options(max_iterations=5, function_tolerance=1e-4, parameter_tolerance=1e-5)
    capi.solve(H.to_product(probs[0]), opt)
    free0, total = torch.cuda.mem_get_info()
    times = []
    ref_cost = {}
    for i in range(n):
        arr = probs[i % len(probs)]
        p = H.to_product(arr)
        t0 = time.perf_counter()
        s = capi.solve(p, opt)
        times.append(time.perf_counter() - t0)
        key = i % len(probs)
        if key in ref_cost:
            assert s.final_cost == ref_cost[key], "results must be bit-reproducible call after call"
        ref_cost[key] = s.final_cost
        if (i + 1) % 500 == 0:
            free, _ = torch.cuda.mem_get_info()
            print(f"{i + 1} calls: {1e3 * np.mean(times[-500:]):.3f} ms/call (median {1e3 * np.median(times[-500:]):.3f}), "
                  f"device memory in use +{(free0 - free) / 2**20:.1f} MiB", flush=True)
    free, _ = torch.cuda.mem_get_info()
    assert free0 - free < 64 * 2**20, "device memory grows with the number of calls"
    print("ok")


if __name__ == "__main__":
    main()
