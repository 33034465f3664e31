#!/usr/bin/env python3
"""Developer aid (GPU box): how long the first tiny kernel takes after the GPU has been idle for a while — the mapper replay's slow
pose refinements (15-27 ms for two LM steps, always the first GPU call after ~10 ms of host-only work) in isolation."""
import time
import torch
x = torch.zeros(1024, device="cuda")
torch.cuda.synchronize()
for idle_ms in (0, 1, 2, 5, 10, 20, 50, 100):
    ts = []
    for _ in range(5):
        (x + 1).sum().item()
        time.sleep(idle_ms * 1e-3)
        t0 = time.perf_counter()
        (x + 1).sum().item()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"idle {idle_ms:4d} ms -> first tiny op + sync: median {sorted(ts)[2]:.3f} ms, max {max(ts):.3f} ms")
