#!/usr/bin/env python3
"""H2D rate of this box: a pinned torch tensor, and the same bytes through hipHostRegister'ed numpy memory (rpvg_hip_host_register)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n = 280 * 1000 * 1000
src = torch.empty(n, dtype=torch.uint8).pin_memory()
dst = torch.empty(n, dtype=torch.uint8, device="cuda")
for label, chunks in (("one copy", 1), ("8 copies", 8)):
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        step = n // chunks
        for c in range(chunks):
            dst[c * step:(c + 1) * step].copy_(src[c * step:(c + 1) * step], non_blocking=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"torch pinned, {label}: {n / dt / 1e9:.1f} GB/s ({dt * 1e3:.2f} ms)")
from rpvg_amd import hip
arr = np.zeros(n // 8, dtype=np.float64)
hip.host_register(arr)
t = torch.from_numpy(arr)
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dst.view(torch.float64).copy_(t, non_blocking=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"hipHostRegister'ed numpy through torch copy_: {n / dt / 1e9:.1f} GB/s ({dt * 1e3:.2f} ms)")
hip.host_unregister(arr)
