#!/usr/bin/env python3
"""micro: device-to-device copy / fill / read bandwidth on this GPU (calibrates the HBM roof for mixed traffic)"""
import torch, time
n = 4_000_000_000  # 32 GB of int64
a = torch.empty(n, dtype=torch.int64, device="cuda")
b = torch.empty(n, dtype=torch.int64, device="cuda")
a.fill_(1); torch.cuda.synchronize()
def t(f, reps=3):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
dt = t(lambda: b.copy_(a)); print("copy  32 GB -> 32 GB: %.2f ms, %.2f TB/s (read+write)" % (dt * 1e3, 2 * n * 8 / dt / 1e12))
dt = t(lambda: a.fill_(3)); print("fill  32 GB: %.2f ms, %.2f TB/s" % (dt * 1e3, n * 8 / dt / 1e12))
dt = t(lambda: a.sum()); print("read  32 GB (sum): %.2f ms, %.2f TB/s" % (dt * 1e3, n * 8 / dt / 1e12))
