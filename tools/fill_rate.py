#!/usr/bin/env python3
"""Pure-write, pure-read and copy rates of this box (torch fill_ / sum / copy_ on buffers of 128 MB and 1.28 GB: the sizes the leaf
kernel writes at 1M and 10M triangles).  python3 tools/fill_rate.py"""
import torch
for mb in (128, 1280):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, dtype=torch.float32, device="cuda")
    b = torch.empty(n, dtype=torch.float32, device="cuda")
    def t(f, reps=20):
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps):
            f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3
    tf = t(lambda: a.fill_(1.0))
    tr = t(lambda: a.sum())
    tc = t(lambda: b.copy_(a))
    nb = n * 4
    print(f"{nb / 1e6:7.1f} MB: fill {nb / tf / 1e12:5.2f} TB/s ({tf * 1e6:6.1f} us)   read (sum) {nb / tr / 1e12:5.2f} TB/s ({tr * 1e6:6.1f} us)   "
          f"copy {2 * nb / tc / 1e12:5.2f} TB/s of read + write ({tc * 1e6:6.1f} us)")
