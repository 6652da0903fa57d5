#!/usr/bin/env python3
"""Calibration: what a plain streaming copy / read / write reaches on this box (torch elementwise kernels), to put the
streaming kernels' 3.3-4.3 TB/s into perspective."""
import torch
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3
for mb in (64, 335, 1024):
    n = mb * 1024 * 1024 // 2
    x = torch.empty(n, dtype=torch.bfloat16, device='cuda').normal_()
    y = torch.empty_like(x)
    dt = t(lambda: y.copy_(x));           print(f'{mb:5d} MB copy      : {2 * mb / 1024 / dt / 1e3 * 1.073741824:.2f} TB/s (read+write)')
    dt = t(lambda: y.fill_(1.0));         print(f'{mb:5d} MB fill      : {mb / 1024 / dt / 1e3 * 1.073741824:.2f} TB/s (write)')
    dt = t(lambda: x.float().sum()) if mb <= 335 else None
    dt = t(lambda: torch.add(x, x, out=y)); print(f'{mb:5d} MB y = x + x : {2 * mb / 1024 / dt / 1e3 * 1.073741824:.2f} TB/s (read+write)')
    xs = x.view(torch.int16)
    dt = t(lambda: xs.sum());             print(f'{mb:5d} MB sum (read): {mb / 1024 / dt / 1e3 * 1.073741824:.2f} TB/s')
