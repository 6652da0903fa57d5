#!/usr/bin/env python3
"""Floor of a dependent kernel launch on this box: N tiny kernels back to back on one stream, with and without a kernel in
front of each that leaves ~400 MB of freshly written lines in the L2s (what a front kernel leaves for se_kernel)."""
import torch
def t(fn, n=200):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
tiny = torch.zeros(256, device='cuda')
big = torch.empty(200 * 1024 * 1024, dtype=torch.bfloat16, device='cuda')
print(f'tiny kernel, back to back          : {t(lambda: tiny.add_(1.0)):7.2f} us per launch')
fill = t(lambda: big.fill_(1.0))
both = t(lambda: (big.fill_(1.0), tiny.add_(1.0)))
print(f'400 MB fill                        : {fill:7.2f} us')
print(f'400 MB fill + dependent tiny kernel: {both:7.2f} us  (tiny kernel costs {both - fill:5.2f} us behind it)')
