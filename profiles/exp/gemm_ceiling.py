#!/usr/bin/env python3
"""What the vendor library (hipBLASLt through torch.matmul, bf16) reaches on the late 1x1-conv GEMM shapes at 256 crops --
a ceiling reference for profiles/r02_pmc_summary.txt's MfmaUtil of pw_gemm_dma_kernel on the same shapes (ours carry the BN
epilogue, SiLU or the SE gate and the residual on top).  Prints us, TFLOP/s and the fraction of the 2.5 PFLOP/s dense bf16 peak."""
import torch
SHAPES = [('b9-12 project', 65536, 576, 96), ('b13 project', 65536, 576, 136), ('b14-17 project', 65536, 816, 136),
          ('b18 expand', 65536, 136, 816), ('b18 project', 16384, 816, 232), ('b19-23 project', 16384, 1392, 232),
          ('b24 project', 16384, 1392, 384), ('b25 project', 16384, 2304, 384), ('head', 16384, 384, 1536),
          ('b19-23 expand (fused in ours)', 16384, 232, 1392), ('b25 expand (fused in ours)', 16384, 384, 2304)]
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
print(f'{"layer":32s} {"M":>6s} {"K":>5s} {"N":>5s} {"us":>8s} {"TFLOP/s":>8s} {"of peak":>8s}')
for name, M, K, N in SHAPES:
    A = torch.randn(M, K, device='cuda', dtype=torch.bfloat16); W = torch.randn(N, K, device='cuda', dtype=torch.bfloat16)
    out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    us = t(lambda: torch.matmul(A, W.t(), out=out))
    tf = 2.0 * M * K * N / us / 1e6
    print(f'{name:32s} {M:6d} {K:5d} {N:5d} {us:8.1f} {tf:8.1f} {tf / 2500:8.3f}')
