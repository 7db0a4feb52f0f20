#!/usr/bin/env python3
"""What the vendor GEMM library sustains on this chip (random bf16 data, fp32 accumulate): context for the fused MLP's
fraction of the 2.5 PF nominal peak.  Large square GEMMs, and the MLP's own shape (points x 256 x 256) as 8 chained
library GEMMs with their activations going through HBM, which is what the fused kernel replaces."""
import json, sys, time
import torch
dev = torch.device("cuda:0")
def bench(fn, flop, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return ms, flop / ms / 1e9
torch.manual_seed(0)
for n in (4096, 8192, 16384):
    a = torch.randn(n, n, device=dev, dtype=torch.bfloat16); b = torch.randn(n, n, device=dev, dtype=torch.bfloat16)
    ms, tf = bench(lambda: a @ b, 2 * n ** 3)
    print(json.dumps({"gemm": f"{n}^3 bf16 random", "ms": ms, "tflops": tf, "frac_of_2500": tf / 2500}))
    z = torch.zeros_like(a)
    ms, tf = bench(lambda: z @ z, 2 * n ** 3)
    print(json.dumps({"gemm": f"{n}^3 bf16 zeros", "ms": ms, "tflops": tf, "frac_of_2500": tf / 2500}))
M = 1 << 22
x = torch.randn(M, 256, device=dev, dtype=torch.bfloat16)
ws = [torch.randn(256, 256, device=dev, dtype=torch.bfloat16) * 0.06 for _ in range(8)]
def chain():
    h = x
    for w in ws:
        h = torch.relu_(h @ w)
    return h
ms, tf = bench(chain, 8 * 2 * M * 256 * 256, reps=5)
print(json.dumps({"gemm": "8 chained [4M,256]x[256,256] bf16 + relu (library, activations via HBM)", "ms": ms, "tflops": tf, "frac_of_2500": tf / 2500}))
