#!/usr/bin/env python
"""dev probe: what do the vendor GEMM libraries (through torch) reach at the model's shapes?"""
import torch, time
M = 8064
shapes = [("ffn_up", 2048, 512), ("ffn_down", 512, 2048), ("qkv", 1536, 512), ("out", 512, 512), ("glu", 1024, 512)]
for Mx in (8064, 32256):
    for name, N, K in shapes:
        a = torch.randn(Mx, K, device="cuda", dtype=torch.float16)
        w = torch.randn(N, K, device="cuda", dtype=torch.float16) * 0.05
        b = torch.randn(N, device="cuda", dtype=torch.float16)
        for _ in range(5):
            y = torch.nn.functional.linear(a, w, b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            y = torch.nn.functional.linear(a, w, b)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        print(f"M={Mx} {name:9s} N={N} K={K}: {us:7.2f} us  {2*Mx*N*K/us/1e6:7.1f} TF/s")
