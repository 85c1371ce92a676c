"""What the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS, bf16 in, bf16 out, no epilogue) takes on the conv GEMM shapes of the step:
a yardstick for the hand-written kernels' K-loops, not a product path.  python tools/blas_ref.py"""
import torch
for (M, N, K) in ((76288, 512, 512), (76288, 1536, 1536), (76288, 128, 1536), (76288, 512, 1536)):
    x = torch.randn(M, K, device='cuda').to(torch.bfloat16)
    w = torch.randn(N, K, device='cuda').to(torch.bfloat16)
    for name, f in (('x @ w.T', lambda: torch.matmul(x, w.t())),):
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print(f'M={M} N={N} K={K} {name}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s', flush=True)
