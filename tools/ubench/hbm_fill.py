"""Microbenchmark: what this GPU sustains for the input-projection GEMM's memory side alone -- 315 MB written, 79 MB read -- with ideal
access patterns (torch elementwise kernels): write-only fill, copy, and a 1:4 read:write mix."""
import time, torch
M, N, K = 76800, 1024, 256
out = torch.empty((M, N), device="cuda"); a = torch.randn((M, K), device="cuda"); b = torch.empty((M, N), device="cuda")
def t(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
ms = t(lambda: out.fill_(1.0)); print(f"fill 315 MB: {ms:.4f} ms = {315 / ms:.0f} GB/s written")
ms = t(lambda: out.copy_(b)); print(f"copy 315 MB -> 315 MB: {ms:.4f} ms = {630 / ms:.0f} GB/s moved")
av = a.view(M, K, 1).expand(M, K, 4).reshape(M, N)
ms = t(lambda: torch.mul(av, 2.0, out=out)); print(f"read 79 MB (each value 4 times), write 315 MB: {ms:.4f} ms = {394 / ms:.0f} GB/s of HBM traffic")
ms = t(lambda: a.mul_(1.0)); print(f"read + write 79 MB in place: {ms:.4f} ms = {157 / ms:.0f} GB/s")
