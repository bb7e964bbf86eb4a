"""What the vendor GEMM libraries reach on this box (torch.matmul -> hipBLASLt / rocBLAS), as an independent reference for
the MFMA rates the K2 kernels are compared with: square 8192^3 GEMMs (the libraries' best case) and the search's own shape
(1024 queries x 768 x a 262144-row corpus slice, output written).  f16, bf16, int8 (torch._int_mm), fp8 (torch._scaled_mm)."""
import time, torch
dev = "cuda:0"
def bench(fn, flops, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / iters
    return t * 1e3, flops / t / 1e12
for (M, N, K) in ((8192, 8192, 8192), (1024, 262144, 768), (1024, 262144, 1024), (256, 262144, 768)):
    fl = 2.0 * M * N * K
    out = []
    for dt in (torch.float16, torch.bfloat16):
        a = torch.randn(M, K, device=dev, dtype=dt); b = torch.randn(N, K, device=dev, dtype=dt)
        ms, tf = bench(lambda: torch.matmul(a, b.t()), fl)
        out.append(f"{str(dt)[6:]} {ms:7.3f} ms {tf:7.1f} TF")
    try:
        a = torch.randint(-127, 127, (M, K), device=dev, dtype=torch.int8); b = torch.randint(-127, 127, (N, K), device=dev, dtype=torch.int8)
        ms, tf = bench(lambda: torch._int_mm(a, b.t()), fl)
        out.append(f"int8 {ms:7.3f} ms {tf:7.1f} TOP")
    except Exception as e:
        out.append(f"int8 failed: {str(e)[:60]}")
    try:
        f8 = torch.float8_e4m3fn
        a = torch.randn(M, K, device=dev).to(f8); b = torch.randn(N, K, device=dev).to(f8)
        one = torch.tensor(1.0, device=dev)
        ms, tf = bench(lambda: torch._scaled_mm(a, b.t(), scale_a=one, scale_b=one, out_dtype=torch.bfloat16), fl)
        out.append(f"fp8 {ms:7.3f} ms {tf:7.1f} TF")
    except Exception as e:
        out.append(f"fp8 failed: {str(e)[:60]}")
    print(f"M={M} N={N} K={K}: " + " | ".join(out), flush=True)
