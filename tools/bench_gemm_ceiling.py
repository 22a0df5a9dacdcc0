"""What a dense bf16 GEMM reaches on this chip with the vendor library (hipBLASLt through torch.matmul), random vs all-zero operands:
the practical matrix-pipe ceiling under the clock the power governor holds (the roofline in bench.py stays priced at the 2.5 PFLOP/s
spec peak; this number says how much of the gap is the chip, not the kernels)."""
import torch, time
dev = torch.device("cuda:0")
for n in (4096, 8192, 16384):
    for zero in (False, True):
        a = torch.zeros(n, n, device=dev, dtype=torch.bfloat16) if zero else torch.randn(n, n, device=dev).bfloat16()
        b = torch.zeros(n, n, device=dev, dtype=torch.bfloat16) if zero else torch.randn(n, n, device=dev).bfloat16()
        for _ in range(3):
            torch.matmul(a, b)
        torch.cuda.synchronize()
        iters = max(3, int(2e14 / (2 * n ** 3)))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            torch.matmul(a, b)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        print(f"n={n:6d} {'zeros ' if zero else 'random'}: {ms:8.3f} ms  {2 * n ** 3 / ms / 1e9:7.1f} TFLOP/s", flush=True)
