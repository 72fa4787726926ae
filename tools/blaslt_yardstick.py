"""Yardstick only (not on the product path): hipBLASLt bf16 TN GEMM rate via torch.matmul on the
shapes the benchmark workload launches, to compare with the in-tree kernel's per-launch log."""
import os, torch, time
ZERO = bool(os.environ.get("ZERO"))
shapes = [(65536, 3072, 1024), (65536, 1024, 1024), (65536, 4096, 1024), (65536, 1024, 4096), (32768, 8192, 1024), (65536, 1024, 1536), (8192, 8192, 8192)]
if os.environ.get("SHAPES"):      # SHAPES="M,N,K M,N,K ...": e.g. the launch shapes of the narrow workloads (XLM-R, TinyLlama) or of a 4 096-row shard
    shapes = [tuple(int(x) for x in s.split(",")) for s in os.environ["SHAPES"].split()]
for M, N, K in shapes:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    if ZERO:
        a.zero_(); w.zero_()
    for _ in range(3): torch.matmul(a, w.t())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): torch.matmul(a, w.t())
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"M={M} N={N} K={K}: {ms:.3f} ms  {2*M*N*K/ms/1e9:.0f} TFLOP/s", flush=True)
