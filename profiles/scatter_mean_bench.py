import torch, sys, time
sys.path.insert(0, '.')
from jamun_amd import native
dev = torch.device('cuda', 0)
for walkers, n, deg in [(256, 17, 17), (2048, 17, 17), (2048, 33, 32), (8192, 17, 17)]:
    N = walkers * n
    E = N * deg
    src = torch.randn(E, 248, device=dev)
    seg = torch.arange(0, N + 1, device=dev, dtype=torch.int32) * deg
    for _ in range(3): out = native.scatter_mean(src, seg, N)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps): out = native.scatter_mean(src, seg, N)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    byt = E * 248 * 4 + N * 248 * 4 + (N + 1) * 4
    ref = src.view(N, deg, 248).sum(1) / deg
    print(f"walkers {walkers} n {n} deg {deg}: E={E} bytes={byt/1e6:.1f} MB  {ms*1e3:.1f} us  {byt/ms/1e6:.0f} GB/s  = {byt/ms/1e6/8000:.1%} of 8 TB/s  maxerr {(out-ref).abs().max().item():.2e}")
