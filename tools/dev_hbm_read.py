"""Diagnostic: what a plain streaming read of the BCD pass's 617 MB reaches on this part (torch reduction kernels)."""
import torch, time
x = torch.randn(617_349_120 // 4, device="cuda")
for fn, name in ((lambda: x.sum(), "sum (read only)"), (lambda: x.abs().max(), "abs+max (read+write+read)"), (lambda: torch.max(x), "max (read only)")):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name}: {ms:.4f} ms -> {x.numel()*4/ms/1e9:.2f} TB/s of input")
