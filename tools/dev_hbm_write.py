import torch
x = torch.empty(604_000_000 // 4, device="cuda")
y = torch.empty(302_000_000, dtype=torch.uint8, device="cuda")
def t(fn, name, nbytes):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name}: {ms:.4f} ms -> {nbytes/ms/1e9:.2f} TB/s")
t(lambda: x.zero_(), "write 604 MB (zero_)", 604e6)
t(lambda: x.copy_(y[:x.numel()].to(torch.float32)) if False else x.fill_(1.5), "write 604 MB (fill_)", 604e6)
z = torch.empty_like(x)
t(lambda: z.copy_(x), "copy 604 MB -> 604 MB", 1208e6)
