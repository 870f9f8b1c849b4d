import torch, time
x = torch.randn(700000, 768, device="cuda", dtype=torch.float16)
y = torch.empty_like(x)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
gb = x.numel() * 2 / 1e9
t = timeit(lambda: x.view(torch.int32).sum()); print(f"int32 sum read: {t*1e6:.1f} us  {gb/t/1e3:.2f} TB/s")
t = timeit(lambda: x.view(torch.float32).sum()); print(f"f32 sum read: {t*1e6:.1f} us  {gb/t/1e3:.2f} TB/s")
t = timeit(lambda: x.view(torch.float32).max()); print(f"f32 max read: {t*1e6:.1f} us  {gb/t/1e3:.2f} TB/s")
t = timeit(lambda: y.copy_(x)); print(f"copy: {t*1e6:.1f} us  {2*gb/t/1e3:.2f} TB/s (read+write)")
