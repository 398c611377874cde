import torch, time
x = torch.empty(3_000_000_000, dtype=torch.uint8).pin_memory()
d = torch.empty_like(x, device="cuda")
for chunk in (32 << 20, 256 << 20, 3_000_000_000):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for a in range(0, x.numel(), chunk):
        d[a:a + chunk].copy_(x[a:a + chunk], non_blocking=True)
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    print("pinned H2D, chunks of %d MB: %.1f ms = %.1f GB/s" % (chunk >> 20, t * 1e3, x.numel() / t / 1e9))
y = torch.empty(3_000_000_000, dtype=torch.uint8); y.fill_(65)
torch.cuda.synchronize(); t0 = time.perf_counter(); d.copy_(y); torch.cuda.synchronize(); t = time.perf_counter() - t0
print("pageable H2D (torch): %.1f ms = %.1f GB/s" % (t * 1e3, y.numel() / t / 1e9))
