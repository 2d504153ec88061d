"""Dev helper (GPU box): write-only and copy bandwidth of the GPU as torch sees them (fill / copy of 8 GB)."""
import time, torch
dev = torch.device("cuda:0")
n = 8 * 1024 ** 3
a = torch.empty(n, dtype=torch.uint8, device=dev)
b = torch.empty(n, dtype=torch.uint8, device=dev)
for name, fn, bytes_ in (("fill", lambda: a.fill_(7), n), ("copy", lambda: b.copy_(a), 2 * n), ("read (sum)", lambda: a.view(torch.int32).sum(), n)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(name, round(bytes_ / dt / 1e12, 2), "TB/s", flush=True)
