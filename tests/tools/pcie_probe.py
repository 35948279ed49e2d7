"""One-off probe: device->host and host->device copy rates into pinned memory for a few sizes (what bounds the ICP finish
and the host-image path)."""
import time, torch
dev = torch.device("cuda:0")
for mb in (1, 7.3, 32, 128):
    n = int(mb * 1e6) // 4
    d = torch.empty(n, dtype=torch.float32, device=dev)
    h = torch.empty(n, dtype=torch.float32).pin_memory()
    for name, fn in (("D2H", lambda: h.copy_(d, non_blocking=True)), ("H2D", lambda: d.copy_(h, non_blocking=True))):
        fn(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20): fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 20
        print("%s %6.1f MB: %7.1f us  %5.1f GB/s" % (name, mb, dt * 1e6, mb * 1e6 / dt / 1e9))
