import torch, time
torch.cuda.init()
dev = torch.device("cuda:0")
sizes = [256, 1024, 2048, 4096, 8192, 12288, 16384, 18432, 20480, 24576, 32768, 65536, 262144]
dst = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
src = torch.empty(1 << 20, dtype=torch.uint8).pin_memory()
x = torch.zeros(1024, device=dev)
for n in sizes:
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ts = []
    for it in range(50):
        x.add_(1.0)                       # a kernel before (stream busy)
        ev[0].record()
        dst[:n].copy_(src[:n], non_blocking=True)
        ev[1].record()
        x.add_(1.0)                       # dependent kernel behind
        ev[2].record()
        torch.cuda.synchronize()
        ts.append((ev[0].elapsed_time(ev[1]) * 1e3, ev[0].elapsed_time(ev[2]) * 1e3))
    ts = ts[10:]
    import statistics
    print(f"{n:7d} B: copy {statistics.median(t[0] for t in ts):7.1f} us (median), copy + kernel {statistics.median(t[1] for t in ts):7.1f} us")
# host wall per iteration when back to back (no sync): 200 copies + kernels
for n in (1024, 18432, 24576):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for it in range(400):
        dst[:n].copy_(src[:n], non_blocking=True); x.add_(1.0)
    torch.cuda.synchronize(); print(n, "B back-to-back:", (time.perf_counter() - t0) / 400 * 1e6, "us per (copy + kernel)")
