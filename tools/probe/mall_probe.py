"""Does the 256 MiB Infinity Cache keep a tensor between the kernel that WRITES it and the kernel that reads it next?
For sizes from 16 MB to 1 GB: time a streaming read (torch sum over bf16) of a tensor (a) right after a kernel wrote it, (b) after a
1.5 GB unrelated stream has gone through the memory system in between.  Read bandwidth in TB/s."""
import torch
dev = torch.device("cuda:0")


def timed(fn, n=5):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e-3)
    return sorted(ts)[len(ts) // 2]


flush = torch.empty(768 * 1024 * 1024, dtype=torch.bfloat16, device=dev)  # 1.5 GB
for mb in (16, 32, 64, 128, 192, 256, 384, 512, 1024):
    n = mb * 1024 * 1024 // 2
    src = torch.randn(n, device=dev).to(torch.bfloat16)
    y = torch.empty_like(src)
    out = torch.empty((), dtype=torch.float32, device=dev)
    hot, cold = [], []
    for _ in range(5):
        y.copy_(src)                     # the producer kernel writes y
        hot.append(timed(lambda: torch.sum(y, dtype=torch.float32), 1))
        y.copy_(src)
        flush.zero_()                    # 1.5 GB of unrelated writes
        cold.append(timed(lambda: torch.sum(y, dtype=torch.float32), 1))
    h, c = sorted(hot)[2], sorted(cold)[2]
    print("%5d MB   read right after the write: %6.2f TB/s (%7.1f us)    after a 1.5 GB flush: %6.2f TB/s (%7.1f us)" % (mb, mb * 1.048576e6 / h / 1e12, h * 1e6, mb * 1.048576e6 / c / 1e12, c * 1e6))
