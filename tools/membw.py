"""HBM bandwidth sanity numbers on this GPU (torch fill / copy / read-reduce), for calibrating the roofline."""
import torch, sys
dev = torch.device("cuda:0")
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e-3
for mb in (58, 256, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
    tf = t(lambda: a.fill_(1.0)); tc = t(lambda: b.copy_(a)); tr = t(lambda: a.sum())
    print("%5d MB: fill %.0f GB/s   copy (r+w) %.0f GB/s   sum-read %.0f GB/s" % (mb, n*4/tf/1e9, 2*n*4/tc/1e9, n*4/tr/1e9))
