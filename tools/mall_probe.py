#!/usr/bin/env python3
"""Does a buffer written by one kernel get read faster by the next one (256 MB Infinity Cache)?
Times y = x + 1 (read n, write n bytes) on a buffer that was just written vs one evicted by 2 GB of other traffic."""
import torch, time
torch.cuda.init()
dev = "cuda"
big = torch.empty(512 * 1024 * 1024, dtype=torch.float32, device=dev)  # 2 GB evictor
def timed(fn, reps=1):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); fn(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)
for mb in (16, 32, 64, 128, 192, 256, 512):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, dtype=torch.float32, device=dev); b = torch.empty_like(a); c = torch.empty_like(a)
    a.uniform_()
    res = {}
    for mode in ("cold", "hot"):
        ts = []
        for _ in range(5):
            torch.add(a, 1.0, out=b)            # producer: writes b
            if mode == "cold":
                big.fill_(0.0)                  # evict
            torch.cuda.synchronize()
            ts.append(timed(lambda: torch.add(b, 1.0, out=c)))   # consumer: reads b, writes c
        res[mode] = sorted(ts)[2]
    gb = 2 * n * 4 / 1e9
    print(f"{mb:4d} MB: consumer cold {res['cold']*1e3:8.1f} us ({gb/res['cold']*1e3:6.0f} GB/s)   hot {res['hot']*1e3:8.1f} us ({gb/res['hot']*1e3:6.0f} GB/s)")
