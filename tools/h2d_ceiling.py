"""Host->device copy rate from pinned memory, as config C5's pipeline issues it (2 MB descriptor blocks), by number of
streams: the ceiling the end-to-end batch figure is read against.   python tools/h2d_ceiling.py"""
import time

import torch

dev = torch.device("cuda", 0)
for chunk_mb in (2, 8):
    n = chunk_mb * (1 << 20) // 4
    for streams in (1, 2, 3, 4, 6):
        ss = [torch.cuda.Stream(device=dev) for _ in range(streams)]
        src = [torch.empty(n, dtype=torch.float32).pin_memory() for _ in range(streams)]
        dst = [torch.empty(n, dtype=torch.float32, device=dev) for _ in range(streams)]
        reps = 200
        for warm in (True, False):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for r in range(20 if warm else reps):
                for i, s in enumerate(ss):
                    with torch.cuda.stream(s):
                        dst[i].copy_(src[i], non_blocking=True)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print("chunk %d MB, %d stream(s): %.1f GB/s" % (chunk_mb, streams, reps * streams * n * 4 / dt / 1e9))
