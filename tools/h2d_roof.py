"""What the file route's FASTA -> HBM phase can reach on this box: host-to-device copy rates (pinned, pageable) and the host's own
copy rate into a pinned buffer with T threads (the staging step of ingest.hip: page cache -> pinned -> HBM).
   python tools/h2d_roof.py   -> one JSON line"""
import json
import threading
import time

import numpy as np
import torch

GB = 1 << 30
dev = torch.device("cuda", 0)
dst = torch.empty(GB, dtype=torch.uint8, device=dev)
pin = torch.empty(GB, dtype=torch.uint8).pin_memory()
page = torch.empty(GB, dtype=torch.uint8)
pin.fill_(1); page.fill_(2)
torch.cuda.synchronize()


def rate(src, reps=5):
    dst.copy_(src, non_blocking=True); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    return GB * reps / (time.perf_counter() - t0) / 1e9


def host_copy(threads, reps=3):
    src = page.numpy(); out = pin.numpy()
    def work(t):
        lo, hi = GB * t // threads, GB * (t + 1) // threads
        out[lo:hi] = src[lo:hi]
    t0 = time.perf_counter()
    for _ in range(reps):
        th = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
        [x.start() for x in th]; [x.join() for x in th]
    return GB * reps / (time.perf_counter() - t0) / 1e9


out = {"h2d_pinned_gbs": round(rate(pin), 2), "h2d_pageable_gbs": round(rate(page), 2),
       "host_copy_into_pinned_gbs": {str(t): round(host_copy(t), 2) for t in (1, 4, 8, 16)},
       "note": "1 GiB per copy; the file route stages page-cache text through pinned buffers (host copy) and uploads them (pinned H2D), "
               "the two overlapped: its ceiling is the smaller of the two rates"}
print(json.dumps(out))
