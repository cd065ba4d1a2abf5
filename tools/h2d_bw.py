"""Development aid: host -> device copy rates on this box (pinned vs pageable, first copy of a fresh buffer vs repeats), to judge
the read upload of the CLI."""
import time
import numpy as np
import torch
n = 512 << 20
dev = torch.device("cuda", 0)
dst = torch.empty(n, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
for kind in ("pinned", "pageable", "pageable, fresh buffer each time"):
    src = None
    for rep in range(3):
        if src is None or "fresh" in kind:
            t0 = time.perf_counter()
            src = torch.empty(n, dtype=torch.uint8, pin_memory=(kind == "pinned"))
            src.fill_(1)
            t_alloc = time.perf_counter() - t0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{kind} rep {rep}: {n / dt / 1e9:.1f} GB/s ({dt * 1e3:.1f} ms for 512 MiB; alloc+fill {t_alloc * 1e3:.0f} ms)", flush=True)
