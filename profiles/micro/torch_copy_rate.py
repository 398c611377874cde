#!/usr/bin/env python3
"""What torch's own device-to-device copy and fill reach on this box (read + write / write-only GB/s): context for
roofline.box.copy_read_plus_write_gbs (ts_box_probe's plain 16-byte copy kernel)."""
import torch
dev = torch.device("cuda", 0)
for gb in (1, 3):
    n = gb << 30
    a = torch.zeros(n, dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    best_c = best_f = 0.0
    for _ in range(6):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(); b.copy_(a); e1.record(); b.fill_(7); e2.record()
        torch.cuda.synchronize()
        best_c = max(best_c, 2 * n / e0.elapsed_time(e1) / 1e6)
        best_f = max(best_f, n / e1.elapsed_time(e2) / 1e6)
    print("%d GiB: torch copy_ %.0f GB/s (read + write), fill_ %.0f GB/s (write only)" % (gb, best_c, best_f))
