"""Multi-GPU sharding of the scan path: one process per GPU, torch.distributed (RCCL on
ROCm; gloo for CPU rehearsal).

The reference parallelises by path (one thread-pool job per path, src/input.cpp:719-724);
segments are independent units here too, so ranks scan disjoint sets of segments with no
data-path collective.  The only exchange is ONE gather of the per-segment hit summaries
({windows, matches, canonical, forward} x int64, produced on the device by
ts_batch_segment_summary) to rank 0, which needs them for the path summary report; window
records and match streams stay on the rank that produced them.
"""
from typing import List, Sequence

import numpy as np


def lpt_partition(lengths: Sequence[int], world_size: int) -> List[List[int]]:
    """Longest-processing-time-first assignment of segments to ranks (deterministic):
    returns, per rank, the ascending list of segment indices it scans."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    load = [0] * world_size
    shards: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda j: (load[j], j))
        shards[r].append(i)
        load[r] += int(lengths[i])
    return [sorted(s) for s in shards]


def gather_segment_summaries(local_summary, local_indices, n_total, dst=0, group=None):
    """One fixed-size gather of per-segment summaries to `dst`.

    local_summary: tensor [n_local, 4] int64 on the rank's device (cuda for nccl, cpu for gloo),
    rows in the order of local_indices.  Returns on dst a numpy array [n_total, 4] in global
    segment order, None elsewhere."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = local_summary.device
    n_local = torch.tensor([local_summary.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)              # 8 bytes per rank
    max_n = max(int(c.item()) for c in counts)
    # rows: [global index, windows, matches, canonical, forward]; padded to the largest shard
    msg = torch.full((max_n, 5), -1, dtype=torch.int64, device=dev)
    if local_summary.shape[0]:
        msg[:local_summary.shape[0], 0] = torch.as_tensor(list(local_indices), dtype=torch.int64, device=dev)
        msg[:local_summary.shape[0], 1:] = local_summary
    bufs = [torch.empty_like(msg) for _ in range(world)] if rank == dst else None
    dist.gather(msg, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    out = np.zeros((n_total, 4), dtype=np.int64)
    seen = np.zeros(n_total, dtype=bool)
    for b in bufs:
        a = b.cpu().numpy()
        a = a[a[:, 0] >= 0]
        out[a[:, 0]] = a[:, 1:]
        seen[a[:, 0]] = True
    if not seen.all():
        raise RuntimeError("segment summaries missing after gather: %s" % np.flatnonzero(~seen)[:8])
    return out
