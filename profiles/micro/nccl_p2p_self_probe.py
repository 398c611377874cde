"""Probe (GPU box, one GPU): the exchange's point-to-point calls on the NCCL / RCCL process group as far as one rank can take
them — a batch of isend + irecv of u16 data viewed as bytes, rank 0 to itself, through batch_isend_irecv (the coalesced path
gather_shards uses), then widened.  Two ranks cannot share a GPU under RCCL ("Duplicate GPU detected"), so this and the gloo
tests are what can run before the driver's 8-GPU job."""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29556")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dist.barrier()
src = (torch.arange(100000, device="cuda", dtype=torch.int32) * 7) & 0xFFFF
wire = src.to(torch.int16)
land = torch.empty(100000 + 16, dtype=torch.int16, device="cuda")[:100000]
cnt = [torch.zeros(1, dtype=torch.int64, device="cuda")]
dist.all_gather(cnt, torch.tensor([100000], dtype=torch.int64, device="cuda"))
ops = [dist.P2POp(dist.isend, wire.view(torch.uint8), 0, tag=3), dist.P2POp(dist.irecv, land.view(torch.uint8), 0, tag=3)]
for w in dist.batch_isend_irecv(ops):
    w.wait()
torch.cuda.synchronize()
got = land.to(torch.int32) & 0xFFFF
print("all_gather:", int(cnt[0].item()), " p2p to self over the NCCL group:", "ok" if torch.equal(got, src) else "MISMATCH")
# the other collectives bench.py issues at N > 1: gather of pass bytes / int64 summaries (async), MAX all-reduce of a double
pb = torch.ones(1000, dtype=torch.uint8, device="cuda"); outb = [torch.zeros_like(pb)]
dist.gather(pb, outb, dst=0)
sm = torch.arange(800, dtype=torch.int64, device="cuda"); outs = [torch.zeros_like(sm)]
dist.gather(sm, outs, dst=0, async_op=True).wait()
t = torch.tensor([1.25], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
print("gather u8:", bool(outb[0].all()), " gather i64 async:", torch.equal(outs[0], sm), " all_reduce MAX f64:", float(t.item()))
dist.destroy_process_group()
