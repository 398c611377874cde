import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29555")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
for dt in (torch.int16, torch.uint8, torch.int32, torch.int64, torch.float64):
    t = torch.ones(8, dtype=dt, device="cuda")
    out = [torch.zeros_like(t)]
    try:
        dist.all_gather(out, t); torch.cuda.synchronize(); print(dt, "ok")
    except Exception as e:
        print(dt, "FAILS:", str(e)[:120])
dist.destroy_process_group()
