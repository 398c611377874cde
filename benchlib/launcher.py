"""benchlib.launcher — arguments, rank processes, NUMA binding and main() of bench.py."""
from . import common
from .common import *  # noqa: F401,F403  (the standard modules bench.py always imported, and its constants)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--verify", action="store_true",
                    help="untimed full-size parity properties (independent torch computation; at N > 1 also "
                         "bit-equality of the assembled arrays with a single-GPU scan of the whole assembly on rank 0)")
    ap.add_argument("--blocks", action="store_true", help="also time device block calling (untimed in value)")
    ap.add_argument("--no-e2e", action="store_true",
                    help="skip the PCIe-inclusive leg (host buffers through the C-ABI entry points; never in value)")
    ap.add_argument("--e2e", action="store_true", help="(kept for compatibility: the PCIe-inclusive leg is on by default at N = 1)")
    ap.add_argument("--flags", default=FLAGS, help="Teloscope flags of the workload (default: configs[1])")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--gbases", type=float, default=3.0, help="bases of the assembly (Gb); 3.0 = BASELINE config")
    ap.add_argument("--contigs", type=int, default=200)
    ap.add_argument("--full-exchange", action="store_true",
                    help="N > 1: round 2's exchange (every window, directory entry and match record assembled on rank 0) instead of "
                         "the shard results (blocks called per rank; packed windows, writer-visible records and blocks travel)")
    ap.add_argument("--weak", action="store_true", help="N > 1: every rank scans its own --gbases assembly (weak scaling; "
                                                          "only per-segment hit summaries are gathered)")
    ap.add_argument("--no-reads", action="store_true", help="skip the `reads` sub-record of the default line (configs[3] at 500 k reads)")
    ap.add_argument("--cpu-sample-mb", type=float, default=384.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--reads", action="store_true", help="benchmark the read filter (configs[3]) instead of the assembly scan")
    ap.add_argument("--n-reads", type=float, default=5e6, help="--reads: reads filtered per step across all GPUs")
    return ap.parse_args()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def ensure_built():
    """A fresh checkout has no built artefacts: build them BEFORE any GPU or torch.distributed call (hipcc, gcc;
    the first local rank builds, the others wait for the files).  Nothing here is a fallback — without the HIP
    library the import of teloscope_amd raises."""
    if os.path.exists(LIB) and os.path.exists(ORACLE):
        return
    if int(os.environ.get("LOCAL_RANK", "0")) == 0:
        import __graft_entry__ as entry
        entry.build()
        return
    t0 = time.time()
    while not (os.path.exists(LIB) and os.path.exists(ORACLE)):
        if time.time() - t0 > 900:
            raise RuntimeError("libteloscan.so was not built by local rank 0")
        time.sleep(1.0)
    time.sleep(2.0)                                            # let the linker finish writing


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N rank processes of this script (the parent makes no
    GPU call).  Rank 0 writes the JSON line to our stdout; every rank's stderr is ours.  The first rank that fails takes
    the others down with it (they would wait in a collective for ever), and its exit code is ours."""
    ensure_built()
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py")] + sys.argv[1:], env=e,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    live = set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = abs(code) or 1
                sys.stderr.write("bench.py: rank %d exited with code %d; stopping the other ranks\n" % (r, code))
                for q in sorted(live):
                    procs[q].terminate()
        if live:
            time.sleep(0.05)
    return rc

def bind_to_gpu_node(dev_index):
    """One process per GPU, on the CPUs of the NUMA node the GPU hangs off (what `numactl --cpunodebind` does for a rank):
    on a two-socket host a pageable buffer or a staging thread on the other socket costs 15-20 % of the PCIe-inclusive
    rate.  Returns the node, or None when the topology cannot be read (then nothing is bound)."""
    try:
        import torch
        p = torch.cuda.get_device_properties(dev_index)
        bus = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bus).read())
        if node < 0 or os.environ.get("TS_NO_NUMA_BIND"):
            return None
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        want = os.sched_getaffinity(0) & cpus
        if not want:
            return None
        os.sched_setaffinity(0, want)
        return node
    except Exception:
        return None


def main():
    args = parse_args()
    common.ORIG_AFFINITY = os.sched_getaffinity(0)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    ensure_built()                                              # before any GPU / torch.distributed call

    import torch  # noqa: E402  (imported before libteloscan so both share one HIP runtime)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    ndev = max(1, torch.cuda.device_count())
    dev_index = local_rank % ndev
    # one rank per GPU over RCCL; "gloo" only to rehearse N > 1 where ranks have to share a GPU
    backend = os.environ.get("TS_BENCH_BACKEND") or ("nccl" if ndev >= int(os.environ.get("LOCAL_WORLD_SIZE", world)) else "gloo")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    common.HOST_NUMA_NODE = bind_to_gpu_node(dev_index)
    # TS_BENCH_FORCE_STRONG=1: the N > 1 code path (shard object, export, exchange, adopt) with ONE rank on the RCCL group — as
    # far as the sharded path can be taken on real RCCL where two ranks cannot share a GPU (rehearsal hook, not a bench mode)
    forced = world == 1 and bool(os.environ.get("TS_BENCH_FORCE_STRONG"))
    if forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or forced:
        import torch.distributed as dist
        import datetime
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=600))
        else:
            dist.init_process_group(backend, timeout=datetime.timedelta(seconds=600))
    try:
        if args.reads:
            from .reads import run_reads
            run_reads(args, rank, local_rank, world, dev, backend)
        else:
            from .contigs import run_scan
            run_scan(args, rank, local_rank, world, dev, backend)
    finally:
        if world > 1 or forced:
            import torch.distributed as dist
            dist.destroy_process_group()

