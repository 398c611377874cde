"""N > 1 path on CPU: two gloo ranks shard a set of segments, each produces per-segment hit
summaries for its shard (here from the oracle, standing in for ts_batch_segment_summary which
needs a GPU), and the product's gather puts them back in global order on rank 0."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _summaries(opts, seqs):
    from tests.backends import OracleBackend
    ob = OracleBackend(opts)
    rows = []
    for s in seqs:
        r = ob.scan_segment(s, 0, False)
        m = r["all_matches"]
        rows.append([len(r["windows"]), len(m), int(m["is_canonical"].sum()), int(m["is_forward"].sum())])
    return np.array(rows, dtype=np.int64).reshape(-1, 4)


def _worker(rank, world, port, lens, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from teloscope_amd.distributed import gather_segment_summaries, lpt_partition
    from tests import harness as H
    from tests import seqgen

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    opts = H.parse_cli("x -w 1000 -s 500 -r -g -i")
    seqs = [seqgen.chromosome(np.random.default_rng(100 + i), n, n_its=2) for i, n in enumerate(lens)]
    mine = lpt_partition(lens, world)[rank]
    local = torch.from_numpy(_summaries(opts, [seqs[i] for i in mine]))
    got = gather_segment_summaries(local, mine, len(lens), dst=0)
    if rank == 0:
        q.put(got)
    dist.barrier()
    dist.destroy_process_group()


def test_lpt_partition_is_balanced_and_complete():
    from teloscope_amd.distributed import lpt_partition
    rng = np.random.default_rng(3)
    lens = [int(x) for x in np.exp(rng.uniform(np.log(1e6), np.log(250e6), size=200))]
    for world in (1, 2, 4, 8):
        shards = lpt_partition(lens, world)
        assert sorted(i for s in shards for i in s) == list(range(200))
        loads = [sum(lens[i] for i in s) for s in shards]
        assert max(loads) - min(loads) <= max(lens)
    assert lpt_partition([], 4) == [[], [], [], []]


@pytest.mark.timeout(180)
def test_two_rank_gather_matches_single_process():
    import torch.multiprocessing as mp
    from tests import harness as H
    from tests import seqgen
    lens = [30000, 1500, 52000, 7, 999, 20001, 12345]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, lens, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=150)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    opts = H.parse_cli("x -w 1000 -s 500 -r -g -i")
    seqs = [seqgen.chromosome(np.random.default_rng(100 + i), n, n_its=2) for i, n in enumerate(lens)]
    assert np.array_equal(got, _summaries(opts, seqs))
