"""CPU, world_size 2, gloo: the sample-sharding / gather / counter-reduction helpers used for
multi-GPU inference (streamingflow_amd.dist).  No GPU compute here — the per-sample function is
the CPU oracle of one tiny conv-GRU cell — the point is the partition and the collectives."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from util import build_pair, hashfill
    from oracle import ref_torch as R
    from streamingflow_amd import dist as sfd
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    _, sd = build_pair(8, device="cpu")
    n = 5
    samples = [(hashfill.normal(f"x{i}", (1, 8, 6, 6), i), hashfill.normal(f"s{i}", (1, 8, 6, 6), 100 + i)) for i in range(n)]
    fn = lambda xs: R.gru_cell(sd, "spatial_grus.0", xs[0], xs[1])
    with torch.no_grad():
        local = sfd.run_sharded(fn, samples)
        assert sorted(local) == list(range(rank, n, world))
        full = sfd.gather_predictions(local, n)
        want = [fn(s) for s in samples]
    ok = all(torch.equal(a, b) for a, b in zip(full, want))
    # padded shards: 3 samples over 2 ranks (2 + 1), and 1 sample (rank 1 owns nothing and passes `like`)
    with torch.no_grad():
        for m in (3, 1):
            loc = sfd.run_sharded(fn, samples[:m])
            got = sfd.gather_predictions(loc, m, like=torch.empty(1, 8, 6, 6))
            ok = ok and len(got) == m and all(torch.equal(a, b) for a, b in zip(got, want[:m]))
    if rank == 1:
        try:
            sfd.gather_predictions({}, 0)       # nothing to infer the shape from and no `like`
            ok = False
        except ValueError:
            pass
    cnt = sfd.reduce_counters(torch.tensor([float(len(local)), 1.0]))
    ok = ok and cnt.tolist() == [float(n), float(world)]
    q.put((rank, ok))
    dist.destroy_process_group()


def test_sharded_inference_two_ranks():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_shard_indices_cover_everything():
    from streamingflow_amd import dist as sfd
    for n in (0, 1, 7, 8, 19):
        for w in (1, 2, 4, 8):
            got = sorted(i for r in range(w) for i in sfd.shard_indices(n, r, w))
            assert got == list(range(n))
