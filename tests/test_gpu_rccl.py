"""RCCL on one GPU: a one-rank "nccl" (= RCCL on ROCm) process group drives the DEVICE branch of the collectives the batch-sharded
inference path uses — all_gather_into_tensor of per-sample BEV grids and all_reduce(SUM) of the metric counters
(streamingflow_amd/dist.py; reference: evaluate.py has no inference collective, metrics.py:32-35 / 89-92 declare the counters with
dist_reduce_fx='sum').  First evidence that librccl loads and its gather / reduce kernels run on gfx950 with this torch build; the
multi-rank partition logic itself is covered on gloo (tests/test_dist_gloo.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture()
def one_rank_rccl():
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        yield
    finally:
        dist.destroy_process_group()


def test_device_all_gather_and_all_reduce_on_rccl(one_rank_rccl):
    from streamingflow_amd import dist as sfd
    assert dist.get_backend() == "nccl"
    n = 3
    g = torch.Generator(device="cuda").manual_seed(7)
    local = {i: torch.randn((1, 7, 64, 50, 50), device="cuda", generator=g) for i in range(n)}      # per-sample BEV grids
    want = [local[i].clone() for i in range(n)]
    got = sfd.gather_predictions(local, n, force_collective=True)
    torch.cuda.synchronize()
    assert len(got) == n and all(t.is_cuda for t in got)
    assert all(torch.equal(a, b) for a, b in zip(got, want))
    # the returned tensors are views of the gathered buffer, not the inputs: the collective really ran
    assert all(a.data_ptr() != b.data_ptr() for a, b in zip(got, [local[i] for i in range(n)]))
    cnt = torch.tensor([5.0, 7.0, 11.0, 13.0], device="cuda")
    out = sfd.reduce_counters(cnt, force_collective=True)
    torch.cuda.synchronize()
    assert out.tolist() == [5.0, 7.0, 11.0, 13.0]
    # and the plain paths are unchanged: one rank without the flag returns its own tensors, no collective
    same = sfd.gather_predictions(local, n)
    assert all(a.data_ptr() == local[i].data_ptr() for i, a in enumerate(same))
