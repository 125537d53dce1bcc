"""Batch-sharded inference across the GPUs of one node.

One sample of the GRU-ODE path is a strictly serial chain on a 50x50 latent — it does not shard.
Samples do: the reference is batch-1 anyway, so sample i goes to rank i mod W (one process per GPU,
weights replicated, every rank keeps its own packed weights and hipGraph cache).  There is NO
collective on the data path.  Results are collected with one RCCL all-gather of the per-sample BEV
grids (backend "nccl" = RCCL over xGMI; a fully connected 8-GPU node does it in one hop), or, for
evaluation, one all-reduce(SUM) of the IoU / PQ counters.  The same code runs on the gloo backend
with CPU tensors (tests/test_dist_gloo.py).
"""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_indices(n_samples, rank=None, world_size=None):
    """Round-robin sample partition: rank r owns r, r+W, r+2W, ..."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    return list(range(rank, n_samples, world_size))


def run_sharded(model_fn, samples):
    """Run ``model_fn(sample)`` on this rank's share of ``samples``; returns {index: output}."""
    return {i: model_fn(samples[i]) for i in shard_indices(len(samples))}


def gather_predictions(local, n_samples, like=None, force_collective=False):
    """All-gather per-sample outputs (same shape on every rank) so that every rank holds the full,
    index-ordered list.  ``local``: {sample index: tensor}.  One collective for the whole shard:
    the shard is stacked (padded to ceil(n/W) rows) and gathered with all_gather_into_tensor.
    A single rank returns its own tensors without a collective unless ``force_collective`` (a
    one-rank process group then runs the same device all-gather: tests/test_gpu_rccl.py)."""
    rank, w = world()
    if w == 1 and not (force_collective and dist.is_available() and dist.is_initialized()):
        return [local[i] for i in range(n_samples)]
    per = (n_samples + w - 1) // w
    mine = shard_indices(n_samples)
    ref = local[mine[0]] if mine else like
    if ref is None:
        raise ValueError("a rank without samples needs `like` to know the output shape")
    buf = ref.new_zeros((per,) + tuple(ref.shape))
    for slot, i in enumerate(mine):
        buf[slot] = local[i]
    out = ref.new_empty((w * per,) + tuple(ref.shape))
    if ref.is_cuda and dist.get_backend() == "gloo":
        # gloo has no device all-gather: stage the shard through the host (the multi-rank path on a box without RCCL
        # peers, e.g. two ranks sharing one GPU — bench.py SF_BENCH_BACKEND=gloo)
        host = torch.empty(out.shape, dtype=out.dtype, pin_memory=True)
        dist.all_gather_into_tensor(host, buf.cpu())
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, buf)
    out = out.view((w, per) + tuple(ref.shape))
    return [out[i % w, i // w] for i in range(n_samples)]


def reduce_counters(t, force_collective=False):
    """Sum metric counters (IoU intersection/union, PQ tp/fp/fn/iou — reference metrics.py:32-35,
    89-92 declare them with dist_reduce_fx='sum') over all ranks, in place."""
    if world()[1] > 1 or (force_collective and dist.is_available() and dist.is_initialized()):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t
