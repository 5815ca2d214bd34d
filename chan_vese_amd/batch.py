"""Batch sharding of independent images across the GPUs of one node.

The path shards embarrassingly: image b of a batch of B goes to rank b // (B / R) (C5: images
8r .. 8r+7 on GPU r).  There is no data-path collective; torch.distributed is used only for
the barriers around the timed region, the max-over-ranks of the elapsed time and the
end-of-run gather of per-rank throughput records (RCCL via backend "nccl" on GPUs, "gloo" in
the CPU tests).
"""
import os


def shard_images(num_images, rank, world):
    """Contiguous block partition: rank r owns images [r*B/R, (r+1)*B/R)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    lo = (num_images * rank) // world
    hi = (num_images * (rank + 1)) // world
    return list(range(lo, hi))


def init_distributed(backend=None):
    """One process per GPU as launched by torch.distributed.run; returns (dist or None, rank, world, local_rank).
    torch is imported HERE, before the HIP library is loaded, so both share one HIP runtime."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # CHANVESE_DIST_FORCE=1: build the process group even for ONE rank (the RCCL path on a one-GPU box:
    # tests/test_gpu_fullsize.py::test_bench_one_rank_over_rccl); MASTER_ADDR / MASTER_PORT must be set
    if world == 1 and os.environ.get("CHANVESE_DIST_FORCE") != "1":
        return None, 0, 1, 0
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend=backend)
    return dist, rank, world, local_rank


def _tensor(values, dist):
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    return torch.tensor(values, dtype=torch.float64, device=dev)


def barrier(dist):
    if dist is not None:
        dist.barrier()


def max_over_ranks(dist, value):
    if dist is None:
        return float(value)
    t = _tensor([float(value)], dist)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_records(dist, record):
    """End-of-run gather: every rank contributes [images_done, pixel_iterations, seconds];
    returns the list of records (one per rank) on every rank."""
    if dist is None:
        return [list(map(float, record))]
    import torch
    t = _tensor(list(map(float, record)), dist)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[float(x) for x in o.tolist()] for o in out]


def aggregate_throughput(records, elapsed_max):
    """Whole-job Mpixel-iterations/s = all pixel-iterations / slowest rank's time."""
    total = sum(r[1] for r in records)
    return total / elapsed_max / 1e6
