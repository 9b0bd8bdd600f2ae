"""Stream sharding across ranks (one process per GPU).  The reconstruction path shards only
across independent streams: stream i lives on rank i % world with its own frame
store; there is NO data-path collective.  torch.distributed is used for rendezvous, the
barrier and the max-over-ranks clock of bench.py (backend "nccl" = RCCL on the GPU box,
"gloo" in the CPU tests)."""
import os


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def streams_of_rank(n_streams_total, rank, world):
    """Global stream ids owned by `rank` (round-robin, like SURVEY 8e 'stream i -> GPU i mod 8')."""
    return [s for s in range(n_streams_total) if s % world == rank]


def stream_seed(global_stream_id, distinct):
    """Seed of the synthetic stream a global stream id decodes (bench.py): `distinct` different
    streams per rank, the rest are private copies."""
    return 1000 + global_stream_id % max(distinct, 1)


def init(backend, device=None):
    import torch.distributed as dist
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, **kw)
    return rank, local_rank, world


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    """The job's clock is the slowest rank's."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_objects(obj):
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out
