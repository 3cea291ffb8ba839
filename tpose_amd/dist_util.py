"""Small torch.distributed helpers shared by bench.py and the multi-GPU drivers."""


def init(backend=None):
    """Initialise the process group from the torchrun environment; returns (dist, rank, world, device).
    backend None -> "nccl" (RCCL) when CUDA/HIP devices exist, else "gloo"."""
    import os

    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
        dist.init_process_group("nccl", device_id=device)  # RCCL over xGMI
    else:
        device = torch.device("cpu")
        dist.init_process_group(backend)
    return dist, rank, world, device


def max_over_ranks(dist, value, device):
    """MAX all-reduce of a python float (the bench's whole-job time is the slowest rank's)"""
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def replica_seed(rank, base=1234):
    """independent replica per rank: its own synthetic image"""
    return base + rank
