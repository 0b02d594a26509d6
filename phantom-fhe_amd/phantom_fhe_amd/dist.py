"""Ciphertext-batch sharding over the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in CPU tests).
Independent ciphertexts are split into contiguous blocks per rank; there is NO collective on the
data path.  The only communication is the one-time broadcast of evaluation / Galois keys from the
rank that generated them, and the max-over-ranks reduction of the timed region.
"""
import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _group():
    """True when a process group exists (also a one-rank group: its collectives still run through the backend)."""
    return dist.is_available() and dist.is_initialized()


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def shard_range(batch, rank_, world_):
    """Contiguous block of ciphertext indices owned by rank_ (sizes differ by at most one)."""
    base, rem = divmod(batch, world_)
    lo = rank_ * base + min(rank_, rem)
    return range(lo, lo + base + (1 if rank_ < rem else 0))


def broadcast_keys(keys, src=0):
    """Broadcast every key tensor ([2][#QP][N] int64) from src to all ranks, in place."""
    if not _group():
        return keys
    for k in keys:
        dist.broadcast(k, src=src)
    return keys


def max_over_ranks(seconds, device=None):
    """Whole-job time of a region = the slowest rank's time."""
    if not _group():
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_checksums(local, device=None):
    """All-gather one int64 checksum per rank (used to check that N-GPU runs reproduce 1-GPU results)."""
    if not _group():
        return [int(local)]
    t = torch.tensor([local], dtype=torch.int64, device=device)
    out = [torch.zeros_like(t) for _ in range(world())]
    dist.all_gather(out, t)
    return [int(x.item()) for x in out]
