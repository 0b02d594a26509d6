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


def _flat_runs(keys):
    """Merge CONSECUTIVE key tensors of the list that lie back to back in one storage into flat 1-D views (slices of one
    allocation per key set become ONE buffer); every other tensor stays as it is.  The list order is kept -- every rank must
    issue the same sequence of collectives, so the result depends only on how the caller laid the keys out (which must be the
    same on every rank: either one slab per key set everywhere, or separate tensors everywhere)."""
    out, cur = [], None   # cur = [storage, first_byte, end_byte, dtype, device]
    def flush():
        nonlocal cur
        if cur is not None:
            st, lo, hi, dt, dev = cur
            esz = torch.empty(0, dtype=dt).element_size()
            out.append(torch.empty(0, dtype=dt, device=dev).set_(st, (lo - st.data_ptr()) // esz, ((hi - lo) // esz,)))
            cur = None
    for k in keys:
        if not k.is_contiguous():
            flush()
            out.append(k)
            continue
        st = k.untyped_storage()
        lo = k.data_ptr()
        hi = lo + k.numel() * k.element_size()
        if cur is not None and cur[0].data_ptr() == st.data_ptr() and cur[3] == k.dtype and lo == cur[2]:
            cur[2] = hi
        else:
            flush()
            cur = [st, lo, hi, k.dtype, k.device]
    flush()
    return out


def _raw_nccl_comm(device):
    """(ncclComm_t of the default group as an integer, None) -- or (None, reason) when it cannot be reached: gloo, no group, a
    torch without the accessor, or a communicator that does not exist yet because no collective has run on `device`."""
    try:
        pg = dist.distributed_c10d._get_default_group()
    except Exception as e:   # no default group
        return None, f"no default process group ({e})"
    backend = dist.get_backend(pg)
    if backend != "nccl":
        return None, f"the process group's backend is '{backend}', not 'nccl' (RCCL)"
    try:
        be = pg._get_backend(torch.device(device))
    except Exception as e:
        return None, f"ProcessGroup._get_backend({device}) failed: {e}"
    if not hasattr(be, "_comm_ptr"):
        return None, f"this torch build's {type(be).__name__} has no _comm_ptr() accessor"
    try:
        ptr = int(be._comm_ptr())
    except Exception as e:
        return None, f"_comm_ptr() raised {type(e).__name__}: {e}"
    if not ptr:
        return None, "_comm_ptr() returned null (no collective has created the communicator on this device yet)"
    return ptr, None


LAST_BROADCAST_PATH = None        # "pha_broadcast_keys" / "dist.broadcast": which form the last broadcast_keys call took
LAST_BROADCAST_FALLBACK = None    # why a direct=True request was served by dist.broadcast instead (None: it was not, or not asked)
BROADCAST_CHUNK_BYTES = 1 << 31   # one collective call moves at most 2 GiB (a C5 key set is 12 GiB per rank)


def _check_same_layout(mine):
    """Every rank must issue the SAME sequence of collectives: compare a description of it (the (numel, dtype) list of the merged
    runs, or the (numel, count) list of the direct form) across the ranks before the first broadcast -- one small
    all_gather_object per key set; a mismatch would otherwise hang the job.

    On an RCCL group all_gather_object runs on torch.cuda.current_device(): the caller must have called
    torch.cuda.set_device(<this rank's device>) first (bench.py and workloads do), exactly as for any object collective;
    `check_layout=False` skips the check (and its pickle + host round trip) for callers that time the broadcast."""
    every = [None] * dist.get_world_size()
    dist.all_gather_object(every, mine)
    for r, other in enumerate(every):
        if other != mine:
            raise RuntimeError(f"broadcast_keys: rank {r} laid its keys out as {other}, rank {dist.get_rank()} as {mine}: the ranks "
                               "would issue different collective sequences (allocate every key set the same way on every rank)")


def broadcast_keys(keys, src=0, ctx=None, direct=False, check_layout=True):
    """Broadcast every key tensor ([2][#QP][N] int64) from src to all ranks, in place, in as few collective calls as the
    layout allows (192 per-tensor calls for the config-5 leg).

    * default: tensors that lie back to back in one storage (slices of one allocation per key set) are merged into flat
      views and each run is one `dist.broadcast`, split at BROADCAST_CHUNK_BYTES;
    * `direct=True` with an RCCL group and `ctx` (a PhantomContext on this rank's device): all keys go through ONE
      `pha_broadcast_keys` call per key size -- one ncclGroupStart / ncclGroupEnd around the set, on the process group's own
      communicator (csrc/pha_comm.hip), the call a C / C++ job makes.  Opt-in: it has only ever run as a one-rank group
      (no multi-GPU node was available to the build).  When the communicator cannot be reached the call falls back to the
      default form LOUDLY: a RuntimeWarning naming the reason, which is also left in LAST_BROADCAST_FALLBACK.
    Returns the NUMBER OF COLLECTIVE CALLS issued (0 without a process group) -- since r04; r01-r03 returned the key list, which
    is broadcast in place and therefore still what the caller holds."""
    global LAST_BROADCAST_PATH, LAST_BROADCAST_FALLBACK
    LAST_BROADCAST_FALLBACK = None
    if not _group() or not keys:
        return 0
    calls = 0
    if direct:
        comm, why = None, None
        if ctx is None:
            why = "no PhantomContext was passed (ctx=None)"
        elif not all(k.is_cuda and k.is_contiguous() for k in keys):
            why = "the keys are not contiguous tensors on a HIP device"
        elif any(k.element_size() != 8 for k in keys):
            why = "the keys are not 64-bit words"
        else:
            comm, why = _raw_nccl_comm(keys[0].device)
        if comm is not None:
            by_size = {}
            for k in keys:
                by_size.setdefault(k.numel(), []).append(k)
            if check_layout:
                _check_same_layout([(int(numel), len(g)) for numel, g in by_size.items()])
            torch.cuda.current_stream(keys[0].device).synchronize()   # the keys were written on torch's stream
            for group in by_size.values():
                ctx.broadcast_keys(group, src, comm)
                calls += 1
            torch.cuda.current_stream(keys[0].device).synchronize()   # one-time setup: hand the communicator back idle
            LAST_BROADCAST_PATH = "pha_broadcast_keys"
            return calls
        import sys
        import warnings
        LAST_BROADCAST_FALLBACK = why
        msg = f"broadcast_keys(direct=True): falling back to dist.broadcast -- {why}"
        warnings.warn(msg, RuntimeWarning, stacklevel=2)
        print("phantom_fhe_amd.dist: " + msg, file=sys.stderr, flush=True)
    LAST_BROADCAST_PATH = "dist.broadcast"
    runs = _flat_runs(keys)
    if check_layout:
        _check_same_layout([(int(t.numel()), str(t.dtype)) for t in runs])
    for t in runs:
        step = max(1, BROADCAST_CHUNK_BYTES // t.element_size()) if t.dim() == 1 and t.is_contiguous() else None
        if step is None or t.numel() <= step:
            dist.broadcast(t, src=src)
            calls += 1
        else:
            for lo in range(0, t.numel(), step):
                dist.broadcast(t[lo:lo + step], src=src)
                calls += 1
    return calls


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
