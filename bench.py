#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native RNS core (BASELINE.json metric).

A "step" is one forward negacyclic NTT over one batch of ciphertext polynomials of the CKKS set
N = 2^16, 45 RNS limbs (examples/3_ckks.cu:729-739): NTT_BATCH = 16 polynomials x 45 limbs = 720
limb-transforms, 360 MiB, i.e. larger than the 256 MiB MALL, so every step streams from and to HBM.  It is the
reference call nwt_2d_radix8_forward_inplace(data, tables, 45, 0) (src/ntt/fntt_2d.cu:620-653) applied to every
polynomial of the batch in one launch pair (pha_nwt_2d_radix8_forward_inplace_batched; the reference's own
ntt_bench sweeps the limbs per launch the same way, benchmark/ntt_bench.cu:104-117).  `value` is
limb-transforms per second over all ranks.  The same line also carries
  * single_polynomial: the r01 protocol (one 45-limb polynomial per launch pair), in place (MALL-resident) and
    rotating over the 16 buffers (HBM-resident),
  * HomMul + relinearize + rescale per second for the same parameter set, alone and batched,
  * keyswitch_c4: BASELINE config 4 -- BFV relinearize + Galois rotate at N = 2^15, 30 + 15 limbs, a batch of 64
    ciphertexts split over the ranks (strong scaling), with a checksum that is the same for every world size,
  * the roofline object of the forward NTT and the CPU baseline (the oracle, timed on host cores).

Multi-GPU: independent ciphertexts shard across ranks (no data-path collective); evaluation / Galois keys are
generated on rank 0 and broadcast once over RCCL (setup, not timed).  `python bench.py --gpus N` without a
launcher starts the N ranks itself; under torch.distributed.run (RANK / WORLD_SIZE in the environment) it is one rank.
"""
import argparse
import hashlib
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "phantom-fhe_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

LOG_N = 16
BITS = [60] + [50] * 44 + [60] * 15   # 45 data primes + 15 special primes
SIZE_P = 15
NTT_BATCH = 16                        # polynomials per step: 16 x 22.5 MiB = 360 MiB > 256 MiB MALL
PEAK_HBM = 8.0e12                     # MI355X_MICROARCH.md: 8 TB/s HBM3E peak
C4_LOG_N = 15
C4_BITS = [60] + [50] * 29 + [60] * 15  # benchmark/keyswitch_bench.cu:25-34
C4_BATCH = 64
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic.json")   # PMC-derived HBM bytes (tools/traffic.sh)
STAGES_BATCHED_FILE = os.path.join(ROOT, "profiles", "stages_batched.json")   # the same for one op of a batch (tools/stage_table_batched.py)
STAGES_FILE = os.path.join(ROOT, "profiles", "stages.json")     # per-kernel GPU times of one HomMul (tools/stage_table.py over a committed kernel trace)
# BASELINE config 5: 128-slot encrypted mat-vec, one 128-diagonal block per row block = 64 baby x 2 giant steps, eight row blocks per
# pass over the baby keys (tools/time_bsgs.py)
C5_BABY, C5_GIANT = 64, 2
C5_BLOCKS = 32                        # row blocks of the job (split over the ranks: 4 per GPU at 8 GPUs, so the key pass is shared everywhere)


def spawn_ranks(n):
    """`python bench.py --gpus N` with no launcher: start N ranks of this script (one per GPU, RCCL rendezvous on
    127.0.0.1) and pass rank 0's stdout through.  The first rank that fails ends the job: its stderr tail is printed, its
    siblings (which would otherwise wait inside a collective for ever) are terminated, and its exit code is returned."""
    import tempfile
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs, logs = [], []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        log = tempfile.TemporaryFile(mode="w+b")
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL, stderr=log))
    rc, failed = 0, None
    live = set(range(n))
    while live and failed is None:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0:
                rc, failed = code, r
                break
        time.sleep(0.05)
    if failed is not None:
        for r in live:
            procs[r].terminate()
        for r in live:
            try:
                procs[r].wait(timeout=10)
            except subprocess.TimeoutExpired:
                procs[r].kill()
    for r, log in enumerate(logs):   # a failing rank's stderr in full (tail), the others' only if they said anything
        log.seek(0)
        text = log.read().decode("utf-8", "replace")
        if text.strip() and (failed is None or r == failed):
            sys.stderr.write(f"---- rank {r} stderr ----\n{text[-4000:]}\n")
    if failed is not None:
        sys.stderr.write(f"bench: rank {failed} exited with code {rc}; the other ranks were terminated\n")
    return rc


def uniform_residues(primes, n, device, gen):
    """[len(primes)][n] int64 tensor, limb i uniform in [0, primes[i]) (bits = uint64 residues)."""
    import torch
    out = torch.empty((len(primes), n), dtype=torch.int64, device=device)
    for i, q in enumerate(primes):
        out[i] = torch.randint(0, int(q), (n,), dtype=torch.int64, device=device, generator=gen)
    return out


def cpu_baseline(primes, n, seconds=12.0, gpu_forward=None):
    """Time the oracle's forward NTT (C port of the reference semantics) on the host: one core (the reported
    baseline) and, informational, OpenMP over the 45 limbs on every core of the box.  Before the timing is
    accepted the same input goes through the GPU path (gpu_forward) and the two outputs are compared bit for bit."""
    import numpy as np
    os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")   # idle OpenMP threads must not spin on a shared box
    from oracle import oracle as O
    path = O.build(native=True)
    oc = O.Ctx(LOG_N, [int(p) for p in primes[:45]], 0, libpath=path)
    rng = np.random.default_rng(1)
    x = np.stack([rng.integers(0, int(q), n, dtype=np.uint64) for q in primes[:45]]).reshape(-1)
    import ctypes as C
    ptr = x.ctypes.data_as(C.POINTER(C.c_uint64))

    checked = None
    if gpu_forward is not None:
        want = x.copy()
        oc.L.orc_set_threads(1)
        oc.L.orc_nwt_forward(oc.h, want.ctypes.data_as(C.POINTER(C.c_uint64)), 45, 0)
        got = gpu_forward(x.reshape(45, n))
        if not np.array_equal(got.reshape(-1), want):
            raise SystemExit("bench: the GPU forward NTT differs from the CPU restatement -- timing rejected")
        checked = "GPU forward NTT of the baseline's 45-limb input == CPU restatement, bit for bit"

    def run(threads, budget):
        oc.L.orc_set_threads(threads)
        oc.L.orc_nwt_forward(oc.h, ptr, 45, 0)
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < budget:
            oc.L.orc_nwt_forward(oc.h, ptr, 45, 0)
            reps += 1
        return reps, time.perf_counter() - t0

    reps, dt = run(1, seconds)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:   # a cgroup CPU quota caps what the visible cores can deliver
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    threads = min(cores, 45)
    reps_all, dt_all = run(threads, 3.0)
    oc.L.orc_set_threads(1)
    return {"value": 45 * reps / dt, "unit": "NTT/s", "cores": 1, "kind": "port",
            "sample": f"{reps} forward NTTs of 45 limbs at N=2^16 ({dt:.1f} s, oracle/oracle.c -O3 -march=native, 1 thread)",
            "host_cpus": cores, "checked": checked,
            "all_cores": {"value": 45 * reps_all / dt_all, "unit": "NTT/s", "cores": threads,
                          "sample": f"{reps_all} x 45 limbs, OpenMP over limbs ({threads} threads, one limb each), "
                                    f"{dt_all:.1f} s; cores = min(affinity, cgroup quota, 45)"}}


def load_record(path):
    """(parsed body, reference) of a committed OFFLINE record under profiles/ (rocprofv3 counters and kernel traces cannot be
    taken inside this process).  Only the REFERENCE -- file name, sha of the file, collection date -- and the scalars picked
    below go into the bench output: r05 copied the bodies into the line, which grew past what the driver's parser keeps."""
    try:
        raw = open(path, "rb").read()
        body = json.loads(raw)
        return body, {"file": os.path.relpath(path, ROOT), "sha16": hashlib.sha256(raw).hexdigest()[:16],
                      "collected": body.get("collected")}
    except (OSError, ValueError):
        return None, None


def stage_ref(path):
    """Reference to a per-stage table + the one row a reader needs first: the stage furthest below the 8 TB/s line."""
    body, ref = load_record(path)
    if body is None:
        return None
    rows = body.get("stages")
    if rows is None:      # the batched table holds one table per batch size; quote the best batch's
        best = next((b for b in body.get("batches", []) if b.get("batch") == body.get("best_batch")), None) or {}
        rows, ref["batch"] = best.get("stages", []), body.get("best_batch")
        ref["per_op_us"] = best.get("per_op_us_sum_of_kernels")
    else:
        ref["per_op_us"] = body.get("per_op_us_sum_of_kernels")
    fracs = [r["frac_of_8TBps"] for r in rows if isinstance(r.get("frac_of_8TBps"), (int, float))]
    ref["furthest_below_roofline"] = body.get("furthest_below_roofline")
    ref["frac"] = min(fracs) if fracs else None
    return ref


LINE_LIMIT = 8000     # bytes of the ONE stdout line (the r05 line was 20.3 KB and the driver's record had parsed: null)


def compact_line(full, full_path):
    """The ONE stdout line: the contract keys + `roofline` + `cpu_baseline` in full meaning, one scalar (or a handful) per
    extra leg.  Everything else -- sweeps, notes, per-step event statistics, calibration streams -- is in the full record
    written beside it (`full_record`)."""
    def pick(d, *keys):
        return None if d is None else {k: d[k] for k in keys if k in d and d[k] is not None}

    def rnd(x, nd=6):
        if isinstance(x, float):
            return float(f"{x:.{nd}g}")
        if isinstance(x, dict):
            return {k: rnd(v, nd) for k, v in x.items()}
        if isinstance(x, list):
            return [rnd(v, nd) for v in x]
        return x

    line = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                 "vs_baseline", "dtype", "data")}
    line["config"] = pick(full["config"], "workload", "N", "limbs", "special_limbs", "polynomials_per_step", "parallelism", "launch")
    r = full["roofline"]
    line["roofline"] = pick(r, "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_ratio", "traffic_source", "kernel",
                            "algorithmic_bytes_per_launch", "avg_launch_ms", "rmw_GBps", "copy_GBps", "ceiling_two_pass", "frac_of_ceiling")
    line["roofline"]["per_step_median_ms"] = (r.get("per_step_events") or {}).get("median_ms")
    line["roofline"]["sustained"] = pick(r.get("sustained"), "seconds", "steps", "median_ms_per_step", "min_ms_per_step", "max_ms_per_step",
                                         "value", "frac_of_peak", "gpu_busy_fraction")
    c = full.get("cpu_baseline")
    line["cpu_baseline"] = None if c is None else dict(pick(c, "value", "unit", "cores", "kind", "sample", "host_cpus", "checked"),
                                                       all_cores_value=(c.get("all_cores") or {}).get("value"),
                                                       all_cores=(c.get("all_cores") or {}).get("cores"))
    sp = full.get("single_polynomial") or {}
    line["single_polynomial_us"] = {k: 1e3 * v["mean_ms"] for k, v in sp.items() if v}
    hm = full.get("hommul_relin_rescale")
    if hm is not None:
        hb = hm["batched"]
        line["hommul_relin_rescale"] = {
            "value": hm["value"], "unit": hm["unit"], "ms_per_op": hm["ms_per_op"], "gpu_ms_per_op": hm["gpu_ms_per_op"]["median_ms"],
            "frac_of_peak": hm["frac_of_peak"], "algorithmic_bytes_per_op": hm["algorithmic_bytes_per_op"],
            "traffic_ratio": hm.get("traffic_ratio"), "stages": hm.get("stages"),
            "batched": {"value": hb["value"], "ms_per_op": hb["ms_per_op"], "batch": hb["batch"], "frac_of_peak": hb["frac_of_peak"],
                        "ms_per_op_by_batch": {str(e["batch"]): e["ms_per_op"] for e in hb["sweep"]},
                        "sustained_median_ms_per_op": (hb.get("sustained") or {}).get("median_ms_per_op"),
                        "traffic_ratio": hb.get("traffic_ratio"), "stages": hb.get("stages")}}
    line["keyswitch_c4"] = pick(full.get("keyswitch_c4"), "value", "unit", "batch", "ms_per_ciphertext", "per_rank_ciphertexts", "checksum",
                                "checked", "scaling")
    line["matvec_c5"] = pick(full.get("matvec_c5"), "value", "unit", "blocks", "ms_per_block", "per_rank_blocks", "checksum", "scaling",
                             "key_broadcast_s", "key_broadcast_calls")
    ex = full.get("next_rows")
    if ex is not None:
        line["next_rows"] = {"bfv_multiply": pick(ex["bfv_multiply"], "behz_multiply_ms", "hps_multiply_ms", "behz_frac_of_peak",
                                                  "hps_frac_of_peak"),
                             "modular_gemm": pick(ex["modular_gemm"], "us_per_batch", "frac_of_i8_mfma_peak")}
    rc = full.get("rccl")
    line["rccl"] = None if rc is None else {"backend": rc["backend"], "ranks": rc["ranks"],
                                            "devices": sorted({str(p.get("pci_bus_id")) for p in rc["per_rank"]})}
    for k in ("key_broadcast_calls", "key_broadcast_path", "collectives", "preflight"):
        line[k] = full.get(k)
    line["full_record"] = full_path
    text = json.dumps(rnd(line), separators=(",", ":"))
    if len(text) > LINE_LIMIT:      # never again a line the driver cannot keep: shed the optional legs, loudly
        for k in ("next_rows", "matvec_c5", "keyswitch_c4", "single_polynomial_us", "hommul_relin_rescale"):
            line[k] = "dropped: line over %d bytes, see full_record" % LINE_LIMIT
            text = json.dumps(rnd(line), separators=(",", ":"))
            if len(text) <= LINE_LIMIT:
                break
    assert len(text) <= LINE_LIMIT, f"bench line is {len(text)} bytes, limit {LINE_LIMIT}"
    return text


def preflight(args, P, pdist, torch, dist, dev, world, rank, share, force_dist, comm, emit=True):
    """First-contact check of a multi-GPU run: nothing in this repository has ever executed on more than one
    real RCCL rank, so the first 8-GPU run should fail HERE, with a reason, and not twelve GiB into the config-5 key broadcast.
    Checks, per rank and then agreed over the group: visible devices >= world; distinct devices (PCI bus id / uuid) behind the ranks;
    free device memory against the per-rank budget of the full bench (dominated by config 5: 12.1 GB of Galois keys + 4 GB of
    plaintext diagonals + 6 GB of outputs and scratch, next to the headline's 0.4 GB batch and the table replicas); the RCCL
    communicator (created by the caller: init_process_group + the all_gather above); a 64 MiB trial broadcast from rank 0 through
    BOTH key-broadcast paths -- dist.broadcast and pha_broadcast_keys on the process group's own ncclComm_t (_comm_ptr()) -- each
    checked word for word on every rank and timed.  Rank 0 prints ONE JSON line; the exit code is 0 only if every rank passed."""
    import warnings
    errors, notes = [], []
    group = world > 1 or force_dist
    n_vis = torch.cuda.device_count()
    if not share and n_vis < world:
        errors.append(f"{n_vis} HIP device(s) visible, {world} ranks")
    free_b, total_b = torch.cuda.mem_get_info(dev)
    need_b = int(26e9)   # see the docstring; measured peak of the full bench on one rank: 23.4 GB (torch.cuda.max_memory_allocated)
    if free_b < need_b:
        errors.append(f"rank {rank}: {free_b / 1e9:.1f} GB free on {dev}, the bench needs ~{need_b / 1e9:.0f} GB per rank (config 5)")
    ids = [(i.get("pci_bus_id"), i.get("uuid")) for i in (comm or {}).get("per_rank", [])]
    if group and not share and len(set(ids)) != len(ids):
        errors.append(f"ranks share a device: {ids}")
    trial = {}
    if group:
        words = (64 << 20) // 8
        want = torch.arange(words, dtype=torch.int64, device=dev) * 0x9E3779B97F4A7C15 % (1 << 62)   # the pattern every rank can form itself
        small_primes = [int(p) for p in P.coeff_modulus_create(4096, [50, 50, 60])]
        pctx = P.PhantomContext(12, small_primes, 1, device=dev)
        for path in ("dist.broadcast", "pha_broadcast_keys"):
            if share and path == "pha_broadcast_keys":
                trial[path] = {"skipped": "PHA_BENCH_SHARE_GPU: gloo group, no RCCL communicator"}
                continue
            buf = want.clone() if rank == 0 else torch.zeros(words, dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            try:
                with warnings.catch_warnings(record=True) as caught:
                    warnings.simplefilter("always")
                    if share:
                        host = buf.cpu()
                        calls = pdist.broadcast_keys([host], src=0, check_layout=False)
                        buf.copy_(host)
                    else:   # check_layout=False: the trial is timed, and one flat buffer has one layout
                        calls = pdist.broadcast_keys([buf], src=0, ctx=pctx, direct=(path == "pha_broadcast_keys"), check_layout=False)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                same = bool(torch.equal(buf, want))
                took = pdist.LAST_BROADCAST_PATH
                rec = {"ok": same and took == path, "path_taken": took, "calls": calls, "seconds": dt,
                       "GBps": words * 8 / dt / 1e9, "fallback": pdist.LAST_BROADCAST_FALLBACK}
                if not same:
                    errors.append(f"rank {rank}: the {path} trial delivered different words")
                if took != path:
                    errors.append(f"rank {rank}: asked for {path}, got {took}: {pdist.LAST_BROADCAST_FALLBACK}")
                trial[path] = rec
            except Exception as e:   # noqa: BLE001 -- a preflight reports, it does not crash
                errors.append(f"rank {rank}: {path} raised {type(e).__name__}: {e}")
                trial[path] = {"ok": False, "error": f"{type(e).__name__}: {e}"}
            del buf
    else:
        notes.append("one rank, no process group: nothing to broadcast (PHA_BENCH_FORCE_DIST=1 runs the RCCL checks on one device)")
    mine = {"rank": rank, "errors": errors, "free_GB": free_b / 1e9, "total_GB": total_b / 1e9, "trial_broadcast": trial}
    every = [mine]
    if group:
        every = [None] * dist.get_world_size()
        dist.all_gather_object(every, mine)
    ok = all(not e["errors"] for e in every)
    if group and dist.get_world_size() != world:     # the communicator itself must have seen every rank
        ok = False
        every[0]["errors"].append(f"the process group has {dist.get_world_size()} ranks, --gpus {world}")
    rec = {"preflight": True, "ok": ok, "n_gpus": world, "devices_visible": n_vis,
           "backend": (comm or {}).get("backend"), "rccl": comm, "need_GB_per_rank": need_b / 1e9,
           "per_rank": every, "notes": notes, "errors": [x for e in every for x in e["errors"]]}
    if rank == 0 and (emit or not ok):     # stand-alone: always one line; in front of a bench run: only the refusal
        print(json.dumps(rec), flush=True)
    return ok, rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the K steps from one hipGraph")
    ap.add_argument("--no-graph", action="store_true", help="enqueue the K steps one by one from Python")
    ap.add_argument("--only-ntt", action="store_true", help="skip the HomMul and config-4 / config-5 legs (profiling runs)")
    ap.add_argument("--no-c5", action="store_true", help="skip the config-5 leg (23 GB of Galois keys)")
    ap.add_argument("--sustain", type=float, default=6.0,
                    help="seconds of back-to-back headline steps AFTER the K timed ones (a sustained figure, and a GPU that the "
                         "driver's utilisation sampler can see); 0 = off")
    ap.add_argument("--full-out", default=os.path.join("gpurun_out", "bench_full.json"),
                    help="where the FULL record goes (sweeps, notes, calibration streams); stdout carries one compact line")
    ap.add_argument("--preflight", action="store_true",
                    help="first-contact check of an N-GPU run (devices, memory budget, RCCL init, a 64 MiB trial broadcast through "
                         "both key-broadcast paths); prints one JSON line and exits")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # before the HIP runtime starts: RCCL needs dmabuf IPC on this driver
    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench: --gpus {args.gpus} but WORLD_SIZE={world}")
    # test hook (1-GPU boxes): PHA_BENCH_SHARE_GPU=1 runs every rank on cuda:0 over gloo, to exercise the
    # multi-rank code path where RCCL cannot be used (it refuses two ranks on one device)
    share = os.environ.get("PHA_BENCH_SHARE_GPU") == "1"
    if world > 1 and not share and torch.cuda.device_count() < world:
        raise SystemExit(f"bench: --gpus {world} but only {torch.cuda.device_count()} HIP device(s) visible "
                         "(PHA_BENCH_SHARE_GPU=1 puts all ranks on cuda:0 over gloo: a functional test, not a measurement)")
    # PHA_BENCH_FORCE_DIST=1: initialise the process group (RCCL) even for one rank, so that the broadcast / all-reduce /
    # all-gather calls below really go through RCCL on a one-GPU box
    force_dist = os.environ.get("PHA_BENCH_FORCE_DIST") == "1" and world == 1
    if force_dist:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1 or force_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="gloo" if share else "nccl", init_method="env://", rank=rank, world_size=world)   # "nccl" is RCCL on ROCm
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # what the communicator itself reports (so that a multi-GPU record can show that RCCL saw N ranks on N devices)
    comm = None
    if world > 1 or force_dist:
        props = torch.cuda.get_device_properties(dev_index)
        mine_info = {"rank": dist.get_rank(), "device_index": dev_index, "device": props.name,
                     "pci_bus_id": getattr(props, "pci_bus_id", None), "uuid": str(getattr(props, "uuid", "")), "pid": os.getpid()}
        infos = [None] * dist.get_world_size()
        dist.all_gather_object(infos, mine_info)
        comm = {"backend": dist.get_backend(), "ranks": dist.get_world_size(), "per_rank": infos}
    small = os.environ.get("PHA_BENCH_SMALL") == "1"    # functional test of the multi-rank path: smaller batches

    import phantom_fhe_amd as P
    from phantom_fhe_amd import dist as pdist
    from phantom_fhe_amd import workloads as W
    # A/B experiments only: the knobs exist in the test-only library (PHA_LIB_OVERRIDE=.../libphantom_amd_exp.so), results never change
    if os.environ.get("PHA_NTT_VARIANT"):
        P.set_tuning(0, int(os.environ["PHA_NTT_VARIANT"]))
    for kv in filter(None, os.environ.get("PHA_TUNING", "").split(",")):   # "key=value,..."
        P.set_tuning(int(kv.split("=")[0]), int(kv.split("=")[1]))
    n = 1 << LOG_N
    primes = [int(p) for p in P.coeff_modulus_create(n, BITS)]
    size_q = len(primes) - SIZE_P
    red_dev = None if share else dev

    def barrier():
        torch.cuda.synchronize()
        if world > 1 or force_dist:
            dist.barrier()

    # r06: every multi-rank run starts with the first-contact checks (devices, memory, the communicator's own rank count, a trial
    # broadcast through both key paths) and REFUSES, with one JSON line and exit code 1, instead of timing a broken group
    preflight_rec = None
    if args.preflight or world > 1:
        ok, rec = preflight(args, P, pdist, torch, dist, dev, world, rank, share, force_dist, comm, emit=args.preflight)
        if args.preflight or not ok:
            if world > 1 or force_dist:
                dist.barrier()
                dist.destroy_process_group()
            sys.exit(0 if ok else 1)
        preflight_rec = {"ok": True, "ranks": world,
                         "trial_broadcast_GBps": {k: v.get("GBps") for k, v in rec["per_rank"][0]["trial_broadcast"].items()}}

    ctx = P.PhantomContext(LOG_N, primes, SIZE_P, device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x5EED0000 + 3 + rank)

    # ---- CPU baseline FIRST (r05): it needs idle host cores, not an idle GPU, and it is 13 of the run's ~30 s -- at the end it
    #      left the GPU idle for the driver's utilisation sampler; at the start the GPU legs run back to back behind it.  Rank 0
    #      times it (reported at N = 1 by contract; for N > 1 the other ranks wait at the barrier below, so that every rank ramps
    #      its clocks at the same moment afterwards).
    cpu_line = None
    if rank == 0 and not args.no_cpu_baseline:
        def gpu_forward(host_poly):   # the product path on the baseline's own input
            d = P.to_device(host_poly, dev)
            ctx.nwt_2d_radix8_forward_inplace(d, 45, 0)
            return P.to_host(d)
        cpu_line = cpu_baseline(primes, n, seconds=2.0 if small else 10.0, gpu_forward=gpu_forward)
    barrier()

    direct_bcast = os.environ.get("PHA_BCAST_DIRECT") == "1"   # pha_broadcast_keys on the process group's own RCCL communicator

    def key_slab(count, shape):
        """`count` key buffers of `shape` cut out of ONE allocation: the set travels as one flat buffer per collective call
        (phantom_fhe_amd/dist.py merges back-to-back views), not as one call per [2][#QP][N] tensor."""
        slab = torch.empty((count,) + tuple(shape), dtype=torch.int64, device=dev)
        return [slab[i] for i in range(count)]

    def broadcast(keys):
        """Returns the number of collective calls issued (0 without a process group)."""
        if share and world > 1:               # gloo moves host tensors
            host = torch.stack([k.cpu() for k in keys])
            calls = pdist.broadcast_keys([host[i] for i in range(len(keys))], src=0)
            for k, h in zip(keys, host):
                k.copy_(h)
            return calls
        return pdist.broadcast_keys(keys, src=0, ctx=ctx, direct=direct_bcast)  # one-time RCCL broadcast over xGMI; no collective on the data path

    def timed(fn, steps):
        """K calls between barrier + synchronize on both sides; whole-job time = the slowest rank's."""
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        barrier()
        return pdist.max_over_ranks(time.perf_counter() - t0, device=red_dev)

    def per_step_events(fn, steps):
        """GPU-side duration of each of `steps` calls, from event pairs on the launch stream (torch's current stream,
        which is the stream every pha_* call is enqueued on)."""
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        torch.cuda.synchronize()
        for a, b in evs:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in evs]
        return {"mean_ms": statistics.fmean(ms), "median_ms": statistics.median(ms), "min_ms": min(ms), "steps": steps}

    # ---- forward NTT: the timed headline (HBM-resident batch) --------------------------------------------------
    nb = 4 if small else NTT_BATCH
    polys = torch.stack([uniform_residues(primes[:size_q], n, dev, gen) for _ in range(nb)])
    poly_stride = size_q * n

    def ntt_step():
        ctx.nwt_2d_radix8_forward_inplace_batched(polys, size_q, 0, nb, poly_stride)

    for _ in range(args.warmup):
        ntt_step()
    torch.cuda.synchronize()
    # clock ramp: the part idles at a few hundred MHz, and W = 5 steps (1.6 ms) do not bring it up, so a short timed region
    # would mostly measure the ramp.  Untimed steps until 0.2 s have passed (on top of the W requested ones; reported).
    ramp_steps, t_ramp = 0, time.perf_counter()
    while not small and time.perf_counter() - t_ramp < 0.2:
        for _ in range(10):
            ntt_step()
        torch.cuda.synchronize()
        ramp_steps += 10
    # The K timed steps are enqueued by ONE library call (pha_repeat_forward_ntt_batched: 2 K kernel launches from C, no
    # per-step host work); --graph replays them from a hipGraph instead (its replay costs ~0.6 ms of submission on this
    # stack, which a short K would mostly measure), --no-graph enqueues them step by step from Python.
    graph = None
    if args.graph:
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                for _ in range(args.steps):
                    ntt_step()
        torch.cuda.current_stream().wait_stream(side)
        g.replay()                   # one untimed replay (warm instantiation)
        torch.cuda.synchronize()
        graph = g
    barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()                       # same stream the launches go to (torch's current stream)
    if graph is not None:
        graph.replay()
    elif args.no_graph:
        for _ in range(args.steps):
            ntt_step()
    else:
        ctx.repeat_forward_ntt_batched(polys, size_q, 0, nb, poly_stride, args.steps)
    e1.record()
    barrier()
    elapsed = pdist.max_over_ranks(time.perf_counter() - t0, device=red_dev)
    kernel_ms = e0.elapsed_time(e1) / args.steps      # average duration of one launch pair inside the timed region
    ntt_per_s = world * args.steps * nb * size_q / elapsed
    step_stats = per_step_events(ntt_step, max(100, args.steps) if not small else 10)

    # ---- sustained (r05): the same step back to back for --sustain seconds, in chunks of 100 steps enqueued by one library call
    #      each, every chunk bracketed by events on the launch stream.  Reported beside the K timed steps (it is NOT `value`):
    #      median / min / max per-step time over the chunks show what the part holds once clocks and temperature have settled.
    sustained = None
    sustain_s = 0.5 if small and args.sustain > 0 else args.sustain
    if sustain_s > 0:
        chunk = 20 if small else 100
        chunk_ms = []
        t_s = time.perf_counter()
        while time.perf_counter() - t_s < sustain_s:
            a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a_.record()
            ctx.repeat_forward_ntt_batched(polys, size_q, 0, nb, poly_stride, chunk)
            b_.record()
            b_.synchronize()
            chunk_ms.append(a_.elapsed_time(b_) / chunk)
        wall_s = time.perf_counter() - t_s
        med = statistics.median(chunk_ms)
        sustained = {"seconds": wall_s, "steps": chunk * len(chunk_ms), "steps_per_chunk": chunk,
                     "median_ms_per_step": med, "min_ms_per_step": min(chunk_ms), "max_ms_per_step": max(chunk_ms),
                     "first_chunk_ms_per_step": chunk_ms[0], "last_chunk_ms_per_step": chunk_ms[-1],
                     "value": nb * size_q / (med * 1e-3), "unit": "NTT/s (this rank, median chunk)",
                     "frac_of_peak": 16.0 * n * size_q * nb / (med * 1e-3) / PEAK_HBM,
                     "gpu_busy_fraction": sum(chunk_ms) * chunk * 1e-3 / wall_s}

    # ---- informational: one 45-limb polynomial per launch pair (the r01 headline protocol) ------------------
    def single_inplace():
        ctx.nwt_2d_radix8_forward_inplace(polys[0], size_q, 0)

    rot = [0]

    def single_rotating():
        ctx.nwt_2d_radix8_forward_inplace(polys[rot[0] % nb], size_q, 0)
        rot[0] += 1

    single_steps = 20 if small else max(100, args.steps)
    for _ in range(5):
        single_inplace()
    mall_stats = per_step_events(single_inplace, single_steps)
    for _ in range(nb):
        single_rotating()
    rot_stats = per_step_events(single_rotating, single_steps)

    # device-to-device copy of 512 MiB (read + write), the calibrated counterpart of the nominal 8 TB/s
    cal_a = torch.empty(64 << 20, dtype=torch.int64, device=dev)
    cal_b = torch.empty_like(cal_a)
    for _ in range(3):
        cal_b.copy_(cal_a)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(10):
        cal_b.copy_(cal_a)
    ev1.record()
    torch.cuda.synchronize()
    copy_bps = 10 * 2 * cal_a.numel() * 8 / (ev0.elapsed_time(ev1) * 1e-3)
    own_copy_bps = P.stream_copy_rate(cal_b, cal_a, 10)    # the library's own 16-byte-per-lane streaming kernel (nontemporal copy)
    # r04: the independent calibration (tools/stream_calib.hip in library form): read-only, write-only, copy and in-place
    # read-modify-write over 512 MiB, default and nontemporal policy; the best of each kind is what a pass can be held against
    cal = {}
    for name, mode in (("copy", 0), ("read", 1), ("write", 2), ("rmw", 3)):
        cal[name + "_GBps"] = max(P.stream_rate(cal_b, cal_a, mode, nt, 10) for nt in (False, True)) / 1e9
    del cal_a, cal_b
    del polys

    hm = None
    c4 = None
    bcast_calls = {}
    if not args.only_ntt:
        # ---- evaluation key: generated on rank 0, broadcast once over RCCL/xGMI --------
        dnum = size_q // SIZE_P
        evk = key_slab(dnum, (2, len(primes), n))
        if rank == 0:
            for k in evk:
                k[0] = uniform_residues(primes, n, dev, gen)
                k[1] = uniform_residues(primes, n, dev, gen)
        bcast_calls["evk_c3"] = broadcast(evk)
        rlk = P.PhantomRelinKey(evk)

        # ---- HomMul + relinearize + rescale (secondary figure, same parameter set) -------------------------
        ct1 = torch.stack([uniform_residues(primes[:size_q], n, dev, gen) for _ in range(2)])
        ct2 = torch.stack([uniform_residues(primes[:size_q], n, dev, gen) for _ in range(2)])
        buf = torch.zeros((3, size_q, n), dtype=torch.int64, device=dev)
        out = torch.zeros((2, size_q - 1, n), dtype=torch.int64, device=dev)

        def hommul_two_calls():   # the reference's sequence of launchers
            ctx.tensor_prod_2x2_rns_poly(ct1, ct2, buf, size_q)                              # multiply: (ct1, ct2) -> 3 polynomials
            ctx.keyswitch_inplace(size_q, buf, buf[2], rlk.public_keys_ptr, P.scheme_type.ckks)  # relinearize
            ctx.divide_and_round_q_last_ntt(size_q, buf, 2, out)                             # rescale_to_next

        def hommul():             # same result bit for bit (tests/test_gpu_rns.py), key switch + rescale as one entry point
            ctx.tensor_prod_2x2_rns_poly(ct1, ct2, buf, size_q)
            ctx.keyswitch_rescale(size_q, buf, buf[2], rlk.public_keys_ptr, out)

        hm_steps = 3 if small else max(20, args.steps // 2)
        for _ in range(3):
            hommul_two_calls()
        two_stats = per_step_events(hommul_two_calls, hm_steps)
        two_sum = int(out.sum().item())
        for _ in range(3):
            hommul()
        if int(out.sum().item()) != two_sum:
            raise SystemExit("bench: pha_keyswitch_rescale differs from keyswitch_inplace + divide_and_round_q_last_ntt")
        hm_elapsed = timed(hommul, hm_steps)
        hm_stats = per_step_events(hommul, hm_steps)

        # the same operation on a batch of B ciphertext pairs through the batched entry points (one set of launches:
        # key limbs read once, NTT / base-conversion launches B times larger).
        # r04: a sweep over B; the best batch is the figure quoted as sustained HomMul + relinearize + rescale / s
        sweep_B = [2] if small else [int(b) for b in os.environ.get("PHA_BENCH_BATCHES", "1,2,4,8,16,32").split(",")]
        Bmax = max(sweep_B)
        bt1 = torch.stack([torch.stack([uniform_residues(primes[:size_q], n, dev, gen) for _ in range(2)]) for _ in range(min(Bmax, 8))])
        bt2 = torch.stack([torch.stack([uniform_residues(primes[:size_q], n, dev, gen) for _ in range(2)]) for _ in range(min(Bmax, 8))])
        if Bmax > 8:   # the arithmetic is data-independent: the larger batches repeat the 8 seeded pairs
            bt1 = bt1.repeat((Bmax + 7) // 8, 1, 1, 1)[:Bmax].contiguous()
            bt2 = bt2.repeat((Bmax + 7) // 8, 1, 1, 1)[:Bmax].contiguous()
        b01 = torch.zeros_like(bt1)
        b2 = torch.zeros((Bmax, size_q, n), dtype=torch.int64, device=dev)
        bout = torch.zeros((Bmax, 2, size_q - 1, n), dtype=torch.int64, device=dev)
        batch_sweep = []
        for B in sweep_B:
            def hommul_batched():
                ctx.tensor_prod_2x2_batched(bt1, bt2, b01, b2, size_q, B)
                ctx.keyswitch_rescale_batched(size_q, b01, b2, B, rlk.public_keys_ptr, bout)

            hb_steps = 2 if small else max(3, min(10, 64 // B))
            for _ in range(2):
                hommul_batched()
            el = timed(hommul_batched, hb_steps)
            batch_sweep.append({"batch": B, "ms_per_op": 1e3 * el / (B * hb_steps), "ops_per_s": world * B * hb_steps / el})
        best = min(batch_sweep, key=lambda e: e["ms_per_op"])
        B, hb_steps, hm_batched_elapsed = best["batch"], 1, best["ms_per_op"] * 1e-3 * best["batch"]
        fixed = next((e for e in batch_sweep if e["batch"] == 8), None)   # the fixed-batch figure, comparable across rounds
        # r05: the best batch again, back to back for ~--sustain / 2 seconds (each call = one op set of B ciphertext pairs between
        # events): the sustained HomMul + relinearize + rescale rate; the sweep above times 3-10 calls per batch size
        hm_sustained = None
        if sustain_s > 0:
            def hommul_best():
                ctx.tensor_prod_2x2_batched(bt1, bt2, b01, b2, size_q, B)
                ctx.keyswitch_rescale_batched(size_q, b01, b2, B, rlk.public_keys_ptr, bout)
            per_call = []
            t_s = time.perf_counter()
            while time.perf_counter() - t_s < sustain_s / 2:
                a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a_.record()
                for _ in range(4):
                    hommul_best()
                b_.record()
                b_.synchronize()
                per_call.append(a_.elapsed_time(b_) / (4 * B))
            med = statistics.median(per_call)
            hm_sustained = {"batch": B, "seconds": time.perf_counter() - t_s, "ops": 4 * B * len(per_call),
                            "median_ms_per_op": med, "min_ms_per_op": min(per_call), "max_ms_per_op": max(per_call),
                            "value": 1e3 / med, "unit": "ops/s (this rank, median)", "frac_of_peak": 929.0 * (1 << 20) / (med * 1e-3) / PEAK_HBM}
        # minimal per-stage traffic of one HomMul + relinearize + rescale at C3: 929 MiB
        hm_alg_bytes = 929.0 * (1 << 20)
        hm = {"value": world * hm_steps / hm_elapsed, "unit": "ops/s", "ms_per_op": 1e3 * hm_elapsed / hm_steps,
              "steps": hm_steps, "gpu_ms_per_op": hm_stats, "algorithmic_bytes_per_op": hm_alg_bytes,
              "entry_points": "pha_tensor_prod_2x2_rns_poly + pha_keyswitch_rescale (key switch and rescale fused: one forward NTT "
                              "for mod-down + rescale; output checked equal to the three reference launchers in this run)",
              "three_launcher_sequence_gpu_ms_per_op": two_stats,
              "frac_of_peak": hm_alg_bytes / (hm_elapsed / hm_steps) / PEAK_HBM,
              "batched": {"value": world * B * hb_steps / hm_batched_elapsed, "unit": "ops/s",
                          "ms_per_op": 1e3 * hm_batched_elapsed / (B * hb_steps), "batch": B,
                          "frac_of_peak": hm_alg_bytes / (hm_batched_elapsed / (B * hb_steps)) / PEAK_HBM,
                          "sweep": batch_sweep,
                          "fixed_batch_8": fixed,
                          "sustained": hm_sustained,
                          "note": "pha_tensor_prod_2x2_batched + pha_keyswitch_rescale_batched; value / ms_per_op = the best batch of "
                                  "the sweep (3-10 timed calls per batch size); fixed_batch_8 = the B = 8 row of the same sweep (the "
                                  "figure comparable across rounds); sustained = the best batch back to back for seconds"}}
        del ct1, ct2, buf, out, bt1, bt2, b01, b2, bout, rlk, evk

        # ---- BASELINE config 4: BFV relinearize + Galois rotate, N = 2^15, 30 + 15 limbs, batch 64 over the ranks ----
        n4 = 1 << C4_LOG_N
        primes4 = [int(p) for p in P.coeff_modulus_create(n4, C4_BITS)]
        q4 = len(primes4) - SIZE_P
        ctx4 = P.PhantomContext(C4_LOG_N, primes4, SIZE_P, device=dev)
        batch4 = 6 if small else C4_BATCH
        kgen = torch.Generator(device=dev)
        kgen.manual_seed(0x5EED0000 + 4)
        keys4 = key_slab(2 * (q4 // SIZE_P), (2, len(primes4), n4))
        if rank == 0:
            for k in keys4:
                k[0] = uniform_residues(primes4, n4, dev, kgen)
                k[1] = uniform_residues(primes4, n4, dev, kgen)
        bcast_calls["keys_c4"] = broadcast(keys4)           # relin key + one Galois key, RCCL broadcast from rank 0
        rlk4 = P.PhantomRelinKey(keys4[: q4 // SIZE_P])
        glk4 = P.PhantomRelinKey(keys4[q4 // SIZE_P:])
        mine = pdist.shard_range(batch4, rank, world)
        # ciphertext b is generated from its own seed, so the inputs (and the checksum) do not depend on the world size
        ct3 = torch.empty((len(mine), 3, q4, n4), dtype=torch.int64, device=dev)
        for i, b in enumerate(mine):
            kgen.manual_seed(0x5EED4000 + b)
            for p_ in range(3):
                ct3[i, p_] = uniform_residues(primes4[:q4], n4, dev, kgen)
        elt = 3
        # the output buffer exists before the clock starts (W.relinearize_rotate_batch would allocate 1 GiB per call: a fresh
        # block inside the timed region, which on some boxes of the pool halves the figure)
        out4 = torch.empty((len(mine), 2, q4, n4), dtype=torch.int64, device=dev)
        res = [out4]

        def c4_step():
            if len(mine):
                ctx4.relinearize_rotate_batched(q4, ct3, len(mine), rlk4.public_keys_ptr, glk4.public_keys_ptr, elt, P.scheme_type.bfv,
                                                out4, 0)

        c4_step()
        c4_step()
        c4_steps = 1 if small else 3
        c4_elapsed = timed(c4_step, c4_steps)
        local_sum = int(res[0].sum().item()) & ((1 << 64) - 1) if len(mine) else 0
        sums = pdist.gather_checksums(local_sum - (1 << 64) if local_sum >= (1 << 63) else local_sum, device=red_dev)
        # r04: the leg proves its own parity -- two of this rank's ciphertexts (first and last of the shard) against the oracle's
        # composition of the restated reference steps (relinearize: keyswitch of c2 into (c0, c1); rotate: automorphism in the
        # coefficient domain, key switch of the rotated c1), outside the timed region; the oracle is the checker, never the path
        c4_checked = None
        if not args.no_cpu_baseline:
            import numpy as np
            from oracle import oracle as O
            oc4 = O.Ctx(C4_LOG_N, primes4, SIZE_P)
            tool4 = O.Tool(oc4, q4)
            h_keys = [P.to_host(k) for k in keys4]
            half4 = q4 // SIZE_P
            n_checked = 0
            for i in (sorted({0, len(mine) - 1}) if len(mine) else []):
                x = P.to_host(ct3[i])
                ctk = tool4.keyswitch_inplace(x[:2], x[2], [h_keys[j] for j in range(tool4.beta)], O.BFV)
                g_ = [oc4.apply_galois_coeff(ctk[p_], elt, q4) for p_ in range(2)]
                want4 = tool4.keyswitch_inplace(np.stack([g_[0], np.zeros_like(g_[0])]), g_[1],
                                                [h_keys[half4 + j] for j in range(tool4.beta)], O.BFV)
                if not np.array_equal(P.to_host(out4[i]), want4):
                    raise SystemExit(f"bench: config-4 ciphertext {mine[i]} differs from the oracle's composition")
                n_checked += 1
            all_checked = pdist.gather_checksums(n_checked, device=red_dev)
            c4_checked = f"{sum(all_checked)} ciphertexts == oracle (first and last of every rank's shard, bit for bit)"
            del h_keys, oc4, tool4
        c4 = {"value": batch4 * c4_steps / c4_elapsed, "unit": "relinearize+rotate ciphertexts/s (whole job)",
              "checked": c4_checked,
              "batch": batch4, "scaling": "strong", "ms_per_ciphertext": 1e3 * c4_elapsed / (batch4 * c4_steps),
              "per_rank_ciphertexts": [len(pdist.shard_range(batch4, r, world)) for r in range(world)],
              "checksum": f"{sum(sums) & ((1 << 64) - 1):016x}",
              "checksum_note": "sum mod 2^64 of all output words of the 64 ciphertexts; identical for every --gpus",
              "config": "BFV N=2^15, 30 data + 15 special limbs (keyswitch_bench.cu:25-34), keys broadcast from rank 0"}

    extras = None
    if not args.only_ntt:
        # ---- SURVEY 8(f) rows measured beside the headline (this rank's GPU only; no collectives): BFV multiply at the config-4 chain
        #      and the reference's matmul_bench shape (30 x 256^3 modular GEMM, 50-bit moduli) ----
        def local_ms(fn, reps):
            fn()
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize(dev)
            return e0.elapsed_time(e1) / reps

        xg = torch.Generator(device=dev)
        xg.manual_seed(0x5EED0000 + 6)
        ctx4.set_plain_modulus(1032193)
        m1 = torch.stack([uniform_residues(primes4[:q4], n4, dev, xg) for _ in range(2)])
        m2 = torch.stack([uniform_residues(primes4[:q4], n4, dev, xg) for _ in range(2)])
        prod = torch.zeros((3, q4, n4), dtype=torch.int64, device=dev)
        reps = 3 if small else 20
        bfv = {}
        for name, fn in (("behz", ctx4.bfv_multiply_behz), ("hps", ctx4.bfv_multiply_hps)):
            mul = local_ms(lambda: fn(m1, m2, prod), reps)

            def mul_relin():
                fn(m1, m2, prod)
                ctx4.keyswitch_inplace(q4, prod[:2], prod[2], rlk4.public_keys_ptr, P.scheme_type.bfv)
            bfv[name + "_multiply_ms"] = mul
            bfv[name + "_multiply_relinearize_ms"] = local_ms(mul_relin, reps)
        bfv["config"] = ("BFV N=2^15, 30 data limbs (+15 special for the relinearization), 61-bit auxiliary bases; bfv_multiply_behz / "
                         "bfv_multiply_hps, src/evaluate.cu:447-548, :674-818")
        # r06: the rows' roofline figure.  Algorithmic bytes = the minimal per-stage traffic summed over the stages of
        # the reference's own call sequence, in limb-polynomials of N x 8 B (every stage reads its inputs once and writes its outputs once):
        #   BEHZ (src/evaluate.cu:404-548), Q limbs, |Bsk| = Q + 2: per input polynomial NTT over q (2Q), q -> Bsk u {m_tilde} (Q + Bsk + 1),
        #     sm_mrq (2 Bsk + 1), NTT over Bsk (2 Bsk) -- x 4 polynomials; tensor 7 (Q + Bsk); inverse NTTs 3 x 2 (Q + Bsk); fast_floor
        #     3 (Q + 2 Bsk); fastbconv_sk 3 (Bsk + Q)
        #   HPS (:674-818), |R| = Q + 1: per input polynomial bConv_HPS q -> R (Q + R), NTT over Q u R (2 (Q + R)) -- x 4; tensor 7 (Q + R);
        #     inverse 3 x 2 (Q + R); scaleAndRound_HPS_QR_R 3 (Q + 2 R); bConv_HPS R -> Q 3 (R + Q)
        nq, nbsk, nr = q4, q4 + 2, q4 + 1
        limb_bytes = 8.0 * n4
        behz_units = 4 * (2 * nq + (nq + nbsk + 1) + (2 * nbsk + 1) + 2 * nbsk) + 7 * (nq + nbsk) + 6 * (nq + nbsk) + 3 * (nq + 2 * nbsk) + 3 * (nbsk + nq)
        hps_units = 4 * ((nq + nr) + 2 * (nq + nr)) + 7 * (nq + nr) + 6 * (nq + nr) + 3 * (nq + 2 * nr) + 3 * (nr + nq)
        for name, units in (("behz", behz_units), ("hps", hps_units)):
            bfv[name + "_algorithmic_bytes"] = units * limb_bytes
            bfv[name + "_frac_of_peak"] = units * limb_bytes / (bfv[name + "_multiply_ms"] * 1e-3) / PEAK_HBM
        gm = gn = gk = 256
        gbatch = 4 if small else 30
        gprimes = [int(p) for p in P.coeff_modulus_create(4096, [50] * gbatch)]
        gctx = P.PhantomContext(12, gprimes, 0, device=dev)
        ga = torch.stack([torch.randint(0, q, (gm, gk), generator=xg, device=dev, dtype=torch.int64) for q in gprimes])
        gb = torch.stack([torch.randint(0, q, (gk, gn), generator=xg, device=dev, dtype=torch.int64) for q in gprimes])
        gc = torch.zeros((gbatch, gm, gn), dtype=torch.int64, device=dev)
        gemm_ms = local_ms(lambda: gctx.batched_modular_gemm(gc, ga, gb, gm, gn, gk, gbatch), 5 if small else 50)
        z, r_, c_ = gbatch - 1, 17, 203                    # one entry against Python integers
        want = sum(int(x) * int(y) for x, y in zip(ga[z, r_].tolist(), gb[z, :, c_].tolist())) % gprimes[z]
        if int(gc[z, r_, c_].item()) != want:
            raise RuntimeError("modular GEMM: the checked entry differs from the integer dot product")
        i8_macs = gbatch * gm * gn * gk * 49.0             # 7 x 7 signed-byte digit products per modular multiply-add
        extras = {"bfv_multiply": bfv,
                  "modular_gemm": {"us_per_batch": 1e3 * gemm_ms, "batch": gbatch, "m": gm, "n": gn, "k": gk, "modulus_bits": 50,
                                   "modular_mac_per_s": gbatch * gm * gn * gk / (gemm_ms * 1e-3),
                                   "i8_mac_per_s": i8_macs / (gemm_ms * 1e-3),
                                   "frac_of_i8_mfma_peak": 2.0 * i8_macs / (gemm_ms * 1e-3) / 5.0e15,
                                   "peak_note": "5.0 PF/s dense i8 MFMA (MI355X_MICROARCH.md; 4.4 measured); one entry checked against "
                                                "Python integers in this run",
                                   "config": "benchmark/matmul_bench.cu:545-673 shape: 256^3 per modulus, 50-bit moduli; operands as "
                                             "signed-byte digit planes through v_mfma_i32_32x32x32_i8 (DESIGN 4.9)"}}
        del gctx, ga, gb, gc, m1, m2, prod

    c5 = None
    if not args.only_ntt and not args.no_c5:
        # ---- BASELINE config 5: encrypted 128-slot matrix-vector product in diagonal form at the C3 set: C5_BLOCKS row blocks
        #      (split over the ranks), each one 128-diagonal block = 127 hoisted rotations + the main diagonal behind ONE mod-up
        #      and ONE mod-down (pha_hoisting_weighted); the 127 Galois keys are generated on rank 0 and broadcast once ----
        del ctx4
        torch.cuda.empty_cache()
        nbaby, ngiant = (4, 2) if small else (C5_BABY, C5_GIANT)
        n_diag = nbaby * ngiant
        n_blocks = 3 if small else C5_BLOCKS
        dnum5 = size_q // SIZE_P
        kg = torch.Generator(device=dev)
        kg.manual_seed(0x5EED0000 + 5)

        def below_every_prime(shape, g):   # residues below 2^49 < every prime of the set (the arithmetic is data-independent)
            return torch.randint(0, 1 << 49, shape, dtype=torch.int64, device=dev, generator=g)

        # diagonal k = i * nbaby + j: baby step j (rotation by j slots), giant step i (rotation by i * nbaby slots); element 5^r
        baby_elts = [pow(5, j, 2 * n) for j in range(nbaby)]
        giant_elts = [pow(5, nbaby * i, 2 * n) for i in range(ngiant)]
        n_keys = nbaby - 1 + ngiant - 1
        flat5 = key_slab(n_keys * dnum5, (2, len(primes), n))    # ONE allocation for the whole Galois key set
        gkeys = [flat5[i * dnum5:(i + 1) * dnum5] for i in range(n_keys)]
        if rank == 0:
            for key in gkeys:
                for d in key:
                    d.copy_(below_every_prime((2, len(primes), n), kg))
        t_b0 = time.perf_counter()
        bcast_calls5 = broadcast([d for key in gkeys for d in key])   # one-time RCCL broadcast of the Galois keys from rank 0
        torch.cuda.synchronize()
        bcast_s = time.perf_counter() - t_b0
        glks = [P.PhantomRelinKey(key) for key in gkeys]
        baby_keys = [None] + glks[:nbaby - 1]
        giant_keys = [None] + glks[nbaby - 1:]
        kg.manual_seed(0x5EED5000)
        ct5 = below_every_prime((2, size_q, n), kg)        # the same input ciphertext on every rank
        mine5 = pdist.shard_range(n_blocks, rank, world)
        # the plaintext diagonals: a pool of n_diag encoded diagonals (3.75 GiB at the C3 set), block b takes them rotated by b -- every
        # (block, giant, baby) triple of one launch still streams its own tensor, the inputs do not depend on the world size, and
        # 32 blocks do not need 120 GiB of synthetic plaintexts
        kg.manual_seed(0x5EED5100)
        pool5 = [below_every_prime((size_q + SIZE_P, n), kg) for _ in range(n_diag)]
        blocks5 = [[[pool5[(i * nbaby + j + b) % n_diag] for j in range(nbaby)] for i in range(ngiant)] for b in mine5]
        res5 = [None]

        def c5_step():   # this rank's row blocks against the one ciphertext; the baby keys are streamed once per 16 / ngiant blocks
            res5[0] = W.diag_matvec_bsgs_blocks(ctx, size_q, ct5, baby_elts, baby_keys, giant_elts, giant_keys, blocks5,
                                                P.scheme_type.ckks) if blocks5 else []

        c5_step()
        c5_step()     # twice: the wrapper allocates its 1.5 GiB output while the previous one is still held -- both blocks exist now
        c5_steps = 1 if small else 3
        c5_elapsed = timed(c5_step, c5_steps)
        local5 = int(res5[0].sum().item()) & ((1 << 64) - 1) if blocks5 else 0
        sums5 = pdist.gather_checksums(local5 - (1 << 64) if local5 >= (1 << 63) else local5, device=red_dev)
        key_bytes = n_keys * dnum5 * 2 * len(primes) * n * 8
        c5 = {"value": n_blocks * c5_steps / c5_elapsed, "unit": "128-diagonal blocks/s (whole job)", "blocks": n_blocks,
              "diagonals_per_block": n_diag, "baby_steps": nbaby, "giant_steps": ngiant, "scaling": "strong",
              "ms_per_block": 1e3 * c5_elapsed / (c5_steps * max(len(pdist.shard_range(n_blocks, r, world)) for r in range(world))),
              "per_rank_blocks": [len(pdist.shard_range(n_blocks, r, world)) for r in range(world)],
              "galois_keys": n_keys, "galois_key_bytes": key_bytes,
              "key_broadcast_s": bcast_s if (world > 1 or force_dist) else None,
              "key_broadcast_calls": bcast_calls5 if (world > 1 or force_dist) else None,
              "checksum": f"{sum(sums5) & ((1 << 64) - 1):016x}",
              "checksum_note": "sum mod 2^64 of all output words of the row blocks; identical for every --gpus",
              "config": f"CKKS N=2^16, 45 + 15 limbs; out_b = sum_i rot_(nb i)(sum_j diag_(b, nb i + j) (.) rot_j(ct)), nb = {nbaby}: baby-step / "
                        f"giant-step with double hoisting (pha_hoisting_weighted_bsgs), {n_keys} Galois keys instead of {n_diag - 1}; no reference counterpart "
                        "(SURVEY 8(0) row C5; building blocks src/evaluate.cu:1670-1866, :1297-1340)"}
        del gkeys, glks, blocks5, res5, pool5, flat5

    if rank == 0:
        alg_bytes = 16.0 * n * size_q * nb             # 8 B read + 8 B write per coefficient
        achieved = alg_bytes / (kernel_ms * 1e-3)
        one = 16.0 * n * size_q
        traffic, traffic_ref = load_record(TRAFFIC_FILE)
        traffic = traffic or {}
        # two in-place passes: every coefficient is read and written twice, so the algorithmic rate cannot exceed half the rate of an
        # in-place read-modify-write stream (r04: measured with the calibration kernels, not with the r01-r03 grid-stride copy)
        ceiling = cal["rmw_GBps"] * 1e9 / 2.0 / PEAK_HBM
        full = {
            "metric": "forward NTT limb-transforms/s at N=2^16, 45 RNS moduli",
            "value": ntt_per_s, "unit": "NTT/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"CKKS N=2^16, 45 data limbs (+15 special), forward NTT of a batch of {nb} ciphertext "
                                   f"polynomials per step ({nb * size_q} limb-transforms, {nb * size_q * n * 8 >> 20} MiB, "
                                   "HBM-resident; configs[2] parameter set)",
                       "N": n, "limbs": size_q, "special_limbs": SIZE_P, "polynomials_per_step": nb,
                       "parallelism": f"ciphertext-batch x{world}",
                       "launch": ("hipGraph replay of the K steps" if graph is not None else "K steps enqueued one by one from Python"
                                  if args.no_graph else "K steps enqueued by one library call (2 K eager kernel launches)"),
                       "untimed_clock_ramp_steps": ramp_steps},
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s",
                         "frac": achieved / PEAK_HBM,
                         "traffic": traffic.get("ntt_batched_bytes_per_launch"),
                         "traffic_ratio": (traffic["ntt_batched_bytes_per_launch"] / alg_bytes * (NTT_BATCH / nb)
                                           if traffic.get("ntt_batched_bytes_per_launch") else None),
                         "traffic_source": traffic_ref,   # OFFLINE: PMC FETCH_SIZE x 2 + WRITE_SIZE of the same step (tools/traffic.sh)
                         "kernel": "ntt_pass_kernel (strided pass, 64 x 32 tiles) + ntt_zloop_kernel (contiguous pass, 1024-point rows, twiddles resident "
                      "across the polynomials of the batch)",
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": kernel_ms,
                         "per_step_events": step_stats,
                         "calibrated_copy_GBps": own_copy_bps / 1e9,
                         "read_GBps": cal["read_GBps"], "write_GBps": cal["write_GBps"], "copy_GBps": cal["copy_GBps"],
                         "rmw_GBps": cal["rmw_GBps"],
                         "calibrated_note": "512 MiB per operand, one 16-byte word per lane, one trip per thread (pha_time_stream; the "
                                            "form of tools/stream_calib.hip that reaches the guide's float4-copy rate), best of default / "
                                            "nontemporal policy, (bytes read + bytes written) / time, this run; rmw = in-place "
                                            "read-modify-write, what an in-place pass does",
                         "sustained": sustained,
                         "kernel_memory_floor_ms_offline": 0.239,
                         "kernel_memory_floor_note": "OFFLINE figure, not measured in this run (profiles/r04j_bygrid_nocompute.csv, round 4): the two pass kernels with their butterflies compiled out (-DPHA_X_NOCOMPUTE build, "
                                                     "profiles/r04j_bygrid_nocompute.csv): strided 110 + contiguous 129 us per step = "
                                                     "0.395 of 8 TB/s; the FP64 work of the same step is ~84 operations per coefficient "
                                                     "= 131 us of pure issue at the 1.85 GHz the part holds under this load, so the "
                                                     "passes are issue- and memory-bound at once (profiles/r04_experiments.md)",
                         "torch_copy_GBps": copy_bps / 1e9, "guide_copy_GBps": 6290.0,
                         "ceiling_two_pass": ceiling, "frac_of_ceiling": achieved / PEAK_HBM / ceiling,
                         "ceiling_two_pass_copy_rate": cal["copy_GBps"] * 1e9 / 2.0 / PEAK_HBM,
                         "ceiling_two_pass_guide": 6.29e12 / 2.0 / PEAK_HBM,
                         "frac_of_ceiling_guide": achieved / PEAK_HBM / (6.29e12 / 2.0 / PEAK_HBM),
                         "ceiling_note": "a two-pass transform moves every coefficient through the fabric twice in each "
                                         "direction: algorithmic rate <= measured in-place read-modify-write rate / 2; N = 2^16 (512 "
                                         "KiB per limb) does not fit one CU's 160 KiB LDS, so no single-pass plan exists for it "
                                         "(DESIGN 4.1)"},
            "single_polynomial": {
                "mall_resident": dict(mall_stats, value=size_q / (mall_stats["mean_ms"] * 1e-3), unit="NTT/s (this rank)",
                                      frac_of_peak=one / (mall_stats["mean_ms"] * 1e-3) / PEAK_HBM,
                                      note="one 45-limb polynomial transformed in place again and again (22.5 MiB: "
                                           "stays in the 256 MiB MALL); the r01 headline protocol"),
                "hbm_resident": dict(rot_stats, value=size_q / (rot_stats["mean_ms"] * 1e-3), unit="NTT/s (this rank)",
                                     frac_of_peak=one / (rot_stats["mean_ms"] * 1e-3) / PEAK_HBM,
                                     note=f"one 45-limb polynomial per launch pair, rotating over {nb} buffers")},
            "hommul_relin_rescale": hm,
            "keyswitch_c4": c4,
            "matvec_c5": c5,
            "next_rows": extras,
            "rccl": comm,
            "key_broadcast_calls": (bcast_calls if (world > 1 or force_dist) and not args.only_ntt else None),
            "key_broadcast_path": pdist.LAST_BROADCAST_PATH,
            "collectives": ("RCCL (torch.distributed backend nccl)" if (world > 1 or force_dist) and not share else
                            "gloo (PHA_BENCH_SHARE_GPU)" if share and world > 1 else "none (one rank, no process group)"),
        }
        if hm is not None:
            if traffic.get("hommul_bytes_per_op"):
                hm["traffic"] = traffic["hommul_bytes_per_op"]
                hm["traffic_ratio"] = traffic["hommul_bytes_per_op"] / hm["algorithmic_bytes_per_op"]
            # per-kernel us / algorithmic bytes / fraction of 8 TB/s of one op, alone and inside the best batch: OFFLINE kernel traces
            # (tools/stage_table.py, tools/stage_table_batched.py); the line carries the reference and the furthest-below row only
            hm["stages"] = stage_ref(STAGES_FILE)
            hm["batched"]["stages"] = stage_ref(STAGES_BATCHED_FILE)
            if traffic.get("hommul_batched_bytes_per_op"):
                hm["batched"]["traffic"] = traffic["hommul_batched_bytes_per_op"]
                hm["batched"]["traffic_batch"] = traffic.get("hommul_batched_batch")
                hm["batched"]["traffic_ratio"] = traffic["hommul_batched_bytes_per_op"] / hm["algorithmic_bytes_per_op"]
        full["cpu_baseline"] = cpu_line    # timed first (see above); None with --no-cpu-baseline
        full["preflight"] = preflight_rec
        full_path = args.full_out
        try:
            os.makedirs(os.path.dirname(os.path.abspath(full_path)), exist_ok=True)
            with open(full_path, "w") as fh:
                json.dump(full, fh, indent=1)
        except OSError as e:               # a read-only tree must not cost the line
            full_path = f"not written: {e}"
        print(compact_line(full, full_path), flush=True)
    if world > 1 or force_dist:
        if not share:
            torch.cuda.synchronize()
        dist.barrier()                  # the other ranks leave only after rank 0 has printed the line
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
