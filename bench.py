#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native RNS core (BASELINE.json metric).

A "step" is one forward negacyclic NTT over one ciphertext polynomial of the CKKS set
N = 2^16, 45 RNS limbs (examples/3_ckks.cu:729-739) -- exactly the reference call
nwt_2d_radix8_forward_inplace(data, tables, 45, 0) (src/ntt/fntt_2d.cu:620-653) -- on synthetic
uniform residues already resident in HBM.  `value` is limb-transforms per second over all ranks.
The same line also carries HomMul+relinearize+rescale/s for the same parameter set,
the roofline object for the forward NTT, and the CPU baseline (the oracle, timed on host cores).

Multi-GPU: independent ciphertexts shard across ranks (weak scaling, no data-path collective); the
evaluation key is generated on rank 0 and broadcast once over RCCL (setup, not timed).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "phantom-fhe_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist

LOG_N = 16
BITS = [60] + [50] * 44 + [60] * 15   # 45 data primes + 15 special primes
SIZE_P = 15
PEAK_HBM = 8.0e12                     # MI355X_MICROARCH.md: 8 TB/s HBM3E peak
# HBM-side bytes of one 45-limb forward NTT from the PMC passes of profiles/r01j_pmc_{fetch,write}.csv
# (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs; FETCH_SIZE doubled per the gfx950 correction of
# MI355X_MICROARCH.md): strided pass 2*12023 + 23040 KiB, contiguous pass 2*24142 + 23040 KiB (it streams the
# twiddle table: 8-byte entries for the 44 FP64 limbs, 16-byte pairs for the 60-bit limb).  Not measurable inside this process, hence a recorded constant.
NTT_TRAFFIC_BYTES = (2 * (12023 + 24142) + 23040 + 23040) * 1024


def uniform_residues(primes, n, device, gen):
    """[len(primes)][n] int64 tensor, limb i uniform in [0, primes[i]) (bits = uint64 residues)."""
    out = torch.empty((len(primes), n), dtype=torch.int64, device=device)
    for i, q in enumerate(primes):
        out[i] = torch.randint(0, int(q), (n,), dtype=torch.int64, device=device, generator=gen)
    return out


def cpu_baseline(primes, n, seconds=12.0, gpu_forward=None):
    """Time the oracle's forward NTT (C port of the reference semantics) on the host: one core (the reported
    baseline) and, informational, OpenMP over the 45 limbs on every core of the box.  Before the timing is
    accepted the same input goes through the GPU path (gpu_forward) and the two outputs are compared bit for bit."""
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
    reps_all, dt_all = run(threads, 4.0)
    oc.L.orc_set_threads(1)
    return {"value": 45 * reps / dt, "unit": "NTT/s", "cores": 1, "kind": "port",
            "sample": f"{reps} forward NTTs of 45 limbs at N=2^16 ({dt:.1f} s, oracle/oracle.c -O3 -march=native, 1 thread)",
            "host_cpus": cores, "checked": checked,
            "all_cores": {"value": 45 * reps_all / dt_all, "unit": "NTT/s", "cores": threads,
                          "sample": f"{reps_all} x 45 limbs, OpenMP over limbs ({threads} threads, one limb each), "
                                    f"{dt_all:.1f} s; cores = min(affinity, cgroup quota, 45)"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="time eager launches instead of one hipGraph replay")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hook (1-GPU boxes): PHA_BENCH_SHARE_GPU=1 runs every rank on cuda:0 over gloo, to exercise the
    # multi-rank code path where RCCL cannot be used (it refuses two ranks on one device)
    share = os.environ.get("PHA_BENCH_SHARE_GPU") == "1"
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="gloo" if share else "nccl", init_method="env://")   # "nccl" is RCCL on ROCm
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    import phantom_fhe_amd as P
    if os.environ.get("PHA_NTT_VARIANT"):   # A/B experiments only (pha_set_tuning key 0); results never change
        P.set_tuning(0, int(os.environ["PHA_NTT_VARIANT"]))
    n = 1 << LOG_N
    primes = [int(p) for p in P.coeff_modulus_create(n, BITS)]
    size_q = len(primes) - SIZE_P
    ctx = P.PhantomContext(LOG_N, primes, SIZE_P, device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x5EED0000 + 3 + rank)

    # ---- evaluation key: generated on rank 0, broadcast once over RCCL/xGMI --------
    dnum = size_q // SIZE_P
    evk = [torch.empty((2, len(primes), n), dtype=torch.int64, device=dev) for _ in range(dnum)]
    if rank == 0:
        for k in evk:
            k[0] = uniform_residues(primes, n, dev, gen)
            k[1] = uniform_residues(primes, n, dev, gen)
    from phantom_fhe_amd import dist as pdist
    if share and world > 1:               # gloo moves host tensors
        host = [k.cpu() for k in evk]
        pdist.broadcast_keys(host, src=0)
        for k, h in zip(evk, host):
            k.copy_(h)
    else:
        pdist.broadcast_keys(evk, src=0)  # one-time RCCL broadcast; no collective on the data path
    rlk = P.PhantomRelinKey(evk)

    # ---- forward NTT: the timed headline ---------------------------------------------------------------
    poly = uniform_residues(primes[:size_q], n, dev, gen)
    for _ in range(args.warmup):
        ctx.nwt_2d_radix8_forward_inplace(poly, size_q, 0)
    torch.cuda.synchronize()
    # The K timed steps are captured once into a hipGraph (the launch-bound inner loop: 2 kernels of
    # ~14 us each per step) and replayed inside the timed region; eager launches are the fallback.
    graph = None
    if not args.no_graph:
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    for _ in range(args.steps):
                        ctx.nwt_2d_radix8_forward_inplace(poly, size_q, 0)
            torch.cuda.current_stream().wait_stream(side)
            g.replay()                   # one untimed replay (warm instantiation)
            torch.cuda.synchronize()
            graph = g
        except Exception as exc:         # pragma: no cover - depends on the runtime
            print(f"[bench] hipGraph capture unavailable ({exc}); timing eager launches", file=sys.stderr)
            graph = None
    if world > 1:
        dist.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()                       # same stream the launches go to (torch's current stream)
    if graph is not None:
        graph.replay()
    else:
        for _ in range(args.steps):
            ctx.nwt_2d_radix8_forward_inplace(poly, size_q, 0)
    e1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = e0.elapsed_time(e1) / args.steps      # average duration of one forward NTT (2 kernels)
    elapsed = pdist.max_over_ranks(elapsed, device=None if share else dev)
    ntt_per_s = world * args.steps * size_q / elapsed

    # ---- informational: the same transform over a batch of 4 polynomials in one launch (extension API) ----
    batch = 4
    polys = torch.stack([uniform_residues(primes[:size_q], n, dev, gen) for _ in range(batch)])
    for _ in range(5):
        ctx.nwt_2d_radix8_forward_inplace_batched(polys, size_q, 0, batch, size_q * n)
    torch.cuda.synchronize()
    b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b_steps = max(10, args.steps // 4)
    b0.record()
    for _ in range(b_steps):
        ctx.nwt_2d_radix8_forward_inplace_batched(polys, size_q, 0, batch, size_q * n)
    b1.record()
    torch.cuda.synchronize()
    batched_ms = b0.elapsed_time(b1) / b_steps
    del polys

    # ---- HomMul + relinearize + rescale (secondary figure, same parameter set) -------------------------
    ct1 = torch.stack([uniform_residues(primes[:size_q], n, dev, gen) for _ in range(2)])
    ct2 = torch.stack([uniform_residues(primes[:size_q], n, dev, gen) for _ in range(2)])
    buf = torch.zeros((3, size_q, n), dtype=torch.int64, device=dev)
    out = torch.zeros((2, size_q - 1, n), dtype=torch.int64, device=dev)

    def hommul():
        buf[:2].copy_(ct1)
        ctx.tensor_prod_2x2_rns_poly(buf, ct2, buf, size_q)                              # multiply_inplace
        ctx.keyswitch_inplace(size_q, buf, buf[2], rlk.public_keys_ptr, P.scheme_type.ckks)  # relinearize
        ctx.divide_and_round_q_last_ntt(size_q, buf, 2, out)                             # rescale_to_next

    hm_steps = max(5, args.steps // 10)
    for _ in range(3):
        hommul()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(hm_steps):
        hommul()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    hm_elapsed = time.perf_counter() - t0
    hm_elapsed = pdist.max_over_ranks(hm_elapsed, device=None if share else dev)

    # the same operation on S independent ciphertext pairs, one HIP stream each (per-stream scratch arenas):
    # the latency-bound kernels of different ciphertexts overlap on the chip.  Informational (this rank).
    S = 4
    lanes = []
    for i in range(S):
        a = torch.stack([uniform_residues(primes[:size_q], n, dev, gen) for _ in range(2)])
        lanes.append((torch.cuda.Stream(device=dev), a, torch.zeros((3, size_q, n), dtype=torch.int64, device=dev),
                      torch.zeros((2, size_q - 1, n), dtype=torch.int64, device=dev)))

    def hommul_lanes():
        for st, a, b3, o in lanes:
            with torch.cuda.stream(st):
                b3[:2].copy_(a)
                ctx.tensor_prod_2x2_rns_poly(b3, ct2, b3, size_q)
                ctx.keyswitch_inplace(size_q, b3, b3[2], rlk.public_keys_ptr, P.scheme_type.ckks)
                ctx.divide_and_round_q_last_ntt(size_q, b3, 2, o)

    torch.cuda.synchronize()
    for _ in range(2):
        hommul_lanes()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(hm_steps):
        hommul_lanes()
    torch.cuda.synchronize()
    hm_lanes_elapsed = time.perf_counter() - t0

    # the same operation on a batch of B ciphertext pairs through the batched entry points (one set of launches:
    # key limbs read once, NTT / base-conversion launches B times larger).  Informational (this rank).
    del lanes
    B = int(os.environ.get("PHA_BENCH_BATCH", "8"))
    bt1 = torch.stack([torch.stack([uniform_residues(primes[:size_q], n, dev, gen) for _ in range(2)]) for _ in range(B)])
    bt2 = torch.stack([torch.stack([uniform_residues(primes[:size_q], n, dev, gen) for _ in range(2)]) for _ in range(B)])
    b01 = torch.zeros_like(bt1)
    b2 = torch.zeros((B, size_q, n), dtype=torch.int64, device=dev)
    bout = torch.zeros((B, 2, size_q - 1, n), dtype=torch.int64, device=dev)

    def hommul_batched():
        ctx.tensor_prod_2x2_batched(bt1, bt2, b01, b2, size_q, B)
        ctx.keyswitch_inplace_batched(size_q, b01, b2, B, rlk.public_keys_ptr, P.scheme_type.ckks)
        ctx.divide_and_round_q_last_ntt(size_q, b01, 2 * B, bout)

    for _ in range(2):
        hommul_batched()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(hm_steps):
        hommul_batched()
    torch.cuda.synchronize()
    hm_batched_elapsed = time.perf_counter() - t0

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
    copy_gbps = 10 * 2 * cal_a.numel() * 8 / (ev0.elapsed_time(ev1) * 1e-3) / 1e9
    del cal_a, cal_b
    # minimal per-stage traffic of one HomMul + relinearize + rescale at C3: 929 MiB
    hm_alg_bytes = 929.0 * (1 << 20)

    if rank == 0:
        alg_bytes = 16.0 * n * size_q                  # 8 B read + 8 B write per coefficient
        achieved = alg_bytes / (kernel_ms * 1e-3)
        line = {
            "metric": "forward NTT limb-transforms/s at N=2^16, 45 RNS moduli",
            "value": ntt_per_s, "unit": "NTT/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "CKKS N=2^16, 45 data limbs (+15 special), forward NTT of one ciphertext "
                                   "polynomial per step (configs[2] parameter set)",
                       "N": n, "limbs": size_q, "special_limbs": SIZE_P, "parallelism": f"ciphertext-batch x{world}",
                       "launch": "hipGraph replay of the K steps" if graph is not None else "eager"},
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s",
                         "frac": achieved / PEAK_HBM, "traffic": NTT_TRAFFIC_BYTES,
                         "traffic_source": "profiles/r01m_pmc_fetch.csv + r01m_pmc_write.csv (same figures as r01j / r01l) (rocprofv3 PMC, per launch pair)",
                         "kernel": "ntt_pass_kernel pair (strided pass + contiguous pass)",
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": kernel_ms,
                         "calibrated_copy_GBps": copy_gbps,
                         "calibrated_note": "512 MiB device-to-device copy, read + write bytes / time, this run"},
            "batched_ntt": {"polynomials_per_launch": batch, "ms_per_launch": batched_ms,
                            "value": batch * size_q / (batched_ms * 1e-3), "unit": "NTT/s (this rank)",
                            "frac_of_peak": batch * alg_bytes / (batched_ms * 1e-3) / PEAK_HBM,
                            "note": "pha_nwt_2d_radix8_forward_inplace_batched: 4 x 45 limbs per launch pair"},
            "hommul_relin_rescale": {"value": world * hm_steps / hm_elapsed, "unit": "ops/s",
                                     "ms_per_op": 1e3 * hm_elapsed / hm_steps, "steps": hm_steps,
                                     "algorithmic_bytes_per_op": hm_alg_bytes,
                                     "frac_of_peak": hm_alg_bytes / (hm_elapsed / hm_steps) / PEAK_HBM},
            "hommul_relin_rescale_4_streams": {"value": S * hm_steps / hm_lanes_elapsed, "unit": "ops/s (this rank)",
                                               "ms_per_op": 1e3 * hm_lanes_elapsed / (S * hm_steps),
                                               "note": "4 independent ciphertext pairs, one HIP stream each"},
            "hommul_relin_rescale_batched": {"value": B * hm_steps / hm_batched_elapsed, "unit": "ops/s (this rank)",
                                             "ms_per_op": 1e3 * hm_batched_elapsed / (B * hm_steps), "batch": B,
                                             "frac_of_peak": hm_alg_bytes / (hm_batched_elapsed / (B * hm_steps)) / PEAK_HBM,
                                             "note": "pha_tensor_prod_2x2_batched + pha_keyswitch_inplace_batched + rescale of the batch"},
        }
        if not args.no_cpu_baseline and world == 1:
            def gpu_forward(host_poly):   # the product path on the baseline's own input
                d = P.to_device(host_poly, dev)
                ctx.nwt_2d_radix8_forward_inplace(d, 45, 0)
                return P.to_host(d)
            line["cpu_baseline"] = cpu_baseline(primes, n, gpu_forward=gpu_forward)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
