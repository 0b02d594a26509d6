"""Generates tests/golden/hotpath_golden.json -- committed known-answer vectors for the hot path.

The reference cannot be executed in this image (no CUDA, no SEAL) and its own tests
hold no golden vectors, so these vectors come from the CPU oracle (oracle/oracle.c), after the oracle
itself has been pinned by tests/test_oracle.py.  They freeze: the prime chains CoeffModulus::Create
yields for every BASELINE.json configuration, per-prime constants (minimal 2N-th root, Barrett ratio,
N^-1, first twiddles), and SHA-256 digests (+ a few sampled coefficients) of every hot-path stage on
seeded inputs.  Inputs are regenerated from the seed by numpy's PCG64 (np.random.default_rng), so the
file stays small.  Run:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import oracle as O  # noqa: E402
from util import CONFIGS, oracle_ctx, primes_of, rng_for, uniform_poly  # noqa: E402


def digest(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return {"sha256": hashlib.sha256(a.tobytes()).hexdigest(), "shape": list(a.shape),
            "samples": [int(v) for v in a.reshape(-1)[:: max(1, a.size // 8)][:8]]}


def stage_vectors(name, scheme, ql, seed):
    log_n, primes, size_p = primes_of(name)
    n = 1 << log_n
    size_q = len(primes) - size_p
    oc = oracle_ctx(name)
    r = rng_for(seed)
    out = {"config": name, "scheme": int(scheme), "size_Ql": ql, "seed": seed}
    x = uniform_poly(r, primes[:ql], n)
    out["ntt_forward"] = digest(oc.nwt_forward(x, ql, 0))
    out["ntt_backward"] = digest(oc.nwt_backward(x, ql, 0))
    y = uniform_poly(r, primes[:ql], n)
    out["tensor_prod_2x2"] = digest(oc.tensor_prod_2x2(np.stack([x, y]), np.stack([y, x]), ql))
    if size_p:
        tool = O.Tool(oc, ql)
        dnum = size_q // size_p
        evk = np.stack([np.stack([uniform_poly(r, primes, n), uniform_poly(r, primes, n)]) for _ in range(dnum)])
        c2 = uniform_poly(r, primes[:ql], n)
        mu = tool.modup(c2, scheme)
        out["modup"] = digest(mu)
        cx = tool.key_switch_inner_prod(mu, [evk[i] for i in range(tool.beta)])
        out["inner_prod"] = digest(cx)
        out["moddown"] = digest(tool.moddown_from_ntt(cx[0], scheme))
        ct = np.stack([x, y])
        ks = tool.keyswitch_inplace(ct, c2, [evk[i] for i in range(tool.beta)], scheme)
        out["keyswitch_inplace"] = digest(ks)
        if ql > 1 and scheme == O.CKKS:
            out["rescale"] = digest(tool.rescale_ntt(ks, 2))
    return out


def main():
    g = {"generator": "tests/golden/make_golden.py (oracle/oracle.c)", "configs": {}, "stages": []}
    for name in CONFIGS:
        log_n, primes, size_p = primes_of(name)
        oc = oracle_ctx(name) if log_n <= 14 else None
        consts = []
        for i in (0, len(primes) - 1):
            q = int(primes[i])
            tw, tws, itw, itws, ni, nis = O.ntt_tables(log_n, q)
            consts.append({"index": i, "q": q, "const_ratio": list(O.const_ratio(q)),
                           "root": O.minimal_primitive_root(2 << log_n, q), "n_inv": ni,
                           "twiddle_1_2_3": [int(v) for v in tw[1:4]], "twiddle_shoup_1": int(tws[1]),
                           "itwiddle_1_folded": int(itw[1])})
        g["configs"][name] = {"log_n": log_n, "size_P": size_p, "primes": [int(p) for p in primes], "consts": consts}
    g["stages"].append(stage_vectors("c1_bfv4096", O.BFV, 2, 101))
    g["stages"].append(stage_vectors("c1_bfv4096", O.CKKS, 2, 102))
    g["stages"].append(stage_vectors("hyb12_a2", O.CKKS, 6, 103))
    g["stages"].append(stage_vectors("hyb12_a2", O.BFV, 6, 104))
    g["stages"].append(stage_vectors("hyb13_a3", O.CKKS, 7, 105))
    g["stages"].append(stage_vectors("c2_ntt14", O.CKKS, 8, 106))
    g["stages"].append(stage_vectors("c4_bfv15", O.BFV, 30, 107))
    g["stages"].append(stage_vectors("c3_ckks16", O.CKKS, 45, 108))
    with open(os.path.join(HERE, "hotpath_golden.json"), "w") as f:
        json.dump(g, f, indent=1)
    print("wrote", os.path.join(HERE, "hotpath_golden.json"))


if __name__ == "__main__":
    main()
