"""Committed golden vectors (tests/golden/hotpath_golden.json, made by tests/golden/make_golden.py):
CPU: the oracle still reproduces them (small configs); GPU: the HIP path reproduces every stage digest at
every configuration, including the full-size C3 (N=2^16, 45+15 limbs) and C4 (N=2^15, 30+15 limbs)."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import oracle as O
from util import oracle_ctx, primes_of, rng_for, uniform_poly

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hotpath_golden.json")))


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint64).tobytes()).hexdigest()


def test_prime_chains_and_constants():
    for name, cfg in GOLD["configs"].items():
        log_n, primes, size_p = primes_of(name)
        assert [int(p) for p in primes] == cfg["primes"] and size_p == cfg["size_P"] and log_n == cfg["log_n"]
        for c in cfg["consts"]:
            q = c["q"]
            assert list(O.const_ratio(q)) == c["const_ratio"]
            assert O.minimal_primitive_root(2 << log_n, q) == c["root"]
            if log_n <= 13:
                tw, tws, itw, itws, ni, _ = O.ntt_tables(log_n, q)
                assert ni == c["n_inv"] and [int(v) for v in tw[1:4]] == c["twiddle_1_2_3"]
                assert int(tws[1]) == c["twiddle_shoup_1"] and int(itw[1]) == c["itwiddle_1_folded"]


def _inputs(st):
    log_n, primes, size_p = primes_of(st["config"])
    n, ql = 1 << log_n, st["size_Ql"]
    size_q = len(primes) - size_p
    r = rng_for(st["seed"])
    x = uniform_poly(r, primes[:ql], n)
    y = uniform_poly(r, primes[:ql], n)
    evk = c2 = None
    if size_p:
        evk = np.stack([np.stack([uniform_poly(r, primes, n), uniform_poly(r, primes, n)]) for _ in range(size_q // size_p)])
        c2 = uniform_poly(r, primes[:ql], n)
    return log_n, primes, size_p, n, ql, x, y, evk, c2


@pytest.mark.parametrize("st", [s for s in GOLD["stages"] if s["config"] in ("c1_bfv4096", "hyb12_a2")],
                         ids=lambda s: f"{s['config']}-{s['scheme']}")
def test_oracle_reproduces_golden(st):
    log_n, primes, size_p, n, ql, x, y, evk, c2 = _inputs(st)
    oc = oracle_ctx(st["config"])
    assert _sha(oc.nwt_forward(x, ql, 0)) == st["ntt_forward"]["sha256"]
    assert _sha(oc.nwt_backward(x, ql, 0)) == st["ntt_backward"]["sha256"]
    tool = O.Tool(oc, ql)
    ks = tool.keyswitch_inplace(np.stack([x, y]), c2, [evk[i] for i in range(tool.beta)], st["scheme"])
    assert _sha(ks) == st["keyswitch_inplace"]["sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("st", GOLD["stages"], ids=lambda s: f"{s['config']}-{s['scheme']}")
def test_gpu_reproduces_golden(st, gpu):
    import phantom_fhe_amd as P
    log_n, primes, size_p, n, ql, x, y, evk, c2 = _inputs(st)
    ctx = P.PhantomContext(log_n, list(primes), size_p, device=gpu)
    d = P.to_device(x, gpu)
    ctx.nwt_2d_radix8_forward_inplace(d, ql, 0)
    assert _sha(P.to_host(d)) == st["ntt_forward"]["sha256"]
    d = P.to_device(x, gpu)
    ctx.nwt_2d_radix8_backward_inplace(d, ql, 0)
    assert _sha(P.to_host(d)) == st["ntt_backward"]["sha256"]
    buf = P.to_device(np.concatenate([np.stack([x, y]), np.zeros((1, ql, n), dtype=np.uint64)]), gpu)
    ctx.tensor_prod_2x2_rns_poly(buf, P.to_device(np.stack([y, x]), gpu), buf, ql)
    assert _sha(P.to_host(buf)) == st["tensor_prod_2x2"]["sha256"]
    if not size_p:
        return
    rlk = P.PhantomRelinKey.from_numpy(evk, gpu)
    beta, qlp = ctx.beta(ql), ql + size_p
    mu = P.to_device(np.zeros((beta, qlp, n), dtype=np.uint64), gpu)
    ctx.modup(ql, mu, P.to_device(c2, gpu), st["scheme"])
    assert _sha(P.to_host(mu)) == st["modup"]["sha256"]
    cx = P.to_device(np.zeros((2, qlp, n), dtype=np.uint64), gpu)
    ctx.key_switch_inner_prod(ql, cx, mu, rlk.public_keys_ptr)
    assert _sha(P.to_host(cx)) == st["inner_prod"]["sha256"]
    md = P.to_device(np.zeros((ql, n), dtype=np.uint64), gpu)
    ctx.moddown_from_NTT(ql, md, cx[0], st["scheme"])
    assert _sha(P.to_host(md)) == st["moddown"]["sha256"]
    ct = P.to_device(np.stack([x, y]), gpu)
    ctx.keyswitch_inplace(ql, ct, P.to_device(c2, gpu), rlk.public_keys_ptr, st["scheme"])
    assert _sha(P.to_host(ct)) == st["keyswitch_inplace"]["sha256"]
    if "rescale" in st:
        out = P.to_device(np.zeros((2, ql - 1, n), dtype=np.uint64), gpu)
        ctx.divide_and_round_q_last_ntt(ql, ct, 2, out)
        assert _sha(P.to_host(out)) == st["rescale"]["sha256"]
