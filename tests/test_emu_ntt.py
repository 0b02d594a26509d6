"""Replays the HIP NTT thread program (phantom-fhe_amd/csrc/pha_ntt_core.h) on the CPU, workgroup by
workgroup, and compares with the oracle -- catches indexing / LDS-layout / twiddle-addressing bugs
without a GPU.  The replay harness is tests/emu/emu_ntt.cpp (test-only)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
from util import rng_for

HERE = os.path.dirname(os.path.abspath(__file__))
u64p = C.POINTER(C.c_uint64)


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("emu") / "libemu_ntt.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", out, os.path.join(HERE, "emu", "emu_ntt.cpp")])
    L = C.CDLL(out)
    L.emu_ntt.argtypes = [C.c_int, C.c_int, C.c_int, u64p, u64p, C.c_uint64] + [u64p] * 5
    return L


def p(a):
    return a.ctypes.data_as(u64p)


def pair(v, q):
    return np.array([v, O.compute_shoup(v, q)], dtype=np.uint64)


@pytest.mark.parametrize("log_n", [12, 13, 14, 15, 16, 17])
@pytest.mark.parametrize("bits", [40, 42, 43, 47, 48, 50, 60, 61])   # 42 / 47: just under the light-butterfly thresholds
@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 12])   # 6: one-launch plans; 5 / 7 / 8: NttPlan variants 5 / 6 / 7; 9 / 11 / 12: the r04 splits of 2^16 (plans 8 / 10 / 12)
def test_thread_program_matches_oracle(emu, log_n, bits, variant):
    if variant == 6 and log_n not in (12, 13, 14):
        pytest.skip("the one-launch plans exist for N = 4096, 8192 and 16384 only")
    if variant in (5, 7, 8) and log_n not in (14, 15, 16):
        pytest.skip("the four-coefficients-per-thread contiguous pass exists for N = 2^14 .. 2^16")
    if variant in (9, 11, 12) and log_n != 16:
        pytest.skip("the 128 x 512 and 64 x 1024 plans exist for N = 2^16")
    n = 1 << log_n
    code = log_n | (variant << 8)   # variant 1 = 8 coefficients per thread (512-thread workgroups); 2 = 1 + on-the-fly twiddles
    q = int(O.get_primes(n, bits, 1)[0])
    tw, tws, itw, itws, ni, nis = O.ntt_tables(log_n, q)
    c = O.Ctx(log_n, [q], 0)
    r = rng_for(log_n * 100 + bits)
    x = r.integers(0, q, n, dtype=np.uint64)
    if log_n == 13:
        x[:] = q - 1          # extreme input: exercises the top of the [0,8q) lazy range (8q < 2^64 needs q < 2^61)
    if log_n == 14:
        x[::2] = 0
        x[1::2] = q - 1
    ref = c.nwt_forward(x.reshape(1, n), 1)[0]
    twi = np.ascontiguousarray(np.stack([tw, tws], axis=1).reshape(-1))
    z = np.zeros(2, dtype=np.uint64)
    out = np.zeros(n, dtype=np.uint64)
    assert emu.emu_ntt(code, 1, 1, p(x), p(out), q, p(twi), p(z), p(z), p(z), p(z)) == 0
    assert np.array_equal(out, ref)
    # the product keeps a pristine inverse table plus (N^-1, itw[1]*N^-1); rebuild that from the oracle's
    itw1 = int(itw[1]) * n % q
    itw_p, itws_p = itw.copy(), itws.copy()
    itw_p[1], itws_p[1] = itw1, O.compute_shoup(itw1, q)
    itwi = np.ascontiguousarray(np.stack([itw_p, itws_p], axis=1).reshape(-1))
    back = np.zeros(n, dtype=np.uint64)
    assert emu.emu_ntt(code, 0, 3, p(ref), p(back), q, p(itwi), p(pair(ni, q)), p(pair(int(itw[1]), q)), p(z), p(z)) == 0
    assert np.array_equal(back, x)
    if bits <= 50:
        # FP64 path (primes below 2^50): tables of doubles W (8 bytes per entry); results must stay bit-identical
        def fpairs(w):
            return np.ascontiguousarray(w.astype(np.float64)).view(np.uint64).copy()
        def fpair1(v):
            d = np.array([float(v), 0.0], dtype=np.float64)
            return d.view(np.uint64).copy()
        fcode = code | (1 << 16)
        out_f = np.zeros(n, dtype=np.uint64)
        assert emu.emu_ntt(fcode, 1, 1, p(x), p(out_f), q, p(fpairs(tw)), p(z), p(z), p(z), p(z)) == 0
        assert np.array_equal(out_f, ref)
        back_f = np.zeros(n, dtype=np.uint64)
        assert emu.emu_ntt(fcode, 0, 3, p(ref), p(back_f), q, p(fpairs(itw_p)), p(fpair1(ni)), p(fpair1(int(itw[1]))), p(z), p(z)) == 0
        assert np.array_equal(back_f, x)
        if log_n in (15, 17) or variant == 6:   # the inverse's sums double every stage: feed it the largest residues everywhere
            near_q = (q - 1 - r.integers(0, 1 << 16, n)).astype(np.uint64)   # large sums with busy low bits: inexact past 2^53
            for y in (np.full(n, q - 1, dtype=np.uint64), np.where(np.arange(n) % 2 == 0, q - 1, 0).astype(np.uint64), near_q):
                assert emu.emu_ntt(fcode, 0, 3, p(y), p(back_f), q, p(fpairs(itw_p)), p(fpair1(ni)), p(fpair1(int(itw[1]))), p(z), p(z)) == 0
                assert np.array_equal(back_f, c.nwt_backward(y.reshape(1, n), 1)[0])
            if variant == 6:   # the forward's magnitudes grow by q/2 per stage: 14 stages in one pass need the re-centring
                for y in (np.full(n, q - 1, dtype=np.uint64), near_q):
                    assert emu.emu_ntt(fcode, 1, 1, p(y), p(out_f), q, p(fpairs(tw)), p(z), p(z), p(z), p(z)) == 0
                    assert np.array_equal(out_f, c.nwt_forward(y.reshape(1, n), 1)[0])
        s2 = int(r.integers(1, q))
        assert emu.emu_ntt(fcode, 0, 4, p(ref), p(back_f), q, p(fpairs(itw_p)), p(fpair1(ni)), p(fpair1(int(itw[1]))), p(pair(s2, q)), p(z)) == 0
        assert np.array_equal(back_f, c.multiply_scalar(x.reshape(1, n), np.array([s2], dtype=np.uint64), 1)[0])
    if log_n in (12, 16):
        # inverse with fused scale (EPI_INV_SCALE) and forward with fused mod-down epilogue (EPI_FWD_MODDOWN)
        s = int(r.integers(1, q))
        assert emu.emu_ntt(code, 0, 4, p(ref), p(back), q, p(itwi), p(pair(ni, q)), p(pair(int(itw[1]), q)), p(pair(s, q)), p(z)) == 0
        assert np.array_equal(back, c.multiply_scalar(x.reshape(1, n), np.array([s], dtype=np.uint64), 1)[0])
        cx = r.integers(0, q, n, dtype=np.uint64)
        assert emu.emu_ntt(code, 1, 2, p(x), p(out), q, p(twi), p(z), p(z), p(pair(s, q)), p(cx)) == 0
        want = c.multiply_scalar(c.sub(cx.reshape(1, n), ref.reshape(1, n), 1), np.array([s], dtype=np.uint64), 1)[0]
        assert np.array_equal(out, want)
        if bits <= 50:   # the FP64 path computes this epilogue on the doubles (apply_epilogue_fp): same words
            fcode = code | (1 << 16)
            ftw = np.ascontiguousarray(tw.astype(np.float64)).view(np.uint64).copy()
            assert emu.emu_ntt(fcode, 1, 2, p(x), p(out), q, p(ftw), p(z), p(z), p(pair(s, q)), p(cx)) == 0
            assert np.array_equal(out, want)


@pytest.mark.parametrize("log_n", [14, 15, 16, 17])
@pytest.mark.parametrize("bits", [40, 47, 50, 60])
def test_fused_modup_hands_its_registers_to_the_inverse_pass(emu, log_n, bits):
    """modup_ip_body (pha_ntt.hip): the forward contiguous pass keeps its last round's outputs in registers; for the special limbs the
    inverse transform's contiguous pass starts from those registers (ModupIpArgs::inv_from) and the caller launches the strided pass
    alone.  Replay of exactly that call sequence: the values at the kernel's key addresses are the oracle's forward transform, and
    the finished inverse is the input again -- integer and FP64 back ends, plan 3 (the product) and plan 5."""
    n = 1 << log_n
    q = int(O.get_primes(n, bits, 1)[0])
    tw, tws, itw, itws, ni, nis = O.ntt_tables(log_n, q)
    c = O.Ctx(log_n, [q], 0)
    r = rng_for(log_n * 7 + bits)
    x = r.integers(0, q, n, dtype=np.uint64)
    if log_n == 15:
        x[::3] = q - 1
    ref = c.nwt_forward(x.reshape(1, n), 1)[0]
    itw1 = int(itw[1]) * n % q
    itw_p, itws_p = itw.copy(), itws.copy()
    itw_p[1], itws_p[1] = itw1, O.compute_shoup(itw1, q)
    twi = np.ascontiguousarray(np.stack([tw, tws], axis=1).reshape(-1))
    itwi = np.ascontiguousarray(np.stack([itw_p, itws_p], axis=1).reshape(-1))
    emu.emu_keep_then_inverse.argtypes = [C.c_int, C.c_int, C.c_int, u64p, u64p, u64p, C.c_uint64] + [u64p] * 4
    for variant in (3, 5):
        if variant == 5 and log_n == 17:
            continue
        mid, back = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
        assert emu.emu_keep_then_inverse(log_n, variant, 0, p(x), p(mid), p(back), q, p(twi), p(itwi), p(pair(ni, q)), p(pair(int(itw[1]), q))) == 0
        assert np.array_equal(mid, ref) and np.array_equal(back, x), (variant, "integer")
        if bits <= 50:
            f = lambda w: np.ascontiguousarray(w.astype(np.float64)).view(np.uint64).copy()
            f1 = lambda v: np.array([float(v), 0.0], dtype=np.float64).view(np.uint64).copy()
            mid[:] = 0
            back[:] = 0
            assert emu.emu_keep_then_inverse(log_n, variant, 1, p(x), p(mid), p(back), q, p(f(tw)), p(f(itw_p)), p(f1(ni)), p(f1(int(itw[1])))) == 0
            assert np.array_equal(mid, ref) and np.array_equal(back, x), (variant, "fp64")


@pytest.mark.parametrize("log_n", [12, 13, 14, 15, 16, 17])
def test_barrier_free_plans_are_wave_local(emu, log_n):
    """The plans whose rounds hand over without a workgroup barrier (one-wavefront contiguous tiles) may only do so if no wavefront touches another wavefront's LDS words in any round."""
    assert emu.emu_plan_is_wave_local(log_n, 1) == 0           # the barrier plans do not claim it
    assert emu.emu_plan_is_wave_local(log_n, 3) == 2 and emu.emu_plan_is_wave_local(log_n, 4) == 2
    for variant in (3, 4):
        assert emu.emu_check_wave_local(log_n, variant) == 0
    if log_n in (14, 15, 16):      # the four-coefficients-per-thread contiguous pass: one row per wavefront
        assert emu.emu_plan_is_wave_local(log_n, 5) == 2 and emu.emu_check_wave_local(log_n, 5) == 0


def _has_fma():
    try:
        return " fma " in open("/proc/cpuinfo").read()
    except OSError:
        return False


@pytest.mark.skipif(not _has_fma(), reason="host CPU has no FMA unit: contraction cannot happen")
@pytest.mark.parametrize("cxx", ["g++", "/opt/rocm/lib/llvm/bin/clang++"])
def test_fp64_path_does_not_depend_on_the_contract_flag(cxx, tmp_path):
    """pha_arith.h states `fp contract(off)` itself: a build whose flags ask for contraction (-ffp-contract=fast with an
    FMA unit to contract into) must still replay the FP64 butterflies bit-exactly for 40..50-bit primes, forward and
    inverse, incl. the extreme-residue inputs."""
    if not (os.path.exists(cxx) or cxx == "g++"):
        pytest.skip(cxx + " not present")
    out = str(tmp_path / "libemu_fast.so")
    subprocess.check_call([cxx, "-O3", "-std=c++17", "-fPIC", "-shared", "-mfma", "-ffp-contract=fast", "-o", out,
                           os.path.join(HERE, "emu", "emu_ntt.cpp")])
    L = C.CDLL(out)
    L.emu_ntt.argtypes = [C.c_int, C.c_int, C.c_int, u64p, u64p, C.c_uint64] + [u64p] * 5
    z = np.zeros(2, dtype=np.uint64)
    for log_n, variant in ((12, 6), (14, 4), (16, 4), (16, 5)):
        n = 1 << log_n
        for bits in (40, 43, 47, 50):
            q = int(O.get_primes(n, bits, 1)[0])
            tw, tws, itw, itws, ni, nis = O.ntt_tables(log_n, q)
            c = O.Ctx(log_n, [q], 0)
            r = rng_for(log_n * 1000 + bits)
            itw_p = itw.copy()
            itw_p[1] = int(itw[1]) * n % q
            f = lambda w: np.ascontiguousarray(w.astype(np.float64)).view(np.uint64).copy()
            f1 = lambda v: np.array([float(v), 0.0], dtype=np.float64).view(np.uint64).copy()
            fcode = log_n | (variant << 8) | (1 << 16)
            near_q = (q - 1 - r.integers(0, 1 << 16, n)).astype(np.uint64)
            for x in (r.integers(0, q, n, dtype=np.uint64), np.full(n, q - 1, dtype=np.uint64), near_q):
                got = np.zeros(n, dtype=np.uint64)
                assert L.emu_ntt(fcode, 1, 1, p(x), p(got), q, p(f(tw)), p(z), p(z), p(z), p(z)) == 0
                assert np.array_equal(got, c.nwt_forward(x.reshape(1, n), 1)[0])
                assert L.emu_ntt(fcode, 0, 3, p(x), p(got), q, p(f(itw_p)), p(f1(ni)), p(f1(int(itw[1]))), p(z), p(z)) == 0
                assert np.array_equal(got, c.nwt_backward(x.reshape(1, n), 1)[0])


@pytest.mark.parametrize("log_n,variant", [(14, 3), (15, 3), (16, 11), (17, 3)])
@pytest.mark.parametrize("bits", [43, 50, 60])
def test_batched_kernel_form_of_the_contiguous_pass(emu, log_n, variant, bits):
    """ntt_zloop_kernel (r04) runs the contiguous pass as: every round's twiddles requested once, then per polynomial the first round's
    global loads (`prefetch`) and `run_prefetched` for each segment.  Bit 17 of the replay's code selects that form: forward and
    inverse, both back ends, the product's plans (3 for N != 2^16, 10 for N = 2^16), must equal the oracle."""
    n = 1 << log_n
    q = int(O.get_primes(n, bits, 1)[0])
    tw, tws, itw, itws, ni, nis = O.ntt_tables(log_n, q)
    c = O.Ctx(log_n, [q], 0)
    r = rng_for(log_n * 7 + bits)
    z = np.zeros(2, dtype=np.uint64)
    itw_p, itws_p = itw.copy(), itws.copy()
    itw_p[1] = int(itw[1]) * n % q
    itws_p[1] = O.compute_shoup(int(itw_p[1]), q)
    twi = np.ascontiguousarray(np.stack([tw, tws], axis=1).reshape(-1))
    itwi = np.ascontiguousarray(np.stack([itw_p, itws_p], axis=1).reshape(-1))
    code = log_n | (variant << 8) | (1 << 17)
    for x in (r.integers(0, q, n, dtype=np.uint64), np.full(n, q - 1, dtype=np.uint64)):
        got = np.zeros(n, dtype=np.uint64)
        assert emu.emu_ntt(code, 1, 1, p(x), p(got), q, p(twi), p(z), p(z), p(z), p(z)) == 0
        assert np.array_equal(got, c.nwt_forward(x.reshape(1, n), 1)[0])
        assert emu.emu_ntt(code, 0, 3, p(x), p(got), q, p(itwi), p(pair(ni, q)), p(pair(int(itw[1]), q)), p(z), p(z)) == 0
        assert np.array_equal(got, c.nwt_backward(x.reshape(1, n), 1)[0])
        if bits <= 50:
            f = lambda w: np.ascontiguousarray(w.astype(np.float64)).view(np.uint64).copy()
            f1 = lambda v: np.array([float(v), 0.0], dtype=np.float64).view(np.uint64).copy()
            fcode = code | (1 << 16)
            assert emu.emu_ntt(fcode, 1, 1, p(x), p(got), q, p(f(tw)), p(z), p(z), p(z), p(z)) == 0
            assert np.array_equal(got, c.nwt_forward(x.reshape(1, n), 1)[0])
            assert emu.emu_ntt(fcode, 0, 3, p(x), p(got), q, p(f(itw_p)), p(f1(ni)), p(f1(int(itw[1]))), p(z), p(z)) == 0
            assert np.array_equal(got, c.nwt_backward(x.reshape(1, n), 1)[0])


@pytest.mark.parametrize("bits", [43, 47, 50, 60])
def test_strided_pass_of_the_fused_modup_conversion(emu, bits):
    """r05: modup_conv_s1_kernel runs the forward transform's strided pass from registers that its base conversion fills: NttPlan<16, 13>::P1
    (four coefficients per thread, radix-4 rounds 2-2-2, 64 x 16 tiles), every round's twiddles requested up front, `run_prefetched` for
    each segment; plan 10's contiguous pass (in the batched kernel's form) finishes the transform.  Code 13 of the replay runs exactly
    that pair: integer and FP64 back ends, random and extreme inputs, against the oracle's forward transform."""
    log_n = 16
    n = 1 << log_n
    q = int(O.get_primes(n, bits, 1)[0])
    tw, tws, itw, itws, ni, nis = O.ntt_tables(log_n, q)
    c = O.Ctx(log_n, [q], 0)
    r = rng_for(1300 + bits)
    z = np.zeros(2, dtype=np.uint64)
    twi = np.ascontiguousarray(np.stack([tw, tws], axis=1).reshape(-1))
    code = log_n | (13 << 8) | (1 << 17)
    for x in (r.integers(0, q, n, dtype=np.uint64), np.full(n, q - 1, dtype=np.uint64), np.zeros(n, dtype=np.uint64)):
        want = c.nwt_forward(x.reshape(1, n), 1)[0]
        got = np.zeros(n, dtype=np.uint64)
        assert emu.emu_ntt(code, 1, 1, p(x), p(got), q, p(twi), p(z), p(z), p(z), p(z)) == 0
        assert np.array_equal(got, want)
        if bits <= 50:
            f = lambda w: np.ascontiguousarray(w.astype(np.float64)).view(np.uint64).copy()
            assert emu.emu_ntt(code | (1 << 16), 1, 1, p(x), p(got), q, p(f(tw)), p(z), p(z), p(z), p(z)) == 0
            assert np.array_equal(got, want)
    assert emu.emu_ntt(code, 0, 3, p(x), p(got), q, p(twi), p(z), p(z), p(z), p(z)) == -3     # forward only
