"""ctypes binding of oracle/liboracle.so (numpy uint64 arrays in/out).

TEST INFRASTRUCTURE ONLY -- see oracle/oracle.h.  Each wrapper mirrors one C entry point; the
C code cites the reference file:line it restates.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

BFV, CKKS, BGV = 1, 2, 3

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)


def build(native=False, out=None):
    """Compile the oracle with gcc.  native=True adds -march=native (used for the CPU baseline)."""
    out = out or os.path.join(_HERE, "liboracle_native.so" if native else "liboracle.so")
    src = os.path.join(_HERE, "oracle.c")
    if os.path.exists(out) and os.path.getmtime(out) >= max(
            os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "oracle.h"))):
        return out
    cmd = ["gcc", "-O3", "-fPIC", "-std=c11", "-fopenmp", "-ffp-contract=off", "-shared", "-o", out, src, "-lm"]
    if native:
        cmd.insert(2, "-march=native")
    subprocess.check_call(cmd)
    return out


def lib(path=None):
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = path or build()
    L = C.CDLL(p)
    L.orc_is_prime.restype = C.c_int
    L.orc_is_prime.argtypes = [C.c_uint64]
    L.orc_get_primes.argtypes = [C.c_uint64, C.c_int, C.c_size_t, u64p]
    L.orc_coeff_modulus_create.argtypes = [C.c_uint64, C.POINTER(C.c_int), C.c_size_t, u64p]
    L.orc_const_ratio.argtypes = [C.c_uint64, u64p]
    L.orc_minimal_primitive_root.argtypes = [C.c_uint64, C.c_uint64, u64p]
    for f in ("orc_compute_shoup", "orc_invmod", "orc_mulmod"):
        getattr(L, f).restype = C.c_uint64
    L.orc_compute_shoup.argtypes = [C.c_uint64, C.c_uint64]
    L.orc_invmod.argtypes = [C.c_uint64, C.c_uint64]
    L.orc_mulmod.argtypes = [C.c_uint64] * 3
    L.orc_powmod.restype = C.c_uint64
    L.orc_powmod.argtypes = [C.c_uint64] * 3
    L.orc_ntt_tables.argtypes = [C.c_int, C.c_uint64, u64p, u64p, u64p, u64p, u64p, u64p]
    L.orc_ntt_forward.argtypes = [u64p, C.c_int, C.c_uint64, u64p, u64p]
    L.orc_ntt_inverse.argtypes = [u64p, C.c_int, C.c_uint64, u64p, u64p, C.c_uint64, C.c_uint64]
    L.orc_ctx_create.restype = C.c_void_p
    L.orc_ctx_create.argtypes = [C.c_int, u64p, C.c_size_t, C.c_size_t]
    L.orc_ctx_destroy.argtypes = [C.c_void_p]
    L.orc_ctx_twiddle.restype = u64p
    L.orc_ctx_twiddle.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    L.orc_ctx_n_inv.restype = C.c_uint64
    L.orc_ctx_n_inv.argtypes = [C.c_void_p, C.c_size_t]
    L.orc_nwt_forward.argtypes = [C.c_void_p, u64p, C.c_size_t, C.c_size_t]
    L.orc_nwt_backward.argtypes = [C.c_void_p, u64p, C.c_size_t, C.c_size_t]
    L.orc_nwt_forward_map.argtypes = [C.c_void_p, u64p, u32p, C.c_size_t]
    L.orc_nwt_backward_map.argtypes = [C.c_void_p, u64p, u32p, C.c_size_t]
    for f in ("orc_add_rns_poly", "orc_sub_rns_poly", "orc_multiply_rns_poly", "orc_multiply_scalar_rns_poly"):
        getattr(L, f).argtypes = [C.c_void_p, u64p, u64p, u64p, C.c_size_t, C.c_size_t]
    L.orc_negate_rns_poly.argtypes = [C.c_void_p, u64p, u64p, C.c_size_t, C.c_size_t]
    L.orc_multiply_and_add_rns_poly.argtypes = [C.c_void_p, u64p, u64p, u64p, u64p, C.c_size_t, C.c_size_t]
    L.orc_tensor_prod_2x2.argtypes = [C.c_void_p, u64p, u64p, u64p, C.c_size_t]
    L.orc_tensor_square_2x2.argtypes = [C.c_void_p, u64p, u64p, C.c_size_t]
    L.orc_bconv.argtypes = [u64p, C.c_size_t, u64p, C.c_size_t, u64p, u64p, C.c_size_t]
    L.orc_tool_create.restype = C.c_void_p
    L.orc_tool_create.argtypes = [C.c_void_p, C.c_size_t]
    L.orc_tool_destroy.argtypes = [C.c_void_p]
    L.orc_tool_beta.restype = C.c_size_t
    L.orc_tool_beta.argtypes = [C.c_void_p]
    L.orc_modup.argtypes = [C.c_void_p, u64p, u64p, C.c_int]
    L.orc_key_switch_inner_prod.argtypes = [C.c_void_p, u64p, u64p, C.POINTER(u64p)]
    L.orc_moddown_from_ntt.argtypes = [C.c_void_p, u64p, u64p, C.c_int]
    L.orc_keyswitch_inplace.argtypes = [C.c_void_p, u64p, u64p, C.POINTER(u64p), C.c_int]
    L.orc_bconv_hps.argtypes = [u64p, C.c_size_t, u64p, C.c_size_t, u64p, u64p, C.c_size_t]
    L.orc_bfv_add_plain.argtypes = [C.c_void_p, C.c_size_t, u64p, u64p, C.c_uint64, C.c_int]
    L.orc_bgv_lift_plain.argtypes = [C.c_void_p, C.c_size_t, u64p, u64p]
    L.orc_bfv_multiply_plain.argtypes = [C.c_void_p, C.c_size_t, u64p, C.c_size_t, u64p, C.c_uint64]
    L.orc_hoisting.argtypes = [C.c_void_p, u64p, u32p, C.c_size_t, C.POINTER(C.POINTER(u64p)), C.c_int]
    L.orc_hoisting_weighted.argtypes = [C.c_void_p, u64p, u32p, C.c_size_t, C.POINTER(C.POINTER(u64p)), C.POINTER(u64p), C.c_int]
    L.orc_rescale_ntt.argtypes = [C.c_void_p, u64p, C.c_size_t, u64p]
    L.orc_set_threads.argtypes = [C.c_int]
    L.orc_gemm_mod.argtypes = [C.c_uint64, u64p, u64p, u64p, C.c_size_t, C.c_size_t, C.c_size_t]
    L.orc_gemm_mod_ref_quirk.argtypes = [C.c_uint64, u64p, u64p, u64p, C.c_size_t, C.c_size_t, C.c_size_t]
    L.orc_hps_create.restype = C.c_void_p
    L.orc_hps_create.argtypes = [C.c_void_p, C.c_uint64]
    L.orc_hps_destroy.argtypes = [C.c_void_p]
    L.orc_hps_r_size.restype = C.c_size_t
    L.orc_hps_r_size.argtypes = [C.c_void_p]
    L.orc_hps_base.argtypes = [C.c_void_p, u64p]
    L.orc_bfv_multiply_hps.argtypes = [C.c_void_p, u64p, u64p, u64p]
    L.orc_hpsq_create.restype = C.c_void_p
    L.orc_hpsq_create.argtypes = [C.c_void_p, C.c_uint64]
    L.orc_hpsq_destroy.argtypes = [C.c_void_p]
    L.orc_hpsq_create_level.restype = C.c_void_p
    L.orc_hpsq_create_level.argtypes = [C.c_void_p, C.c_uint64, C.c_size_t]
    L.orc_hps_scale_q_ql.argtypes = [C.c_void_p, u64p, u64p]
    L.orc_hps_expand_ql_q.argtypes = [C.c_void_p, u64p, u64p]
    L.orc_bfv_mul_relin_hps_overq_leveled.argtypes = [C.c_void_p, C.c_void_p, u64p, u64p, C.POINTER(u64p), u64p]
    L.orc_keyswitch_bfv_leveled.argtypes = [C.c_void_p, C.c_void_p, u64p, u64p, C.POINTER(u64p)]
    L.orc_hpsq_r_size.restype = C.c_size_t
    L.orc_hpsq_r_size.argtypes = [C.c_void_p]
    L.orc_hpsq_base.argtypes = [C.c_void_p, u64p]
    L.orc_bfv_multiply_hps_overq.argtypes = [C.c_void_p, u64p, u64p, u64p]
    L.orc_behz_create.restype = C.c_void_p
    L.orc_behz_create.argtypes = [C.c_void_p, C.c_uint64]
    L.orc_behz_destroy.argtypes = [C.c_void_p]
    L.orc_behz_bsk_size.restype = C.c_size_t
    L.orc_behz_bsk_size.argtypes = [C.c_void_p]
    L.orc_behz_base.argtypes = [C.c_void_p, u64p]
    L.orc_bfv_multiply_behz.argtypes = [C.c_void_p, u64p, u64p, u64p]
    L.orc_set_threads.restype = None
    L.orc_moddown.argtypes = [C.c_void_p, u64p, u64p, C.c_int]
    L.orc_bconv_behz_var1.argtypes = [u64p, C.c_size_t, u64p, C.c_size_t, u64p, u64p, C.c_size_t]
    L.orc_exact_convert_array.argtypes = [u64p, C.c_size_t, C.c_uint64, u64p, u64p, C.c_size_t]
    for f in ("orc_behz_fastbconv_m_tilde", "orc_behz_sm_mrq", "orc_behz_fastbconv_sk", "orc_hps_scale_round_qr_r",
              "orc_hpsq_scale_round_qlrl_ql", "orc_hpsq_expand_add_to_ct"):
        getattr(L, f).argtypes = [C.c_void_p, u64p, u64p]
    L.orc_behz_fast_floor.argtypes = [C.c_void_p, u64p, u64p, u64p]
    L.orc_tool_set_plain_modulus.restype = C.c_int
    L.orc_tool_set_plain_modulus.argtypes = [C.c_void_p, C.c_uint64]
    L.orc_mod_t_divide_q_last_ntt.argtypes = [C.c_void_p, u64p, C.c_size_t, u64p]
    L.orc_divide_and_round_q_last.argtypes = [C.c_void_p, u64p, C.c_size_t, u64p]
    L.orc_galois_ntt_table.argtypes = [C.c_int, C.c_uint32, u32p]
    L.orc_apply_galois_ntt.argtypes = [u64p, u64p, u32p, C.c_size_t, C.c_size_t]
    L.orc_apply_galois_coeff.argtypes = [C.c_void_p, u64p, u64p, C.c_uint32, C.c_size_t, C.c_size_t]
    L.orc_gen_kswitch_key.argtypes = [C.c_void_p, u64p, u64p, u64p, u64p, u64p]
    if path is None:
        _LIB = L
    return L


def _p(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u64p)


def _p32(a):
    assert a.dtype == np.uint32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u32p)


def is_prime(v):
    return bool(lib().orc_is_prime(int(v)))


def get_primes(n, bit_size, count):
    out = np.zeros(count, dtype=np.uint64)
    if lib().orc_get_primes(n, bit_size, count, _p(out)):
        raise ValueError("failed to find enough qualifying primes")
    return out


def coeff_modulus_create(n, bit_sizes):
    bits = (C.c_int * len(bit_sizes))(*bit_sizes)
    out = np.zeros(len(bit_sizes), dtype=np.uint64)
    if lib().orc_coeff_modulus_create(n, bits, len(bit_sizes), _p(out)):
        raise ValueError("failed to find enough qualifying primes")
    return out


def const_ratio(q):
    r = np.zeros(2, dtype=np.uint64)
    lib().orc_const_ratio(int(q), _p(r))
    return int(r[0]), int(r[1])


def minimal_primitive_root(degree, q):
    r = C.c_uint64(0)
    if lib().orc_minimal_primitive_root(int(degree), int(q), C.byref(r)):
        raise ValueError("no primitive root")
    return r.value


def compute_shoup(w, q):
    return lib().orc_compute_shoup(int(w), int(q))


def ntt_tables(log_n, q):
    n = 1 << log_n
    tw, tws, itw, itws = (np.zeros(n, dtype=np.uint64) for _ in range(4))
    ni, nis = C.c_uint64(0), C.c_uint64(0)
    if lib().orc_ntt_tables(log_n, int(q), _p(tw), _p(tws), _p(itw), _p(itws), C.byref(ni), C.byref(nis)):
        raise ValueError("invalid modulus")
    return tw, tws, itw, itws, ni.value, nis.value


class Ctx:
    """Tables for a QP chain (mirrors the DNTTTable a PhantomContext uploads, context.cu:170-183)."""

    def __init__(self, log_n, primes_qp, size_p, libpath=None):
        self.L = lib(libpath)
        self.log_n = log_n
        self.n = 1 << log_n
        self.primes = np.ascontiguousarray(primes_qp, dtype=np.uint64)
        self.size_qp = len(self.primes)
        self.size_p = size_p
        self.size_q = self.size_qp - size_p
        self.h = self.L.orc_ctx_create(log_n, _p(self.primes), self.size_qp, size_p)
        if not self.h:
            raise ValueError("orc_ctx_create failed")

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_ctx_destroy(self.h)
            self.h = None

    def twiddle(self, prime_idx, which):
        ptr = self.L.orc_ctx_twiddle(self.h, prime_idx, which)
        return np.ctypeslib.as_array(ptr, shape=(self.n,)).copy()

    def n_inv(self, prime_idx):
        return self.L.orc_ctx_n_inv(self.h, prime_idx)

    # transforms (in place on a copy, returns the copy)
    def nwt_forward(self, data, limbs, start_idx=0):
        d = np.array(data, dtype=np.uint64, copy=True).reshape(-1)
        self.L.orc_nwt_forward(self.h, _p(d), limbs, start_idx)
        return d.reshape(np.shape(data))

    def nwt_backward(self, data, limbs, start_idx=0):
        d = np.array(data, dtype=np.uint64, copy=True).reshape(-1)
        self.L.orc_nwt_backward(self.h, _p(d), limbs, start_idx)
        return d.reshape(np.shape(data))

    def nwt_forward_map(self, data, prime_idx):
        d = np.array(data, dtype=np.uint64, copy=True).reshape(-1)
        m = np.ascontiguousarray(prime_idx, dtype=np.uint32)
        self.L.orc_nwt_forward_map(self.h, _p(d), _p32(m), len(m))
        return d.reshape(np.shape(data))

    def nwt_backward_map(self, data, prime_idx):
        d = np.array(data, dtype=np.uint64, copy=True).reshape(-1)
        m = np.ascontiguousarray(prime_idx, dtype=np.uint32)
        self.L.orc_nwt_backward_map(self.h, _p(d), _p32(m), len(m))
        return d.reshape(np.shape(data))

    def _bin(self, fn, a, b, limbs, start_idx):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        r = np.zeros(limbs * self.n, dtype=np.uint64)
        fn(self.h, _p(a.reshape(-1)), _p(b.reshape(-1)), _p(r), limbs, start_idx)
        return r.reshape(limbs, self.n)

    def add(self, a, b, limbs, start_idx=0):
        return self._bin(self.L.orc_add_rns_poly, a, b, limbs, start_idx)

    def sub(self, a, b, limbs, start_idx=0):
        return self._bin(self.L.orc_sub_rns_poly, a, b, limbs, start_idx)

    def multiply(self, a, b, limbs, start_idx=0):
        return self._bin(self.L.orc_multiply_rns_poly, a, b, limbs, start_idx)

    def multiply_scalar(self, a, scalar, limbs, start_idx=0):
        return self._bin(self.L.orc_multiply_scalar_rns_poly, a, scalar, limbs, start_idx)

    def negate(self, a, limbs, start_idx=0):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        r = np.zeros(limbs * self.n, dtype=np.uint64)
        self.L.orc_negate_rns_poly(self.h, _p(a.reshape(-1)), _p(r), limbs, start_idx)
        return r.reshape(limbs, self.n)

    def multiply_and_add(self, a, b, d, limbs, start_idx=0):
        a, b, d = (np.ascontiguousarray(x, dtype=np.uint64).reshape(-1) for x in (a, b, d))
        r = np.zeros(limbs * self.n, dtype=np.uint64)
        self.L.orc_multiply_and_add_rns_poly(self.h, _p(a), _p(b), _p(d), _p(r), limbs, start_idx)
        return r.reshape(limbs, self.n)

    def tensor_prod_2x2(self, op1, op2, limbs):
        op1, op2 = (np.ascontiguousarray(x, dtype=np.uint64).reshape(-1) for x in (op1, op2))
        r = np.zeros(3 * limbs * self.n, dtype=np.uint64)
        self.L.orc_tensor_prod_2x2(self.h, _p(op1), _p(op2), _p(r), limbs)
        return r.reshape(3, limbs, self.n)

    def tensor_square_2x2(self, op, limbs):
        op = np.ascontiguousarray(op, dtype=np.uint64).reshape(-1)
        r = np.zeros(3 * limbs * self.n, dtype=np.uint64)
        self.L.orc_tensor_square_2x2(self.h, _p(op), _p(r), limbs)
        return r.reshape(3, limbs, self.n)

    def apply_galois_coeff(self, src, galois_elt, limbs, start_idx=0):
        src = np.ascontiguousarray(src, dtype=np.uint64).reshape(-1)
        r = np.zeros(limbs * self.n, dtype=np.uint64)
        self.L.orc_apply_galois_coeff(self.h, _p(src), _p(r), galois_elt, limbs, start_idx)
        return r.reshape(limbs, self.n)

    def bfv_add_plain(self, ct0, plain, t, subtract=False):
        """multiply_{add,sub}_plain_with_scaling_variant (src/scalingvariant.cu:10-60) on c0 [Ql][N]."""
        ct0 = np.array(ct0, dtype=np.uint64, copy=True)
        ql = ct0.shape[0]
        flat = ct0.reshape(-1)
        self.L.orc_bfv_add_plain(self.h, ql, _p(flat), _p(np.ascontiguousarray(plain, dtype=np.uint64).reshape(-1)), int(t), int(subtract))
        return flat.reshape(ql, self.n)

    def bgv_lift_plain(self, plain, ql):
        out = np.zeros(ql * self.n, dtype=np.uint64)
        self.L.orc_bgv_lift_plain(self.h, ql, _p(np.ascontiguousarray(plain, dtype=np.uint64).reshape(-1)), _p(out))
        return out.reshape(ql, self.n)

    def bfv_multiply_plain(self, ct, plain, t):
        ct = np.array(ct, dtype=np.uint64, copy=True)
        size, ql = ct.shape[0], ct.shape[1]
        flat = ct.reshape(-1)
        self.L.orc_bfv_multiply_plain(self.h, ql, _p(flat), size, _p(np.ascontiguousarray(plain, dtype=np.uint64).reshape(-1)), int(t))
        return flat.reshape(size, ql, self.n)

    def gen_kswitch_key(self, sk_ntt, new_key_ntt, a_ntt, e_ntt):
        dnum = self.size_q // self.size_p
        sk, nk, a, e = (np.ascontiguousarray(x, dtype=np.uint64).reshape(-1) for x in (sk_ntt, new_key_ntt, a_ntt, e_ntt))
        evk = np.zeros(dnum * 2 * self.size_qp * self.n, dtype=np.uint64)
        self.L.orc_gen_kswitch_key(self.h, _p(sk), _p(nk), _p(a), _p(e), _p(evk))
        return evk.reshape(dnum, 2, self.size_qp, self.n)


def bconv(ibase, obase, src, n):
    ibase = np.ascontiguousarray(ibase, dtype=np.uint64)
    obase = np.ascontiguousarray(obase, dtype=np.uint64)
    src = np.ascontiguousarray(src, dtype=np.uint64).reshape(-1)
    dst = np.zeros(len(obase) * n, dtype=np.uint64)
    lib().orc_bconv(_p(ibase), len(ibase), _p(obase), len(obase), _p(src), _p(dst), n)
    return dst.reshape(len(obase), n)


def bconv_behz_var1(ibase, obase, src, n):
    """DBaseConverter::bConv_BEHZ_var1 (src/rns_bconv.cu:231-246) for arbitrary bases (prime output moduli)."""
    ib = np.array([int(q) for q in ibase], dtype=np.uint64)
    ob = np.array([int(q) for q in obase], dtype=np.uint64)
    src = np.ascontiguousarray(src, dtype=np.uint64).reshape(-1)
    dst = np.zeros(len(ob) * n, dtype=np.uint64)
    lib().orc_bconv_behz_var1(_p(ib), len(ib), _p(ob), len(ob), _p(src), _p(dst), n)
    return dst.reshape(len(ob), n)


def exact_convert_array(ibase, t, src, n):
    """DBaseConverter::exact_convert_array (src/rns_bconv.cu:374-431): [ibase][N] -> [N] modulo t."""
    ib = np.array([int(q) for q in ibase], dtype=np.uint64)
    src = np.ascontiguousarray(src, dtype=np.uint64).reshape(-1)
    dst = np.zeros(n, dtype=np.uint64)
    lib().orc_exact_convert_array(_p(ib), len(ib), int(t), _p(src), _p(dst), n)
    return dst


def bconv_hps(ibase, obase, src, n):
    """DBaseConverter::bConv_HPS (src/rns_bconv.cu:248-372) for arbitrary bases."""
    ib = np.array([int(q) for q in ibase], dtype=np.uint64)
    ob = np.array([int(q) for q in obase], dtype=np.uint64)
    src = np.ascontiguousarray(src, dtype=np.uint64).reshape(-1)
    dst = np.zeros(len(ob) * n, dtype=np.uint64)
    lib().orc_bconv_hps(_p(ib), len(ib), _p(ob), len(ob), _p(src), _p(dst), n)
    return dst.reshape(len(ob), n)


class Behz:
    """BFV multiply, BEHZ variant, at the top data level (src/evaluate.cu:404-548)."""

    def __init__(self, ctx, plain_t):
        self.ctx, self.L = ctx, ctx.L
        self.h = self.L.orc_behz_create(ctx.h, int(plain_t))
        if not self.h:
            raise ValueError("cannot set up the BEHZ bases")
        self.size_bsk = self.L.orc_behz_bsk_size(self.h)
        bsk = np.zeros(self.size_bsk, dtype=np.uint64)
        self.L.orc_behz_base(self.h, _p(bsk))
        self.bsk = [int(v) for v in bsk]

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_behz_destroy(self.h)
            self.h = None

    def multiply(self, ct1, ct2):
        c = self.ctx
        a = np.ascontiguousarray(ct1, dtype=np.uint64).reshape(-1)
        b = np.ascontiguousarray(ct2, dtype=np.uint64).reshape(-1)
        out = np.zeros(3 * c.size_q * c.n, dtype=np.uint64)
        self.L.orc_bfv_multiply_behz(self.h, _p(a), _p(b), _p(out))
        return out.reshape(3, c.size_q, c.n)

    def _step(self, fn, src, out_limbs):
        src = np.ascontiguousarray(src, dtype=np.uint64).reshape(-1)
        dst = np.zeros(out_limbs * self.ctx.n, dtype=np.uint64)
        fn(self.h, _p(src), _p(dst))
        return dst.reshape(out_limbs, self.ctx.n)

    def fastbconv_m_tilde(self, src):
        """DRNSTool::fastbconv_m_tilde (src/rns.cu:1249-1278): [Q][N] -> [Bsk + 1][N], last limb modulo m_tilde."""
        return self._step(self.L.orc_behz_fastbconv_m_tilde, src, self.size_bsk + 1)

    def sm_mrq(self, src):
        """DRNSTool::sm_mrq (src/rns.cu:1290-1338): [Bsk + 1][N] -> [Bsk][N]."""
        return self._step(self.L.orc_behz_sm_mrq, src, self.size_bsk)

    def fast_floor(self, in_q, in_bsk):
        """DRNSTool::fast_floor (src/rns.cu:1394-1419): ([Q][N], [Bsk][N]) -> [Bsk][N]."""
        a = np.ascontiguousarray(in_q, dtype=np.uint64).reshape(-1)
        b = np.ascontiguousarray(in_bsk, dtype=np.uint64).reshape(-1)
        dst = np.zeros(self.size_bsk * self.ctx.n, dtype=np.uint64)
        self.L.orc_behz_fast_floor(self.h, _p(a), _p(b), _p(dst))
        return dst.reshape(self.size_bsk, self.ctx.n)

    def fastbconv_sk(self, in_bsk):
        """DRNSTool::fastbconv_sk (src/rns.cu:1470-1510): [Bsk][N] -> [Q][N]."""
        return self._step(self.L.orc_behz_fastbconv_sk, in_bsk, self.ctx.size_q)


def gemm_mod(q, A, B, quirk=False):
    """C = A @ B mod q (exact), or with the reference benchmark kernel's dropped carries (quirk=True)."""
    L = lib()
    A = np.ascontiguousarray(A, dtype=np.uint64)
    B = np.ascontiguousarray(B, dtype=np.uint64)
    m, k = A.shape
    n = B.shape[1]
    out = np.zeros((m, n), dtype=np.uint64)
    (L.orc_gemm_mod_ref_quirk if quirk else L.orc_gemm_mod)(int(q), _p(A.reshape(-1)), _p(B.reshape(-1)), _p(out.reshape(-1)), m, n, k)
    return out


class Hps:
    """BFV multiply, HPS variant (mul_tech_type::hps), at the top data level (src/evaluate.cu:674-818)."""

    def __init__(self, ctx, plain_t):
        self.ctx, self.L = ctx, ctx.L
        self.h = self.L.orc_hps_create(ctx.h, int(plain_t))
        if not self.h:
            raise ValueError("cannot set up the HPS bases")
        self.size_r = self.L.orc_hps_r_size(self.h)
        r = np.zeros(self.size_r, dtype=np.uint64)
        self.L.orc_hps_base(self.h, _p(r))
        self.r = [int(v) for v in r]

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_hps_destroy(self.h)
            self.h = None

    def multiply(self, ct1, ct2):
        c = self.ctx
        a = np.ascontiguousarray(ct1, dtype=np.uint64).reshape(-1)
        b = np.ascontiguousarray(ct2, dtype=np.uint64).reshape(-1)
        out = np.zeros(3 * c.size_q * c.n, dtype=np.uint64)
        self.L.orc_bfv_multiply_hps(self.h, _p(a), _p(b), _p(out))
        return out.reshape(3, c.size_q, c.n)

    def scale_round_qr_r(self, src):
        """DRNSTool::scaleAndRound_HPS_QR_R (src/rns.cu:1700-1746): [Q + R][N] -> [R][N]."""
        src = np.ascontiguousarray(src, dtype=np.uint64).reshape(-1)
        dst = np.zeros(self.size_r * self.ctx.n, dtype=np.uint64)
        self.L.orc_hps_scale_round_qr_r(self.h, _p(src), _p(dst))
        return dst.reshape(self.size_r, self.ctx.n)


class HpsOverQ:
    """BFV multiply, hps_overq variant (mul_tech_type::hps_overq; src/evaluate.cu:674-818), and with size_ql < size_Q
    the hps_overq_leveled form with size_Q - size_ql levels dropped (src/rns.cu:897-975)."""

    def __init__(self, ctx, plain_t, size_ql=None):
        self.ctx, self.L = ctx, ctx.L
        self.size_ql = ctx.size_q if size_ql is None else int(size_ql)
        self.h = self.L.orc_hpsq_create_level(ctx.h, int(plain_t), self.size_ql)
        if not self.h:
            raise ValueError("cannot set up the HPS bases")
        self.size_r = self.L.orc_hpsq_r_size(self.h)
        r = np.zeros(self.size_r, dtype=np.uint64)
        self.L.orc_hpsq_base(self.h, _p(r))
        self.r = [int(v) for v in r]

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_hpsq_destroy(self.h)
            self.h = None

    def multiply(self, ct1, ct2):
        """Operands and result over the full base Q; `ct2 is ct1` selects the reference's squaring path."""
        c = self.ctx
        a = np.ascontiguousarray(ct1, dtype=np.uint64).reshape(-1)
        b = a if ct2 is ct1 else np.ascontiguousarray(ct2, dtype=np.uint64).reshape(-1)
        out = np.zeros(3 * c.size_q * c.n, dtype=np.uint64)
        self.L.orc_bfv_multiply_hps_overq(self.h, _p(a), _p(b), _p(out))
        return out.reshape(3, c.size_q, c.n)

    def scale_q_ql(self, src):
        """scaleAndRound_HPS_Q_Ql (src/rns.cu:1798-1808): [Q][N] -> [Ql][N]."""
        src = np.ascontiguousarray(src, dtype=np.uint64).reshape(-1)
        dst = np.zeros(self.size_ql * self.ctx.n, dtype=np.uint64)
        self.L.orc_hps_scale_q_ql(self.h, _p(src), _p(dst))
        return dst.reshape(self.size_ql, self.ctx.n)

    def scale_round_qlrl_ql(self, src):
        """DRNSTool::scaleAndRound_HPS_QlRl_Ql (src/rns.cu:1748-1796): [Ql + Rl][N] -> [Ql][N]."""
        src = np.ascontiguousarray(src, dtype=np.uint64).reshape(-1)
        dst = np.zeros(self.size_ql * self.ctx.n, dtype=np.uint64)
        self.L.orc_hpsq_scale_round_qlrl_ql(self.h, _p(src), _p(dst))
        return dst.reshape(self.size_ql, self.ctx.n)

    def expand_add_to_ct(self, dst, src):
        """DRNSTool::ExpandCRTBasis_Ql_Q_add_to_ct (src/rns.cu:1838-1858): dst [Ql][N] += src [Ql][N] * prod(dropped primes)."""
        dst = np.array(dst, dtype=np.uint64, copy=True).reshape(-1)
        src = np.ascontiguousarray(src, dtype=np.uint64).reshape(-1)
        self.L.orc_hpsq_expand_add_to_ct(self.h, _p(src), _p(dst))
        return dst.reshape(self.size_ql, self.ctx.n)

    def expand_ql_q(self, src):
        """ExpandCRTBasis_Ql_Q (src/rns.cu:1810-1836): [Ql][N] -> [Q][N]."""
        src = np.ascontiguousarray(src, dtype=np.uint64).reshape(-1)
        dst = np.zeros(self.ctx.size_q * self.ctx.n, dtype=np.uint64)
        self.L.orc_hps_expand_ql_q(self.h, _p(src), _p(dst))
        return dst.reshape(self.ctx.size_q, self.ctx.n)

    def mul_relin_leveled(self, tool, ct1, ct2, evks):
        """bfv_mul_relin_hps with levels dropped (src/evaluate.cu:822-1027); `ct2 is ct1` is the squaring path."""
        c = self.ctx
        a = np.ascontiguousarray(ct1, dtype=np.uint64).reshape(-1)
        b = a if ct2 is ct1 else np.ascontiguousarray(ct2, dtype=np.uint64).reshape(-1)
        out = np.zeros(2 * c.size_q * c.n, dtype=np.uint64)
        arr, keep = tool._evk_ptrs(evks)
        self.L.orc_bfv_mul_relin_hps_overq_leveled(tool.h, self.h, _p(a), _p(b), arr, _p(out))
        return out.reshape(2, c.size_q, c.n)

    def keyswitch_leveled(self, tool, ct, c2, evks):
        """BFV key switch with levels dropped (src/eval_key_switch.cu:142-147, 170-175); tool = Tool(ctx, size_ql)."""
        ct = np.array(ct, dtype=np.uint64, copy=True).reshape(-1)
        c2 = np.ascontiguousarray(c2, dtype=np.uint64).reshape(-1)
        arr, keep = tool._evk_ptrs(evks)
        self.L.orc_keyswitch_bfv_leveled(tool.h, self.h, _p(ct), _p(c2), arr)
        return ct.reshape(2, self.ctx.size_q, self.ctx.n)


class Tool:
    """DRNSTool of one data level (src/rns.cu:11-200): size_ql current data limbs."""

    def __init__(self, ctx, size_ql):
        self.ctx = ctx
        self.L = ctx.L
        self.size_ql = size_ql
        self.size_qlp = size_ql + ctx.size_p
        self.n = ctx.n
        self.h = self.L.orc_tool_create(ctx.h, size_ql)
        self.beta = self.L.orc_tool_beta(self.h) if ctx.size_p else 0

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_tool_destroy(self.h)
            self.h = None

    def _evk_ptrs(self, evks):
        keep = [np.ascontiguousarray(e, dtype=np.uint64).reshape(-1) for e in evks]
        arr = (u64p * len(keep))(*[_p(k) for k in keep])
        return arr, keep

    def modup(self, cks, scheme):
        cks = np.ascontiguousarray(cks, dtype=np.uint64).reshape(-1)
        dst = np.zeros(self.beta * self.size_qlp * self.n, dtype=np.uint64)
        self.L.orc_modup(self.h, _p(dst), _p(cks), scheme)
        return dst.reshape(self.beta, self.size_qlp, self.n)

    def key_switch_inner_prod(self, t_mod_up, evks):
        t = np.ascontiguousarray(t_mod_up, dtype=np.uint64).reshape(-1)
        cx = np.zeros(2 * self.size_qlp * self.n, dtype=np.uint64)
        arr, keep = self._evk_ptrs(evks)
        self.L.orc_key_switch_inner_prod(self.h, _p(cx), _p(t), arr)
        return cx.reshape(2, self.size_qlp, self.n)

    def moddown_from_ntt(self, cx, scheme):
        cx = np.array(cx, dtype=np.uint64, copy=True).reshape(-1)
        ct = np.zeros(self.size_ql * self.n, dtype=np.uint64)
        self.L.orc_moddown_from_ntt(self.h, _p(ct), _p(cx), scheme)
        return ct.reshape(self.size_ql, self.n)

    def moddown(self, cx, scheme):
        """DRNSTool::moddown (src/rns_bconv.cu:712-761): BFV input in coefficient form, CKKS / BGV in NTT form."""
        cx = np.array(cx, dtype=np.uint64, copy=True).reshape(-1)
        ct = np.zeros(self.size_ql * self.n, dtype=np.uint64)
        self.L.orc_moddown(self.h, _p(ct), _p(cx), scheme)
        return ct.reshape(self.size_ql, self.n)

    def keyswitch_inplace(self, ct, c2, evks, scheme):
        ct = np.array(ct, dtype=np.uint64, copy=True).reshape(-1)
        c2 = np.ascontiguousarray(c2, dtype=np.uint64).reshape(-1)
        arr, keep = self._evk_ptrs(evks)
        self.L.orc_keyswitch_inplace(self.h, _p(ct), _p(c2), arr, scheme)
        return ct.reshape(2, self.size_ql, self.n)

    def hoisting(self, ct, galois_elts, glk, scheme):
        """glk[e] = list of beta keys ([2][QP][N]) of Galois element e; returns sum_e rotate_e(ct)."""
        ct = np.array(ct, dtype=np.uint64, copy=True).reshape(-1)
        elts = np.ascontiguousarray(galois_elts, dtype=np.uint32)
        keep, tabs = [], []
        for keys in glk:
            arr, k = self._evk_ptrs(keys)
            keep.append(k)
            tabs.append(arr)
        outer = (C.POINTER(u64p) * len(tabs))(*[C.cast(a, C.POINTER(u64p)) for a in tabs])
        self.L.orc_hoisting(self.h, _p(ct), _p32(elts), len(elts), outer, scheme)
        return ct.reshape(2, self.size_ql, self.n)

    def hoisting_weighted(self, ct, galois_elts, glk, weights, scheme):
        """sum_e w_e (.) rotate_e(ct); weights[e] is [QlP][N] in NTT form; glk[e] may be None for element 1."""
        ct = np.array(ct, dtype=np.uint64, copy=True).reshape(-1)
        elts = np.ascontiguousarray(galois_elts, dtype=np.uint32)
        keep, tabs = [], []
        for keys in glk:
            if keys is None:
                tabs.append(None)
                continue
            arr, k = self._evk_ptrs(keys)
            keep.append(k)
            tabs.append(arr)
        outer = (C.POINTER(u64p) * len(tabs))(*[C.cast(a, C.POINTER(u64p)) if a is not None else C.POINTER(u64p)() for a in tabs])
        ws = [np.ascontiguousarray(w, dtype=np.uint64).reshape(-1) for w in weights]
        warr = (u64p * len(ws))(*[_p(w) for w in ws])
        self.L.orc_hoisting_weighted(self.h, _p(ct), _p32(elts), len(elts), outer, warr, scheme)
        return ct.reshape(2, self.size_ql, self.n)

    def hoisting_weighted_bsgs(self, ct, baby_elts, baby_glk, giant_elts, giant_glk, weights, scheme):
        """Baby-step / giant-step form of hoisting_weighted (build-defined, BASELINE config 5), composed from the restated
        reference steps: B_i = hoisting_weighted(ct, baby steps, weights[i]) for every giant step i (hoisting_inplace
        src/evaluate.cu:1670-1866 with plaintext weights), then the giant rotations with ONE shared mod-down:
        (sum_i perm_i(B_i0), sum_{identity i} B_i1) + moddown(sum_{keyed i} key_switch_inner_prod(modup(perm_i(B_i1)), key_i))
        (apply_galois_ntt src/galois.cu:11-39, modup rns_bconv.cu:530-627, inner product eval_key_switch.cu:14-69, mod-down
        rns_bconv.cu:776-828).  weights[i][j] may be None (no such term); keys may be None for element 1."""
        n, ql, qlp = self.n, self.size_ql, self.size_qlp
        primes_ql = [int(self.ctx.primes[i]) for i in range(ql)]
        primes_qlp = primes_ql + [int(self.ctx.primes[self.ctx.size_q + i]) for i in range(self.ctx.size_p)]
        zero = np.zeros((qlp, n), dtype=np.uint64)
        ct = np.ascontiguousarray(ct, dtype=np.uint64).reshape(2, ql, n)
        r0 = np.zeros((ql, n), dtype=np.uint64)
        r1 = np.zeros((ql, n), dtype=np.uint64)
        cx = np.zeros((2, qlp, n), dtype=np.uint64)

        def add(a, b, primes):
            out = np.empty_like(a)
            for j, q in enumerate(primes):
                s_ = a[j] + b[j]                       # both below 2^61: no wrap
                out[j] = np.where(s_ >= np.uint64(q), s_ - np.uint64(q), s_)
            return out

        keyed = False
        for i, ge in enumerate(giant_elts):
            Bi = self.hoisting_weighted(ct, baby_elts, baby_glk, [w if w is not None else zero for w in weights[i]], scheme)
            table = galois_ntt_table(self.ctx.log_n, int(ge))
            r0 = add(r0, apply_galois_ntt(Bi[0], table, n, ql), primes_ql)
            if int(ge) == 1:
                r1 = add(r1, Bi[1], primes_ql)
                continue
            keyed = True
            mu = self.modup(apply_galois_ntt(Bi[1], table, n, ql), scheme)
            ip = self.key_switch_inner_prod(mu, giant_glk[i])
            cx = np.stack([add(cx[p], ip[p], primes_qlp) for p in range(2)])
        if keyed:
            r0 = add(r0, self.moddown_from_ntt(cx[0], scheme), primes_ql)
            r1 = add(r1, self.moddown_from_ntt(cx[1], scheme), primes_ql)
        return np.stack([r0, r1])

    def set_plain_modulus(self, t):
        """BGV constants of the tool (rns.cu:196-285)."""
        if self.L.orc_tool_set_plain_modulus(self.h, int(t)) != 0:
            raise ValueError("plain modulus not invertible modulo the chain")
        self.plain_t = int(t)
        return self

    def mod_t_divide_q_last_ntt(self, src, cipher_size):
        src = np.array(src, dtype=np.uint64, copy=True).reshape(-1)
        dst = np.zeros(cipher_size * (self.size_ql - 1) * self.n, dtype=np.uint64)
        self.L.orc_mod_t_divide_q_last_ntt(self.h, _p(src), cipher_size, _p(dst))
        return dst.reshape(cipher_size, self.size_ql - 1, self.n)

    def rescale_ntt(self, src, cipher_size):
        src = np.array(src, dtype=np.uint64, copy=True).reshape(-1)
        dst = np.zeros(cipher_size * (self.size_ql - 1) * self.n, dtype=np.uint64)
        self.L.orc_rescale_ntt(self.h, _p(src), cipher_size, _p(dst))
        return dst.reshape(cipher_size, self.size_ql - 1, self.n)

    def divide_and_round_q_last(self, src, cipher_size):
        src = np.ascontiguousarray(src, dtype=np.uint64).reshape(-1)
        dst = np.zeros(cipher_size * (self.size_ql - 1) * self.n, dtype=np.uint64)
        self.L.orc_divide_and_round_q_last(self.h, _p(src), cipher_size, _p(dst))
        return dst.reshape(cipher_size, self.size_ql - 1, self.n)


def galois_ntt_table(log_n, galois_elt):
    t = np.zeros(1 << log_n, dtype=np.uint32)
    lib().orc_galois_ntt_table(log_n, galois_elt, _p32(t))
    return t


def apply_galois_ntt(src, table, n, limbs):
    src = np.ascontiguousarray(src, dtype=np.uint64).reshape(-1)
    dst = np.zeros(limbs * n, dtype=np.uint64)
    lib().orc_apply_galois_ntt(_p(src), _p(dst), _p32(np.ascontiguousarray(table, dtype=np.uint32)), n, limbs)
    return dst.reshape(limbs, n)
