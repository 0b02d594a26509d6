"""Host-side mirror of the reference's hot-path objects over the C ABI.

Names follow the reference (PhantomContext / DNTTTable launchers in include/ntt.cuh:157-226,
DRNSTool methods in include/rns.cuh:156-205, key_switch_inner_prod / keyswitch_inplace in
include/evaluate.cuh:18-32).  torch is used only as the owner of device memory and of the HIP
stream; every call goes straight to libphantom_amd.so with raw device pointers.
Polynomials are torch.int64 CUDA tensors whose bit patterns are the uint64 residues, limb-major
[limb][coeff] (a ciphertext is [poly][limb][coeff], include/ciphertext.h:15-25).
"""
import ctypes as C
from enum import IntEnum

import numpy as np
import torch

from . import lib as _lib


class scheme_type(IntEnum):  # include/host/encryptionparams.h:19-27
    none = 0
    bfv = 1
    ckks = 2
    bgv = 3


def to_device(arr, device="cuda:0"):
    """numpy uint64 array -> int64 device tensor with the same bits."""
    a = np.ascontiguousarray(arr, dtype=np.uint64)
    return torch.from_numpy(a.view(np.int64)).to(device)


def to_host(t):
    """int64 device tensor -> numpy uint64 array with the same bits."""
    return t.detach().cpu().contiguous().numpy().view(np.uint64)


def _ptr(t):
    if t is None:
        return None
    if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.int64):
        raise ValueError("expected a contiguous torch.int64 CUDA tensor")
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def stream_copy_rate(dst, src, iters=10):
    """bytes per second (read + write) of the library's own 16-byte-per-lane device-to-device copy (phantom_amd_bench.h)."""
    rate = C.c_double()
    nbytes = min(dst.numel() * dst.element_size(), src.numel() * src.element_size())
    _lib.check(_lib.load().pha_time_stream_copy(_ptr(dst), _ptr(src), nbytes - nbytes % 16, iters, _stream(), C.byref(rate)))
    return rate.value


def stream_rate(dst, src, mode, nontemporal=False, iters=10):
    """bytes per second (read + written) of the library's streaming calibration kernels (phantom_amd_bench.h: pha_time_stream):
    mode 0 copy src -> dst, 1 read-only, 2 write-only, 3 in-place read-modify-write of dst."""
    rate = C.c_double()
    nbytes = min(dst.numel() * dst.element_size(), src.numel() * src.element_size())
    _lib.check(_lib.load().pha_time_stream(_ptr(dst), _ptr(src), nbytes - nbytes % 16, int(mode), int(bool(nontemporal)), iters, _stream(),
                                           C.byref(rate)))
    return rate.value


def has_tuning():
    """True when the loaded library is the test-only experiments build (PHA_LIB_OVERRIDE=.../libphantom_amd_exp.so)."""
    return hasattr(_lib.load(), "pha_set_tuning")


def set_tuning(key, value):
    """A/B knob of the experiments library (csrc/pha_experiments.h); results never change.  The product library's kernel
    selection is fixed: it does not export the symbol."""
    if not has_tuning():
        raise RuntimeError("pha_set_tuning exists only in libphantom_amd_exp.so (set PHA_LIB_OVERRIDE to it): "
                           "the product library has one plan per launch shape")
    _lib.check(_lib.load().pha_set_tuning(int(key), int(value)))


def set_strict(on):
    """Strict mode of the library (pha_set_strict): entry points check caller-supplied operands for words >= their modulus and
    raise ValueError instead of computing.  Returns the previous state.  Process-wide; default from PHA_STRICT=1."""
    return bool(_lib.load().pha_set_strict(1 if on else 0))


def coeff_modulus_create(poly_modulus_degree, bit_sizes):
    """CoeffModulus::Create (src/host/modulus.cu:82-111)."""
    L = _lib.load()
    bits = (C.c_int * len(bit_sizes))(*bit_sizes)
    out = np.zeros(len(bit_sizes), dtype=np.uint64)
    _lib.check(L.pha_coeff_modulus_create(poly_modulus_degree, bits, len(bit_sizes),
                                          out.ctypes.data_as(_lib.u64p)))
    return out


class PhantomContext:
    """Hot-path state of PhantomContext (src/context.cu:121-232): NTT tables for all QP primes on
    one device plus a lazily built DRNSTool per level."""

    def __init__(self, log_n, coeff_modulus, special_modulus_size, device=0):
        self._L = _lib.load()
        self.log_n = int(log_n)
        self.n = 1 << self.log_n
        self.coeff_modulus = np.ascontiguousarray(coeff_modulus, dtype=np.uint64)
        self.size_QP = len(self.coeff_modulus)
        self.size_P = int(special_modulus_size)
        self.size_Q = self.size_QP - self.size_P
        self.device = torch.device("cuda", device) if not isinstance(device, torch.device) else device
        if not torch.cuda.is_available():
            raise RuntimeError("phantom_fhe_amd needs a HIP device; there is no CPU path")
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._L.pha_context_create(C.byref(h), self.log_n,
                                                  self.coeff_modulus.ctypes.data_as(_lib.u64p),
                                                  self.size_QP, self.size_P, self.device.index or 0))
        self._h = h

    def set_plain_modulus(self, plain_modulus):
        """EncryptionParameters::set_plain_modulus as DRNSTool sees it (src/rns.cu:196-285, BGV constants)."""
        _lib.check(self._L.pha_context_set_plain_modulus(self._h, int(plain_modulus)))
        self.plain_modulus = int(plain_modulus)
        return self

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._L.pha_context_destroy(h)
            self._h = None

    # -- the canonical-operand precondition, checkable (include/phantom_amd.h; csrc/pha_check.hip) --------------------
    def check_canonical(self, data, cms, start=0, size_P_tail=0, polys=1, poly_stride=0):
        """Number of words of data [polys][cms][N] that are >= their limb's modulus (synchronises the stream)."""
        bad = C.c_uint64()
        _lib.check(self._L.pha_check_canonical(self._h, _ptr(data), cms, start, size_P_tail, polys, poly_stride, C.byref(bad), _stream()))
        return bad.value

    def check_canonical_keys(self, size_Ql, keys_ptr, n_keys):
        """The same for the limbs a key switch at level size_Ql reads of n_keys keys (keys_ptr: PhantomRelinKey.public_keys_ptr)."""
        bad = C.c_uint64()
        _lib.check(self._L.pha_check_canonical_keys(self._h, size_Ql, _ptr(keys_ptr), n_keys, C.byref(bad), _stream()))
        return bad.value

    # -- table queries ------------------------------------------------------------------------
    def prime_info(self, idx):
        v, root, ninv = C.c_uint64(), C.c_uint64(), C.c_uint64()
        ratio = (C.c_uint64 * 2)()
        _lib.check(self._L.pha_context_prime_info(self._h, idx, C.byref(v), ratio, C.byref(root), C.byref(ninv)))
        return {"value": v.value, "const_ratio": (ratio[0], ratio[1]), "root": root.value, "n_inv": ninv.value}

    def twiddle_row(self, idx, which):
        out = np.zeros(self.n, dtype=np.uint64)
        _lib.check(self._L.pha_context_download_twiddle(self._h, idx, which, out.ctypes.data_as(_lib.u64p)))
        return out

    def beta(self, size_Ql):
        b = C.c_uint32()
        _lib.check(self._L.pha_tool_beta(self._h, size_Ql, C.byref(b)))
        return b.value

    # -- NTT launchers (include/ntt.cuh:178-226) ------------------------------------------------
    def nwt_2d_radix8_forward_inplace(self, inout, coeff_modulus_size, start_modulus_idx=0):
        _lib.check(self._L.pha_nwt_2d_radix8_forward_inplace(self._h, _ptr(inout), coeff_modulus_size,
                                                             start_modulus_idx, _stream()))

    def nwt_2d_radix8_forward_inplace_include_special_mod(self, inout, cms, start, size_QP, size_P):
        _lib.check(self._L.pha_nwt_2d_radix8_forward_inplace_include_special_mod(
            self._h, _ptr(inout), cms, start, size_QP, size_P, _stream()))

    def nwt_2d_radix8_forward_inplace_include_special_mod_exclude_range(self, inout, cms, start, size_QP,
                                                                       size_P, ex_start, ex_end):
        _lib.check(self._L.pha_nwt_2d_radix8_forward_inplace_include_special_mod_exclude_range(
            self._h, _ptr(inout), cms, start, size_QP, size_P, ex_start, ex_end, _stream()))

    def nwt_2d_radix8_forward_inplace_fuse_moddown(self, ct, cx, bigPInv_mod_q, bigPInv_mod_q_shoup, delta,
                                                   cms, start=0):
        _lib.check(self._L.pha_nwt_2d_radix8_forward_inplace_fuse_moddown(
            self._h, _ptr(ct), _ptr(cx), _ptr(bigPInv_mod_q), _ptr(bigPInv_mod_q_shoup), _ptr(delta), cms,
            start, _stream()))

    def nwt_2d_radix8_backward_inplace(self, inout, coeff_modulus_size, start_modulus_idx=0):
        _lib.check(self._L.pha_nwt_2d_radix8_backward_inplace(self._h, _ptr(inout), coeff_modulus_size,
                                                              start_modulus_idx, _stream()))

    def nwt_2d_radix8_forward_inplace_batched(self, inout, coeff_modulus_size, start_modulus_idx, batch, poly_stride):
        """Extension: the same limbs of `batch` polynomials (poly_stride elements apart) in one launch."""
        _lib.check(self._L.pha_nwt_2d_radix8_forward_inplace_batched(self._h, _ptr(inout), coeff_modulus_size,
                                                                     start_modulus_idx, batch, poly_stride, _stream()))

    def nwt_2d_radix8_backward_inplace_batched(self, inout, coeff_modulus_size, start_modulus_idx, batch, poly_stride):
        _lib.check(self._L.pha_nwt_2d_radix8_backward_inplace_batched(self._h, _ptr(inout), coeff_modulus_size,
                                                                      start_modulus_idx, batch, poly_stride, _stream()))

    def nwt_2d_radix8_backward(self, out, inp, coeff_modulus_size, start_modulus_idx=0):
        _lib.check(self._L.pha_nwt_2d_radix8_backward(self._h, _ptr(out), _ptr(inp), coeff_modulus_size,
                                                      start_modulus_idx, _stream()))

    def nwt_2d_radix8_backward_scale(self, out, inp, cms, start, scale, scale_shoup):
        _lib.check(self._L.pha_nwt_2d_radix8_backward_scale(self._h, _ptr(out), _ptr(inp), cms, start,
                                                            _ptr(scale), _ptr(scale_shoup), _stream()))

    def nwt_2d_radix8_backward_inplace_scale(self, inout, cms, start, scale, scale_shoup):
        """include/ntt.cuh:217 (src/ntt/intt_2d.cu:759-794): inverse NTT in place, every output times the per-limb scale[i]."""
        _lib.check(self._L.pha_nwt_2d_radix8_backward_inplace_scale(self._h, _ptr(inout), cms, start,
                                                                    _ptr(scale), _ptr(scale_shoup), _stream()))

    def nwt_2d_radix8_backward_inplace_include_special_mod(self, inout, cms, start, size_QP, size_P):
        _lib.check(self._L.pha_nwt_2d_radix8_backward_inplace_include_special_mod(
            self._h, _ptr(inout), cms, start, size_QP, size_P, _stream()))

    # -- dyadic kernels (include/polymath.cuh) ------------------------------------------------
    def add_rns_poly(self, a, b, r, cms, mod_start=0):
        _lib.check(self._L.pha_add_rns_poly(self._h, _ptr(a), _ptr(b), _ptr(r), cms, mod_start, _stream()))

    def sub_rns_poly(self, a, b, r, cms, mod_start=0):
        _lib.check(self._L.pha_sub_rns_poly(self._h, _ptr(a), _ptr(b), _ptr(r), cms, mod_start, _stream()))

    def negate_rns_poly(self, a, r, cms, mod_start=0):
        _lib.check(self._L.pha_negate_rns_poly(self._h, _ptr(a), _ptr(r), cms, mod_start, _stream()))

    def multiply_rns_poly(self, a, b, r, cms, mod_start=0):
        _lib.check(self._L.pha_multiply_rns_poly(self._h, _ptr(a), _ptr(b), _ptr(r), cms, mod_start, _stream()))

    def multiply_and_add_rns_poly(self, a, b, d, r, cms, mod_start=0):
        _lib.check(self._L.pha_multiply_and_add_rns_poly(self._h, _ptr(a), _ptr(b), _ptr(d), _ptr(r), cms,
                                                         mod_start, _stream()))

    def multiply_scalar_rns_poly(self, a, scalar, scalar_shoup, r, cms, mod_start=0):
        _lib.check(self._L.pha_multiply_scalar_rns_poly(self._h, _ptr(a), _ptr(scalar), _ptr(scalar_shoup),
                                                        _ptr(r), cms, mod_start, _stream()))

    # -- the rest of polymath.cu (kernels around the hot path: encryption / decryption / plaintext layers) --------
    def add_std_cipher(self, c1, c2, r, cms):
        _lib.check(self._L.pha_add_std_cipher(self._h, _ptr(c1), _ptr(c2), _ptr(r), cms, _stream()))

    def add_and_negate_rns_poly(self, a, b, r, cms, mod_start=0):
        _lib.check(self._L.pha_add_and_negate_rns_poly(self._h, _ptr(a), _ptr(b), _ptr(r), cms, mod_start, _stream()))

    def add_many_rns_poly(self, operands, r, poly_index, cms):
        tab = (C.c_void_p * len(operands))(*[_ptr(o) for o in operands])
        _lib.check(self._L.pha_add_many_rns_poly(self._h, tab, len(operands), _ptr(r), poly_index, cms, _stream()))

    def multiply_uniform_scalar_rns_poly(self, a, scale, r, cms, mod_start=0):
        _lib.check(self._L.pha_multiply_uniform_scalar_rns_poly(self._h, _ptr(a), int(scale), _ptr(r), cms, mod_start, _stream()))

    def multiply_scalar_and_add_rns_poly(self, a, b, scalar, r, cms, mod_start=0):
        _lib.check(self._L.pha_multiply_scalar_and_add_rns_poly(self._h, _ptr(a), _ptr(b), int(scalar), _ptr(r), cms, mod_start, _stream()))

    def multiply_scalar_and_sub_rns_poly(self, a, b, scalar, r, cms, mod_start=0):
        _lib.check(self._L.pha_multiply_scalar_and_sub_rns_poly(self._h, _ptr(a), _ptr(b), int(scalar), _ptr(r), cms, mod_start, _stream()))

    def multiply_and_scale_add_rns_poly(self, a, b, d, scale, r, cms, mod_start=0):
        _lib.check(self._L.pha_multiply_and_scale_add_rns_poly(self._h, _ptr(a), _ptr(b), _ptr(d), int(scale), _ptr(r), cms, mod_start, _stream()))

    def multiply_and_add_negate_rns_poly(self, a, b, d, r, cms, mod_start=0):
        _lib.check(self._L.pha_multiply_and_add_negate_rns_poly(self._h, _ptr(a), _ptr(b), _ptr(d), _ptr(r), cms, mod_start, _stream()))

    def sub_and_scale_rns_poly(self, a, b, scale, scale_shoup, r, cms, mod_start=0):
        _lib.check(self._L.pha_sub_and_scale_rns_poly(self._h, _ptr(a), _ptr(b), _ptr(scale), _ptr(scale_shoup), _ptr(r), cms, mod_start, _stream()))

    def sub_and_scale_single_mod_poly(self, a, b, scale, scale_shoup, modulus, r):
        _lib.check(self._L.pha_sub_and_scale_single_mod_poly(self._h, _ptr(a), _ptr(b), int(scale), int(scale_shoup), int(modulus), _ptr(r), _stream()))

    def bfv_add_timesQ_overt(self, ct, pt, neg_ql_mod_t, neg_ql_mod_t_shoup, t_inv, t_inv_shoup, t, size_Ql, sub=False):
        f = self._L.pha_bfv_sub_timesQ_overt if sub else self._L.pha_bfv_add_timesQ_overt
        _lib.check(f(self._h, _ptr(ct), _ptr(pt), int(neg_ql_mod_t), int(neg_ql_mod_t_shoup), _ptr(t_inv), _ptr(t_inv_shoup), int(t), size_Ql, _stream()))

    def abs_plain_rns_poly(self, operand, threshold, increment, r, cms):
        _lib.check(self._L.pha_abs_plain_rns_poly(self._h, _ptr(operand), int(threshold), _ptr(increment), _ptr(r), cms, _stream()))

    def tensor_prod_mxn_rns_poly(self, op1, m, op2, n_polys, r, cms):
        _lib.check(self._L.pha_tensor_prod_mxn_rns_poly(self._h, _ptr(op1), m, _ptr(op2), n_polys, _ptr(r), m + n_polys - 1, cms, _stream()))

    def multiply_and_negated_add_rns_poly(self, alpha_sk, m_sk, prod_b_mod_q, operand3, r, cms):
        _lib.check(self._L.pha_multiply_and_negated_add_rns_poly(self._h, _ptr(alpha_sk), int(m_sk), _ptr(prod_b_mod_q), _ptr(operand3), _ptr(r), cms, _stream()))

    def bfv_add_plain(self, size_Ql, ct, plain, subtract=False):
        """multiply_{add,sub}_plain_with_scaling_variant (src/scalingvariant.cu:10-60) on ct[0]."""
        _lib.check(self._L.pha_bfv_add_plain(self._h, size_Ql, _ptr(ct), _ptr(plain), int(bool(subtract)), _stream()))

    def bfv_multiply_plain(self, size_Ql, ct, cipher_size, plain):
        """multiply_plain_normal (src/evaluate.cu:1256-1300)."""
        _lib.check(self._L.pha_bfv_multiply_plain(self._h, size_Ql, _ptr(ct), cipher_size, _ptr(plain), _stream()))

    def bgv_lift_plain(self, size_Ql, plain, out):
        """NTT of a plaintext modulo every q_i (the modup_fuse loop of src/evaluate.cu:1150-1154)."""
        _lib.check(self._L.pha_bgv_lift_plain(self._h, size_Ql, _ptr(plain), _ptr(out), _stream()))

    def tensor_prod_2x2_rns_poly_at(self, op1, op2, result, cms, mod_start):
        """tensor_prod_2x2_rns_poly over table rows mod_start .. (the reference's `modulus` pointer: base Bsk / base R callers)."""
        _lib.check(self._L.pha_tensor_prod_2x2_rns_poly_at(self._h, _ptr(op1), _ptr(op2), _ptr(result), cms, mod_start, _stream()))

    def tensor_square_2x2_rns_poly_at(self, op, result, cms, mod_start):
        _lib.check(self._L.pha_tensor_square_2x2_rns_poly_at(self._h, _ptr(op), _ptr(result), cms, mod_start, _stream()))

    def tensor_prod_2x2_rns_poly(self, op1, op2, result, cms):
        _lib.check(self._L.pha_tensor_prod_2x2_rns_poly(self._h, _ptr(op1), _ptr(op2), _ptr(result), cms, _stream()))

    def tensor_square_2x2_rns_poly(self, op, result, cms):
        _lib.check(self._L.pha_tensor_square_2x2_rns_poly(self._h, _ptr(op), _ptr(result), cms, _stream()))

    def add_to_ct(self, ct, cx, size_Ql):
        _lib.check(self._L.pha_add_to_ct(self._h, _ptr(ct), _ptr(cx), size_Ql, _stream()))

    # -- DRNSTool at level size_Ql (include/rns.cuh:156-205) --------------------------------------
    def bconv_P_to_Ql(self, size_Ql, dst, src):
        _lib.check(self._L.pha_bconv_P_to_Ql(self._h, size_Ql, _ptr(dst), _ptr(src), _stream()))

    def modup(self, size_Ql, dst, cks, scheme):
        _lib.check(self._L.pha_modup(self._h, size_Ql, _ptr(dst), _ptr(cks), int(scheme), _stream()))

    def key_switch_inner_prod(self, size_Ql, p_cx, p_t_mod_up, rlk_ptrs):
        """rlk_ptrs: int64 CUDA tensor holding beta device pointers (PhantomRelinKey::public_keys_ptr())."""
        _lib.check(self._L.pha_key_switch_inner_prod(self._h, size_Ql, _ptr(p_cx), _ptr(p_t_mod_up),
                                                     _ptr(rlk_ptrs), _stream()))

    def moddown_from_NTT(self, size_Ql, ct_i, cx_i, scheme):
        _lib.check(self._L.pha_moddown_from_NTT(self._h, size_Ql, _ptr(ct_i), _ptr(cx_i), int(scheme), _stream()))

    def keyswitch_inplace(self, size_Ql, ct, c2, rlk_ptrs, scheme):
        _lib.check(self._L.pha_keyswitch_inplace(self._h, size_Ql, _ptr(ct), _ptr(c2), _ptr(rlk_ptrs),
                                                 int(scheme), _stream()))

    def keyswitch_inplace_batched(self, size_Ql, ct, c2, batch, rlk_ptrs, scheme):
        """`batch` ciphertexts through one set of launches: ct [batch][2][Ql][N] += KS(c2 [batch][Ql][N])."""
        _lib.check(self._L.pha_keyswitch_inplace_batched(self._h, size_Ql, _ptr(ct), _ptr(c2), batch, _ptr(rlk_ptrs),
                                                         int(scheme), _stream()))

    def keyswitch_rescale(self, size_Ql, ct, c2, rlk_ptrs, dst):
        """dst [2][Ql-1][N] = rescale(ct + keyswitch(c2)) in one call (ckks): bit-identical to keyswitch_inplace followed by
        divide_and_round_q_last_ntt, one forward NTT fewer; ct and c2 are only read."""
        _lib.check(self._L.pha_keyswitch_rescale(self._h, size_Ql, _ptr(ct), _ptr(c2), _ptr(rlk_ptrs), _ptr(dst), _stream()))

    def keyswitch_rescale_batched(self, size_Ql, ct, c2, batch, rlk_ptrs, dst):
        _lib.check(self._L.pha_keyswitch_rescale_batched(self._h, size_Ql, _ptr(ct), _ptr(c2), batch, _ptr(rlk_ptrs), _ptr(dst),
                                                         _stream()))

    def tensor_prod_2x2_batched(self, op1, op2, res01, res2, cms, batch):
        _lib.check(self._L.pha_tensor_prod_2x2_batched(self._h, _ptr(op1), _ptr(op2), _ptr(res01), _ptr(res2), cms,
                                                       batch, _stream()))

    def bfv_multiply_behz(self, ct1, ct2, dst):
        """bfv_multiply_behz (src/evaluate.cu:447-548): [2][Q][N] x [2][Q][N] -> [3][Q][N], coefficient form."""
        _lib.check(self._L.pha_bfv_multiply_behz(self._h, _ptr(ct1), _ptr(ct2), _ptr(dst), _stream()))

    def bfv_multiply_hps(self, ct1, ct2, dst):
        """bfv_multiply_hps, mul_tech hps (src/evaluate.cu:674-818): same shapes as bfv_multiply_behz."""
        _lib.check(self._L.pha_bfv_multiply_hps(self._h, _ptr(ct1), _ptr(ct2), _ptr(dst), _stream()))

    def bfv_multiply_hps_overq(self, ct1, ct2, dst):
        """bfv_multiply_hps, mul_tech hps_overq without dropped levels (src/evaluate.cu:674-818); passing the same
        tensor twice takes the reference's squaring shortcut."""
        _lib.check(self._L.pha_bfv_multiply_hps_overq(self._h, _ptr(ct1), _ptr(ct2), _ptr(dst), _stream()))

    def bfv_multiply_hps_overq_leveled(self, size_Ql, ct1, ct2, dst):
        """bfv_multiply_hps under hps_overq_leveled with size_Q - size_Ql levels dropped; buffers over the full base Q."""
        _lib.check(self._L.pha_bfv_multiply_hps_overq_leveled(self._h, size_Ql, _ptr(ct1), _ptr(ct2), _ptr(dst), _stream()))

    def bfv_mul_relin_hps_overq_leveled(self, size_Ql, ct1, ct2, rlk_ptrs, dst):
        """bfv_mul_relin_hps with levels dropped (src/evaluate.cu:822-1027); dst [2][Q][N]."""
        _lib.check(self._L.pha_bfv_mul_relin_hps_overq_leveled(self._h, size_Ql, _ptr(ct1), _ptr(ct2), _ptr(rlk_ptrs), _ptr(dst), _stream()))

    # ---- the DRNSTool steps of the BFV multiplies, one polynomial per call (include/rns.cuh:159-200) ----
    def moddown(self, size_Ql, ct_i, cx_i, scheme):
        """DRNSTool::moddown (src/rns_bconv.cu:712-761): BFV input in coefficient form."""
        _lib.check(self._L.pha_moddown(self._h, size_Ql, _ptr(ct_i), _ptr(cx_i), int(scheme), _stream()))

    def tool_aux_sizes(self, size_Ql):
        """(|Bsk|, |R|, |Rl|) of the level's tool; 0 where the base does not exist at that level."""
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _lib.check(self._L.pha_tool_aux_sizes(self._h, size_Ql, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def fastbconv_m_tilde(self, size_Ql, dst, src):
        _lib.check(self._L.pha_fastbconv_m_tilde(self._h, size_Ql, _ptr(dst), _ptr(src), _stream()))

    def sm_mrq(self, size_Ql, dst, src):
        _lib.check(self._L.pha_sm_mrq(self._h, size_Ql, _ptr(dst), _ptr(src), _stream()))

    def fast_floor(self, size_Ql, input_base_q, input_base_Bsk, out_base_Bsk):
        _lib.check(self._L.pha_fast_floor(self._h, size_Ql, _ptr(input_base_q), _ptr(input_base_Bsk), _ptr(out_base_Bsk), _stream()))

    def fastbconv_sk(self, size_Ql, input_base_Bsk, out_base_q):
        _lib.check(self._L.pha_fastbconv_sk(self._h, size_Ql, _ptr(input_base_Bsk), _ptr(out_base_q), _stream()))

    def scaleAndRound_HPS_QR_R(self, size_Ql, dst, src):
        _lib.check(self._L.pha_scaleAndRound_HPS_QR_R(self._h, size_Ql, _ptr(dst), _ptr(src), _stream()))

    def scaleAndRound_HPS_QlRl_Ql(self, size_Ql, dst, src):
        _lib.check(self._L.pha_scaleAndRound_HPS_QlRl_Ql(self._h, size_Ql, _ptr(dst), _ptr(src), _stream()))

    def ExpandCRTBasis_Ql_Q_add_to_ct(self, size_Ql, dst, src):
        _lib.check(self._L.pha_ExpandCRTBasis_Ql_Q_add_to_ct(self._h, size_Ql, _ptr(dst), _ptr(src), _stream()))

    def scaleAndRound_HPS_Q_Ql(self, size_Ql, dst, src):
        _lib.check(self._L.pha_scaleAndRound_HPS_Q_Ql(self._h, size_Ql, _ptr(dst), _ptr(src), _stream()))

    def ExpandCRTBasis_Ql_Q(self, size_Ql, dst, src):
        _lib.check(self._L.pha_ExpandCRTBasis_Ql_Q(self._h, size_Ql, _ptr(dst), _ptr(src), _stream()))

    def keyswitch_inplace_bfv_leveled(self, size_Ql, ct, c2, rlk_ptrs):
        """keyswitch_inplace for BFV with levels dropped (src/eval_key_switch.cu:142-147, 170-175); ct [2][Q][N], c2 [Q][N]."""
        _lib.check(self._L.pha_keyswitch_inplace_bfv_leveled(self._h, size_Ql, _ptr(ct), _ptr(c2), _ptr(rlk_ptrs), _stream()))

    def batched_modular_gemm(self, C, A, B, m, n, k, batch, mod_start=0):
        """C[z] = A[z] @ B[z] mod q_{mod_start + z}, row-major [batch][m][k] x [batch][k][n] (benchmark/matmul_bench.cu)."""
        _lib.check(self._L.pha_batched_modular_gemm(self._h, _ptr(C), n, _ptr(A), k, _ptr(B), n, m, n, k, batch, mod_start,
                                                    _stream()))

    def hoisting(self, size_Ql, ct, galois_elts, galois_keys, scheme):
        """hoisting_inplace (src/evaluate.cu:1670-1866): ct <- sum_e rotate_e(ct); galois_keys[e] is the
        PhantomRelinKey of Galois element galois_elts[e]."""
        elts = (C.c_uint32 * len(galois_elts))(*[int(e) for e in galois_elts])
        tabs = (C.c_void_p * len(galois_keys))(*[k.public_keys_ptr.data_ptr() for k in galois_keys])
        _lib.check(self._L.pha_hoisting(self._h, size_Ql, _ptr(ct), elts, len(galois_elts), tabs, int(scheme), _stream()))

    def hoisting_weighted(self, size_Ql, ct, galois_elts, galois_keys, weights, scheme):
        """Build-defined (BASELINE config 5): ct <- sum_e weights[e] (.) rotate_e(ct); weights[e] is a device
        tensor [Ql + size_P][N] (the plaintext over [Q_l || P], NTT form); galois_keys[e] may be None for element 1."""
        elts = (C.c_uint32 * len(galois_elts))(*[int(e) for e in galois_elts])
        tabs = (C.c_void_p * len(galois_keys))(*[k.public_keys_ptr.data_ptr() if k is not None else None for k in galois_keys])
        ws = (C.c_void_p * len(weights))(*[_ptr(w) for w in weights])
        _lib.check(self._L.pha_hoisting_weighted(self._h, size_Ql, _ptr(ct), elts, len(galois_elts), tabs, ws, int(scheme),
                                                 _stream()))

    def hoisting_weighted_bsgs(self, size_Ql, ct, baby_elts, baby_keys, giant_elts, giant_keys, weights, scheme):
        """Baby-step / giant-step form (pha_hoisting_weighted_bsgs): ct <- sum_i rot_{giant_elts[i]}(sum_j weights[i][j] (.)
        rot_{baby_elts[j]}(ct)); weights is a list (per giant step) of lists (per baby step) of device tensors [Ql + size_P][N] or
        None; keys may be None for element 1."""
        be = (C.c_uint32 * len(baby_elts))(*[int(e) for e in baby_elts])
        ge = (C.c_uint32 * len(giant_elts))(*[int(e) for e in giant_elts])
        bk = (C.c_void_p * len(baby_keys))(*[k.public_keys_ptr.data_ptr() if k is not None else None for k in baby_keys])
        gk = (C.c_void_p * len(giant_keys))(*[k.public_keys_ptr.data_ptr() if k is not None else None for k in giant_keys])
        flat = [w for row in weights for w in row]
        if len(flat) != len(baby_elts) * len(giant_elts):
            raise ValueError("weights must be [n_giant][n_baby]")
        ws = (C.c_void_p * len(flat))(*[_ptr(w) for w in flat])
        _lib.check(self._L.pha_hoisting_weighted_bsgs(self._h, size_Ql, _ptr(ct), be, len(baby_elts), bk, ge, len(giant_elts), gk, ws,
                                                      int(scheme), _stream()))

    def hoisting_weighted_bsgs_blocks(self, size_Ql, ct, baby_elts, baby_keys, giant_elts, giant_keys, weights, out, scheme):
        """Several row blocks that share ct and the keys (pha_hoisting_weighted_bsgs_blocks): weights[block][giant][baby] device
        tensors or None, out [blocks][2][Ql][N]; ct is only read."""
        be = (C.c_uint32 * len(baby_elts))(*[int(e) for e in baby_elts])
        ge = (C.c_uint32 * len(giant_elts))(*[int(e) for e in giant_elts])
        bk = (C.c_void_p * len(baby_keys))(*[k.public_keys_ptr.data_ptr() if k is not None else None for k in baby_keys])
        gk = (C.c_void_p * len(giant_keys))(*[k.public_keys_ptr.data_ptr() if k is not None else None for k in giant_keys])
        flat = [w for blk in weights for row in blk for w in row]
        if len(flat) != len(weights) * len(baby_elts) * len(giant_elts):
            raise ValueError("weights must be [n_blocks][n_giant][n_baby]")
        ws = (C.c_void_p * len(flat))(*[_ptr(w) for w in flat])
        _lib.check(self._L.pha_hoisting_weighted_bsgs_blocks(self._h, size_Ql, _ptr(ct), len(weights), be, len(baby_elts), bk, ge,
                                                             len(giant_elts), gk, ws, _ptr(out), int(scheme), _stream()))

    def divide_and_round_q_last_ntt(self, size_Ql, src, cipher_size, dst):
        _lib.check(self._L.pha_divide_and_round_q_last_ntt(self._h, size_Ql, _ptr(src), cipher_size, _ptr(dst),
                                                           _stream()))

    def generate_one_kswitch_key(self, sk_ntt, new_key_ntt, a, e, scheme):
        """PhantomSecretKey::generate_one_kswitch_key (src/secretkey.cu:297-341), randomness (a, e) supplied by
        the caller; e is in coefficient form and is clobbered.  Returns a PhantomRelinKey."""
        dnum = self.size_Q // self.size_P
        keys = [torch.empty((2, self.size_QP, self.n), dtype=torch.int64, device=self.device) for _ in range(dnum)]
        rlk = PhantomRelinKey(keys)
        _lib.check(self._L.pha_generate_one_kswitch_key(self._h, _ptr(sk_ntt), _ptr(new_key_ntt), _ptr(a), _ptr(e),
                                                        _ptr(rlk.public_keys_ptr), int(scheme), _stream()))
        return rlk

    def mod_t_and_divide_q_last_ntt(self, size_Ql, src, cipher_size, dst):
        """BGV modulus switch (DRNSTool::mod_t_and_divide_q_last_ntt, src/rns.cu:1210-1236)."""
        _lib.check(self._L.pha_mod_t_and_divide_q_last_ntt(self._h, size_Ql, _ptr(src), cipher_size, _ptr(dst),
                                                           _stream()))

    def divide_and_round_q_last(self, size_Ql, src, cipher_size, dst):
        _lib.check(self._L.pha_divide_and_round_q_last(self._h, size_Ql, _ptr(src), cipher_size, _ptr(dst),
                                                       _stream()))

    # -- Galois (src/galois.cu:67-102) --------------------------------------------------------
    def apply_galois_ntt(self, src, dst, galois_elt, cms):
        _lib.check(self._L.pha_apply_galois_ntt(self._h, _ptr(src), _ptr(dst), galois_elt, cms, _stream()))

    def apply_galois(self, src, dst, galois_elt, cms, mod_start=0):
        _lib.check(self._L.pha_apply_galois(self._h, _ptr(src), _ptr(dst), galois_elt, cms, mod_start, _stream()))

    def nwt_2d_radix8_forward_inplace_include_temp_mod(self, inout, cms, start, total_modulus_size):
        _lib.check(self._L.pha_nwt_2d_radix8_forward_inplace_include_temp_mod(self._h, _ptr(inout), cms, start,
                                                                              total_modulus_size, _stream()))

    def nwt_2d_radix8_backward_inplace_include_temp_mod_scale(self, inout, cms, start, total_modulus_size, scale, scale_shoup):
        _lib.check(self._L.pha_nwt_2d_radix8_backward_inplace_include_temp_mod_scale(
            self._h, _ptr(inout), cms, start, total_modulus_size, _ptr(scale), _ptr(scale_shoup), _stream()))

    def nwt_2d_radix8_forward_modup_fuse(self, out, inp, modulus_index, cms, start=0):
        _lib.check(self._L.pha_nwt_2d_radix8_forward_modup_fuse(self._h, _ptr(out), _ptr(inp), modulus_index, cms, start,
                                                                _stream()))

    def apply_galois_batched(self, src, dst, galois_elt, cms, polys, ntt_form):
        _lib.check(self._L.pha_apply_galois_batched(self._h, _ptr(src), _ptr(dst), galois_elt, cms, polys, int(bool(ntt_form)), _stream()))

    def apply_galois_for_keyswitch(self, src, dst_ct, dst_c2, galois_elt, size_Ql, batch, ntt_form):
        """src [batch][2][Ql][N] -> dst_ct = (galois(c0), 0), dst_c2 [batch][Ql][N] = galois(c1): the operands of the key switch
        of a rotation, in one kernel (build-defined; BASELINE config 4)."""
        _lib.check(self._L.pha_apply_galois_for_keyswitch(self._h, _ptr(src), _ptr(dst_ct), _ptr(dst_c2), galois_elt, size_Ql,
                                                          batch, int(bool(ntt_form)), _stream()))

    def relinearize_rotate_batched(self, size_Ql, ct3, batch, rlk_ptrs, glk_ptrs, galois_elt, scheme, out, chunk=0):
        """BASELINE config 4 in one call: out [batch][2][Ql][N] = rotate(relinearize(ct3 [batch][3][Ql][N])); no copies."""
        _lib.check(self._L.pha_relinearize_rotate_batched(self._h, size_Ql, _ptr(ct3), batch, _ptr(rlk_ptrs), _ptr(glk_ptrs),
                                                          galois_elt, int(scheme), _ptr(out), chunk, _stream()))

    def broadcast_keys(self, key_tensors, root, nccl_comm):
        """RCCL-direct broadcast (pha_broadcast_keys) of a list of equally sized key tensors from rank `root` of the raw
        ncclComm_t handle `nccl_comm` (an integer / ctypes pointer), in place."""
        if not key_tensors:
            return
        words = key_tensors[0].numel()
        if any(k.numel() != words for k in key_tensors):
            raise ValueError("keys of one broadcast must have the same size")
        arr = (C.c_void_p * len(key_tensors))(*[_ptr(k) for k in key_tensors])
        _lib.check(self._L.pha_broadcast_keys(self._h, arr, len(key_tensors), words, int(root), C.c_void_p(nccl_comm), _stream()))

    def arena_count(self):
        """Scratch arenas currently held (phantom_amd_bench.h): one per explicit stream / per live host thread."""
        cnt = C.c_size_t()
        _lib.check(self._L.pha_context_arena_count(self._h, C.byref(cnt)))
        return cnt.value

    # -- measurement ------------------------------------------------------------------------------
    def repeat_forward_ntt_batched(self, inout, cms, start, batch, poly_stride, repeats):
        """`repeats` back-to-back batched forward transforms enqueued from C (bench.py's timed region)."""
        _lib.check(self._L.pha_repeat_forward_ntt_batched(self._h, _ptr(inout), cms, start, batch, poly_stride, repeats, _stream()))

    def time_forward_ntt(self, inout, cms, iters):
        ms = C.c_float()
        _lib.check(self._L.pha_time_forward_ntt(self._h, _ptr(inout), cms, iters, _stream(), C.byref(ms)))
        return ms.value


class DBaseConverter:
    """DBaseConverter (include/rns_bconv.cuh:13-87) between two bases given as rows of the context's prime table."""

    def __init__(self, ctx, ibase, obase=None, out_modulus=None):
        """obase: rows of the prime table; or out_modulus: ONE raw output modulus (the plain modulus t of base_q_to_t_conv_,
        src/rns.cu:283-284) -- such a converter serves exact_convert_array only."""
        self._ctx, self._L = ctx, _lib.load()
        self.ibase = [int(i) for i in ibase]
        ib = (C.c_uint32 * len(self.ibase))(*self.ibase)
        h = C.c_void_p()
        if out_modulus is not None:
            self.obase = None
            _lib.check(self._L.pha_base_converter_create_modulus(ctx._h, ib, len(self.ibase), int(out_modulus), C.byref(h)))
        else:
            self.obase = [int(o) for o in obase]
            ob = (C.c_uint32 * len(self.obase))(*self.obase)
            _lib.check(self._L.pha_base_converter_create(ctx._h, ib, len(self.ibase), ob, len(self.obase), C.byref(h)))
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.pha_base_converter_destroy(self._h)
            self._h = None

    def bConv_BEHZ(self, dst, src):
        _lib.check(self._L.pha_bConv_BEHZ(self._h, _ptr(dst), _ptr(src), _stream()))

    def bConv_HPS(self, dst, src):
        _lib.check(self._L.pha_bConv_HPS(self._h, _ptr(dst), _ptr(src), _stream()))

    def bConv_BEHZ_var1(self, dst, src):
        """DBaseConverter::bConv_BEHZ_var1 (src/rns_bconv.cu:231-246)."""
        _lib.check(self._L.pha_bConv_BEHZ_var1(self._h, _ptr(dst), _ptr(src), _stream()))

    def exact_convert_array(self, dst, src):
        """DBaseConverter::exact_convert_array (src/rns_bconv.cu:374-431): [ibase][N] -> [N] modulo the one output modulus."""
        _lib.check(self._L.pha_exact_convert_array(self._h, _ptr(dst), _ptr(src), _stream()))


class PhantomRelinKey:
    """Device layout of PhantomRelinKey (include/secretkey.h:102-165): dnum public keys, each
    [2][size_QP][N], plus a device array of their pointers."""

    def __init__(self, keys):
        self.public_keys = [k.contiguous() for k in keys]
        self.public_keys_ptr = torch.tensor([k.data_ptr() for k in self.public_keys], dtype=torch.int64,
                                            device=self.public_keys[0].device)

    @classmethod
    def from_numpy(cls, evk, device="cuda:0"):
        return cls([to_device(evk[i], device) for i in range(evk.shape[0])])


def fnwt_1d(inout, twiddles, twiddles_shoup, modulus, dim, cms, start=0, opt=False):
    """fnwt_1d / fnwt_1d_opt (src/ntt/ntt_1d.cu): modulus is a device tensor of (value, const_ratio lo, hi) triples."""
    L = _lib.load()
    f = L.pha_fnwt_1d_opt if opt else L.pha_fnwt_1d
    _lib.check(f(_ptr(inout), _ptr(twiddles), _ptr(twiddles_shoup), _ptr(modulus), dim, cms, start, _stream()))


def inwt_1d(inout, itwiddles, itwiddles_shoup, modulus, scalar, scalar_shoup, dim, cms, start=0, opt=False):
    L = _lib.load()
    f = L.pha_inwt_1d_opt if opt else L.pha_inwt_1d
    _lib.check(f(_ptr(inout), _ptr(itwiddles), _ptr(itwiddles_shoup), _ptr(modulus), _ptr(scalar), _ptr(scalar_shoup), dim,
                 cms, start, _stream()))
