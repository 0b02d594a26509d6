"""ctypes binding of libphantom_amd.so (the C ABI declared in include/phantom_amd.h).

The HIP library is the product; there is NO CPU fallback.  Importing this module without the
built library raises immediately (run `python -c "import __graft_entry__ as g; g.build()"` or
`make -C phantom-fhe_amd/csrc`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PHA_LIB_OVERRIDE") or os.path.join(_HERE, "libphantom_amd.so")  # override: experiments only
EXP_LIB_PATH = os.path.join(_HERE, "libphantom_amd_exp.so")   # test-only build with every NTT variant and pha_set_tuning

u64p = C.POINTER(C.c_uint64)
vp = C.c_void_p
sz = C.c_size_t
u64 = C.c_uint64

# name -> argtypes (restype is int status unless listed in _SPECIAL)
_SIGS = {
    "pha_coeff_modulus_create": [C.c_uint64, C.POINTER(C.c_int), sz, u64p],
    "pha_context_create": [C.POINTER(vp), C.c_uint32, u64p, C.c_uint32, C.c_uint32, C.c_int],
    "pha_context_set_plain_modulus": [vp, C.c_uint64],
    "pha_context_prime_info": [vp, C.c_uint32, u64p, u64p, u64p, u64p],
    "pha_context_download_twiddle": [vp, C.c_uint32, C.c_int, u64p],
    "pha_tool_beta": [vp, C.c_uint32, C.POINTER(C.c_uint32)],
    "pha_nwt_2d_radix8_forward_inplace": [vp, vp, sz, sz, vp],
    "pha_nwt_2d_radix8_forward_inplace_include_special_mod": [vp, vp, sz, sz, sz, sz, vp],
    "pha_nwt_2d_radix8_forward_inplace_include_special_mod_exclude_range": [vp, vp, sz, sz, sz, sz, sz, sz, vp],
    "pha_nwt_2d_radix8_forward_inplace_fuse_moddown": [vp, vp, vp, vp, vp, vp, sz, sz, vp],
    "pha_nwt_2d_radix8_backward_inplace": [vp, vp, sz, sz, vp],
    "pha_nwt_2d_radix8_forward_inplace_batched": [vp, vp, sz, sz, sz, sz, vp],
    "pha_nwt_2d_radix8_backward_inplace_batched": [vp, vp, sz, sz, sz, sz, vp],
    "pha_nwt_2d_radix8_backward": [vp, vp, vp, sz, sz, vp],
    "pha_nwt_2d_radix8_backward_scale": [vp, vp, vp, sz, sz, vp, vp, vp],
    "pha_nwt_2d_radix8_backward_inplace_scale": [vp, vp, sz, sz, vp, vp, vp],
    "pha_nwt_2d_radix8_backward_inplace_include_special_mod": [vp, vp, sz, sz, sz, sz, vp],
    "pha_add_rns_poly": [vp, vp, vp, vp, sz, sz, vp],
    "pha_sub_rns_poly": [vp, vp, vp, vp, sz, sz, vp],
    "pha_negate_rns_poly": [vp, vp, vp, sz, sz, vp],
    "pha_multiply_rns_poly": [vp, vp, vp, vp, sz, sz, vp],
    "pha_multiply_and_add_rns_poly": [vp, vp, vp, vp, vp, sz, sz, vp],
    "pha_multiply_scalar_rns_poly": [vp, vp, vp, vp, vp, sz, sz, vp],
    "pha_tensor_prod_2x2_rns_poly": [vp, vp, vp, vp, sz, vp],
    "pha_tensor_square_2x2_rns_poly": [vp, vp, vp, sz, vp],
    "pha_tensor_prod_2x2_rns_poly_at": [vp, vp, vp, vp, sz, sz, vp],
    "pha_tensor_square_2x2_rns_poly_at": [vp, vp, vp, sz, sz, vp],
    "pha_add_to_ct": [vp, vp, vp, sz, vp],
    "pha_bconv_P_to_Ql": [vp, sz, vp, vp, vp],
    "pha_modup": [vp, sz, vp, vp, C.c_int, vp],
    "pha_key_switch_inner_prod": [vp, sz, vp, vp, vp, vp],
    "pha_moddown_from_NTT": [vp, sz, vp, vp, C.c_int, vp],
    "pha_keyswitch_inplace": [vp, sz, vp, vp, vp, C.c_int, vp],
    "pha_keyswitch_inplace_batched": [vp, sz, vp, vp, sz, vp, C.c_int, vp],
    "pha_keyswitch_rescale": [vp, sz, vp, vp, vp, vp, vp],
    "pha_keyswitch_rescale_batched": [vp, sz, vp, vp, sz, vp, vp, vp],
    "pha_tensor_prod_2x2_batched": [vp, vp, vp, vp, vp, sz, sz, vp],
    "pha_bfv_multiply_behz": [vp, vp, vp, vp, vp],
    "pha_bfv_multiply_hps": [vp, vp, vp, vp, vp],
    "pha_bfv_multiply_hps_overq": [vp, vp, vp, vp, vp],
    "pha_bfv_multiply_hps_overq_leveled": [vp, sz, vp, vp, vp, vp],
    "pha_bfv_mul_relin_hps_overq_leveled": [vp, sz, vp, vp, vp, vp, vp],
    "pha_scaleAndRound_HPS_Q_Ql": [vp, sz, vp, vp, vp],
    "pha_ExpandCRTBasis_Ql_Q": [vp, sz, vp, vp, vp],
    "pha_keyswitch_inplace_bfv_leveled": [vp, sz, vp, vp, vp, vp],
    "pha_batched_modular_gemm": [vp, vp, sz, vp, sz, vp, sz, sz, sz, sz, sz, sz, vp],
    "pha_nwt_2d_radix8_forward_inplace_include_temp_mod": [vp, vp, sz, sz, sz, vp],
    "pha_nwt_2d_radix8_backward_inplace_include_temp_mod_scale": [vp, vp, sz, sz, sz, vp, vp, vp],
    "pha_nwt_2d_radix8_forward_modup_fuse": [vp, vp, vp, sz, sz, sz, vp],
    "pha_fnwt_1d": [vp, vp, vp, vp, sz, sz, sz, vp],
    "pha_fnwt_1d_opt": [vp, vp, vp, vp, sz, sz, sz, vp],
    "pha_inwt_1d": [vp, vp, vp, vp, vp, vp, sz, sz, sz, vp],
    "pha_inwt_1d_opt": [vp, vp, vp, vp, vp, vp, sz, sz, sz, vp],
    "pha_base_converter_create": [vp, C.POINTER(C.c_uint32), sz, C.POINTER(C.c_uint32), sz, C.POINTER(vp)],
    "pha_bConv_BEHZ": [vp, vp, vp, vp],
    "pha_bConv_HPS": [vp, vp, vp, vp],
    "pha_bConv_BEHZ_var1": [vp, vp, vp, vp],
    "pha_base_converter_create_modulus": [vp, C.POINTER(C.c_uint32), sz, u64, C.POINTER(vp)],
    "pha_exact_convert_array": [vp, vp, vp, vp],
    "pha_moddown": [vp, sz, vp, vp, C.c_int, vp],
    "pha_check_canonical": [vp, vp, sz, sz, sz, sz, sz, u64p, vp],
    "pha_check_canonical_keys": [vp, sz, vp, sz, u64p, vp],
    "pha_tool_aux_sizes": [vp, sz, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)],
    "pha_fastbconv_m_tilde": [vp, sz, vp, vp, vp],
    "pha_sm_mrq": [vp, sz, vp, vp, vp],
    "pha_fast_floor": [vp, sz, vp, vp, vp, vp],
    "pha_fastbconv_sk": [vp, sz, vp, vp, vp],
    "pha_scaleAndRound_HPS_QR_R": [vp, sz, vp, vp, vp],
    "pha_scaleAndRound_HPS_QlRl_Ql": [vp, sz, vp, vp, vp],
    "pha_ExpandCRTBasis_Ql_Q_add_to_ct": [vp, sz, vp, vp, vp],
    "pha_bfv_add_plain": [vp, sz, vp, vp, C.c_int, vp],
    "pha_bfv_multiply_plain": [vp, sz, vp, sz, vp, vp],
    "pha_bgv_lift_plain": [vp, sz, vp, vp, vp],
    "pha_add_std_cipher": [vp, vp, vp, vp, sz, vp],
    "pha_add_and_negate_rns_poly": [vp, vp, vp, vp, sz, sz, vp],
    "pha_add_many_rns_poly": [vp, C.POINTER(vp), sz, vp, sz, sz, vp],
    "pha_multiply_uniform_scalar_rns_poly": [vp, vp, u64, vp, sz, sz, vp],
    "pha_multiply_scalar_and_add_rns_poly": [vp, vp, vp, u64, vp, sz, sz, vp],
    "pha_multiply_scalar_and_sub_rns_poly": [vp, vp, vp, u64, vp, sz, sz, vp],
    "pha_multiply_and_scale_add_rns_poly": [vp, vp, vp, vp, u64, vp, sz, sz, vp],
    "pha_multiply_and_add_negate_rns_poly": [vp, vp, vp, vp, vp, sz, sz, vp],
    "pha_sub_and_scale_rns_poly": [vp, vp, vp, vp, vp, vp, sz, sz, vp],
    "pha_sub_and_scale_single_mod_poly": [vp, vp, vp, u64, u64, u64, vp, vp],
    "pha_bfv_add_timesQ_overt": [vp, vp, vp, u64, u64, vp, vp, u64, sz, vp],
    "pha_bfv_sub_timesQ_overt": [vp, vp, vp, u64, u64, vp, vp, u64, sz, vp],
    "pha_abs_plain_rns_poly": [vp, vp, u64, vp, vp, sz, vp],
    "pha_tensor_prod_mxn_rns_poly": [vp, vp, sz, vp, sz, vp, sz, sz, vp],
    "pha_multiply_and_negated_add_rns_poly": [vp, vp, u64, vp, vp, vp, sz, vp],
    "pha_hoisting": [vp, sz, vp, C.POINTER(C.c_uint32), sz, C.POINTER(vp), C.c_int, vp],
    "pha_hoisting_weighted": [vp, sz, vp, C.POINTER(C.c_uint32), sz, C.POINTER(vp), C.POINTER(vp), C.c_int, vp],
    "pha_hoisting_weighted_bsgs": [vp, sz, vp, C.POINTER(C.c_uint32), sz, C.POINTER(vp), C.POINTER(C.c_uint32), sz, C.POINTER(vp),
                                   C.POINTER(vp), C.c_int, vp],
    "pha_hoisting_weighted_bsgs_blocks": [vp, sz, vp, sz, C.POINTER(C.c_uint32), sz, C.POINTER(vp), C.POINTER(C.c_uint32), sz, C.POINTER(vp),
                                          C.POINTER(vp), vp, C.c_int, vp],
    "pha_divide_and_round_q_last_ntt": [vp, sz, vp, sz, vp, vp],
    "pha_generate_one_kswitch_key": [vp, vp, vp, vp, vp, vp, C.c_int, vp],
    "pha_mod_t_and_divide_q_last_ntt": [vp, sz, vp, sz, vp, vp],
    "pha_divide_and_round_q_last": [vp, sz, vp, sz, vp, vp],
    "pha_apply_galois_ntt": [vp, vp, vp, C.c_uint32, sz, vp],
    "pha_apply_galois": [vp, vp, vp, C.c_uint32, sz, sz, vp],
    "pha_apply_galois_batched": [vp, vp, vp, C.c_uint32, sz, sz, C.c_int, vp],
    "pha_apply_galois_for_keyswitch": [vp, vp, vp, vp, C.c_uint32, sz, sz, C.c_int, vp],
    "pha_broadcast_keys": [vp, C.POINTER(vp), sz, sz, C.c_int, vp, vp],
    "pha_relinearize_rotate_batched": [vp, sz, vp, sz, vp, vp, C.c_uint32, C.c_int, vp, sz, vp],
    # include/phantom_amd_bench.h (measurement hooks)
    "pha_time_forward_ntt": [vp, vp, sz, C.c_int, vp, C.POINTER(C.c_float)],
    "pha_repeat_forward_ntt_batched": [vp, vp, sz, sz, sz, sz, C.c_int, vp],
    "pha_time_stream_copy": [vp, vp, sz, C.c_int, vp, C.POINTER(C.c_double)],
    "pha_time_stream": [vp, vp, sz, C.c_int, C.c_int, C.c_int, vp, C.POINTER(C.c_double)],
    "pha_context_arena_count": [vp, C.POINTER(C.c_size_t)],
}
# exported by the test-only experiments library alone (csrc/pha_experiments.h); bound when present
_OPTIONAL = {
    "pha_set_tuning": [C.c_int, C.c_int],
}
_SPECIAL = {
    "pha_last_error": (C.c_char_p, []),
    "pha_context_destroy": (None, [vp]),
    "pha_base_converter_destroy": (None, [vp]),
    "pha_context_log_n": (C.c_uint32, [vp]),
    "pha_context_size_qp": (C.c_uint32, [vp]),
    "pha_context_size_p": (C.c_uint32, [vp]),
    "pha_set_strict": (C.c_int, [C.c_int]),
}

EXPORTED = sorted(list(_SIGS) + list(_SPECIAL))

_lib = None


class PhantomError(RuntimeError):
    pass


_EXC = {-1: ValueError, -2: ArithmeticError, -3: PhantomError}  # invalid_argument, logic_error, runtime_error


def load():
    """dlopen the HIP library and declare every prototype.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(make -C phantom-fhe_amd/csrc).  There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    for name, args in _SIGS.items():
        f = getattr(L, name)
        f.argtypes = args
        f.restype = C.c_int
    for name, (res, args) in _SPECIAL.items():
        f = getattr(L, name)
        f.argtypes = args
        f.restype = res
    for name, args in _OPTIONAL.items():
        f = getattr(L, name, None)
        if f is not None:
            f.argtypes = args
            f.restype = C.c_int
    _lib = L
    return L


def check(status):
    if status != 0:
        msg = load().pha_last_error().decode("utf-8", "replace")
        raise _EXC.get(status, PhantomError)(msg)
