"""Thread-synchronous emulation of the reference's four 2-D radix-8 NTT kernels (TEST INFRASTRUCTURE, like everything under oracle/).

Why it exists: `oracle/oracle.c` restates the negacyclic transform as the textbook SEAL-order stage
loops; the reference computes it with two-phase radix-8 kernels whose index arithmetic (padded shared-memory tile, `remain_iters` /
`tail` stages, twiddle indices `i; 2i, 2i+1; 4i..4i+3`) is where a restatement could silently diverge.  The reference cannot be
executed here (CUDA + inline PTX, no libcu++), so this file restates the KERNELS THEMSELVES -- every index expression, every
shared-memory slot, every barrier-delimited section, the lazy 64-bit wrap-around arithmetic of the butterflies -- and
`tests/test_ref_kernel_emu.py` checks that they produce, word for word, what the oracle's textbook loops produce, on the tables of
`src/host/ntt.cu:11-56`, for N = 2^12 .. 2^17 (n1 in {64, 128, 256}, n2 in {64, 128, 256, 512}: every `remain_iters` / `tail` case).
It is a restatement, not an execution: parity stays "unpinned against an executed reference" (DESIGN.md section 2).

Emulation model: a kernel's grid-stride loop visits `tid` in [0, n / 8) per limb; `threadIdx.x = tid % blockDim` (the stride is a
multiple of the block size) and every trip of the loop is one "virtual block" tid // blockDim with its own shared buffer (the real
buffer is reused between trips, after a barrier).  All threads of all virtual blocks advance together from barrier to barrier --
which is what `__syncthreads()` guarantees inside a block, and blocks never communicate -- as numpy vectors; uint64 arithmetic wraps
modulo 2^64 exactly like the device's.

Sources restated (file:line under /root/reference):
  include/uintmodmath.cuh:18-21 (csub_q), :223-231 (multiply_and_reduce_shoup_lazy)
  include/butterfly.cuh:10-22 (ct_butterfly), :28-37 (gs_butterfly), :39-59 (fntt8), :61-71 (fntt4), :74-96 (intt8), :98-108 (intt4)
  src/ntt/fntt_2d.cu:9-99 (inplace_fnwt_radix8_phase1), :101-198 (phase2), :620-653 (launcher)
  src/ntt/intt_2d.cu:9-104 (inplace_inwt_radix8_phase1), :106-207 (phase2), :724-757 (launcher)
  include/ntt.cuh:131-153 (SAMPLE_SIZE), include/common.h:23-30 (blockDimNTT = 128, per_block_pad = 4)

r06 -- the kernels with INDEX MAPS, where an unpinned restatement could be wrong without a transform being wrong:
  src/ntt/ntt_modup.cu:395-657   nwt_2d_radix8_forward_inplace_include_special_mod_exclude_range: the grid-stride tid -> (data limb
                                 twr_idx, table row twr_idx2) map of :421-425 / :521-525 (phase 2 walks the limbs backwards), the
                                 `continue` over the digit's own range.  The per-limb bodies are token for token those of
                                 fntt_2d.cu:9-198 apart from the row used for psi / modulus (checked with diff), so they are the
                                 functions above, fed the remapped row.
  src/ntt/ntt_moddown.cu:106-261 inplace_fnwt_radix8_phase2_fuse_moddown: phase 2 reads `delta`, never writes it back, and stores
                                 sub_negate_const_mult(NTT(delta), cx, PInv) into ct (:203-208; uintmodmath.cuh:233-241).
  src/rns_bconv.cu:455-485       bconv_matmul_padded_unroll2_kernel with base_convert_acc_unroll2 (include/rns_bconv.cuh:127-143): thread
                                 tid -> (degree_idx = 2 (tid / obase), out_prime = tid % obase), 128-bit accumulation with carries,
                                 Barrett-128 as the PTX of uintmodmath.cuh:108-126 spells it, the leap of the output limb index over
                                 the digit's own range; :143-170 (bconv_matmul_unroll2_kernel) is the same without the leap; :22-29,
                                 :522-528 (bconv_mult_kernel, modup_copy_partQl_kernel).
  src/eval_key_switch.cu:14-69   key_switch_inner_prod_c2_and_evk: nid -> key limb twr (keys at full QP width, data at QlP), the dead
                                 `i && reduction_threshold == 0` branch, one Barrett-128 per accumulator.
"""
import numpy as np

U64 = np.uint64
_M32 = U64(0xFFFFFFFF)
_S32, _S63 = U64(32), U64(63)
BLOCK_DIM_NTT = 128     # include/common.h:24
PER_BLOCK_PAD = 4       # include/common.h:30


def sample_size(n):
    """SAMPLE_SIZE(n), include/ntt.cuh:131-153."""
    if n in (2048, 4096):
        return 64
    if n == 8192:
        return 128
    if n in (16384, 32768, 65536, 131072):
        return 256
    raise ValueError("unsupported polynomial degree when selecting sample size")


def _umul64hi(a, b):
    """__umul64hi on uint64 vectors."""
    a0, a1, b0, b1 = a & _M32, a >> _S32, b & _M32, b >> _S32
    p00, p01, p10, p11 = a0 * b0, a0 * b1, a1 * b0, a1 * b1
    mid = (p00 >> _S32) + (p01 & _M32) + (p10 & _M32)
    return p11 + (p01 >> _S32) + (p10 >> _S32) + (mid >> _S32)


def _csub_q(x, q):
    """uintmodmath.cuh:18-21: tmp = x - q; x = tmp + (tmp >> 63) * q."""
    tmp = x - q
    return tmp + (tmp >> _S63) * q


def _shoup_lazy(y, tw, tws, q):
    """multiply_and_reduce_shoup_lazy (uintmodmath.cuh:223-231) / the inlined form of butterfly.cuh:14-15: y * tw - hi(y * tw') * q, in [0, 2q)."""
    return y * tw - _umul64hi(y, tws) * q


def _ct(s, i, j, tw, tws, q):
    """ct_butterfly on s[i], s[j] (butterfly.cuh:10-22)."""
    x, y = s[i], s[j]
    tw_y = _shoup_lazy(y, tw, tws, q)
    mod2 = U64(2) * q
    tmp = x - mod2
    x = tmp + (tmp >> _S63) * mod2
    s[j] = x + mod2 - tw_y
    s[i] = x + tw_y


def _gs(s, i, j, tw, tws, q):
    """gs_butterfly on s[i], s[j] (butterfly.cuh:28-37)."""
    x, y = s[i], s[j]
    mod2 = U64(2) * q
    t = x + mod2 - y
    sm = _csub_q(x + y, mod2)
    s[i] = sm
    s[j] = _shoup_lazy(t, tw, tws, q)


def _fntt8(s, psi, psis, ti, q):
    """butterfly.cuh:39-59; ti = vector of twiddle indices."""
    for a, b in ((0, 4), (1, 5), (2, 6), (3, 7)):
        _ct(s, a, b, psi[ti], psis[ti], q)
    for a, b, o in ((0, 2, 0), (1, 3, 0), (4, 6, 1), (5, 7, 1)):
        _ct(s, a, b, psi[2 * ti + o], psis[2 * ti + o], q)
    for a, b, o in ((0, 1, 0), (2, 3, 1), (4, 5, 2), (6, 7, 3)):
        _ct(s, a, b, psi[4 * ti + o], psis[4 * ti + o], q)


def _fntt4(s, base, psi, psis, ti, q):
    """butterfly.cuh:61-71 on s[base .. base + 3]."""
    _ct(s, base, base + 2, psi[ti], psis[ti], q)
    _ct(s, base + 1, base + 3, psi[ti], psis[ti], q)
    _ct(s, base, base + 1, psi[2 * ti], psis[2 * ti], q)
    _ct(s, base + 2, base + 3, psi[2 * ti + 1], psis[2 * ti + 1], q)


def _intt8(s, psi, psis, ti, q):
    """butterfly.cuh:74-96."""
    for a, b, o in ((0, 1, 0), (2, 3, 1), (4, 5, 2), (6, 7, 3)):
        _gs(s, a, b, psi[4 * ti + o], psis[4 * ti + o], q)
    for a, b, o in ((0, 2, 0), (1, 3, 0), (4, 6, 1), (5, 7, 1)):
        _gs(s, a, b, psi[2 * ti + o], psis[2 * ti + o], q)
    for a, b in ((0, 4), (1, 5), (2, 6), (3, 7)):
        _gs(s, a, b, psi[ti], psis[ti], q)


def _intt4(s, base, psi, psis, ti, q):
    """butterfly.cuh:98-108 on s[base], s[base + 2], s[base + 4], s[base + 6]."""
    _gs(s, base, base + 2, psi[2 * ti], psis[2 * ti], q)
    _gs(s, base + 4, base + 6, psi[2 * ti + 1], psis[2 * ti + 1], q)
    _gs(s, base, base + 4, psi[ti], psis[ti], q)
    _gs(s, base + 2, base + 6, psi[ti], psis[ti], q)


def _threads(n, block):
    tid = np.arange(n // 8, dtype=np.int64)
    return tid, tid % block, tid // block


def fnwt_phase1(data, psi, psis, q, n, n1, pad):
    """inplace_fnwt_radix8_phase1 (src/ntt/fntt_2d.cu:9-99) on one limb, in place."""
    q = U64(q)
    block = (n1 // 8) * pad                                   # launcher :629
    n_idx, thr, vb = _threads(n, block)
    buf = np.zeros((int(vb.max()) + 1, (n1 + pad + 1) * pad), dtype=U64)
    pad_tid, pad_idx = thr % pad, thr // pad                   # :22-23
    group, t = n1 // 8, n // 2                                 # :25,28
    n_init = t // 4 // group * pad_idx + pad_tid + pad * (n_idx // (group * pad))   # :44
    s = [data[n_init + t // 4 * j].copy() for j in range(8)]  # :46-48
    one = np.ones_like(n_idx)
    _fntt8(s, psi, psis, one, q)                               # :49-50
    for j in range(8):                                         # :51-53
        buf[vb, pad_tid * (n1 + pad) + pad_idx + group * j] = s[j]
    remain_iters = 0
    j, k = 8, group // 2                                       # :56
    while j < group + 1:
        m_idx2, t_idx2 = pad_idx // (k // 4), pad_idx % (k // 4)
        idx = [(n1 + pad) * pad_tid + 2 * m_idx2 * k + t_idx2 + (k // 4) * l for l in range(8)]
        s = [buf[vb, idx[l]] for l in range(8)]
        _fntt8(s, psi, psis, j * one + m_idx2, q)              # :62-63
        for l in range(8):
            buf[vb, idx[l]] = s[l]
        if j == group // 2:                                    # :67-70
            remain_iters = 1
        if j == group // 4:
            remain_iters = 2
        j, k = j * 8, k >> 3
    if group < 8:                                              # :74-75
        remain_iters = 2 if group == 4 else 1
    idx = [(n1 + pad) * pad_tid + 8 * pad_idx + l for l in range(8)]
    s = [buf[vb, idx[l]] for l in range(8)]                    # :76-78
    if remain_iters == 1:                                      # :79-84
        ti = 4 * group * one + 4 * pad_idx
        for o in range(4):
            _ct(s, 2 * o, 2 * o + 1, psi[ti + o], psis[ti + o], q)
    elif remain_iters == 2:                                    # :85-89
        ti = 2 * group * one + 2 * pad_idx
        _fntt4(s, 0, psi, psis, ti, q)
        _fntt4(s, 4, psi, psis, ti + 1, q)
    for l in range(8):                                         # :90-92
        buf[vb, idx[l]] = s[l]
    for j in range(8):                                         # :94-97
        data[n_init + t // 4 * j] = buf[vb, pad_tid * (n1 + pad) + pad_idx + group * j]


def fnwt_phase2(data, psi, psis, q, n, n1, n2):
    """inplace_fnwt_radix8_phase2 (src/ntt/fntt_2d.cu:101-198) on one limb, in place (the reversed limb order of :124 is immaterial here)."""
    q = U64(q)
    n_idx, thr, vb = _threads(n, BLOCK_DIM_NTT)
    buf = np.zeros((int(vb.max()) + 1, BLOCK_DIM_NTT * 8), dtype=U64)   # per_block_memory, launcher :627
    group = n2 // 8                                            # :113
    st = thr // group                                          # :114
    t = n2 // 2                                                # :117
    m_idx, t_idx = n_idx // (t // 4), n_idx % (t // 4)         # :128-129
    n_init = 2 * m_idx * t + t_idx                             # :136
    s = [data[n_init + t // 4 * j].copy() for j in range(8)]
    tw_idx = n1 + m_idx                                        # :140
    _fntt8(s, psi, psis, tw_idx, q)
    for j in range(8):                                         # :142-144
        buf[vb, st * n2 + t_idx + t // 4 * j] = s[j]
    tail = 0
    j, k = 8, t // 8                                           # :148
    while j < t // 4 + 1:
        m_idx2, t_idx2 = t_idx // (k // 4), t_idx % (k // 4)
        idx = [st * n2 + 2 * m_idx2 * k + t_idx2 + (k // 4) * l for l in range(8)]
        s = [buf[vb, idx[l]] for l in range(8)]
        _fntt8(s, psi, psis, j * tw_idx + m_idx2, q)           # :155-156
        for l in range(8):
            buf[vb, idx[l]] = s[l]
        if j == t // 8:                                        # :161-164
            tail = 1
        if j == t // 16:
            tail = 2
        j, k = j * 8, k >> 3
    idx = [st * n2 + 8 * t_idx + l for l in range(8)]
    s = [buf[vb, idx[l]] for l in range(8)]                    # :168-170
    if tail == 1:                                              # :171-176
        ti = t * tw_idx + 4 * t_idx
        for o in range(4):
            _ct(s, 2 * o, 2 * o + 1, psi[ti + o], psis[ti + o], q)
    elif tail == 2:                                            # :177-181
        ti = (t // 2) * tw_idx + 2 * t_idx
        _fntt4(s, 0, psi, psis, ti, q)
        _fntt4(s, 4, psi, psis, ti + 1, q)
    for l in range(8):
        buf[vb, idx[l]] = s[l]
    for j in range(8):                                         # :187-196: final reduction, canonical output
        v = buf[vb, st * n2 + t_idx + t // 4 * j]
        v = _csub_q(v, U64(2) * q)
        v = _csub_q(v, q)
        data[n_init + t // 4 * j] = v


def inwt_phase1(data, psi, psis, q, n, n1, n2):
    """inplace_inwt_radix8_phase1 (src/ntt/intt_2d.cu:9-104) on one limb, in place."""
    q = U64(q)
    n_idx, thr, vb = _threads(n, BLOCK_DIM_NTT)
    buf = np.zeros((int(vb.max()) + 1, BLOCK_DIM_NTT * 8), dtype=U64)
    group = n2 // 8                                            # :23
    st = thr // group                                          # :24
    t = n // 2 // n1                                           # :27
    m_idx, t_idx = n_idx // (t // 4), n_idx % (t // 4)         # :33-34
    n_init = 2 * m_idx * t + t_idx                             # :41
    for j in range(8):                                         # :43-46
        buf[vb, st * n2 + t_idx + t // 4 * j] = data[n_init + t // 4 * j]
    idx = [st * n2 + 8 * t_idx + l for l in range(8)]
    s = [buf[vb, idx[l]] for l in range(8)]                    # :49-52
    tw_idx = n1 + m_idx                                        # :53
    _intt8(s, psi, psis, (t // 4) * tw_idx + t_idx, q)         # :54-55
    for l in range(8):
        buf[vb, idx[l]] = s[l]
    tail = 0
    j, k = t // 32, 32                                         # :63
    while j > 0:
        m_idx2, t_idx2 = t_idx // (k // 4), t_idx % (k // 4)
        idx = [st * n2 + 2 * m_idx2 * k + t_idx2 + (k // 4) * l for l in range(8)]
        s = [buf[vb, idx[l]] for l in range(8)]
        _intt8(s, psi, psis, j * tw_idx + m_idx2, q)           # :71-72
        for l in range(8):
            buf[vb, idx[l]] = s[l]
        if j == 2:                                             # :78-81
            tail = 1
        if j == 4:
            tail = 2
        j, k = j >> 3, k * 8
    s = [buf[vb, st * n2 + t_idx + t // 4 * jj] for jj in range(8)]   # :85-88
    if tail == 1:                                              # :89-93
        for a in range(4):
            _gs(s, a, a + 4, psi[tw_idx], psis[tw_idx], q)
    elif tail == 2:                                            # :94-97
        _intt4(s, 0, psi, psis, tw_idx, q)
        _intt4(s, 1, psi, psis, tw_idx, q)
    for jj in range(8):                                        # :99-102
        data[n_init + t // 4 * jj] = s[jj]


def inwt_phase2(data, psi, psis, n_inv, n_inv_shoup, q, n, n1, pad):
    """inplace_inwt_radix8_phase2 (src/ntt/intt_2d.cu:106-207) on one limb, in place."""
    q = U64(q)
    block = (n1 // 8) * pad                                    # launcher :746
    n_idx, thr, vb = _threads(n, block)
    buf = np.zeros((int(vb.max()) + 1, (n1 + pad + 1) * pad), dtype=U64)
    pad_tid, pad_idx = thr % pad, thr // pad                   # :122-123
    group, t = n1 // 8, n // 2                                 # :125,128
    n_init = 2 * t // group * pad_idx + pad_tid + pad * (n_idx // (group * pad))    # :142
    s = [data[n_init + t // 4 // group * j].copy() for j in range(8)]              # :144-147
    one = np.ones_like(n_idx)
    _intt8(s, psi, psis, group * one + pad_idx, q)             # :148-150
    for j in range(8):                                         # :151-154
        buf[vb, pad_tid * (n1 + pad) + 8 * pad_idx + j] = s[j]
    tail = 0
    j, k = group // 8, 32                                      # :158
    while j > 0:
        m_idx2, t_idx2 = pad_idx // (k // 4), pad_idx % (k // 4)
        idx = [(n1 + pad) * pad_tid + 2 * m_idx2 * k + t_idx2 + (k // 4) * l for l in range(8)]
        s = [buf[vb, idx[l]] for l in range(8)]
        _intt8(s, psi, psis, j * one + m_idx2, q)              # :166-167
        for l in range(8):
            buf[vb, idx[l]] = s[l]
        if j == 2:                                             # :172-175
            tail = 1
        if j == 4:
            tail = 2
        j, k = j >> 3, k * 8
    if group < 8:                                              # :178-179
        tail = 2 if group == 4 else 1
    s = [buf[vb, pad_tid * (n1 + pad) + pad_idx + group * l] for l in range(8)]    # :180-183
    if tail == 1:                                              # :184-188
        for a in range(4):
            _gs(s, a, a + 4, psi[one], psis[one], q)
    elif tail == 2:                                            # :189-192
        _intt4(s, 0, psi, psis, one, q)
        _intt4(s, 1, psi, psis, one, q)
    ninv, ninvs = U64(n_inv), U64(n_inv_shoup)
    for j in range(4):                                         # :194-197: N^-1 on the first half only (the table's slot 1 carries it for the other)
        s[j] = _shoup_lazy(s[j], ninv, ninvs, q)
    n_init = t // 4 // group * pad_idx + pad_tid + pad * (n_idx // (group * pad))  # :199
    for j in range(8):                                         # :200-205
        data[n_init + t // 4 * j] = _csub_q(s[j], q)


def nwt_2d_radix8_forward_inplace(x, psi, psis, q):
    """nwt_2d_radix8_forward_inplace (src/ntt/fntt_2d.cu:620-653) on ONE limb: x (canonical, natural order) -> canonical, bit-reversed order."""
    n = len(x)
    n1 = sample_size(n)
    d = np.array(x, dtype=U64, copy=True)
    old = np.seterr(over="ignore")
    try:
        fnwt_phase1(d, psi, psis, q, n, n1, PER_BLOCK_PAD)
        fnwt_phase2(d, psi, psis, q, n, n1, n // n1)
    finally:
        np.seterr(**old)
    return d


def nwt_2d_radix8_backward_inplace(x, ipsi, ipsis, n_inv, n_inv_shoup, q):
    """nwt_2d_radix8_backward_inplace (src/ntt/intt_2d.cu:724-757) on ONE limb; ipsi = the inverse table of src/host/ntt.cu:38-55 (slot 1
    pre-multiplied by N^-1)."""
    n = len(x)
    n2 = sample_size(n)            # phase2_sample_size, :729
    n1 = n // n2                   # phase1_sample_size, :731
    d = np.array(x, dtype=U64, copy=True)
    old = np.seterr(over="ignore")
    try:
        inwt_phase1(d, ipsi, ipsis, q, n, n1, n2)
        inwt_phase2(d, ipsi, ipsis, n_inv, n_inv_shoup, q, n, n1, PER_BLOCK_PAD)
    finally:
        np.seterr(**old)
    return d


# ---- r06: kernels with index maps ------------------------------------------------------------------------------------------------------

def _mul128(a, b):
    """multiply_uint64_uint64 (uintmath.cuh): (lo, hi) of a * b on uint64 vectors."""
    return a * b, _umul64hi(a, b)


def _add128(alo, ahi, blo, bhi):
    """add_uint128_uint128: 128-bit sum modulo 2^128 with the carry out of the low word."""
    lo = alo + blo
    return lo, ahi + bhi + (lo < alo).astype(U64)


def _barrett128(lo, hi, q, mu0, mu1):
    """barrett_reduce_uint128_uint64, the PTX sequence of uintmodmath.cuh:108-126 (mul.hi / mad.lo.cc / madc.hi ...), then csub_q (:135)."""
    tmp = _umul64hi(lo, mu0)                       # mul.hi.u64 tmp, lo, ratio0
    p = lo * mu1                                   # mad.lo.cc.u64 tmp, lo, ratio1, tmp
    t2 = p + tmp
    carry = (t2 < p).astype(U64)
    r = _umul64hi(lo, mu1) + carry                 # madc.hi.u64 r, lo, ratio1, 0
    p = hi * mu0                                   # mad.lo.cc.u64 tmp, hi, ratio0, tmp
    t3 = p + t2
    carry = (t3 < p).astype(U64)
    r = _umul64hi(hi, mu0) + r + carry             # madc.hi.u64 r, hi, ratio0, r
    r = hi * mu1 + r                               # mad.lo.u64 r, hi, ratio1, r
    return _csub_q(lo - r * q, q)                  # mul.lo, sub, csub_q


def _shoup(x, w, ws, q):
    """multiply_and_reduce_shoup (uintmodmath.cuh:207-215): lazy product, then csub_q."""
    return _csub_q(_shoup_lazy(x, w, ws, q), q)


def _quiet(fn):
    def run(*a, **k):
        old = np.seterr(over="ignore")
        try:
            return fn(*a, **k)
        finally:
            np.seterr(**old)
    run.__doc__ = fn.__doc__
    return run


@_quiet
def nwt_2d_radix8_forward_inplace_include_special_mod_exclude_range(data, tw, tws, mod, n, cms, start, size_qp, size_p, ex_start, ex_end):
    """The launcher of src/ntt/ntt_modup.cu:606-657 on data [.][n] (limb-major; rows `start .. start + cms` are visited), tw / tws / mod =
    the DNTTTable rows ([size_QP][n] twiddles, [size_QP] moduli).  In place; returns the list of (phase, twr_idx, twr_idx2) visits."""
    if ex_start < start or ex_end > start + cms:
        raise ValueError("Excluded range in NTT is invalid.")       # :617-620
    n1 = sample_size(n)
    n2 = n // n1
    visits = []
    per_limb = n // 8
    for phase in (1, 2):
        # the grid-stride loop covers tid in [0, n / 8 * cms); tid / (n / 8) takes every value 0 .. cms - 1, all threads of a (virtual)
        # block share it (n / 8 is a multiple of both block sizes), so the `continue` of :422 / :522 is block-uniform
        for k in range(cms):
            tid0 = k * per_limb
            twr_idx = tid0 // per_limb + start if phase == 1 else cms - 1 - tid0 // per_limb + start     # :421 / :521
            if ex_start <= twr_idx < ex_end:                                                                # :422 / :522
                continue
            twr_idx2 = size_qp - (start + cms - twr_idx) if twr_idx >= start + cms - size_p else twr_idx   # :423-425 / :523-525
            visits.append((phase, twr_idx, twr_idx2))
            limb = data[twr_idx]                                     # data_ptr = inout + twr_idx * n
            if phase == 1:
                fnwt_phase1(limb, tw[twr_idx2], tws[twr_idx2], mod[twr_idx2], n, n1, PER_BLOCK_PAD)
            else:
                fnwt_phase2(limb, tw[twr_idx2], tws[twr_idx2], mod[twr_idx2], n, n1, n2)
    return visits


@_quiet
def nwt_2d_radix8_forward_inplace_fuse_moddown(ct, cx, pinv, pinv_shoup, delta, tw, tws, mod, n, cms, start):
    """src/ntt/ntt_moddown.cu:222-261: phase 1 of the plain transform on delta (in place), then inplace_fnwt_radix8_phase2_fuse_moddown
    (:106-210): the plain phase 2 reading delta, whose canonical outputs are NOT written back but go through
    ct[...] = sub_negate_const_mult(sample, cx[...], PInv[twr], PInv_shoup[twr], q) (:203-208)."""
    n1 = sample_size(n)
    n2 = n // n1
    for k in range(cms):
        twr = k + start                                              # fntt_2d.cu:27
        fnwt_phase1(delta[twr], tw[twr], tws[twr], mod[twr], n, n1, PER_BLOCK_PAD)
    for k in range(cms):
        twr = cms - 1 - k + start                                    # :130
        q = U64(mod[twr])
        out = delta[twr].copy()                                      # phase 2 on a scratch copy: the kernel keeps its results in registers
        fnwt_phase2(out, tw[twr], tws[twr], mod[twr], n, n1, n2)     # ... :131-200 == fntt_2d.cu:101-193 (same statements)
        temp = cx[twr] + q - out                                     # sub_negate_const_mult uintmodmath.cuh:233-241: op2 + modulus - op1
        temp = _csub_q(temp, q)
        ct[twr] = _shoup(temp, U64(pinv[twr]), U64(pinv_shoup[twr]), q)


@_quiet
def bconv_mult_kernel(src, scale, scale_shoup, base, n):
    """src/rns_bconv.cu:22-29: dst[tid] = multiply_and_reduce_shoup(src[tid], scale[i], scale_shoup[i], base[i]), i = tid / n."""
    isz = len(base)
    tid = np.arange(n * isz, dtype=np.int64)
    i = tid // n
    flat = np.ascontiguousarray(src, dtype=U64).reshape(-1)
    sc, ss, bq = (np.asarray(v, dtype=U64) for v in (scale, scale_shoup, base))
    return _shoup(flat[tid], sc[i], ss[i], bq[i]).reshape(isz, n)


@_quiet
def bconv_matmul_padded_unroll2_kernel(dst, y, mat, obase, omu, isz, n, start_part_idx, size_part_ql):
    """src/rns_bconv.cu:455-485 with base_convert_acc_unroll2 (include/rns_bconv.cuh:127-143).  dst [.][n] is written at the PADDED limb
    index; y [isz][n] = x_i qhat_i^-1 mod q_i; mat [osz][isz] = QHatModp (row-major, as the shared copy of :461-464 holds it);
    obase / omu = output moduli and their (ratio0, ratio1).  start_part_idx >= osz and size_part_ql = 0 give bconv_matmul_unroll2_kernel
    (:143-170)."""
    osz = len(obase)
    yf = np.ascontiguousarray(y, dtype=U64).reshape(-1)
    m = np.ascontiguousarray(mat, dtype=U64).reshape(-1)
    ob = np.asarray(obase, dtype=U64)
    mu = np.asarray(omu, dtype=U64).reshape(osz, 2)
    tid = np.arange((n * osz + 1) // 2, dtype=np.int64)               # :467
    degree_idx = 2 * (tid // osz)                                      # :469
    out_prime_idx = tid % osz                                          # :470
    xlo = np.zeros(len(tid), dtype=U64)
    xhi, ylo, yhi = xlo.copy(), xlo.copy(), xlo.copy()
    for i in range(isz):                                               # rns_bconv.cuh:131-141
        op2 = m[out_prime_idx * isz + i]
        op1_x = yf[i * n + degree_idx]                                 # ld_two_uint64(ptr + i * degree + degree_idx)
        op1_y = yf[i * n + degree_idx + 1]
        lo, hi = _mul128(op1_x, op2)
        xlo, xhi = _add128(lo, hi, xlo, xhi)
        lo, hi = _mul128(op1_y, op2)
        ylo, yhi = _add128(lo, hi, ylo, yhi)
    q, mu0, mu1 = ob[out_prime_idx], mu[out_prime_idx, 0], mu[out_prime_idx, 1]
    padded = out_prime_idx + np.where(out_prime_idx >= start_part_idx, size_part_ql, 0)   # :479
    out1 = _barrett128(xlo, xhi, q, mu0, mu1)
    out2 = _barrett128(ylo, yhi, q, mu0, mu1)
    dst[padded, degree_idx] = out1                                     # st_two_uint64(dst + padded * n + degree_idx, out, out2)
    dst[padded, degree_idx + 1] = out2


@_quiet
def modup_copy_part_ql_kernel(t_mod_up, cks, size_ql, size_qlp, alpha, n):
    """src/rns_bconv.cu:522-528: t_mod_up[beta_idx * size_QlP_n + tid] = cks[tid], beta_idx = tid / (alpha n); t_mod_up flat [beta][QlP][n]."""
    tid = np.arange(size_ql * n, dtype=np.int64)
    beta_idx = tid // (alpha * n)
    t_mod_up.reshape(-1)[beta_idx * (size_qlp * n) + tid] = np.ascontiguousarray(cks, dtype=U64).reshape(-1)[tid]


@_quiet
def key_switch_inner_prod_c2_and_evk(c2, evks, mod, mu, n, size_qp, size_qlp, size_q, size_ql, beta, reduction_threshold=1 << 8):
    """src/eval_key_switch.cu:14-69.  c2 flat [beta][QlP][n]; evks = list of beta flat keys [2][QP][n]; mod / mu = DModulus rows of the
    QP table.  Returns dst flat [2][QlP][n]."""
    size_qp_n, size_qlp_n = size_qp * n, size_qlp * n
    c2 = np.ascontiguousarray(c2, dtype=U64).reshape(-1)
    ev = [np.ascontiguousarray(e, dtype=U64).reshape(-1) for e in evks]
    md = np.asarray(mod, dtype=U64)
    ratio = np.asarray(mu, dtype=U64).reshape(-1, 2)
    tid = np.arange(size_qlp_n, dtype=np.int64)
    nid = tid // n                                                     # :20
    twr = np.where(nid >= size_ql, size_q + (nid - size_ql), nid)      # :21
    q, mu0, mu1 = md[twr], ratio[twr, 0], ratio[twr, 1]
    evk_id = (tid % n) + twr * n                                       # :24
    c2_id = (tid % n) + nid * n                                        # :25
    a0lo, a0hi = _mul128(c2[c2_id], ev[0][evk_id])                     # :44
    a1lo, a1hi = _mul128(c2[c2_id], ev[0][evk_id + size_qp_n])         # :46
    for i in range(1, beta):
        if i and reduction_threshold == 0:                             # :49-55 (never true for the threshold callers pass)
            a0lo, a0hi = _barrett128(a0lo, a0hi, q, mu0, mu1), np.zeros_like(a0hi)
            a1lo, a1hi = _barrett128(a1lo, a1hi, q, mu0, mu1), np.zeros_like(a1hi)
        lo, hi = _mul128(c2[c2_id + i * size_qlp_n], ev[i][evk_id])
        a0lo, a0hi = _add128(a0lo, a0hi, lo, hi)                       # :57-58
        lo, hi = _mul128(c2[c2_id + i * size_qlp_n], ev[i][evk_id + size_qp_n])
        a1lo, a1hi = _add128(a1lo, a1hi, lo, hi)                       # :60-61
    dst = np.zeros(2 * size_qlp_n, dtype=U64)
    dst[tid] = _barrett128(a0lo, a0hi, q, mu0, mu1)                    # :64-65
    dst[tid + size_qlp_n] = _barrett128(a1lo, a1hi, q, mu0, mu1)       # :67-68
    return dst


@_quiet
def nwt_2d_radix8_backward_inplace_include_special_mod(data, itw, itws, n_inv, n_inv_shoup, mod, n, cms, start, size_qp, size_p):
    """src/ntt/intt_2d.cu:796-834 (kernels :411-617): the plain inverse kernels with the same twr_idx2 remap (:421, :494; diffed against
    :9-207 -- only the row used for the tables differs), limbs twr_idx = tid / (n / 8) + start in both phases."""
    n2 = sample_size(n)
    n1 = n // n2
    for k in range(cms):
        twr_idx = k + start
        twr_idx2 = size_qp - (start + cms - twr_idx) if twr_idx >= start + cms - size_p else twr_idx
        inwt_phase1(data[twr_idx], itw[twr_idx2], itws[twr_idx2], mod[twr_idx2], n, n1, n2)
    for k in range(cms):
        twr_idx = k + start
        twr_idx2 = size_qp - (start + cms - twr_idx) if twr_idx >= start + cms - size_p else twr_idx
        inwt_phase2(data[twr_idx], itw[twr_idx2], itws[twr_idx2], n_inv[twr_idx2], n_inv_shoup[twr_idx2], mod[twr_idx2], n, n1, PER_BLOCK_PAD)
