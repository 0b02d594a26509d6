"""CPU oracle (test infrastructure only -- nothing under phantom-fhe_amd/ imports this) for the remaining
residue-wise kernels of the reference's src/polymath.cu, restated with Python integers: one function per kernel,
each citing the lines it follows.  Arrays are numpy uint64 of shape [limbs][N] (ciphertexts [polys][limbs][N]);
`primes` are the moduli of the limbs.  Parity status: unpinned like the rest of oracle/ (the reference's kernels
cannot be built here); these are direct transcriptions of one-line formulas.
"""
import numpy as np


def _obj(a):
    return np.asarray(a, dtype=np.uint64).astype(object)


def _u64(a):
    return np.asarray(a, dtype=object).astype(np.uint64)


def _col(primes):
    return np.array([int(q) for q in primes], dtype=object)[:, None]


def add_std_cipher(c1, c2, primes):                       # polymath.cu:56-73
    return _u64((_obj(c1) + _obj(c2)) % _col(primes)[None])


def add_and_negate(a, b, primes):                         # :82-98
    return _u64((-(_obj(a) + _obj(b))) % _col(primes))


def add_many(operands, poly_index, primes):               # :126-147 (sum of the poly_index-th polynomials)
    acc = _obj(operands[0][poly_index])
    for o in operands[1:]:
        acc = (acc + _obj(o[poly_index])) % _col(primes)
    return _u64(acc)


def multiply_uniform_scalar(a, scale, primes):            # :181-196
    return _u64(_obj(a) * int(scale) % _col(primes))


def multiply_scalar_and_add(a, b, scalar, primes):        # :246-264
    return _u64((_obj(a) + _obj(b) * int(scalar)) % _col(primes))


def multiply_scalar_and_sub(a, b, scalar, primes):        # :266-283
    return _u64((_obj(a) - _obj(b) * int(scalar)) % _col(primes))


def multiply_and_scale_add(a, b, d, scale, primes):       # :294-315
    return _u64((_obj(a) * _obj(b) + _obj(d) * int(scale)) % _col(primes))


def multiply_and_add_negate(a, b, d, primes):             # :350-371
    return _u64((-(_obj(a) * _obj(b) + _obj(d))) % _col(primes))


def sub_and_scale(a, b, scale, primes):                   # :392-411 (and :374-390 for a single modulus)
    s = np.array([int(v) for v in scale], dtype=object)[:, None]
    return _u64((_obj(a) - _obj(b)) * s % _col(primes))


def bfv_timesQ_overt(ct, pt, neg_ql_mod_t, t_inv_mod_q, t, primes, sub=False):   # :413-461
    m = _obj(pt).reshape(1, -1) * int(neg_ql_mod_t) % int(t)
    v = m * np.array([int(x) for x in t_inv_mod_q], dtype=object)[:, None] % _col(primes)
    return _u64((_obj(ct) - v) % _col(primes) if sub else (_obj(ct) + v) % _col(primes))


def abs_plain(operand, threshold, increment):             # :645-664
    op = _obj(operand).reshape(1, -1)
    inc = np.array([int(x) for x in increment], dtype=object)[:, None]
    return _u64(np.where(op >= int(threshold), op + inc, op + 0 * inc))


def tensor_prod_mxn(op1, op2, primes):                    # :546-592
    m, n = len(op1), len(op2)
    a, b = _obj(op1), _obj(op2)
    out = []
    for j in range(m + n - 1):
        acc = 0
        for i in range(m):
            if 0 <= j - i < n:
                acc = acc + a[i] * b[j - i]
        out.append(acc % _col(primes))
    return _u64(np.stack(out))


def multiply_and_negated_add(alpha_sk, m_sk, prod_b_mod_q, operand3, primes):    # :606-634
    al = _obj(alpha_sk).reshape(1, -1)
    centred = np.where(al > (int(m_sk) >> 1), al - int(m_sk), al)      # alpha_sk in (-m_sk/2, m_sk/2]
    pb = np.array([int(x) for x in prod_b_mod_q], dtype=object)[:, None]
    return _u64((_obj(operand3) - centred * pb) % _col(primes))
