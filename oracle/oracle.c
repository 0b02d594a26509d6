/*
 * oracle.c -- CPU restatement of the PhantomFHE RNS polynomial-arithmetic hot path.
 * TEST INFRASTRUCTURE ONLY (see oracle.h header).  Plain C11 + unsigned __int128.
 * "parity unpinned against an executed reference" -- see oracle.h for how it is pinned instead.
 *
 * Style: every routine is the straightforward scalar statement of the reference semantics
 * (SURVEY.md Appendix A).  Lazy ranges inside the transforms follow include/butterfly.cuh:10-37;
 * every stored output is canonical [0,q) at the points listed in SURVEY.md A.2.
 */
#include "oracle.h"
#include <stdlib.h>
#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
typedef uint64_t u64;

/* ------------------------------------------------------------------------------------------------
 * modular primitives (include/uintmodmath.cuh:18-242)
 * ---------------------------------------------------------------------------------------------- */
static inline u64 mulhi64(u64 a, u64 b) { return (u64)(((u128)a * b) >> 64); }

/* csub_q uintmodmath.cuh:18-21 */
static inline u64 csub(u64 x, u64 q) { return x >= q ? x - q : x; }
static inline u64 addmod(u64 a, u64 b, u64 q) { return csub(a + b, q); }       /* :32-38 */
static inline u64 submod(u64 a, u64 b, u64 q) { return csub(a + q - b, q); }   /* :43-49 */
static inline u64 negmod(u64 a, u64 q) { return a ? q - a : 0; }               /* :24-28 */

/* multiply_and_reduce_shoup_lazy :223-231, result in [0,2q) */
static inline u64 shoup_lazy(u64 a, u64 w, u64 ws, u64 q) { return a * w - mulhi64(a, ws) * q; }
/* multiply_and_reduce_shoup :207-215, canonical */
static inline u64 shoup(u64 a, u64 w, u64 ws, u64 q) { return csub(shoup_lazy(a, w, ws, q), q); }

/* barrett_reduce_uint128_uint64 :96-136 -- canonical 128-bit -> [0,q).
 * Restated exactly as the SEAL-lineage algorithm (only the needed partial products of p*mu >> 128,
 * one conditional subtraction); checked against p % q in tests. */
static inline u64 barrett128(u128 p, u64 q, const u64 mu[2]) {
    u64 lo = (u64)p, hi = (u64)(p >> 64);
    u64 carry = mulhi64(lo, mu[0]);
    u128 t = (u128)lo * mu[1];
    u64 tmp2lo = (u64)t, tmp2hi = (u64)(t >> 64);
    u64 tmp1 = tmp2lo + carry;
    u64 tmp3 = tmp2hi + (tmp1 < tmp2lo);
    t = (u128)hi * mu[0];
    tmp2lo = (u64)t; tmp2hi = (u64)(t >> 64);
    u64 s = tmp1 + tmp2lo;
    carry = tmp2hi + (s < tmp1);
    u64 quo = hi * mu[1] + tmp3 + carry;
    return csub(lo - quo * q, q);
}
/* barrett_reduce_uint64_uint64 :144-151 */
static inline u64 barrett64(u64 x, u64 q, u64 mu_hi) { return csub(x - mulhi64(x, mu_hi) * q, q); }

u64 orc_mulmod(u64 a, u64 b, u64 q) { return (u64)(((u128)a * b) % q); }
u64 orc_powmod(u64 a, u64 e, u64 q) {
    u64 r = 1 % q;
    a %= q;
    while (e) {
        if (e & 1) r = orc_mulmod(r, a, q);
        a = orc_mulmod(a, a, q);
        e >>= 1;
    }
    return r;
}
/* try_invert_uint_mod (src/host/numth.cu xgcd): q is prime on this path, so Fermat is equivalent. */
u64 orc_invmod(u64 a, u64 q) { return orc_powmod(a % q, q - 2, q); }

u64 orc_compute_shoup(u64 w, u64 q) { return (u64)(((u128)w << 64) / q); }

void orc_const_ratio(u64 q, u64 ratio[2]) {
    /* floor(2^128/q): q is never a power of two here, so floor((2^128-1)/q) is the same value */
    u128 r = (~(u128)0) / q;
    ratio[0] = (u64)r;
    ratio[1] = (u64)(r >> 64);
}

/* ------------------------------------------------------------------------------------------------
 * primes and roots (src/host/numth.cu:150-233,260-331; src/host/modulus.cu:82-111)
 * ---------------------------------------------------------------------------------------------- */
int orc_is_prime(u64 n) {
    /* deterministic Miller-Rabin for 64-bit (reference uses 40 random rounds, numth.cu:150-205;
     * same answer on every input up to negligible error) */
    static const u64 bases[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    if (n < 2) return 0;
    for (size_t i = 0; i < 12; i++) {
        if (n == bases[i]) return 1;
        if (n % bases[i] == 0) return 0;
    }
    u64 d = n - 1;
    int r = 0;
    while (!(d & 1)) { d >>= 1; r++; }
    for (size_t i = 0; i < 12; i++) {
        u64 x = orc_powmod(bases[i], d, n);
        if (x == 1 || x == n - 1) continue;
        int comp = 1;
        for (int j = 1; j < r; j++) {
            x = orc_mulmod(x, x, n);
            if (x == n - 1) { comp = 0; break; }
        }
        if (comp) return 0;
    }
    return 1;
}

int orc_get_primes(u64 ntt_size, int bit_size, size_t count, u64 *out) {
    u64 factor = 2 * ntt_size;
    u64 value = ((u64)1 << bit_size);
    if (value < factor) return -1;
    value = value - factor + 1;
    u64 lower = (u64)1 << (bit_size - 1);
    size_t k = 0;
    while (k < count && value > lower) {
        if (orc_is_prime(value)) out[k++] = value;
        value -= factor;
    }
    return k == count ? 0 : -1;
}

int orc_coeff_modulus_create(u64 n, const int *bit_sizes, size_t count, u64 *out) {
    /* count_table / prime_table of modulus.cu:98-109: for each distinct size, find as many primes as
     * requested (descending), then hand them out from the BACK of that list in request order. */
    int sizes[64];
    size_t need[64], used[64];
    u64 *tabs[64];
    size_t nd = 0;
    for (size_t i = 0; i < count; i++) {
        size_t j = 0;
        for (; j < nd; j++) if (sizes[j] == bit_sizes[i]) break;
        if (j == nd) { sizes[nd] = bit_sizes[i]; need[nd] = 0; used[nd] = 0; nd++; }
        need[j]++;
    }
    int rc = 0;
    for (size_t j = 0; j < nd; j++) {
        tabs[j] = (u64 *)malloc(sizeof(u64) * need[j]);
        if (orc_get_primes(n, sizes[j], need[j], tabs[j])) rc = -1;
    }
    if (!rc)
        for (size_t i = 0; i < count; i++) {
            size_t j = 0;
            for (; j < nd; j++) if (sizes[j] == bit_sizes[i]) break;
            out[i] = tabs[j][need[j] - 1 - used[j]];
            used[j]++;
        }
    for (size_t j = 0; j < nd; j++) free(tabs[j]);
    return rc;
}

int orc_minimal_primitive_root(u64 degree, u64 q, u64 *root_out) {
    /* numth.cu:260-331.  Any primitive degree-th root generates the same set of primitive roots
     * (its odd powers), so the minimum over that set does not depend on the random start. */
    if ((q - 1) % degree) return -1;
    u64 quotient = (q - 1) / degree;
    u64 root = 0;
    for (u64 g = 2; g < 1000; g++) {
        u64 r = orc_powmod(g, quotient, q);
        if (orc_powmod(r, degree >> 1, q) == q - 1) { root = r; break; }
    }
    if (!root) return -1;
    u64 gen_sq = orc_mulmod(root, root, q);
    u64 cur = root;
    for (u64 i = 0; i < degree; i++) {
        if (cur < root) root = cur;
        cur = orc_mulmod(cur, gen_sq, q);
    }
    *root_out = root;
    return 0;
}

static uint32_t brev(uint32_t x, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

int orc_ntt_tables(int log_n, u64 q, u64 *tw, u64 *tws, u64 *itw, u64 *itws, u64 *n_inv, u64 *n_inv_s) {
    /* src/host/ntt.cu:11-56 */
    size_t n = (size_t)1 << log_n;
    u64 root;
    if (orc_minimal_primitive_root(2 * n, q, &root)) return -1;
    u64 inv_root = orc_invmod(root, q);
    u64 power = root;
    for (size_t i = 1; i < n; i++) {
        size_t k = brev((uint32_t)i, log_n);
        tw[k] = power;
        tws[k] = orc_compute_shoup(power, q);
        power = orc_mulmod(power, root, q);
    }
    tw[0] = 1;
    tws[0] = orc_compute_shoup(1, q);
    power = inv_root;
    for (size_t i = 1; i < n; i++) {
        size_t k = brev((uint32_t)i, log_n);
        itw[k] = power;
        itws[k] = orc_compute_shoup(power, q);
        power = orc_mulmod(power, inv_root, q);
    }
    itw[0] = 1;
    itws[0] = orc_compute_shoup(1, q);
    *n_inv = orc_invmod(n % q, q);
    *n_inv_s = orc_compute_shoup(*n_inv, q);
    /* :53-55 fold n^-1 into slot 1 */
    itw[1] = orc_mulmod(itw[1], *n_inv, q);
    itws[1] = orc_compute_shoup(itw[1], q);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * transforms.  The reference's 2-phase radix-8 kernels (fntt_2d.cu:9-198) evaluate exactly the
 * SEAL-order Cooley-Tukey loop below: stage with m groups uses tw[m+i]
 * (ntt_1d.cu:50), butterflies are include/butterfly.cuh:10-22 (CT) / :28-37 (GS).
 * ---------------------------------------------------------------------------------------------- */
void orc_ntt_forward(u64 *x, int log_n, u64 q, const u64 *tw, const u64 *tws) {
    size_t n = (size_t)1 << log_n;
    u64 two_q = q << 1;
    size_t t = n;
    for (size_t m = 1; m < n; m <<= 1) {
        t >>= 1;
        for (size_t i = 0; i < m; i++) {
            u64 w = tw[m + i], ws = tws[m + i];
            size_t j1 = 2 * i * t;
            for (size_t j = j1; j < j1 + t; j++) {
                /* ct_butterfly: x in [0,4q) -> csub 2q ; t = y*w lazy [0,2q) ; (x+t, x+2q-t) */
                u64 X = csub(x[j], two_q);
                u64 T = shoup_lazy(x[j + t], w, ws, q);
                x[j] = X + T;
                x[j + t] = X + two_q - T;
            }
        }
    }
    /* final canonicalisation fntt_2d.cu:187-193 */
    for (size_t j = 0; j < n; j++) x[j] = csub(csub(x[j], two_q), q);
}

void orc_ntt_inverse(u64 *x, int log_n, u64 q, const u64 *itw, const u64 *itws, u64 n_inv, u64 n_inv_s) {
    /* GS loop, itw[1] already carries n^-1 (ntt.cu:53-55); first half is scaled explicitly
     * (intt_2d.cu:195-198); final csub q (:201-205). */
    size_t n = (size_t)1 << log_n;
    u64 two_q = q << 1;
    size_t t = 1;
    for (size_t m = n >> 1; m >= 1; m >>= 1) {
        for (size_t i = 0; i < m; i++) {
            u64 w = itw[m + i], ws = itws[m + i];
            size_t j1 = 2 * i * t;
            for (size_t j = j1; j < j1 + t; j++) {
                /* gs_butterfly: (csub2q(x+y), (x+2q-y)*w lazy) */
                u64 X = x[j], Y = x[j + t];
                u64 S = csub(X + Y, two_q);
                u64 D = shoup_lazy(X + two_q - Y, w, ws, q);
                if (m == 1) S = shoup_lazy(S, n_inv, n_inv_s, q);
                x[j] = S;
                x[j + t] = D;
            }
        }
        t <<= 1;
    }
    for (size_t j = 0; j < n; j++) x[j] = csub(csub(x[j], two_q), q);
}

/* ------------------------------------------------------------------------------------------------
 * context: per-prime tables for the whole QP chain (include/ntt.cuh:34-129; context.cu:170-183)
 * ---------------------------------------------------------------------------------------------- */
struct orc_ctx {
    int log_n;
    size_t n, size_qp, size_p, size_q;
    u64 *q;          /* [size_qp] */
    u64 (*mu)[2];    /* const_ratio */
    u64 *tw, *tws, *itw, *itws; /* [size_qp][n] */
    u64 *n_inv, *n_inv_s;
};

orc_ctx *orc_ctx_create(int log_n, const u64 *primes, size_t size_qp, size_t size_p) {
    orc_ctx *c = (orc_ctx *)calloc(1, sizeof(*c));
    c->log_n = log_n;
    c->n = (size_t)1 << log_n;
    c->size_qp = size_qp;
    c->size_p = size_p;
    c->size_q = size_qp - size_p;
    c->q = (u64 *)malloc(sizeof(u64) * size_qp);
    c->mu = malloc(sizeof(u64[2]) * size_qp);
    c->tw = (u64 *)malloc(sizeof(u64) * size_qp * c->n);
    c->tws = (u64 *)malloc(sizeof(u64) * size_qp * c->n);
    c->itw = (u64 *)malloc(sizeof(u64) * size_qp * c->n);
    c->itws = (u64 *)malloc(sizeof(u64) * size_qp * c->n);
    c->n_inv = (u64 *)malloc(sizeof(u64) * size_qp);
    c->n_inv_s = (u64 *)malloc(sizeof(u64) * size_qp);
    for (size_t i = 0; i < size_qp; i++) {
        c->q[i] = primes[i];
        orc_const_ratio(primes[i], c->mu[i]);
        if (orc_ntt_tables(log_n, primes[i], c->tw + i * c->n, c->tws + i * c->n, c->itw + i * c->n,
                           c->itws + i * c->n, &c->n_inv[i], &c->n_inv_s[i])) {
            orc_ctx_destroy(c);
            return NULL;
        }
    }
    return c;
}
void orc_ctx_destroy(orc_ctx *c) {
    if (!c) return;
    free(c->q); free(c->mu); free(c->tw); free(c->tws); free(c->itw); free(c->itws);
    free(c->n_inv); free(c->n_inv_s); free(c);
}
size_t orc_ctx_n(const orc_ctx *c) { return c->n; }
const u64 *orc_ctx_twiddle(const orc_ctx *c, size_t p, int which) {
    const u64 *b = which == 0 ? c->tw : which == 1 ? c->tws : which == 2 ? c->itw : c->itws;
    return b + p * c->n;
}
u64 orc_ctx_n_inv(const orc_ctx *c, size_t p) { return c->n_inv[p]; }

static void fwd1(const orc_ctx *c, u64 *x, size_t p) {
    orc_ntt_forward(x, c->log_n, c->q[p], c->tw + p * c->n, c->tws + p * c->n);
}
static void inv1(const orc_ctx *c, u64 *x, size_t p) {
    orc_ntt_inverse(x, c->log_n, c->q[p], c->itw + p * c->n, c->itws + p * c->n, c->n_inv[p], c->n_inv_s[p]);
}
/* limbs are independent: OpenMP over limbs (the all-cores CPU baseline of bench.py; orc_set_threads(1) = scalar) */
void orc_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : omp_get_num_procs());
#else
    (void)n;
#endif
}
void orc_nwt_forward(const orc_ctx *c, u64 *d, size_t limbs, size_t start) {
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t i = 0; i < limbs; i++) fwd1(c, d + i * c->n, start + i);
}
void orc_nwt_backward(const orc_ctx *c, u64 *d, size_t limbs, size_t start) {
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t i = 0; i < limbs; i++) inv1(c, d + i * c->n, start + i);
}
void orc_nwt_forward_map(const orc_ctx *c, u64 *d, const uint32_t *map, size_t limbs) {
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t i = 0; i < limbs; i++) fwd1(c, d + i * c->n, map[i]);
}
void orc_nwt_backward_map(const orc_ctx *c, u64 *d, const uint32_t *map, size_t limbs) {
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t i = 0; i < limbs; i++) inv1(c, d + i * c->n, map[i]);
}

/* ------------------------------------------------------------------------------------------------
 * dyadic kernels (src/polymath.cu)
 * ---------------------------------------------------------------------------------------------- */
#define FOR_LIMB_COEFF                                     \
    for (size_t l = 0; l < limbs; l++) {                   \
        const u64 q = c->q[start + l];                     \
        const u64 *mu = c->mu[start + l];                  \
        (void)mu;                                          \
        for (size_t k = 0; k < c->n; k++) {                \
            const size_t id = l * c->n + k;

#define END_LIMB_COEFF }}

void orc_add_rns_poly(const orc_ctx *c, const u64 *a, const u64 *b, u64 *r, size_t limbs, size_t start) {
    FOR_LIMB_COEFF r[id] = addmod(a[id], b[id], q); END_LIMB_COEFF
}
void orc_sub_rns_poly(const orc_ctx *c, const u64 *a, const u64 *b, u64 *r, size_t limbs, size_t start) {
    FOR_LIMB_COEFF r[id] = submod(a[id], b[id], q); END_LIMB_COEFF
}
void orc_negate_rns_poly(const orc_ctx *c, const u64 *a, u64 *r, size_t limbs, size_t start) {
    FOR_LIMB_COEFF r[id] = negmod(a[id], q); END_LIMB_COEFF
}
void orc_multiply_rns_poly(const orc_ctx *c, const u64 *a, const u64 *b, u64 *r, size_t limbs, size_t start) {
    FOR_LIMB_COEFF r[id] = barrett128((u128)a[id] * b[id], q, mu); END_LIMB_COEFF
}
void orc_multiply_and_add_rns_poly(const orc_ctx *c, const u64 *a, const u64 *b, const u64 *d, u64 *r,
                                   size_t limbs, size_t start) {
    /* polymath.cu:225-244: r = a*b + d, 128-bit sum then one Barrett */
    FOR_LIMB_COEFF r[id] = barrett128((u128)a[id] * b[id] + d[id], q, mu); END_LIMB_COEFF
}
void orc_multiply_scalar_rns_poly(const orc_ctx *c, const u64 *a, const u64 *scalar, u64 *r, size_t limbs,
                                  size_t start) {
    /* Shoup overload polymath.cu:198-213: scalar[l] per limb */
    FOR_LIMB_COEFF
        u64 s = scalar[l];
        r[id] = shoup(a[id], s, orc_compute_shoup(s, q), q);
    END_LIMB_COEFF
}
void orc_tensor_prod_2x2(const orc_ctx *c, const u64 *op1, const u64 *op2, u64 *res, size_t limbs) {
    const size_t start = 0, rc = limbs * c->n;
    FOR_LIMB_COEFF
        u64 c00 = op1[id], c01 = op1[id + rc], c10 = op2[id], c11 = op2[id + rc];
        u64 d0 = barrett128((u128)c00 * c10, q, mu);
        u64 d2 = barrett128((u128)c01 * c11, q, mu);
        /* (c0+c1) is NOT reduced before the multiply (polymath.cu:487) */
        u64 d1 = barrett128((u128)(c00 + c01) * (c10 + c11), q, mu);
        d1 = d1 + 2 * q - d0 - d2;
        d1 = csub(csub(d1, q), q);
        res[id] = d0; res[id + rc] = d1; res[id + 2 * rc] = d2;
    END_LIMB_COEFF
}
void orc_tensor_square_2x2(const orc_ctx *c, const u64 *op, u64 *res, size_t limbs) {
    const size_t start = 0, rc = limbs * c->n;
    FOR_LIMB_COEFF
        u64 c0 = op[id], c1 = op[id + rc];
        res[id] = barrett128((u128)c0 * c0, q, mu);
        res[id + rc] = barrett128(((u128)c0 * c1) << 1, q, mu);
        res[id + 2 * rc] = barrett128((u128)c1 * c1, q, mu);
    END_LIMB_COEFF
}

/* ------------------------------------------------------------------------------------------------
 * fast base conversion (src/host/rns.cu:282-337 RNSBase::initialize, :438-497 BaseConverter;
 * src/rns_bconv.cu:22-60 bconv_mult, :109-170 bconv_matmul, :212-229 bConv_BEHZ)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    size_t isz, osz;
    u64 *ib, *ob;
    u64 (*omu)[2];
    u64 *hat_inv, *hat_inv_s; /* qhat_i^-1 mod q_i  [isz] */
    u64 *mat;                 /* qhat_i mod p_j     [osz][isz] */
} bconv_t;

static void bconv_init(bconv_t *b, const u64 *ib, size_t isz, const u64 *ob, size_t osz) {
    b->isz = isz; b->osz = osz;
    b->ib = (u64 *)malloc(sizeof(u64) * isz);
    b->ob = (u64 *)malloc(sizeof(u64) * osz);
    b->omu = malloc(sizeof(u64[2]) * osz);
    b->hat_inv = (u64 *)malloc(sizeof(u64) * isz);
    b->hat_inv_s = (u64 *)malloc(sizeof(u64) * isz);
    b->mat = (u64 *)malloc(sizeof(u64) * isz * osz);
    memcpy(b->ib, ib, sizeof(u64) * isz);
    memcpy(b->ob, ob, sizeof(u64) * osz);
    for (size_t i = 0; i < isz; i++) {
        /* punctured product mod q_i, then its inverse (rns.cu:307-320); size-1 base -> 1 (:329-334) */
        u64 h = 1;
        for (size_t k = 0; k < isz; k++) if (k != i) h = orc_mulmod(h, ib[k] % ib[i], ib[i]);
        b->hat_inv[i] = orc_invmod(h, ib[i]);
        b->hat_inv_s[i] = orc_compute_shoup(b->hat_inv[i], ib[i]);
    }
    for (size_t j = 0; j < osz; j++) {
        orc_const_ratio(ob[j], b->omu[j]);
        for (size_t i = 0; i < isz; i++) {
            u64 h = 1;
            for (size_t k = 0; k < isz; k++) if (k != i) h = orc_mulmod(h, ib[k] % ob[j], ob[j]);
            b->mat[j * isz + i] = h; /* QHatModp_[j][i] rns.cu:448-457 */
        }
    }
}
static void bconv_free(bconv_t *b) {
    free(b->ib); free(b->ob); free(b->omu); free(b->hat_inv); free(b->hat_inv_s); free(b->mat);
}
/* phase 1: y_i = x_i * qhat_i^-1 mod q_i, canonical (bconv_mult_kernel) */
static void bconv_mult(const bconv_t *b, const u64 *src, u64 *tmp, size_t n) {
    for (size_t i = 0; i < b->isz; i++)
        for (size_t k = 0; k < n; k++)
            tmp[i * n + k] = shoup(src[i * n + k], b->hat_inv[i], b->hat_inv_s[i], b->ib[i]);
}
/* phase 2: out_j = Barrett128(sum_i y_i * (qhat_i mod p_j)); dst limb stride given so that the
 * "padded" variant (rns_bconv.cu:455-485) can leap over the digit's own range */
static void bconv_matmul(const bconv_t *b, const u64 *tmp, u64 *dst, size_t n, size_t pad_start, size_t pad_len) {
    for (size_t j = 0; j < b->osz; j++) {
        size_t jo = j + (j >= pad_start ? pad_len : 0);
        for (size_t k = 0; k < n; k++) {
            u128 acc = 0;
            for (size_t i = 0; i < b->isz; i++) acc += (u128)tmp[i * n + k] * b->mat[j * b->isz + i];
            dst[jo * n + k] = barrett128(acc, b->ob[j], b->omu[j]);
        }
    }
}
void orc_bconv(const u64 *ibase, size_t isz, const u64 *obase, size_t osz, const u64 *src, u64 *dst, size_t n) {
    bconv_t b;
    bconv_init(&b, ibase, isz, obase, osz);
    u64 *tmp = (u64 *)malloc(sizeof(u64) * isz * n);
    bconv_mult(&b, src, tmp, n);
    bconv_matmul(&b, tmp, dst, n, osz, 0);
    free(tmp);
    bconv_free(&b);
}

/* bConv_BEHZ_var1 constants on top of an initialised converter (src/host/rns.cu:469-496): phase-1 factor
 * negPQHatInvModq_i = q_i - (P mod q_i) * qhat_i^-1 mod q_i (P = product of the OUTPUT base), matrix QInvModp[j][i] = q_i^-1 mod p_j */
static void bconv_make_var1(bconv_t *b) {
    for (size_t i = 0; i < b->isz; i++) {
        const u64 qi = b->ib[i];
        u64 pm = 1 % qi;
        for (size_t j = 0; j < b->osz; j++) pm = orc_mulmod(pm, b->ob[j] % qi, qi);
        const u64 v = qi - orc_mulmod(pm, b->hat_inv[i], qi);
        b->hat_inv[i] = v;
        b->hat_inv_s[i] = orc_compute_shoup(v, qi);
    }
    for (size_t j = 0; j < b->osz; j++)
        for (size_t i = 0; i < b->isz; i++) b->mat[j * b->isz + i] = orc_invmod(b->ib[i] % b->ob[j], b->ob[j]);
}
/* DBaseConverter::bConv_BEHZ_var1 src/rns_bconv.cu:231-246: bconv_mult with negPQHatInvModq, bconv_matmul with QInvModp.
 * The output primes must be prime (q_i^-1 mod p_j by Fermat; the reference's try_invert_uint_mod accepts any coprime pair). */
void orc_bconv_behz_var1(const u64 *ibase, size_t isz, const u64 *obase, size_t osz, const u64 *src, u64 *dst, size_t n) {
    bconv_t b;
    bconv_init(&b, ibase, isz, obase, osz);
    bconv_make_var1(&b);
    u64 *tmp = (u64 *)malloc(sizeof(u64) * isz * n);
    bconv_mult(&b, src, tmp, n);
    bconv_matmul(&b, tmp, dst, n, osz, 0);
    free(tmp);
    bconv_free(&b);
}

/* DBaseConverter::exact_convert_array src/rns_bconv.cu:374-431 (one output modulus t, any t >= 2): per coefficient
 * y_i = x_i * qhat_i^-1 mod q_i; inner = sum_i y_i * (qhat_i mod t) in 128 bits (the mid-loop reduction of :393-396 never fires:
 * `i && reduction_threshold == 0` with reduction_threshold = 15), reduced once; v = sum_i double(y_i) / double(q_i) by IEEE
 * divisions and additions in limb order; Q mod t by Horner over the words of prod(ibase), most significant first (:399-400);
 * dst = inner - round(v) * (Q mod t) mod t. */
void orc_exact_convert_array(const u64 *ibase, size_t isz, u64 t, const u64 *src, u64 *dst, size_t n) {
    bconv_t b;
    bconv_init(&b, ibase, isz, &t, 1);
    /* prod(ibase) as isz little-endian words (RNSBase::big_modulus, src/host/rns.cu:296-305) */
    u64 *big = (u64 *)calloc(isz + 1, sizeof(u64));
    size_t len = 1;
    big[0] = 1;
    for (size_t i = 0; i < isz; i++) {
        u64 carry = 0;
        for (size_t w = 0; w < len; w++) { u128 x = (u128)big[w] * ibase[i] + carry; big[w] = (u64)x; carry = (u64)(x >> 64); }
        if (carry) big[len++] = carry;
    }
    for (size_t k = 0; k < n; k++) {
        u128 inner = 0;
        double v = 0.0;
        u64 q_mod_t = 0;
        for (size_t i = 0; i < isz; i++) {
            const u64 yi = shoup(src[i * n + k], b.hat_inv[i], b.hat_inv_s[i], ibase[i]);
            inner += (u128)yi * b.mat[i];
            q_mod_t = barrett128(((u128)q_mod_t << 64) | big[isz - i - 1], t, b.omu[0]);
            v += (double)yi / (double)ibase[i];
        }
        const u64 ip = barrett128(inner, t, b.omu[0]);
        const u64 rounded = (u64)round(v);
        const u64 corr = barrett128((u128)rounded * q_mod_t, t, b.omu[0]);   /* multiply_and_barrett_reduce_uint64 */
        dst[k] = submod(ip, corr, t);
    }
    free(big);
    bconv_free(&b);
}

/* ------------------------------------------------------------------------------------------------
 * DRNSTool at one data level (src/rns.cu:11-200)
 * ---------------------------------------------------------------------------------------------- */
struct orc_tool {
    const orc_ctx *c;
    size_t n, size_ql, size_p, size_qlp, size_q, size_qp, alpha, beta;
    u64 *qlp;            /* primes of [Ql || P] */
    uint32_t *qlp_idx;   /* their index in the QP table (twr remap fntt_2d.cu:434-437) */
    u64 *part_hat_inv, *part_hat_inv_s; /* partQlHatInv_mod_Ql_concat rns.cu:152-182 */
    bconv_t *digit_conv;                /* part Ql -> complement of QlP, per digit */
    bconv_t p_to_ql;                    /* base_P_to_Ql_conv rns.cu:196-198 */
    u64 *pinv, *pinv_s;                 /* bigPInv_mod_q rns.cu:110-123 */
    u64 *inv_q_last, *inv_q_last_s;     /* rns.cu:66-80 */
    /* plain-modulus constants, BGV (rns.cu:196-285); plain_t == 0 until orc_tool_set_plain_modulus */
    u64 plain_t, t_mu[2];
    u64 *q_last_mod_q, *q_last_mod_q_s; /* rns.cu:68-75 */
    u64 inv_q_last_mod_t, inv_q_last_mod_t_s; /* rns.cu:205-212 */
    u64 *p_mod_q, *p_mod_q_s;           /* bigP_mod_q rns.cu:110-123 */
    u64 pinv_mod_t, pinv_mod_t_s;       /* bigPInv_mod_t rns.cu:272-280 */
    bconv_t p_to_t;                     /* base_P_to_t_conv rns.cu:283-284 */
};

orc_tool *orc_tool_create(const orc_ctx *c, size_t size_ql) {
    orc_tool *t = (orc_tool *)calloc(1, sizeof(*t));
    t->c = c; t->n = c->n; t->size_ql = size_ql; t->size_p = c->size_p; t->alpha = c->size_p;
    t->size_q = c->size_q; t->size_qp = c->size_qp; t->size_qlp = size_ql + c->size_p;
    t->qlp = (u64 *)malloc(sizeof(u64) * t->size_qlp);
    t->qlp_idx = (uint32_t *)malloc(sizeof(uint32_t) * t->size_qlp);
    for (size_t i = 0; i < size_ql; i++) { t->qlp[i] = c->q[i]; t->qlp_idx[i] = (uint32_t)i; }
    for (size_t i = 0; i < t->size_p; i++) {
        t->qlp[size_ql + i] = c->q[t->size_q + i];
        t->qlp_idx[size_ql + i] = (uint32_t)(t->size_q + i);
    }
    /* rescale constants */
    if (size_ql > 1) {
        t->inv_q_last = (u64 *)malloc(sizeof(u64) * (size_ql - 1));
        t->inv_q_last_s = (u64 *)malloc(sizeof(u64) * (size_ql - 1));
        for (size_t i = 0; i + 1 < size_ql; i++) {
            t->inv_q_last[i] = orc_invmod(c->q[size_ql - 1] % c->q[i], c->q[i]);
            t->inv_q_last_s[i] = orc_compute_shoup(t->inv_q_last[i], c->q[i]);
        }
    }
    if (t->size_p) {
        t->pinv = (u64 *)malloc(sizeof(u64) * size_ql);
        t->pinv_s = (u64 *)malloc(sizeof(u64) * size_ql);
        for (size_t i = 0; i < size_ql; i++) {
            u64 p = 1;
            for (size_t k = 0; k < t->size_p; k++) p = orc_mulmod(p, c->q[t->size_q + k] % c->q[i], c->q[i]);
            t->pinv[i] = orc_invmod(p, c->q[i]);
            t->pinv_s[i] = orc_compute_shoup(t->pinv[i], c->q[i]);
        }
        t->beta = (size_ql + t->alpha - 1) / t->alpha; /* rns.cu:152 */
        t->digit_conv = (bconv_t *)calloc(t->beta, sizeof(bconv_t));
        t->part_hat_inv = (u64 *)malloc(sizeof(u64) * size_ql);
        t->part_hat_inv_s = (u64 *)malloc(sizeof(u64) * size_ql);
        u64 *compl = (u64 *)malloc(sizeof(u64) * t->size_qlp);
        for (size_t b = 0; b < t->beta; b++) {
            size_t s = t->alpha * b;
            size_t len = (b == t->beta - 1) ? size_ql - t->alpha * (t->beta - 1) : t->alpha;
            size_t nc = 0;
            for (size_t j = 0; j < t->size_qlp; j++) if (j < s || j >= s + len) compl[nc++] = t->qlp[j];
            bconv_init(&t->digit_conv[b], t->qlp + s, len, compl, nc);
            for (size_t i = 0; i < len; i++) {
                t->part_hat_inv[s + i] = t->digit_conv[b].hat_inv[i];
                t->part_hat_inv_s[s + i] = t->digit_conv[b].hat_inv_s[i];
            }
        }
        free(compl);
        bconv_init(&t->p_to_ql, c->q + t->size_q, t->size_p, c->q, size_ql);
    }
    return t;
}
void orc_tool_destroy(orc_tool *t) {
    if (!t) return;
    free(t->qlp); free(t->qlp_idx); free(t->inv_q_last); free(t->inv_q_last_s);
    if (t->size_p) {
        free(t->pinv); free(t->pinv_s); free(t->part_hat_inv); free(t->part_hat_inv_s);
        for (size_t b = 0; b < t->beta; b++) bconv_free(&t->digit_conv[b]);
        free(t->digit_conv);
        bconv_free(&t->p_to_ql);
        if (t->plain_t) bconv_free(&t->p_to_t);
    }
    free(t->q_last_mod_q); free(t->q_last_mod_q_s); free(t->p_mod_q); free(t->p_mod_q_s);
    free(t);
}

int orc_tool_set_plain_modulus(orc_tool *t, u64 plain_t) {
    /* DRNSTool ctor, plain-modulus part: rns.cu:200-212 (q_last^-1 mod t), :270-284 (P^-1 mod t, P -> {t}) */
    const orc_ctx *c = t->c;
    const size_t ql = t->size_ql;
    if (plain_t < 2 || t->plain_t) return -1;
    orc_const_ratio(plain_t, t->t_mu);
    u64 g = c->q[ql - 1] % plain_t;
    if (g == 0) return -1;
    /* t need not be prime: extended Euclid */
    {
        __int128 r0 = plain_t, r1 = g, s0 = 0, s1 = 1;
        while (r1) { __int128 qq = r0 / r1, tmp = r0 - qq * r1; r0 = r1; r1 = tmp; tmp = s0 - qq * s1; s0 = s1; s1 = tmp; }
        if (r0 != 1) return -1;
        t->inv_q_last_mod_t = (u64)((s0 % (__int128)plain_t + plain_t) % plain_t);
    }
    t->inv_q_last_mod_t_s = orc_compute_shoup(t->inv_q_last_mod_t, plain_t);
    if (ql > 1) {
        t->q_last_mod_q = (u64 *)malloc(sizeof(u64) * (ql - 1));
        t->q_last_mod_q_s = (u64 *)malloc(sizeof(u64) * (ql - 1));
        for (size_t i = 0; i + 1 < ql; i++) {
            t->q_last_mod_q[i] = c->q[ql - 1] % c->q[i];
            t->q_last_mod_q_s[i] = orc_compute_shoup(t->q_last_mod_q[i], c->q[i]);
        }
    }
    if (t->size_p) {
        t->p_mod_q = (u64 *)malloc(sizeof(u64) * ql);
        t->p_mod_q_s = (u64 *)malloc(sizeof(u64) * ql);
        for (size_t i = 0; i < ql; i++) {
            u64 p = 1;
            for (size_t k = 0; k < t->size_p; k++) p = orc_mulmod(p, c->q[t->size_q + k] % c->q[i], c->q[i]);
            t->p_mod_q[i] = p;
            t->p_mod_q_s[i] = orc_compute_shoup(p, c->q[i]);
        }
        u64 pt = 1 % plain_t;
        for (size_t k = 0; k < t->size_p; k++) pt = orc_mulmod(pt, c->q[t->size_q + k] % plain_t, plain_t);
        __int128 r0 = plain_t, r1 = pt, s0 = 0, s1 = 1;
        while (r1) { __int128 qq = r0 / r1, tmp = r0 - qq * r1; r0 = r1; r1 = tmp; tmp = s0 - qq * s1; s0 = s1; s1 = tmp; }
        if (r0 != 1) return -1;
        t->pinv_mod_t = (u64)((s0 % (__int128)plain_t + plain_t) % plain_t);
        t->pinv_mod_t_s = orc_compute_shoup(t->pinv_mod_t, plain_t);
        bconv_init(&t->p_to_t, c->q + t->size_q, t->size_p, &plain_t, 1);
    }
    t->plain_t = plain_t;
    return 0;
}
size_t orc_tool_beta(const orc_tool *t) { return t->beta; }

void orc_modup(const orc_tool *t, u64 *dst, const u64 *cks, int scheme) {
    /* DRNSTool::modup rns_bconv.cu:530-627 */
    const orc_ctx *c = t->c;
    const size_t n = t->n, ql = t->size_ql, qlp = t->size_qlp, alpha = t->alpha;
    u64 *t_cks = (u64 *)malloc(sizeof(u64) * ql * n);
    memcpy(t_cks, cks, sizeof(u64) * ql * n);
    if (scheme == ORC_CKKS || scheme == ORC_BGV) {
        orc_nwt_backward(c, t_cks, ql, 0);
        if (alpha != 1) /* nwt_2d_radix8_backward_scale: x partQlHatInv, full Shoup reduce (:558-559) */
            for (size_t i = 0; i < ql; i++)
                for (size_t k = 0; k < n; k++)
                    t_cks[i * n + k] = shoup(t_cks[i * n + k], t->part_hat_inv[i], t->part_hat_inv_s[i], c->q[i]);
    }
    for (size_t b = 0; b < t->beta; b++) {
        const size_t s = alpha * b;
        const size_t len = (b == t->beta - 1) ? ql - alpha * (t->beta - 1) : alpha;
        u64 *out = dst + b * qlp * n;
        if (alpha == 1) {
            /* modup_bconv_single_p_kernel :432-453 */
            const u64 *normal = (scheme == ORC_BFV) ? cks + s * n : t_cks + s * n;
            for (size_t j = 0; j < qlp; j++)
                for (size_t k = 0; k < n; k++) {
                    if (j == s) out[j * n + k] = cks[s * n + k];
                    else {
                        u64 ip = t->qlp[s], op = t->qlp[j], v = normal[k];
                        out[j * n + k] = ip > op ? barrett64(v, op, c->mu[t->qlp_idx[j]][1]) : v;
                    }
                }
        } else {
            /* own limbs copied verbatim (:522-528) */
            memcpy(out + s * n, cks + s * n, sizeof(u64) * len * n);
            const bconv_t *bc = &t->digit_conv[b];
            u64 *tmp = t_cks + s * n;
            u64 *scaled = NULL;
            if (scheme == ORC_BFV) { /* bconv_mult_kernel :603-607 */
                scaled = (u64 *)malloc(sizeof(u64) * len * n);
                bconv_mult(bc, cks + s * n, scaled, n);
                tmp = scaled;
            }
            bconv_matmul(bc, tmp, out, n, s, len);
            free(scaled);
        }
        /* forward NTT: ckks/bgv skip the digit's own range (ntt_modup.cu:422), bfv does all limbs */
        for (size_t j = 0; j < qlp; j++) {
            if ((scheme == ORC_CKKS || scheme == ORC_BGV) && j >= s && j < s + len) continue;
            fwd1(c, out + j * n, t->qlp_idx[j]);
        }
    }
    free(t_cks);
}

void orc_key_switch_inner_prod(const orc_tool *t, u64 *cx, const u64 *mu_in, const u64 *const *evks) {
    /* key_switch_inner_prod_c2_and_evk eval_key_switch.cu:14-69.  128-bit accumulate, one Barrett. */
    const orc_ctx *c = t->c;
    const size_t n = t->n, qlp = t->size_qlp, qp_n = t->size_qp * n, qlp_n = qlp * n;
    for (size_t nid = 0; nid < qlp; nid++) {
        size_t twr = t->qlp_idx[nid];
        for (size_t k = 0; k < n; k++) {
            u128 a0 = 0, a1 = 0;
            for (size_t i = 0; i < t->beta; i++) {
                u64 v = mu_in[i * qlp_n + nid * n + k];
                a0 += (u128)v * evks[i][twr * n + k];
                a1 += (u128)v * evks[i][twr * n + k + qp_n];
            }
            cx[nid * n + k] = barrett128(a0, c->q[twr], c->mu[twr]);
            cx[nid * n + k + qlp_n] = barrett128(a1, c->q[twr], c->mu[twr]);
        }
    }
}

void orc_moddown_from_ntt(const orc_tool *t, u64 *ct, u64 *cx, int scheme) {
    /* DRNSTool::moddown_from_NTT rns_bconv.cu:776-828 */
    const orc_ctx *c = t->c;
    const size_t n = t->n, ql = t->size_ql, qlp = t->size_qlp;
    u64 *delta = (u64 *)malloc(sizeof(u64) * ql * n);
    if (scheme == ORC_CKKS) orc_nwt_backward_map(c, cx + ql * n, t->qlp_idx + ql, t->size_p);
    else orc_nwt_backward_map(c, cx, t->qlp_idx, qlp);
    if (t->alpha == 1) {
        /* moddown_bconv_single_p_kernel :691-707 */
        u64 ip = t->qlp[ql];
        for (size_t j = 0; j < ql; j++)
            for (size_t k = 0; k < n; k++) {
                u64 v = cx[ql * n + k];
                delta[j * n + k] = ip > c->q[j] ? barrett64(v, c->q[j], c->mu[j][1]) : v;
            }
    } else {
        u64 *tmp = (u64 *)malloc(sizeof(u64) * t->size_p * n);
        bconv_mult(&t->p_to_ql, cx + ql * n, tmp, n);
        bconv_matmul(&t->p_to_ql, tmp, delta, n, ql, 0);
        free(tmp);
    }
    if (scheme == ORC_BGV) {
        /* bgv_moddown_kernel rns_bconv.cu:636-652 with temp_t = bConv(P -> {t}) (:805-807), then NTT (:815) */
        const u64 pt = t->plain_t;
        u64 *tmp = (u64 *)malloc(sizeof(u64) * t->size_p * n);
        u64 *cp_t = (u64 *)malloc(sizeof(u64) * n);
        bconv_mult(&t->p_to_t, cx + ql * n, tmp, n);
        bconv_matmul(&t->p_to_t, tmp, cp_t, n, 1, 0);
        for (size_t j = 0; j < ql; j++)
            for (size_t k = 0; k < n; k++) {
                u64 v = shoup(cp_t[k], t->pinv_mod_t, t->pinv_mod_t_s, pt);
                u64 corr = shoup(v, t->p_mod_q[j], t->p_mod_q_s[j], c->q[j]);
                u64 d = submod(cx[j * n + k], delta[j * n + k], c->q[j]);
                d = addmod(d, corr, c->q[j]);
                ct[j * n + k] = shoup(d, t->pinv[j], t->pinv_s[j], c->q[j]);
            }
        orc_nwt_forward(c, ct, ql, 0);
        free(tmp); free(cp_t); free(delta);
        return;
    }
    if (scheme == ORC_CKKS) orc_nwt_forward(c, delta, ql, 0); /* fused in ntt_moddown.cu:106-261 */
    /* (cx - delta) * P^-1 mod q : sub_negate_const_mult uintmodmath.cuh:233-241 / moddown_kernel :680-689 */
    for (size_t j = 0; j < ql; j++)
        for (size_t k = 0; k < n; k++) {
            u64 d = submod(cx[j * n + k], delta[j * n + k], c->q[j]);
            ct[j * n + k] = shoup(d, t->pinv[j], t->pinv_s[j], c->q[j]);
        }
    free(delta);
}

void orc_moddown(const orc_tool *t, u64 *ct, u64 *cx, int scheme) {
    /* DRNSTool::moddown rns_bconv.cu:712-761.  Differs from moddown_from_NTT (:776-828) in three ways: BFV input is already in
     * coefficient form (no inverse transform, :722-730); alpha = 1 has no special kernel (always bConv_BEHZ, :733, :746); CKKS
     * transforms delta forward and then runs the element-wise moddown_kernel (:748-757) instead of the fused epilogue. */
    const orc_ctx *c = t->c;
    const size_t n = t->n, ql = t->size_ql, qlp = t->size_qlp;
    u64 *delta = (u64 *)malloc(sizeof(u64) * ql * n);
    u64 *tmp = (u64 *)calloc(t->size_p * n, sizeof(u64));
    if (scheme == ORC_CKKS) orc_nwt_backward_map(c, cx + ql * n, t->qlp_idx + ql, t->size_p);     /* :722-725 */
    else if (scheme == ORC_BGV) orc_nwt_backward_map(c, cx, t->qlp_idx, qlp);                      /* :726-730 */
    bconv_mult(&t->p_to_ql, cx + ql * n, tmp, n);                                                  /* bConv_BEHZ :733 / :746 */
    bconv_matmul(&t->p_to_ql, tmp, delta, n, ql, 0);
    if (scheme == ORC_BGV) {   /* :732-744 */
        const u64 pt = t->plain_t;
        u64 *cp_t = (u64 *)malloc(sizeof(u64) * n);
        bconv_mult(&t->p_to_t, cx + ql * n, tmp, n);
        bconv_matmul(&t->p_to_t, tmp, cp_t, n, 1, 0);
        for (size_t j = 0; j < ql; j++)
            for (size_t k = 0; k < n; k++) {
                u64 v = shoup(cp_t[k], t->pinv_mod_t, t->pinv_mod_t_s, pt);
                u64 corr = shoup(v, t->p_mod_q[j], t->p_mod_q_s[j], c->q[j]);
                u64 d = submod(cx[j * n + k], delta[j * n + k], c->q[j]);
                d = addmod(d, corr, c->q[j]);
                ct[j * n + k] = shoup(d, t->pinv[j], t->pinv_s[j], c->q[j]);
            }
        orc_nwt_forward(c, ct, ql, 0);
        free(cp_t); free(tmp); free(delta);
        return;
    }
    if (scheme == ORC_CKKS) orc_nwt_forward(c, delta, ql, 0);                                      /* :748-751 */
    for (size_t j = 0; j < ql; j++)                                                                /* moddown_kernel :680-689 */
        for (size_t k = 0; k < n; k++) {
            u64 d = submod(cx[j * n + k], delta[j * n + k], c->q[j]);
            ct[j * n + k] = shoup(d, t->pinv[j], t->pinv_s[j], c->q[j]);
        }
    free(tmp); free(delta);
}

void orc_keyswitch_inplace(const orc_tool *t, u64 *ct, const u64 *c2, const u64 *const *evks, int scheme) {
    /* keyswitch_inplace eval_key_switch.cu:95-182 (mul_tech != hps_overq_leveled) */
    const orc_ctx *c = t->c;
    const size_t n = t->n, ql = t->size_ql, qlp = t->size_qlp;
    u64 *mu = (u64 *)malloc(sizeof(u64) * t->beta * qlp * n);
    u64 *cx = (u64 *)malloc(sizeof(u64) * 2 * qlp * n);
    orc_modup(t, mu, c2, scheme);
    orc_key_switch_inner_prod(t, cx, mu, evks);
    for (int i = 0; i < 2; i++) {
        u64 *cxi = cx + (size_t)i * qlp * n;
        orc_moddown_from_ntt(t, cxi, cxi, scheme);
        orc_add_rns_poly(c, ct + (size_t)i * ql * n, cxi, ct + (size_t)i * ql * n, ql, 0); /* add_to_ct_kernel :763-769 */
    }
    free(mu);
    free(cx);
}

void orc_rescale_ntt(const orc_tool *t, u64 *src, size_t cipher_size, u64 *dst) {
    /* divide_and_round_q_last_ntt rns.cu:1160-1184 (floors; the "+half" of :1118 is not implemented) */
    const orc_ctx *c = t->c;
    const size_t n = t->n, ql = t->size_ql, nl = ql - 1;
    for (size_t p = 0; p < cipher_size; p++) {
        u64 *in = src + p * ql * n, *out = dst + p * nl * n;
        inv1(c, in + nl * n, nl);
        for (size_t j = 0; j < nl; j++)
            for (size_t k = 0; k < n; k++) out[j * n + k] = barrett64(in[nl * n + k], c->q[j], c->mu[j][1]);
        orc_nwt_forward(c, out, nl, 0);
        for (size_t j = 0; j < nl; j++)
            for (size_t k = 0; k < n; k++) {
                u64 d = submod(in[j * n + k], out[j * n + k], c->q[j]);
                out[j * n + k] = shoup(d, t->inv_q_last[j], t->inv_q_last_s[j], c->q[j]);
            }
    }
}

void orc_mod_t_divide_q_last_ntt(const orc_tool *t, u64 *src, size_t cipher_size, u64 *dst) {
    /* DRNSTool::mod_t_and_divide_q_last_ntt rns.cu:1186-1236 (BGV modulus switching; src is clobbered) */
    const orc_ctx *c = t->c;
    const size_t n = t->n, ql = t->size_ql, nl = ql - 1;
    const u64 pt = t->plain_t;
    for (size_t p = 0; p < cipher_size; p++) {
        u64 *in = src + p * ql * n, *out = dst + p * nl * n;
        orc_nwt_backward(c, in, ql, 0);
        for (size_t j = 0; j < nl; j++)
            for (size_t k = 0; k < n; k++) {
                u64 last = in[nl * n + k];
                u64 delta = barrett64(last, c->q[j], c->mu[j][1]);
                u64 last_t = barrett64(last, pt, t->t_mu[1]);
                u64 v = shoup(last_t, t->inv_q_last_mod_t, t->inv_q_last_mod_t_s, pt);
                u64 corr = shoup(v, t->q_last_mod_q[j], t->q_last_mod_q_s[j], c->q[j]);
                u64 d = addmod(submod(in[j * n + k], delta, c->q[j]), corr, c->q[j]);
                out[j * n + k] = shoup(d, t->inv_q_last[j], t->inv_q_last_s[j], c->q[j]);
            }
        orc_nwt_forward(c, out, nl, 0);
    }
}

void orc_divide_and_round_q_last(const orc_tool *t, const u64 *src, size_t cipher_size, u64 *dst) {
    /* rns.cu:1082-1126 */
    const orc_ctx *c = t->c;
    const size_t n = t->n, ql = t->size_ql, nl = ql - 1;
    for (size_t p = 0; p < cipher_size; p++)
        for (size_t j = 0; j < nl; j++)
            for (size_t k = 0; k < n; k++) {
                u64 last = barrett64(src[p * ql * n + nl * n + k], c->q[j], c->mu[j][1]);
                u64 d = submod(src[p * ql * n + j * n + k], last, c->q[j]);
                dst[p * nl * n + j * n + k] = shoup(d, t->inv_q_last[j], t->inv_q_last_s[j], c->q[j]);
            }
}

/* ------------------------------------------------------------------------------------------------
 * BFV multiply, BEHZ variant (src/evaluate.cu:404-548; constants src/rns.cu:392-560; kernels
 * src/rns.cu:1249-1510, src/polymath.cu:606-634).  Top data level only (base q = every data prime).
 * ---------------------------------------------------------------------------------------------- */
static int inv_euclid(u64 a, u64 m, u64 *out) { /* a^-1 mod m for any m >= 2 (try_invert_uint_mod) */
    __int128 r0 = m, r1 = a % m, s0 = 0, s1 = 1;
    while (r1) { __int128 qq = r0 / r1, tmp = r0 - qq * r1; r0 = r1; r1 = tmp; tmp = s0 - qq * s1; s0 = s1; s1 = tmp; }
    if (r0 != 1) return -1;
    *out = (u64)((s0 % (__int128)m + m) % m);
    return 0;
}
static u64 prod_mod(const u64 *v, size_t cnt, u64 m) {
    u64 p = 1 % m;
    for (size_t i = 0; i < cnt; i++) p = orc_mulmod(p, v[i] % m, m);
    return p;
}
static int prod_bit_count(const u64 *v, size_t cnt) { /* get_significant_bit_count_uint(prod v) */
    u64 *acc = (u64 *)calloc(cnt + 1, sizeof(u64));
    size_t len = 1;
    acc[0] = 1;
    for (size_t i = 0; i < cnt; i++) {
        u64 carry = 0;
        for (size_t w = 0; w < len; w++) {
            u128 t = (u128)acc[w] * v[i] + carry;
            acc[w] = (u64)t;
            carry = (u64)(t >> 64);
        }
        if (carry) acc[len++] = carry;
    }
    int bits = (int)(len - 1) * 64 + (64 - __builtin_clzll(acc[len - 1]));
    free(acc);
    return bits;
}
struct orc_behz {
    const orc_ctx *c;
    size_t n, size_q, size_b, size_bsk;
    int log_n;
    u64 plain_t, m_tilde, m_sk;
    u64 *bsk;                 /* [size_bsk] = B primes then m_sk (rns.cu:413-431) */
    u64 (*bsk_mu)[2];
    u64 *tw, *tws, *itw, *itws, *n_inv, *n_inv_s; /* NTT tables of the Bsk primes (:433-448) */
    u64 *mt_qhatinv, *mt_qhatinv_s;               /* m_tilde * QHatInv mod q (:450-465) */
    bconv_t q_to_bsk, q_to_mtilde, b_to_q, b_to_msk;
    u64 *inv_prod_q_mod_bsk, *inv_prod_q_mod_bsk_s, *prod_q_mod_bsk, *inv_mt_mod_bsk, *inv_mt_mod_bsk_s;
    u64 neg_inv_prod_q_mod_mt, neg_inv_prod_q_mod_mt_s, inv_prod_b_mod_msk, inv_prod_b_mod_msk_s;
    u64 *prod_b_mod_q;
};
size_t orc_behz_bsk_size(const orc_behz *b) { return b->size_bsk; }
void orc_behz_base(const orc_behz *b, u64 *out) { memcpy(out, b->bsk, sizeof(u64) * b->size_bsk); }

orc_behz *orc_behz_create(const orc_ctx *c, u64 plain_t) {
    orc_behz *b = (orc_behz *)calloc(1, sizeof(*b));
    b->c = c; b->n = c->n; b->log_n = c->log_n; b->size_q = c->size_q; b->plain_t = plain_t;
    const size_t sq = c->size_q, n = c->n;
    /* bit count of prod(q) and of t: base B grows by one prime when 32 + |t| + |Q| >= 61 * size_q + 61 (:398-406) */
    int total_bits = prod_bit_count(c->q, sq), t_bits = 64 - __builtin_clzll(plain_t);
    b->size_b = sq + ((32 + t_bits + total_bits >= 61 * (int)sq + 61) ? 1 : 0);
    b->size_bsk = b->size_b + 1;
    u64 *aux = (u64 *)malloc(sizeof(u64) * (b->size_b + 1));
    if (orc_get_primes(n, 61, b->size_b + 1, aux)) { free(aux); free(b); return NULL; }
    b->m_sk = aux[0];
    b->m_tilde = (u64)1 << 32;
    b->bsk = (u64 *)malloc(sizeof(u64) * b->size_bsk);
    for (size_t i = 0; i < b->size_b; i++) b->bsk[i] = aux[1 + i];
    b->bsk[b->size_b] = b->m_sk;
    free(aux);
    const size_t sk = b->size_bsk;
    b->bsk_mu = malloc(sizeof(u64[2]) * sk);
    b->tw = (u64 *)malloc(sizeof(u64) * sk * n); b->tws = (u64 *)malloc(sizeof(u64) * sk * n);
    b->itw = (u64 *)malloc(sizeof(u64) * sk * n); b->itws = (u64 *)malloc(sizeof(u64) * sk * n);
    b->n_inv = (u64 *)malloc(sizeof(u64) * sk); b->n_inv_s = (u64 *)malloc(sizeof(u64) * sk);
    for (size_t i = 0; i < sk; i++) {
        orc_const_ratio(b->bsk[i], b->bsk_mu[i]);
        orc_ntt_tables(c->log_n, b->bsk[i], b->tw + i * n, b->tws + i * n, b->itw + i * n, b->itws + i * n,
                       &b->n_inv[i], &b->n_inv_s[i]);
    }
    u64 *bskmt = (u64 *)malloc(sizeof(u64) * (sk + 1));
    memcpy(bskmt, b->bsk, sizeof(u64) * sk);
    bskmt[sk] = b->m_tilde;
    bconv_init(&b->q_to_bsk, c->q, sq, b->bsk, sk);
    bconv_init(&b->q_to_mtilde, c->q, sq, &b->m_tilde, 1);
    bconv_init(&b->b_to_q, b->bsk, b->size_b, c->q, sq);
    bconv_init(&b->b_to_msk, b->bsk, b->size_b, &b->m_sk, 1);
    free(bskmt);
    b->mt_qhatinv = (u64 *)malloc(sizeof(u64) * sq); b->mt_qhatinv_s = (u64 *)malloc(sizeof(u64) * sq);
    for (size_t i = 0; i < sq; i++) {
        b->mt_qhatinv[i] = orc_mulmod(b->m_tilde % c->q[i], b->q_to_bsk.hat_inv[i], c->q[i]);
        b->mt_qhatinv_s[i] = orc_compute_shoup(b->mt_qhatinv[i], c->q[i]);
    }
    b->inv_prod_q_mod_bsk = (u64 *)malloc(sizeof(u64) * sk); b->inv_prod_q_mod_bsk_s = (u64 *)malloc(sizeof(u64) * sk);
    b->prod_q_mod_bsk = (u64 *)malloc(sizeof(u64) * sk);
    b->inv_mt_mod_bsk = (u64 *)malloc(sizeof(u64) * sk); b->inv_mt_mod_bsk_s = (u64 *)malloc(sizeof(u64) * sk);
    for (size_t i = 0; i < sk; i++) {
        const u64 p = b->bsk[i];
        b->prod_q_mod_bsk[i] = prod_mod(c->q, sq, p);                         /* :548-553 */
        inv_euclid(b->prod_q_mod_bsk[i], p, &b->inv_prod_q_mod_bsk[i]);        /* :508-518 */
        b->inv_prod_q_mod_bsk_s[i] = orc_compute_shoup(b->inv_prod_q_mod_bsk[i], p);
        inv_euclid(b->m_tilde % p, p, &b->inv_mt_mod_bsk[i]);                  /* :537-546 */
        b->inv_mt_mod_bsk_s[i] = orc_compute_shoup(b->inv_mt_mod_bsk[i], p);
    }
    u64 inv = 0;
    inv_euclid(prod_mod(c->q, sq, b->m_tilde), b->m_tilde, &inv);              /* -prod(q)^-1 mod m_tilde */
    b->neg_inv_prod_q_mod_mt = (b->m_tilde - inv) % b->m_tilde;
    b->neg_inv_prod_q_mod_mt_s = orc_compute_shoup(b->neg_inv_prod_q_mod_mt, b->m_tilde);
    inv_euclid(prod_mod(b->bsk, b->size_b, b->m_sk), b->m_sk, &b->inv_prod_b_mod_msk); /* :526-535 */
    b->inv_prod_b_mod_msk_s = orc_compute_shoup(b->inv_prod_b_mod_msk, b->m_sk);
    b->prod_b_mod_q = (u64 *)malloc(sizeof(u64) * sq);
    for (size_t i = 0; i < sq; i++) b->prod_b_mod_q[i] = prod_mod(b->bsk, b->size_b, c->q[i]);
    return b;
}
void orc_behz_destroy(orc_behz *b) {
    if (!b) return;
    free(b->bsk); free(b->bsk_mu); free(b->tw); free(b->tws); free(b->itw); free(b->itws); free(b->n_inv); free(b->n_inv_s);
    free(b->mt_qhatinv); free(b->mt_qhatinv_s);
    bconv_free(&b->q_to_bsk); bconv_free(&b->q_to_mtilde); bconv_free(&b->b_to_q); bconv_free(&b->b_to_msk);
    free(b->inv_prod_q_mod_bsk); free(b->inv_prod_q_mod_bsk_s); free(b->prod_q_mod_bsk);
    free(b->inv_mt_mod_bsk); free(b->inv_mt_mod_bsk_s); free(b->prod_b_mod_q);
    free(b);
}

/* DRNSTool::fastbconv_m_tilde rns.cu:1249-1278: src [Q][N] -> dst [Bsk + 1][N] (last limb modulo m_tilde); ONE phase 1 with
 * m_tilde * QHatInv (:1258-1263), then Q -> Bsk (:1265-1270) and Q -> {m_tilde} (:1272-1277) */
void orc_behz_fastbconv_m_tilde(const orc_behz *b, const u64 *src, u64 *dst) {
    const orc_ctx *c = b->c;
    const size_t n = b->n, sq = b->size_q, sk = b->size_bsk;
    u64 *y = (u64 *)malloc(sizeof(u64) * sq * n);
    for (size_t i = 0; i < sq; i++)
        for (size_t k = 0; k < n; k++) y[i * n + k] = shoup(src[i * n + k], b->mt_qhatinv[i], b->mt_qhatinv_s[i], c->q[i]);
    bconv_matmul(&b->q_to_bsk, y, dst, n, sk, 0);
    bconv_matmul(&b->q_to_mtilde, y, dst + sk * n, n, 1, 0);
    free(y);
}
/* DRNSTool::sm_mrq rns.cu:1290-1338 (sm_mrq_kernel :1290-1320): src [Bsk + 1][N] -> dst [Bsk][N] */
void orc_behz_sm_mrq(const orc_behz *b, const u64 *src, u64 *dst) {
    const size_t n = b->n, sk = b->size_bsk;
    const u64 mt = b->m_tilde;
    for (size_t j = 0; j < sk; j++) {
        const u64 p = b->bsk[j];
        for (size_t k = 0; k < n; k++) {
            u64 r = shoup(src[sk * n + k], b->neg_inv_prod_q_mod_mt, b->neg_inv_prod_q_mod_mt_s, mt);
            if (r >= (mt >> 1)) r += p - mt;
            u128 t = (u128)r * b->prod_q_mod_bsk[j] + src[j * n + k];
            u64 v = barrett128(t, p, b->bsk_mu[j]);
            dst[j * n + k] = shoup(v, b->inv_mt_mod_bsk[j], b->inv_mt_mod_bsk_s[j], p);
        }
    }
}
/* DRNSTool::fast_floor rns.cu:1394-1419 (bconv_fuse_sub_mul_unroll2_kernel :1343-1386): (x_Bsk - FastBconv(x_q, q -> Bsk)) *
 * prod(q)^-1 mod Bsk; in_q [Q][N], in_bsk [Bsk][N] -> out_bsk [Bsk][N] */
void orc_behz_fast_floor(const orc_behz *b, const u64 *in_q, const u64 *in_bsk, u64 *out_bsk) {
    const size_t n = b->n, sq = b->size_q, sk = b->size_bsk;
    u64 *y = (u64 *)calloc(sq * n, sizeof(u64)), *conv = (u64 *)malloc(sizeof(u64) * sk * n);
    bconv_mult(&b->q_to_bsk, in_q, y, n);
    bconv_matmul(&b->q_to_bsk, y, conv, n, sk, 0);
    for (size_t j = 0; j < sk; j++)
        for (size_t k = 0; k < n; k++) {
            const u64 v = in_bsk[j * n + k] + (b->bsk[j] - conv[j * n + k]);   /* not reduced before the multiply (:1376-1378) */
            out_bsk[j * n + k] = shoup(v, b->inv_prod_q_mod_bsk[j], b->inv_prod_q_mod_bsk_s[j], b->bsk[j]);
        }
    free(y); free(conv);
}
/* DRNSTool::fastbconv_sk rns.cu:1470-1510: in_bsk [Bsk][N] (B limbs then the m_sk limb) -> out_q [Q][N] */
void orc_behz_fastbconv_sk(const orc_behz *b, const u64 *in_bsk, u64 *out_q) {
    const orc_ctx *c = b->c;
    const size_t n = b->n, sq = b->size_q, sb = b->size_b;
    u64 *yb = (u64 *)calloc(sb * n, sizeof(u64)), *alpha = (u64 *)malloc(sizeof(u64) * n);
    bconv_mult(&b->b_to_q, in_bsk, yb, n);                               /* :1484-1487 (phase 1 of base B, shared) */
    bconv_matmul(&b->b_to_msk, yb, alpha, n, 1, 0);                      /* :1489-1494 with the fused (conv - x_msk) * B^-1 mod m_sk */
    for (size_t k = 0; k < n; k++) {
        const u64 v = alpha[k] + (b->m_sk - in_bsk[sb * n + k]);
        alpha[k] = shoup(v, b->inv_prod_b_mod_msk, b->inv_prod_b_mod_msk_s, b->m_sk);
    }
    bconv_matmul(&b->b_to_q, yb, out_q, n, sq, 0);                       /* :1496-1500 */
    for (size_t i = 0; i < sq; i++)   /* multiply_and_negated_add_rns_poly polymath.cu:606-634 (:1506-1510) */
        for (size_t k = 0; k < n; k++) {
            u64 op1 = alpha[k], pb = b->prod_b_mod_q[i];
            if (op1 > (b->m_sk >> 1)) op1 = b->m_sk - op1;
            else pb = c->q[i] - pb;
            op1 = barrett128((u128)op1 * pb, c->q[i], c->mu[i]);
            out_q[i * n + k] = addmod(out_q[i * n + k], op1, c->q[i]);
        }
    free(yb); free(alpha);
}

/* BEHZ_mul_1 evaluate.cu:404-441: one polynomial x [Q][N] (coefficient form) -> NTT(x) over q and the lifted,
 * Montgomery-reduced polynomial over Bsk in NTT form */
static void behz_lift(const orc_behz *b, const u64 *x, u64 *out_q, u64 *out_bsk) {
    const orc_ctx *c = b->c;
    const size_t n = b->n, sq = b->size_q, sk = b->size_bsk;
    memcpy(out_q, x, sizeof(u64) * sq * n);
    orc_nwt_forward(c, out_q, sq, 0);
    u64 *lift = (u64 *)malloc(sizeof(u64) * (sk + 1) * n);
    orc_behz_fastbconv_m_tilde(b, x, lift);
    orc_behz_sm_mrq(b, lift, out_bsk);
    for (size_t j = 0; j < sk; j++) orc_ntt_forward(out_bsk + j * n, b->log_n, b->bsk[j], b->tw + j * n, b->tws + j * n);
    free(lift);
}
static void tensor_generic(const u64 *a, const u64 *bb, u64 *r, const u64 *q, u64 (*const mu)[2], size_t limbs, size_t n) {
    /* tensor_prod_2x2_rns_poly polymath.cu:463-496 over an arbitrary base; r may alias a */
    const size_t rc = limbs * n;
    for (size_t l = 0; l < limbs; l++)
        for (size_t k = 0; k < n; k++) {
            const size_t id = l * n + k;
            u64 c00 = a[id], c01 = a[id + rc], c10 = bb[id], c11 = bb[id + rc];
            u64 d0 = barrett128((u128)c00 * c10, q[l], mu[l]);
            u64 d2 = barrett128((u128)c01 * c11, q[l], mu[l]);
            u64 d1 = barrett128((u128)(c00 + c01) * (c10 + c11), q[l], mu[l]);
            d1 = csub(csub(d1 + 2 * q[l] - d0 - d2, q[l]), q[l]);
            r[id] = d0; r[id + rc] = d1; r[id + 2 * rc] = d2;
        }
}
void orc_bfv_multiply_behz(const orc_behz *b, const u64 *ct1, const u64 *ct2, u64 *dst) {
    const orc_ctx *c = b->c;
    const size_t n = b->n, sq = b->size_q, sk = b->size_bsk;
    u64 *q1 = (u64 *)calloc(3 * sq * n, 8), *b1 = (u64 *)calloc(3 * sk * n, 8);
    u64 *q2 = (u64 *)calloc(2 * sq * n, 8), *b2 = (u64 *)calloc(2 * sk * n, 8);
    for (int p = 0; p < 2; p++) {
        behz_lift(b, ct1 + p * sq * n, q1 + p * sq * n, b1 + p * sk * n);
        behz_lift(b, ct2 + p * sq * n, q2 + p * sq * n, b2 + p * sk * n);
    }
    /* step 4: dyadic tensor product in both bases (evaluate.cu:479-498) */
    tensor_generic(q1, q2, q1, c->q, c->mu, sq, n);
    tensor_generic(b1, b2, b1, b->bsk, b->bsk_mu, sk, n);
    /* steps 5-6: inverse NTT fused with the multiplication by t, full Shoup reduce (:518-530) */
    for (int p = 0; p < 3; p++) {
        u64 *xq = q1 + p * sq * n, *xb = b1 + p * sk * n;
        orc_nwt_backward(c, xq, sq, 0);
        for (size_t i = 0; i < sq; i++) {
            const u64 ts = orc_compute_shoup(b->plain_t, c->q[i]);
            for (size_t k = 0; k < n; k++) xq[i * n + k] = shoup(xq[i * n + k], b->plain_t, ts, c->q[i]);
        }
        for (size_t j = 0; j < sk; j++) {
            orc_ntt_inverse(xb + j * n, b->log_n, b->bsk[j], b->itw + j * n, b->itws + j * n, b->n_inv[j], b->n_inv_s[j]);
            const u64 ts = orc_compute_shoup(b->plain_t, b->bsk[j]);
            for (size_t k = 0; k < n; k++) xb[j * n + k] = shoup(xb[j * n + k], b->plain_t, ts, b->bsk[j]);
        }
        /* step 7 fast_floor rns.cu:1394-1419, step 8 fastbconv_sk rns.cu:1470-1510 */
        u64 *fl = (u64 *)malloc(sizeof(u64) * sk * n);
        orc_behz_fast_floor(b, xq, xb, fl);
        orc_behz_fastbconv_sk(b, fl, dst + p * sq * n);
        free(fl);
    }
    free(q1); free(b1); free(q2); free(b2);
}

/* ------------------------------------------------------------------------------------------------
 * BFV multiply, HPS variant (mul_tech_type::hps; src/evaluate.cu:674-818, bConv_HPS src/rns_bconv.cu:248-372,
 * scaleAndRound_HPS_QR_R src/rns.cu:1700-1746, constants src/rns.cu:687-790, src/host/rns.cu:321-323,459-466).
 * Floating point: the reference's kernels are built by nvcc with its default -fmad=true (no flag in its CMake
 * files), so `acc += double(x) * c` is ONE fused multiply-add per term; the same fma chain is restated here.
 * ---------------------------------------------------------------------------------------------- */
typedef struct { u64 *w; size_t len; } big_t;
static big_t big_one(size_t cap) { big_t b; b.w = (u64 *)calloc(cap, sizeof(u64)); b.w[0] = 1; b.len = 1; return b; }
static void big_mul_small(big_t *b, u64 m) {
    u64 carry = 0;
    for (size_t i = 0; i < b->len; i++) { u128 t = (u128)b->w[i] * m + carry; b->w[i] = (u64)t; carry = (u64)(t >> 64); }
    if (carry) b->w[b->len++] = carry;
}
static u64 big_mod_small(const big_t *b, u64 m) {
    u128 r = 0;
    for (size_t i = b->len; i-- > 0;) r = ((r << 64) | b->w[i]) % m;
    return (u64)r;
}
static void big_div_small(big_t *b, u64 m) { /* b = floor(b / m) */
    u128 r = 0;
    for (size_t i = b->len; i-- > 0;) { u128 cur = (r << 64) | b->w[i]; b->w[i] = (u64)(cur / m); r = cur % m; }
    while (b->len > 1 && b->w[b->len - 1] == 0) b->len--;
}
typedef struct {
    bconv_t conv;
    double *inv;        /* 1 / q_i (src/host/rns.cu:321-323) */
    u64 *alpha_q_mod_p; /* [isz + 1][osz]  alpha * Q mod p_j (:459-466) */
} hpsconv_t;
static void hpsconv_init(hpsconv_t *h, const u64 *ib, size_t isz, const u64 *ob, size_t osz) {
    bconv_init(&h->conv, ib, isz, ob, osz);
    h->inv = (double *)malloc(sizeof(double) * isz);
    for (size_t i = 0; i < isz; i++) h->inv[i] = 1.0 / (double)ib[i];
    h->alpha_q_mod_p = (u64 *)malloc(sizeof(u64) * (isz + 1) * osz);
    for (size_t j = 0; j < osz; j++) {
        const u64 qm = prod_mod(ib, isz, ob[j]);
        for (size_t a = 0; a <= isz; a++) h->alpha_q_mod_p[a * osz + j] = orc_mulmod(a, qm, ob[j]);
    }
}
static void hpsconv_free(hpsconv_t *h) { bconv_free(&h->conv); free(h->inv); free(h->alpha_q_mod_p); }
/* DBaseConverter::bConv_HPS rns_bconv.cu:354-372 */
static void hpsconv_apply(const hpsconv_t *h, const u64 *src, u64 *dst, size_t n) {
    const bconv_t *b = &h->conv;
    u64 *y = (u64 *)malloc(sizeof(u64) * b->isz * n);
    bconv_mult(b, src, y, n);
    for (size_t k = 0; k < n; k++) {
        double frac = 0.0;
        for (size_t i = 0; i < b->isz; i++) frac = fma((double)y[i * n + k], h->inv[i], frac);
        const size_t v = (size_t)llround(frac);
        for (size_t j = 0; j < b->osz; j++) {
            u128 acc = 0;
            for (size_t i = 0; i < b->isz; i++) acc += (u128)y[i * n + k] * b->mat[j * b->isz + i];
            const u64 out = barrett128(acc, b->ob[j], b->omu[j]);
            dst[j * n + k] = submod(out, h->alpha_q_mod_p[v * b->osz + j], b->ob[j]);
        }
    }
    free(y);
}
/* DBaseConverter::bConv_HPS for arbitrary bases (rns_bconv.cu:248-372), exported for the converter-object parity test */
void orc_bconv_hps(const u64 *ibase, size_t isz, const u64 *obase, size_t osz, const u64 *src, u64 *dst, size_t n) {
    hpsconv_t h;
    hpsconv_init(&h, ibase, isz, obase, osz);
    hpsconv_apply(&h, src, dst, n);
    hpsconv_free(&h);
}

struct orc_hps {
    const orc_ctx *c;
    size_t n, size_q, size_r;
    int log_n;
    u64 *r, *qr;                 /* R primes, then Q || R */
    u64 (*qr_mu)[2];
    u64 *tw, *tws, *itw, *itws, *n_inv, *n_inv_s; /* tables of Q || R */
    hpsconv_t q_to_r, r_to_q;
    double *frac;                /* [Q]  tRSHatInvModsDivsFrac */
    u64 *div_mod_r;              /* [R][Q + 1] tRSHatInvModsDivsModr */
};
size_t orc_hps_r_size(const orc_hps *h) { return h->size_r; }
void orc_hps_base(const orc_hps *h, u64 *out) { memcpy(out, h->r, sizeof(u64) * h->size_r); }
orc_hps *orc_hps_create(const orc_ctx *c, u64 plain_t) {
    orc_hps *h = (orc_hps *)calloc(1, sizeof(*h));
    const size_t sq = c->size_q, sr = sq + 1, sqr = sq + sr, n = c->n;
    h->c = c; h->n = n; h->log_n = c->log_n; h->size_q = sq; h->size_r = sr;
    /* get_primes_below(n, min q, size_R) src/host/numth.cu:235-263 */
    u64 minq = c->q[0];
    for (size_t i = 1; i < sq; i++) if (c->q[i] < minq) minq = c->q[i];
    h->r = (u64 *)malloc(sizeof(u64) * sr);
    {
        const u64 factor = 2 * (u64)n, lower = (u64)1 << (63 - __builtin_clzll(minq));
        size_t found = 0;
        for (u64 v = minq - factor; found < sr && v > lower; v -= factor)
            if (orc_is_prime(v)) h->r[found++] = v;
        if (found < sr) { free(h->r); free(h); return NULL; }
    }
    h->qr = (u64 *)malloc(sizeof(u64) * sqr);
    memcpy(h->qr, c->q, sizeof(u64) * sq);
    memcpy(h->qr + sq, h->r, sizeof(u64) * sr);
    h->qr_mu = malloc(sizeof(u64[2]) * sqr);
    h->tw = (u64 *)malloc(sizeof(u64) * sqr * n); h->tws = (u64 *)malloc(sizeof(u64) * sqr * n);
    h->itw = (u64 *)malloc(sizeof(u64) * sqr * n); h->itws = (u64 *)malloc(sizeof(u64) * sqr * n);
    h->n_inv = (u64 *)malloc(sizeof(u64) * sqr); h->n_inv_s = (u64 *)malloc(sizeof(u64) * sqr);
    for (size_t i = 0; i < sqr; i++) {
        orc_const_ratio(h->qr[i], h->qr_mu[i]);
        orc_ntt_tables(c->log_n, h->qr[i], h->tw + i * n, h->tws + i * n, h->itw + i * n, h->itws + i * n, &h->n_inv[i], &h->n_inv_s[i]);
    }
    hpsconv_init(&h->q_to_r, c->q, sq, h->r, sr);
    hpsconv_init(&h->r_to_q, h->r, sr, c->q, sq);
    /* t/Q scale-and-round tables (rns.cu:727-790): S = Q || R, x_i = t * R * (S/s_i)^-1 mod s_i as big integers */
    h->frac = (double *)malloc(sizeof(double) * sq);
    h->div_mod_r = (u64 *)malloc(sizeof(u64) * sr * (sq + 1));
    for (size_t i = 0; i < sqr; i++) {
        u64 hat = 1;                                    /* (S / s_i) mod s_i, then its inverse */
        for (size_t k = 0; k < sqr; k++) if (k != i) hat = orc_mulmod(hat, h->qr[k] % h->qr[i], h->qr[i]);
        const u64 shat_inv = orc_invmod(hat, h->qr[i]);
        big_t x = big_one(sr + 4);
        for (size_t k = 0; k < sr; k++) big_mul_small(&x, h->r[k]);
        big_mul_small(&x, plain_t);
        big_mul_small(&x, shat_inv);
        if (i < sq) h->frac[i] = (double)big_mod_small(&x, h->qr[i]) / (double)h->qr[i];
        big_div_small(&x, h->qr[i]);
        if (i < sq) {
            for (size_t j = 0; j < sr; j++) h->div_mod_r[j * (sq + 1) + i] = big_mod_small(&x, h->r[j]);
        } else {
            const size_t j = i - sq;
            h->div_mod_r[j * (sq + 1) + sq] = big_mod_small(&x, h->r[j]);
        }
        free(x.w);
    }
    return h;
}
void orc_hps_destroy(orc_hps *h) {
    if (!h) return;
    free(h->r); free(h->qr); free(h->qr_mu); free(h->tw); free(h->tws); free(h->itw); free(h->itws); free(h->n_inv); free(h->n_inv_s);
    hpsconv_free(&h->q_to_r); hpsconv_free(&h->r_to_q); free(h->frac); free(h->div_mod_r); free(h);
}
/* DRNSTool::scaleAndRound_HPS_QR_R rns.cu:1700-1746: src [Q + R][N] coefficient form -> dst [R][N], scaled by t / Q and rounded;
 * nu accumulates by fused multiply-adds (nvcc's default -fmad), alpha is reduced IN PLACE across the R limbs (:1733) */
void orc_hps_scale_round_qr_r(const orc_hps *h, const u64 *x, u64 *dst) {
    const size_t n = h->n, sq = h->size_q, sr = h->size_r;
    for (size_t k = 0; k < n; k++) {
        double nu = 0.5;
        for (size_t i = 0; i < sq; i++) nu = fma((double)x[i * n + k], h->frac[i], nu);
        u64 alpha = (u64)nu;
        for (size_t j = 0; j < sr; j++) {
            const u64 rj = h->qr[sq + j];
            const u64 *tab = h->div_mod_r + j * (sq + 1);
            u128 cur = 0;
            for (size_t i = 0; i < sq; i++) cur += (u128)x[i * n + k] * tab[i];
            cur += (u128)x[(sq + j) * n + k] * tab[sq];
            const u64 v = barrett128(cur, rj, h->qr_mu[sq + j]);
            alpha = barrett64(alpha, rj, h->qr_mu[sq + j][1]);
            dst[j * n + k] = addmod(v, alpha, rj);
        }
    }
}
void orc_bfv_multiply_hps(const orc_hps *h, const u64 *ct1, const u64 *ct2, u64 *dst) {
    const size_t n = h->n, sq = h->size_q, sr = h->size_r, sqr = sq + sr;
    u64 *x1 = (u64 *)calloc(3 * sqr * n, 8), *x2 = (u64 *)calloc(2 * sqr * n, 8);
    for (int p = 0; p < 2; p++) {
        const u64 *src[2] = {ct1 + p * sq * n, ct2 + p * sq * n};
        u64 *dstp[2] = {x1 + p * sqr * n, x2 + p * sqr * n};
        for (int w = 0; w < 2; w++) {
            memcpy(dstp[w], src[w], sizeof(u64) * sq * n);
            hpsconv_apply(&h->q_to_r, dstp[w], dstp[w] + sq * n, n);     /* evaluate.cu:716 */
            for (size_t i = 0; i < sqr; i++)
                orc_ntt_forward(dstp[w] + i * n, h->log_n, h->qr[i], h->tw + i * n, h->tws + i * n);
        }
    }
    tensor_generic(x1, x2, x1, h->qr, h->qr_mu, sqr, n);
    u64 *tmp = (u64 *)malloc(sizeof(u64) * sr * n);
    for (int p = 0; p < 3; p++) {
        u64 *x = x1 + p * sqr * n;
        for (size_t i = 0; i < sqr; i++)
            orc_ntt_inverse(x + i * n, h->log_n, h->qr[i], h->itw + i * n, h->itws + i * n, h->n_inv[i], h->n_inv_s[i]);
        orc_hps_scale_round_qr_r(h, x, tmp);                                /* scaleAndRound_HPS_QR_R evaluate.cu:800-803 */
        hpsconv_apply(&h->r_to_q, tmp, dst + p * sq * n, n);                /* evaluate.cu:806 */
    }
    free(tmp); free(x1); free(x2);
}

/* ------------------------------------------------------------------------------------------------
 * BFV multiply, HPS-over-Q variant (mul_tech_type::hps_overq, no levels dropped; src/evaluate.cu:674-818 with the
 * hps_overq branches :745-751 and :790-792; bConv_BEHZ_var1 src/rns_bconv.cu:231-246 with the constants of
 * src/host/rns.cu:469-496; scaleAndRound_HPS_QlRl_Ql src/rns.cu:1748-1796; constants src/rns.cu:792-885).
 * Base Rl has as many primes as Q (the first |Q| primes below the smallest q_i).  The first operand is lifted
 * exactly (bConv_HPS); the second goes Q -> Rl through the quotient-style conversion var1 and comes back Rl -> Q
 * exactly; the product is scaled by t / Rl and rounded straight into base Q.
 * ---------------------------------------------------------------------------------------------- */
struct orc_hpsq {
    const orc_ctx *c;
    size_t n, size_q, size_r;
    int log_n;
    u64 *r, *qr;
    u64 (*qr_mu)[2];
    u64 *tw, *tws, *itw, *itws, *n_inv, *n_inv_s;
    hpsconv_t q_to_r, r_to_q;
    bconv_t q_to_r_var1;         /* hat_inv := -Rl * qhat_i^-1 mod q_i, mat[j][i] := q_i^-1 mod r_j; from the FULL base Q when levels are dropped */
    double *frac;                /* [R]  tQlSlHatInvModsDivsFrac */
    u64 *div_mod_q;              /* [Q][R + 1] tQlSlHatInvModsDivsModq */
    /* hps_overq_leveled with levels dropped (rns.cu:897-975): size_q above is |Ql|, the chain has size_q_full primes */
    size_t size_q_full, drop;
    double *frac_drop;           /* [drop]          QlQHatInvModqDivqFrac */
    u64 *div_mod_q_drop;         /* [Ql][drop + 1]  QlQHatInvModqDivqModq */
    u64 *drop_mod_q;             /* [Ql]            prod(dropped primes) mod q_i (base_Ql_to_QlDrop_conv.PModq) */
};
size_t orc_hpsq_r_size(const orc_hpsq *h) { return h->size_r; }
void orc_hpsq_base(const orc_hpsq *h, u64 *out) { memcpy(out, h->r, sizeof(u64) * h->size_r); }
orc_hpsq *orc_hpsq_create_level(const orc_ctx *c, u64 plain_t, size_t size_ql) {
    if (size_ql < 1 || size_ql > c->size_q) return NULL;
    orc_hpsq *h = (orc_hpsq *)calloc(1, sizeof(*h));
    const size_t sq = size_ql, sr = sq, sqr = sq + sr, n = c->n, sq_full = c->size_q;
    h->c = c; h->n = n; h->log_n = c->log_n; h->size_q = sq; h->size_r = sr; h->size_q_full = sq_full; h->drop = sq_full - sq;
    u64 minq = c->q[0];                                  /* over the FULL chain (rns.cu:581, :800) */
    for (size_t i = 1; i < sq_full; i++) if (c->q[i] < minq) minq = c->q[i];
    h->r = (u64 *)malloc(sizeof(u64) * sr);
    {
        const u64 factor = 2 * (u64)n, lower = (u64)1 << (63 - __builtin_clzll(minq));
        size_t found = 0;
        for (u64 v = minq - factor; found < sr && v > lower; v -= factor)
            if (orc_is_prime(v)) h->r[found++] = v;
        if (found < sr) { free(h->r); free(h); return NULL; }
    }
    h->qr = (u64 *)malloc(sizeof(u64) * sqr);
    memcpy(h->qr, c->q, sizeof(u64) * sq);
    memcpy(h->qr + sq, h->r, sizeof(u64) * sr);
    h->qr_mu = malloc(sizeof(u64[2]) * sqr);
    h->tw = (u64 *)malloc(sizeof(u64) * sqr * n); h->tws = (u64 *)malloc(sizeof(u64) * sqr * n);
    h->itw = (u64 *)malloc(sizeof(u64) * sqr * n); h->itws = (u64 *)malloc(sizeof(u64) * sqr * n);
    h->n_inv = (u64 *)malloc(sizeof(u64) * sqr); h->n_inv_s = (u64 *)malloc(sizeof(u64) * sqr);
    for (size_t i = 0; i < sqr; i++) {
        orc_const_ratio(h->qr[i], h->qr_mu[i]);
        orc_ntt_tables(c->log_n, h->qr[i], h->tw + i * n, h->tws + i * n, h->itw + i * n, h->itws + i * n, &h->n_inv[i], &h->n_inv_s[i]);
    }
    hpsconv_init(&h->q_to_r, c->q, sq, h->r, sr);
    hpsconv_init(&h->r_to_q, h->r, sr, c->q, sq);
    /* bConv_BEHZ_var1 constants (src/host/rns.cu:469-496): negPQHatInvModq_i = q_i - (P mod q_i) * qhat_i^-1, QInvModp[j][i] =
     * q_i^-1 mod p_j.  With levels dropped the second operand is converted from the FULL base Q (base_Q_to_Rl_conv, rns.cu:911) */
    {
        const size_t isz = h->drop ? sq_full : sq;
        bconv_init(&h->q_to_r_var1, c->q, isz, h->r, sr);
        for (size_t i = 0; i < isz; i++) {
            const u64 qi = c->q[i];
            const u64 pm = prod_mod(h->r, sr, qi);
            const u64 v = qi - orc_mulmod(pm, h->q_to_r_var1.hat_inv[i], qi);
            h->q_to_r_var1.hat_inv[i] = v;
            h->q_to_r_var1.hat_inv_s[i] = orc_compute_shoup(v, qi);
        }
        for (size_t j = 0; j < sr; j++)
            for (size_t i = 0; i < isz; i++) h->q_to_r_var1.mat[j * isz + i] = orc_invmod(c->q[i] % h->r[j], h->r[j]);
    }
    /* t/Rl scale-and-round tables (rns.cu:836-885): S = Ql || Rl, x_i = t * Ql * (S/s_i)^-1 mod s_i as big integers */
    h->frac = (double *)malloc(sizeof(double) * sr);
    h->div_mod_q = (u64 *)malloc(sizeof(u64) * sq * (sr + 1));
    for (size_t i = 0; i < sqr; i++) {
        u64 hat = 1;
        for (size_t k = 0; k < sqr; k++) if (k != i) hat = orc_mulmod(hat, h->qr[k] % h->qr[i], h->qr[i]);
        const u64 shat_inv = orc_invmod(hat, h->qr[i]);
        big_t x = big_one(sq + 4);
        for (size_t k = 0; k < sq; k++) big_mul_small(&x, c->q[k]);
        big_mul_small(&x, plain_t);
        big_mul_small(&x, shat_inv);
        if (i >= sq) h->frac[i - sq] = (double)big_mod_small(&x, h->qr[i]) / (double)h->qr[i];
        big_div_small(&x, h->qr[i]);
        if (i >= sq) {
            for (size_t l = 0; l < sq; l++) h->div_mod_q[l * (sr + 1) + (i - sq)] = big_mod_small(&x, c->q[l]);
        } else {
            h->div_mod_q[i * (sr + 1) + sr] = big_mod_small(&x, c->q[i]);
        }
        free(x.w);
    }
    if (h->drop) {
        /* Ql/Q scale-and-round tables (rns.cu:918-972): x_i = Ql * (Q/q_i)^-1 mod q_i as big integers, i over the full chain */
        const size_t dr = h->drop;
        h->frac_drop = (double *)malloc(sizeof(double) * dr);
        h->div_mod_q_drop = (u64 *)malloc(sizeof(u64) * sq * (dr + 1));
        h->drop_mod_q = (u64 *)malloc(sizeof(u64) * sq);
        for (size_t i = 0; i < sq_full; i++) {
            u64 hat = 1;
            for (size_t k = 0; k < sq_full; k++) if (k != i) hat = orc_mulmod(hat, c->q[k] % c->q[i], c->q[i]);
            const u64 qhat_inv = orc_invmod(hat, c->q[i]);
            big_t x = big_one(sq + 4);
            for (size_t k = 0; k < sq; k++) big_mul_small(&x, c->q[k]);
            big_mul_small(&x, qhat_inv);
            if (i >= sq) h->frac_drop[i - sq] = (double)big_mod_small(&x, c->q[i]) / (double)c->q[i];
            big_div_small(&x, c->q[i]);
            if (i >= sq) {
                for (size_t l = 0; l < sq; l++) h->div_mod_q_drop[l * (dr + 1) + (i - sq)] = big_mod_small(&x, c->q[l]);
            } else {
                h->div_mod_q_drop[i * (dr + 1) + dr] = big_mod_small(&x, c->q[i]);
            }
            free(x.w);
        }
        for (size_t i = 0; i < sq; i++) h->drop_mod_q[i] = prod_mod(c->q + sq, dr, c->q[i]);
    }
    return h;
}
orc_hpsq *orc_hpsq_create(const orc_ctx *c, u64 plain_t) { return orc_hpsq_create_level(c, plain_t, c->size_q); }
void orc_hpsq_destroy(orc_hpsq *h) {
    if (!h) return;
    free(h->r); free(h->qr); free(h->qr_mu); free(h->tw); free(h->tws); free(h->itw); free(h->itws); free(h->n_inv); free(h->n_inv_s);
    hpsconv_free(&h->q_to_r); hpsconv_free(&h->r_to_q); bconv_free(&h->q_to_r_var1); free(h->frac); free(h->div_mod_q);
    free(h->frac_drop); free(h->div_mod_q_drop); free(h->drop_mod_q); free(h);
}
/* ct2 == ct1 (the same pointer) takes the reference's squaring path: one exact lift, tensor_square (evaluate.cu:720-731).
 * That path still scales by t / Rl, so its result is (Q / Rl) times the wanted one (it does not decrypt to the square);
 * restated as is. */
/* the scale-and-round kernel shared by scaleAndRound_HPS_QlRl_Ql and scaleAndRound_HPS_Q_Ql (rns.cu:1749-1808):
 * src = [Ql limbs || extra limbs], dst = [Ql limbs]; alpha is reduced in place across the Ql limbs (:1783) */
static void hps_scale_round_to_ql(const orc_hpsq *h, const u64 *src, u64 *dst, const double *frac, const u64 *tab, size_t extra) {
    const size_t n = h->n, sq = h->size_q;
    for (size_t k = 0; k < n; k++) {
        double nu = 0.5;
        for (size_t j = 0; j < extra; j++) nu = fma((double)src[(sq + j) * n + k], frac[j], nu);
        u64 alpha = (u64)nu;
        for (size_t i = 0; i < sq; i++) {
            const u64 qi = h->qr[i];
            const u64 *row = tab + i * (extra + 1);
            u128 cur = 0;
            for (size_t j = 0; j < extra; j++) cur += (u128)src[(sq + j) * n + k] * row[j];
            cur += (u128)src[i * n + k] * row[extra];
            const u64 v = barrett128(cur, qi, h->qr_mu[i]);
            alpha = barrett64(alpha, qi, h->qr_mu[i][1]);
            dst[i * n + k] = addmod(v, alpha, qi);
        }
    }
}
/* scaleAndRound_HPS_Q_Ql (rns.cu:1798-1808): [Q][N] -> [Ql][N], scaled by Ql / Q and rounded */
void orc_hps_scale_q_ql(const orc_hpsq *h, const u64 *src, u64 *dst) {
    hps_scale_round_to_ql(h, src, dst, h->frac_drop, h->div_mod_q_drop, h->drop);
}
/* ExpandCRTBasis_Ql_Q (rns.cu:1810-1836): [Ql][N] -> [Q][N]: times the product of the dropped primes on the Ql limbs,
 * zero on the dropped limbs; works in place when dst == src (the limb stride is N either way) */
void orc_hps_expand_ql_q(const orc_hpsq *h, const u64 *src, u64 *dst) {
    const size_t n = h->n, sq = h->size_q;
    for (size_t i = 0; i < sq; i++)
        for (size_t k = 0; k < n; k++) dst[i * n + k] = orc_mulmod(src[i * n + k], h->drop_mod_q[i], h->qr[i]);
    memset(dst + sq * n, 0, sizeof(u64) * h->drop * n);
}
/* scaleAndRound_HPS_QlRl_Ql (rns.cu:1748-1796): src [Ql + Rl][N] -> dst [Ql][N], scaled by t / Rl and rounded */
void orc_hpsq_scale_round_qlrl_ql(const orc_hpsq *h, const u64 *src, u64 *dst) {
    hps_scale_round_to_ql(h, src, dst, h->frac, h->div_mod_q, h->size_r);
}
/* ExpandCRTBasis_Ql_Q_add_to_ct (rns.cu:1838-1858): dst[i] += src[i] * prod(dropped primes) on the Ql limbs */
void orc_hpsq_expand_add_to_ct(const orc_hpsq *h, const u64 *src, u64 *dst) {
    const size_t n = h->n, sq = h->size_q;
    for (size_t i = 0; i < sq; i++)
        for (size_t k = 0; k < n; k++)
            dst[i * n + k] = addmod(orc_mulmod(src[i * n + k], h->drop_mod_q[i], h->qr[i]), dst[i * n + k], h->qr[i]);
}
/* bfv_multiply_hps with mul_tech hps_overq (h built at the top level) or hps_overq_leveled with levels dropped (h built at a
 * lower level: evaluate.cu:709-711, :747-748, :794-795).  Operands and result are over the FULL base Q: [.][Q][N]. */
static void hpsq_multiply(const orc_hpsq *h, const u64 *ct1, const u64 *ct2, u64 *dst, int keep_c2_low) {
    const size_t n = h->n, sq = h->size_q, sr = h->size_r, sqr = sq + sr, sqf = h->size_q_full;
    const int square = ct1 == ct2;
    u64 *x1 = (u64 *)calloc(3 * sqr * n, 8), *x2 = (u64 *)calloc(2 * sqr * n, 8), *y = (u64 *)malloc(sizeof(u64) * sqf * n);
    for (int p = 0; p < 2; p++) {
        u64 *a = x1 + p * sqr * n, *b = x2 + p * sqr * n;
        if (h->drop) orc_hps_scale_q_ql(h, ct1 + p * sqf * n, a);             /* evaluate.cu:709-710 */
        else memcpy(a, ct1 + p * sq * n, sizeof(u64) * sq * n);
        hpsconv_apply(&h->q_to_r, a, a + sq * n, n);                           /* :716 */
        for (size_t i = 0; i < sqr; i++) orc_ntt_forward(a + i * n, h->log_n, h->qr[i], h->tw + i * n, h->tws + i * n);
        if (square) { memcpy(b, a, sizeof(u64) * sqr * n); continue; }
        bconv_mult(&h->q_to_r_var1, ct2 + p * sqf * n, y, n);                  /* bConv_BEHZ_var1 :745-749 (from the full Q when leveled) */
        bconv_matmul(&h->q_to_r_var1, y, b + sq * n, n, (size_t)-1, 0);
        hpsconv_apply(&h->r_to_q, b + sq * n, b, n);                           /* :750 */
        for (size_t i = 0; i < sqr; i++) orc_ntt_forward(b + i * n, h->log_n, h->qr[i], h->tw + i * n, h->tws + i * n);
    }
    tensor_generic(x1, x2, x1, h->qr, h->qr_mu, sqr, n);
    for (int p = 0; p < 3; p++) {
        u64 *x = x1 + p * sqr * n, *out = dst + p * sqf * n;
        for (size_t i = 0; i < sqr; i++)
            orc_ntt_inverse(x + i * n, h->log_n, h->qr[i], h->itw + i * n, h->itws + i * n, h->n_inv[i], h->n_inv_s[i]);
        orc_hpsq_scale_round_qlrl_ql(h, x, out);                               /* scaleAndRound_HPS_QlRl_Ql :790-792 */
        if (h->drop && !(keep_c2_low && p == 2)) orc_hps_expand_ql_q(h, out, out);   /* :794-795 (:957-958 leaves c2 at level l) */
    }
    free(x1); free(x2); free(y);
}
void orc_bfv_multiply_hps_overq(const orc_hpsq *h, const u64 *ct1, const u64 *ct2, u64 *dst) { hpsq_multiply(h, ct1, ct2, dst, 0); }
/* bfv_mul_relin_hps with levels dropped (evaluate.cu:822-1027): the product's c2 stays at level l (:957-958), is switched
 * there, and the two results are expanded onto the Ql limbs of (c0, c1) (ExpandCRTBasis_Ql_Q_add_to_ct :1014-1016).
 * dst [2][Q][N]; t = the tool of level l. */
void orc_bfv_mul_relin_hps_overq_leveled(const orc_tool *t, const orc_hpsq *h, const u64 *ct1, const u64 *ct2,
                                         const u64 *const *evks, u64 *dst) {
    const size_t n = t->n, qlp = t->size_qlp, sqf = h->size_q_full;
    u64 *d3 = (u64 *)malloc(sizeof(u64) * 3 * sqf * n);
    u64 *mu = (u64 *)malloc(sizeof(u64) * t->beta * qlp * n);
    u64 *cx = (u64 *)malloc(sizeof(u64) * 2 * qlp * n);
    hpsq_multiply(h, ct1, ct2, d3, 1);
    orc_modup(t, mu, d3 + 2 * sqf * n, ORC_BFV);
    orc_key_switch_inner_prod(t, cx, mu, evks);
    memcpy(dst, d3, sizeof(u64) * 2 * sqf * n);
    for (int i = 0; i < 2; i++) {
        u64 *cxi = cx + (size_t)i * qlp * n, *ct = dst + (size_t)i * sqf * n;
        orc_moddown_from_ntt(t, cxi, cxi, ORC_BFV);
        orc_hpsq_expand_add_to_ct(h, cxi, ct);                                 /* :1014-1016 */
    }
    free(d3); free(mu); free(cx);
}
/* keyswitch_inplace for BFV under hps_overq_leveled with levels dropped (eval_key_switch.cu:142-147, :170-175): c2 is
 * scaled from Q down to Ql, switched at that level, and both results are expanded back to Q before they are added.
 * t is the tool of the level (size_ql = h->size_q); ct [2][Q][N] and c2 [Q][N] are over the full base. */
void orc_keyswitch_bfv_leveled(const orc_tool *t, const orc_hpsq *h, u64 *ct, const u64 *c2, const u64 *const *evks) {
    const orc_ctx *c = t->c;
    const size_t n = t->n, ql = t->size_ql, qlp = t->size_qlp, sqf = h->size_q_full;
    u64 *c2l = (u64 *)malloc(sizeof(u64) * ql * n);
    u64 *mu = (u64 *)malloc(sizeof(u64) * t->beta * qlp * n);
    u64 *cx = (u64 *)malloc(sizeof(u64) * 2 * qlp * n);
    u64 *full = (u64 *)malloc(sizeof(u64) * sqf * n);
    orc_hps_scale_q_ql(h, c2, c2l);
    orc_modup(t, mu, c2l, ORC_BFV);
    orc_key_switch_inner_prod(t, cx, mu, evks);
    for (int i = 0; i < 2; i++) {
        u64 *cxi = cx + (size_t)i * qlp * n;
        orc_moddown_from_ntt(t, cxi, cxi, ORC_BFV);
        orc_hps_expand_ql_q(h, cxi, full);
        orc_add_rns_poly(c, ct + (size_t)i * sqf * n, full, ct + (size_t)i * sqf * n, sqf, 0);
    }
    free(c2l); free(mu); free(cx); free(full);
}

/* ------------------------------------------------------------------------------------------------
 * batched modular GEMM (benchmark/matmul_bench.cu:215-541), exact: C = A * B mod q, row-major.
 * (The reference's kernels add low and high product words separately and lose the low word's carries, :231-232;
 * orc_gemm_mod_ref_quirk restates that arithmetic so that the difference can be shown on data.)
 * ---------------------------------------------------------------------------------------------- */
void orc_gemm_mod(u64 q, const u64 *A, const u64 *B, u64 *C, size_t m, size_t n, size_t k) {
    u64 mu[2];
    orc_const_ratio(q, mu);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < m; i++)
        for (size_t j = 0; j < n; j++) {
            u64 acc = 0;
            for (size_t l = 0; l < k; l++) acc = addmod(acc, barrett128((u128)A[i * k + l] * B[l * n + j], q, mu), q);
            C[i * n + j] = acc;
        }
}
void orc_gemm_mod_ref_quirk(u64 q, const u64 *A, const u64 *B, u64 *C, size_t m, size_t n, size_t k) {
    u64 mu[2];
    orc_const_ratio(q, mu);
    for (size_t i = 0; i < m; i++)
        for (size_t j = 0; j < n; j++) {
            u64 lo = 0, hi = 0;
            for (size_t l = 0; l < k; l++) {
                const u128 p = (u128)A[i * k + l] * B[l * n + j];
                lo += (u64)p;            /* no carry into hi: matmul_bench.cu:231-232 */
                hi += (u64)(p >> 64);
            }
            C[i * n + j] = barrett128(((u128)hi << 64) | lo, q, mu);
        }
}

/* ------------------------------------------------------------------------------------------------
 * Galois (include/galois.cuh:98-130, src/galois.cu:11-39)
 * ---------------------------------------------------------------------------------------------- */
void orc_galois_ntt_table(int log_n, uint32_t elt, uint32_t *table) {
    uint32_t n = 1u << log_n;
    for (uint32_t i = n; i < 2 * n; i++) {
        uint32_t rev = brev(i, log_n + 1);
        u64 raw = ((u64)elt * rev) >> 1;
        raw &= (u64)(n - 1);
        table[i - n] = brev((uint32_t)raw, log_n);
    }
}
void orc_apply_galois_ntt(const u64 *src, u64 *dst, const uint32_t *table, size_t n, size_t limbs) {
    for (size_t l = 0; l < limbs; l++)
        for (size_t k = 0; k < n; k++) dst[l * n + k] = src[l * n + table[k]];
}
void orc_apply_galois_coeff(const orc_ctx *c, const u64 *src, u64 *dst, uint32_t elt, size_t limbs, size_t start) {
    const size_t n = c->n;
    for (size_t l = 0; l < limbs; l++) {
        u64 q = c->q[start + l];
        u64 raw = 0;
        for (size_t k = 0; k < n; k++) {
            u64 v = src[l * n + k];
            if (raw >= n) v = negmod(v, q);
            dst[l * n + (raw % n)] = v;
            raw = (raw + elt) & (2 * n - 1);
        }
    }
}

void orc_hoisting(const orc_tool *t, u64 *ct, const uint32_t *elts, size_t n_elts, const u64 *const *const *glk,
                  int scheme) {
    /* hoisting_inplace src/evaluate.cu:1670-1866 (mul_tech != hps_overq_leveled): ct <- sum over the
     * Galois elements of rotate(ct): one mod-up of c1, per element a permutation of c0 and of every
     * mod-up digit + an inner product with that element's key, one pair of mod-downs at the end. */
    const orc_ctx *c = t->c;
    const size_t n = t->n, ql = t->size_ql, qlp = t->size_qlp, beta = t->beta;
    u64 *c0 = (u64 *)malloc(sizeof(u64) * ql * n), *c1 = (u64 *)malloc(sizeof(u64) * ql * n);
    u64 *acc_c0 = (u64 *)calloc(ql * n, sizeof(u64)), *tmp_c0 = (u64 *)malloc(sizeof(u64) * ql * n);
    u64 *mu = (u64 *)malloc(sizeof(u64) * beta * qlp * n), *pmu = (u64 *)malloc(sizeof(u64) * beta * qlp * n);
    u64 *acc_cx = (u64 *)calloc(2 * qlp * n, sizeof(u64)), *tmp_cx = (u64 *)malloc(sizeof(u64) * 2 * qlp * n);
    uint32_t *table = (uint32_t *)malloc(sizeof(uint32_t) * n);
    memcpy(c0, ct, sizeof(u64) * ql * n);
    memcpy(c1, ct + ql * n, sizeof(u64) * ql * n);
    orc_modup(t, mu, c1, scheme);
    for (size_t e = 0; e < n_elts; e++) {
        orc_galois_ntt_table(c->log_n, elts[e], table);
        if (scheme == ORC_BFV) orc_apply_galois_coeff(c, c0, tmp_c0, elts[e], ql, 0);
        else orc_apply_galois_ntt(c0, tmp_c0, table, n, ql);
        orc_add_rns_poly(c, acc_c0, tmp_c0, acc_c0, ql, 0);
        for (size_t b = 0; b < beta; b++) orc_apply_galois_ntt(mu + b * qlp * n, pmu + b * qlp * n, table, n, qlp);
        orc_key_switch_inner_prod(t, tmp_cx, pmu, glk[e]);
        for (int p = 0; p < 2; p++)
            for (size_t j = 0; j < qlp; j++) {
                const u64 q = c->q[t->qlp_idx[j]];
                for (size_t k = 0; k < n; k++) {
                    const size_t id = (size_t)p * qlp * n + j * n + k;
                    acc_cx[id] = addmod(acc_cx[id], tmp_cx[id], q);
                }
            }
    }
    orc_moddown_from_ntt(t, acc_cx, acc_cx, scheme);
    orc_moddown_from_ntt(t, acc_cx + qlp * n, acc_cx + qlp * n, scheme);
    orc_add_rns_poly(c, acc_c0, acc_cx, ct, ql, 0);
    memcpy(ct + ql * n, acc_cx + qlp * n, sizeof(u64) * ql * n);
    free(c0); free(c1); free(acc_c0); free(tmp_c0); free(mu); free(pmu); free(acc_cx); free(tmp_cx); free(table);
}

void orc_hoisting_weighted(const orc_tool *t, u64 *ct, const uint32_t *elts, size_t n_elts,
                           const u64 *const *const *glk, const u64 *const *weights, int scheme) {
    /* Build-defined composition for BASELINE config 5 (no reference counterpart): the
     * hoisted rotations of hoisting_inplace (src/evaluate.cu:1670-1866) with a plaintext weight per rotation,
     *     ct <- sum_e w_e (.) rotate_e(ct),
     * the diagonal form of a plaintext-matrix x encrypted-vector product.  The weights are multiplied in BEFORE
     * the shared mod-down, so they are given over the extended base [Q_l || P] (NTT form, [QlP][N]):
     *   acc_c0 += w_e (.) galois_e(c0)           over Q_l
     *   acc_cx += w_e (.) <galois_e(modup(c1)), key_e>   over Q_l P   (key_switch_inner_prod, eval_key_switch.cu:14-69)
     *   ct <- (acc_c0 + moddown(acc_cx0), moddown(acc_cx1)).
     * Galois element 1 (the main diagonal) needs no key: it adds w_e (.) (c0, c1).  NTT-domain schemes only. */
    const orc_ctx *c = t->c;
    const size_t n = t->n, ql = t->size_ql, qlp = t->size_qlp, beta = t->beta;
    u64 *c0 = (u64 *)malloc(sizeof(u64) * ql * n), *c1 = (u64 *)malloc(sizeof(u64) * ql * n);
    u64 *acc_c0 = (u64 *)calloc(ql * n, sizeof(u64)), *acc_c1 = (u64 *)calloc(ql * n, sizeof(u64));
    u64 *tmp_c0 = (u64 *)malloc(sizeof(u64) * ql * n);
    u64 *mu = (u64 *)malloc(sizeof(u64) * beta * qlp * n), *pmu = (u64 *)malloc(sizeof(u64) * beta * qlp * n);
    u64 *acc_cx = (u64 *)calloc(2 * qlp * n, sizeof(u64)), *tmp_cx = (u64 *)malloc(sizeof(u64) * 2 * qlp * n);
    uint32_t *table = (uint32_t *)malloc(sizeof(uint32_t) * n);
    memcpy(c0, ct, sizeof(u64) * ql * n);
    memcpy(c1, ct + ql * n, sizeof(u64) * ql * n);
    orc_modup(t, mu, c1, scheme);
    for (size_t e = 0; e < n_elts; e++) {
        const u64 *w = weights[e];
        if (elts[e] == 1) {
            orc_multiply_rns_poly(c, c0, w, tmp_c0, ql, 0);
            orc_add_rns_poly(c, acc_c0, tmp_c0, acc_c0, ql, 0);
            orc_multiply_rns_poly(c, c1, w, tmp_c0, ql, 0);
            orc_add_rns_poly(c, acc_c1, tmp_c0, acc_c1, ql, 0);
            continue;
        }
        orc_galois_ntt_table(c->log_n, elts[e], table);
        orc_apply_galois_ntt(c0, tmp_c0, table, n, ql);
        orc_multiply_rns_poly(c, tmp_c0, w, tmp_c0, ql, 0);
        orc_add_rns_poly(c, acc_c0, tmp_c0, acc_c0, ql, 0);
        for (size_t b = 0; b < beta; b++) orc_apply_galois_ntt(mu + b * qlp * n, pmu + b * qlp * n, table, n, qlp);
        orc_key_switch_inner_prod(t, tmp_cx, pmu, glk[e]);
        for (int p = 0; p < 2; p++)
            for (size_t j = 0; j < qlp; j++) {
                const u64 q = c->q[t->qlp_idx[j]];
                for (size_t k = 0; k < n; k++) {
                    const size_t id = (size_t)p * qlp * n + j * n + k;
                    acc_cx[id] = addmod(acc_cx[id], orc_mulmod(tmp_cx[id], w[j * n + k], q), q);
                }
            }
    }
    orc_moddown_from_ntt(t, acc_cx, acc_cx, scheme);
    orc_moddown_from_ntt(t, acc_cx + qlp * n, acc_cx + qlp * n, scheme);
    orc_add_rns_poly(c, acc_c0, acc_cx, ct, ql, 0);
    orc_add_rns_poly(c, acc_c1, acc_cx + qlp * n, ct + ql * n, ql, 0);
    free(c0); free(c1); free(acc_c0); free(acc_c1); free(tmp_c0); free(mu); free(pmu); free(acc_cx); free(tmp_cx); free(table);
}

/* ------------------------------------------------------------------------------------------------
 * ciphertext (+|-|*) plaintext (src/evaluate.cu:1105-1340, src/scalingvariant.cu:10-60)
 * ---------------------------------------------------------------------------------------------- */
void orc_bfv_add_plain(const orc_ctx *c, size_t size_ql, u64 *ct, const u64 *plain, u64 t, int subtract) {
    /* bfv_{add,sub}_timesQ_overt_kernel polymath.cu:413-461 with negQl_mod_t = t - (Ql mod t), tInv_mod_q (rns.cu:292-324) */
    const size_t n = c->n;
    u64 ql_t = 1 % t;
    for (size_t i = 0; i < size_ql; i++) ql_t = orc_mulmod(ql_t, c->q[i] % t, t);
    const u64 neg = t - ql_t;
    for (size_t i = 0; i < size_ql; i++) {
        const u64 q = c->q[i];
        /* t^-1 mod q_i (q_i prime) */
        const u64 tinv = orc_invmod(t % q, q);
        for (size_t k = 0; k < n; k++) {
            const u64 m = orc_mulmod(plain[k], neg, t);
            const u64 v = orc_mulmod(m, tinv, q);
            ct[i * n + k] = subtract ? submod(ct[i * n + k], v, q) : addmod(ct[i * n + k], v, q);
        }
    }
}
void orc_bgv_lift_plain(const orc_ctx *c, size_t size_ql, const u64 *plain, u64 *out) {
    /* nwt_2d_radix8_forward_modup_fuse per limb (evaluate.cu:1150-1154): NTT of the coefficients modulo every q_i */
    const size_t n = c->n;
    for (size_t i = 0; i < size_ql; i++)
        for (size_t k = 0; k < n; k++) out[i * n + k] = plain[k] % c->q[i];
    orc_nwt_forward(c, out, size_ql, 0);
}
void orc_bfv_multiply_plain(const orc_ctx *c, size_t size_ql, u64 *ct, size_t cipher_size, const u64 *plain, u64 t) {
    /* multiply_plain_normal evaluate.cu:1256-1300: centred lift (abs_plain_rns_poly polymath.cu:645-664), NTT, product, iNTT */
    const size_t n = c->n;
    u64 *tmp = (u64 *)malloc(sizeof(u64) * size_ql * n);
    const u64 threshold = (t + 1) >> 1;
    for (size_t i = 0; i < size_ql; i++)
        for (size_t k = 0; k < n; k++) tmp[i * n + k] = plain[k] >= threshold ? plain[k] + (c->q[i] - t) : plain[k];
    orc_nwt_forward(c, tmp, size_ql, 0);
    for (size_t p = 0; p < cipher_size; p++) {
        u64 *ci = ct + p * size_ql * n;
        orc_nwt_forward(c, ci, size_ql, 0);
        orc_multiply_rns_poly(c, ci, tmp, ci, size_ql, 0);
        orc_nwt_backward(c, ci, size_ql, 0);
    }
    free(tmp);
}

/* ------------------------------------------------------------------------------------------------
 * evaluation-key generation (src/secretkey.cu:232-341, polymath.cu:318-338)
 * evk_i = ( -(a_i*s + e_i) + P*new_key on limbs [i*alpha,(i+1)*alpha) , a_i ), all NTT form, over QP
 * ---------------------------------------------------------------------------------------------- */
void orc_gen_kswitch_key(const orc_ctx *c, const u64 *sk, const u64 *new_key, const u64 *a, const u64 *e, u64 *evk) {
    const size_t n = c->n, qp = c->size_qp, alpha = c->size_p, dnum = c->size_q / alpha;
    for (size_t d = 0; d < dnum; d++) {
        u64 *b_out = evk + d * 2 * qp * n, *a_out = b_out + qp * n;
        for (size_t j = 0; j < qp; j++) {
            u64 q = c->q[j];
            u64 pmod = 0;
            int in_digit = (j >= d * alpha && j < (d + 1) * alpha);
            if (in_digit) {
                pmod = 1;
                for (size_t k = 0; k < alpha; k++) pmod = orc_mulmod(pmod, c->q[c->size_q + k] % q, q);
            }
            for (size_t k = 0; k < n; k++) {
                size_t id = d * qp * n + j * n + k;
                u64 as = orc_mulmod(a[id], sk[j * n + k], q);
                u64 b = negmod(addmod(as, e[id], q), q);
                if (in_digit) b = addmod(b, orc_mulmod(new_key[j * n + k], pmod, q), q);
                b_out[j * n + k] = b;
                a_out[j * n + k] = a[id];
            }
        }
    }
}
