// test_host_api.cpp -- exercises the C++ host mirror (phantom-fhe_amd/host/phantom.h) the way the
// reference's examples drive evaluate.* (examples/3_ckks.cu:447-520: multiply, relinearize, rescale,
// rotate), on synthetic ciphertexts/keys, and compares every result bit for bit with the CPU oracle.
// Needs a GPU; built and run by tests/test_gpu_host_api.py.
#include <phantom.h>

#include <cstdio>
#include <cstring>
#include <random>
#include <sstream>

extern "C" {
#include "../../oracle/oracle.h"
}

using namespace phantom;
using namespace phantom::arith;

static std::vector<uint64_t> uniform(std::mt19937_64 &g, const std::vector<uint64_t> &primes, size_t n) {
    std::vector<uint64_t> v(primes.size() * n);
    for (size_t i = 0; i < primes.size(); i++)
        for (size_t k = 0; k < n; k++) v[i * n + k] = g() % primes[i];
    return v;
}
#define REQUIRE(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)
template <class F>
static bool throws_invalid(F f) {
    try { f(); } catch (const std::invalid_argument &) { return true; } catch (...) { return false; }
    return false;
}

int main() {
    const size_t n = 4096, alpha = 2;
    const int log_n = 12;
    EncryptionParameters parms(scheme_type::ckks);
    parms.set_poly_modulus_degree(n);
    parms.set_special_modulus_size(alpha);
    parms.set_coeff_modulus(CoeffModulus::Create(n, {60, 40, 40, 40, 40, 40, 60, 60}));
    PhantomContext context(parms);
    const size_t size_qp = 8, size_q = 6, dnum = size_q / alpha;
    REQUIRE(context.using_keyswitching() && context.get_first_index() == 1 && context.total_parm_size() == 1 + size_q);
    REQUIRE(context.get_context_data(3).parms().coeff_modulus().size() == 4);
    std::vector<uint64_t> qp;
    for (auto &m : parms.coeff_modulus()) qp.push_back(m.value());
    const std::vector<uint64_t> q(qp.begin(), qp.begin() + size_q);

    orc_ctx *oc = orc_ctx_create(log_n, qp.data(), size_qp, alpha);
    orc_tool *tool = orc_tool_create(oc, size_q);
    std::mt19937_64 g(0x5EED0000 + 7);

    // synthetic keys: relin key + one Galois key (step 1)
    auto make_key = [&](std::vector<std::vector<uint64_t>> &host, PhantomRelinKey &key) {
        std::vector<uint64_t> flat;
        for (size_t d = 0; d < dnum; d++) {
            std::vector<uint64_t> k;
            for (int h = 0; h < 2; h++) { auto part = uniform(g, qp, n); k.insert(k.end(), part.begin(), part.end()); }
            flat.insert(flat.end(), k.begin(), k.end());
            host.push_back(std::move(k));
        }
        key.load_from_host(context, flat.data(), dnum);
    };
    std::vector<std::vector<uint64_t>> rlk_host, glk_host;
    PhantomRelinKey rlk, glk1;
    make_key(rlk_host, rlk);
    make_key(glk_host, glk1);
    const uint32_t elt1 = util::get_elt_from_step(1, n);
    REQUIRE(elt1 == 5 && util::get_elt_from_step(0, n) == 2 * n - 1 && util::get_elt_from_step(-1, n) != 5);
    PhantomGaloisKey glk;
    glk.add(elt1, std::move(glk1));
    std::vector<const uint64_t *> rlk_ptrs, glk_ptrs;
    for (auto &k : rlk_host) rlk_ptrs.push_back(k.data());
    for (auto &k : glk_host) glk_ptrs.push_back(k.data());

    // ciphertexts at the top data level (chain index 1), NTT form
    std::vector<uint64_t> h1, h2;
    for (int p = 0; p < 2; p++) { auto a = uniform(g, q, n); h1.insert(h1.end(), a.begin(), a.end()); }
    for (int p = 0; p < 2; p++) { auto a = uniform(g, q, n); h2.insert(h2.end(), a.begin(), a.end()); }
    PhantomCiphertext ct1, ct2;
    ct1.load_from_host(context, 1, 2, h1.data());
    ct2.load_from_host(context, 1, 2, h2.data());
    ct1.set_scale(std::pow(2.0, 40));
    ct2.set_scale(std::pow(2.0, 40));
    const size_t ln = size_q * n;

    // add / sub / negate
    {
        auto sum = add(context, ct1, ct2), diff = sub(context, ct1, ct2), neg = negate(context, ct1);
        std::vector<uint64_t> got(2 * ln), ref(2 * ln);
        sum.store_to_host(got.data());
        for (int p = 0; p < 2; p++) orc_add_rns_poly(oc, h1.data() + p * ln, h2.data() + p * ln, ref.data() + p * ln, size_q, 0);
        REQUIRE(got == ref);
        diff.store_to_host(got.data());
        for (int p = 0; p < 2; p++) orc_sub_rns_poly(oc, h1.data() + p * ln, h2.data() + p * ln, ref.data() + p * ln, size_q, 0);
        REQUIRE(got == ref);
        neg.store_to_host(got.data());
        for (int p = 0; p < 2; p++) orc_negate_rns_poly(oc, h1.data() + p * ln, ref.data() + p * ln, size_q, 0);
        REQUIRE(got == ref);
    }

    // add_plain / sub_plain / multiply_plain (CKKS: residue-wise on NTT-form plaintexts), add_many, mod_switch_to
    {
        auto hp = uniform(g, q, n);
        PhantomPlaintext pt;
        pt.load_from_host(hp.data(), size_q, n, 1, std::pow(2.0, 40));
        std::vector<uint64_t> got(2 * ln), ref = h1;
        auto s1 = add_plain(context, ct1, pt);
        orc_add_rns_poly(oc, h1.data(), hp.data(), ref.data(), size_q, 0);
        s1.store_to_host(got.data());
        REQUIRE(got == ref);
        auto s2 = sub_plain(context, ct1, pt);
        orc_sub_rns_poly(oc, h1.data(), hp.data(), ref.data(), size_q, 0);
        s2.store_to_host(got.data());
        REQUIRE(got == ref);
        auto s3 = multiply_plain(context, ct1, pt);
        for (int p = 0; p < 2; p++) orc_multiply_rns_poly(oc, h1.data() + p * ln, hp.data(), ref.data() + p * ln, size_q, 0);
        s3.store_to_host(got.data());
        REQUIRE(got == ref && s3.scale() == std::pow(2.0, 80));
        PhantomPlaintext wrong = pt;
        wrong.set_scale(2.0);
        REQUIRE(throws_invalid([&] { PhantomCiphertext c = ct1; add_plain_inplace(context, c, wrong); }));
        // plaintext file round trip and level switch (keeps the leading limbs)
        std::stringstream io;
        pt.save(io);
        PhantomPlaintext back;
        back.load(io);
        REQUIRE(back.chain_index() == 1 && back.coeff_modulus_size() == size_q && back.scale() == pt.scale());
        PhantomPlaintext lower = mod_switch_to(context, back, 3);
        REQUIRE(lower.chain_index() == 3 && lower.coeff_modulus_size() == size_q - 2);
        std::vector<uint64_t> low((size_q - 2) * n);
        lower.store_to_host(low.data());
        REQUIRE(std::equal(low.begin(), low.end(), hp.begin()));
        REQUIRE(throws_invalid([&] { PhantomPlaintext x = lower; mod_switch_to_inplace(context, x, 1); }));
        // add_many: one kernel over the operand table
        std::vector<PhantomCiphertext> many{ct1, ct2, ct1};
        PhantomCiphertext total;
        add_many(context, many, total);
        for (int p = 0; p < 2; p++) {
            orc_add_rns_poly(oc, h1.data() + p * ln, h2.data() + p * ln, ref.data() + p * ln, size_q, 0);
            orc_add_rns_poly(oc, ref.data() + p * ln, h1.data() + p * ln, ref.data() + p * ln, size_q, 0);
        }
        total.store_to_host(got.data());
        REQUIRE(got == ref && total.chain_index() == 1);
        REQUIRE(throws_invalid([&] { std::vector<PhantomCiphertext> none; add_many(context, none, total); }));
        // ciphertext mod_switch_to: CKKS drops limbs
        PhantomCiphertext dropped = mod_switch_to(context, ct1, 3);
        REQUIRE(dropped.chain_index() == 3 && dropped.coeff_modulus_size() == size_q - 2);
        std::vector<uint64_t> dd(2 * (size_q - 2) * n);
        dropped.store_to_host(dd.data());
        for (int p = 0; p < 2; p++) REQUIRE(std::equal(dd.begin() + p * (size_q - 2) * n, dd.begin() + (p + 1) * (size_q - 2) * n, h1.begin() + p * ln));
        REQUIRE(throws_invalid([&] { PhantomCiphertext x = dropped; mod_switch_to_inplace(context, x, 1); }));
    }

    // multiply -> relinearize -> rescale (examples/3_ckks.cu:496-498)
    std::vector<uint64_t> ref3(3 * ln);
    orc_tensor_prod_2x2(oc, h1.data(), h2.data(), ref3.data(), size_q);
    PhantomCiphertext prod = multiply(context, ct1, ct2);
    REQUIRE(prod.size() == 3 && prod.scale() == std::pow(2.0, 80));
    {
        std::vector<uint64_t> got(3 * ln);
        prod.store_to_host(got.data());
        REQUIRE(got == ref3);
    }
    relinearize_inplace(context, prod, rlk);
    REQUIRE(prod.size() == 2);
    std::vector<uint64_t> ref2(ref3.begin(), ref3.begin() + 2 * ln);
    orc_keyswitch_inplace(tool, ref2.data(), ref3.data() + 2 * ln, rlk_ptrs.data(), ORC_CKKS);
    {
        std::vector<uint64_t> got(2 * ln);
        prod.store_to_host(got.data());
        REQUIRE(got == ref2);
    }
    PhantomCiphertext sq = ct1;   // square path + multiply_and_relin
    multiply_and_relin_inplace(context, sq, sq, rlk);
    {
        std::vector<uint64_t> r3(3 * ln), got(2 * ln);
        orc_tensor_square_2x2(oc, h1.data(), r3.data(), size_q);
        std::vector<uint64_t> r2(r3.begin(), r3.begin() + 2 * ln);
        orc_keyswitch_inplace(tool, r2.data(), r3.data() + 2 * ln, rlk_ptrs.data(), ORC_CKKS);
        sq.store_to_host(got.data());
        REQUIRE(got == r2);
    }
    PhantomCiphertext rescaled = rescale_to_next(context, prod);
    REQUIRE(rescaled.chain_index() == 2 && rescaled.coeff_modulus_size() == size_q - 1);
    REQUIRE(rescaled.scale() == std::pow(2.0, 80) / static_cast<double>(q.back()));
    {
        std::vector<uint64_t> src = ref2, ref((size_q - 1) * n * 2), got((size_q - 1) * n * 2);
        orc_rescale_ntt(tool, src.data(), 2, ref.data());
        rescaled.store_to_host(got.data());
        REQUIRE(got == ref);
        std::vector<uint64_t> still(2 * ln);
        prod.store_to_host(still.data());
        REQUIRE(still == ref2);  // rescale_to_next leaves its input untouched (it works on a copy)
        // the fused second half (extension): multiply -> relinearize_rescale gives the same ciphertext in one call
        PhantomCiphertext fused = multiply_relin_rescale(context, ct1, ct2, rlk);
        REQUIRE(fused.size() == 2 && fused.chain_index() == 2 && fused.coeff_modulus_size() == size_q - 1);
        REQUIRE(fused.scale() == rescaled.scale() && fused.is_ntt_form());
        fused.store_to_host(got.data());
        REQUIRE(got == ref);
        REQUIRE(throws_invalid([&] { (void)relinearize_rescale(context, prod, rlk); }));   // needs a size-3 ciphertext
    }

    // rotate by one slot with a synthetic Galois key (apply_galois_inplace evaluate.cu:1567-1630)
    {
        PhantomCiphertext rot = rotate(context, ct1, 1, glk);
        std::vector<uint32_t> table(n);
        orc_galois_ntt_table(log_n, elt1, table.data());
        std::vector<uint64_t> ref(2 * ln, 0), c1g(ln), got(2 * ln);
        orc_apply_galois_ntt(h1.data(), ref.data(), table.data(), n, size_q);          // c0 <- galois(c0)
        orc_apply_galois_ntt(h1.data() + ln, c1g.data(), table.data(), n, size_q);     // temp <- galois(c1); c1 <- 0
        orc_keyswitch_inplace(tool, ref.data(), c1g.data(), glk_ptrs.data(), ORC_CKKS);
        rot.store_to_host(got.data());
        REQUIRE(got == ref);
        REQUIRE(throws_invalid([&] { rotate_inplace(context, rot, 2, glk); }));   // power of two without a key
        REQUIRE(throws_invalid([&] { apply_galois_inplace(context, rot, 25, glk); }));
    }

    // hoisted rotations: sum of rotate(ct, 1) and rotate(ct, 2) with one mod-up (evaluate.cu:1670-1866)
    {
        std::vector<std::vector<uint64_t>> glk2_host;
        PhantomRelinKey glk2;
        make_key(glk2_host, glk2);
        const uint32_t elt2 = util::get_elt_from_step(2, n);
        PhantomGaloisKey both;
        PhantomRelinKey glk1_again;
        {
            std::vector<uint64_t> flat;
            for (auto &k : glk_host) flat.insert(flat.end(), k.begin(), k.end());
            glk1_again.load_from_host(context, flat.data(), dnum);
        }
        both.add(elt1, std::move(glk1_again));
        both.add(elt2, std::move(glk2));
        PhantomCiphertext h = hoisting(context, ct1, both, {1, 2});
        std::vector<const uint64_t *> g2_ptrs;
        for (auto &k : glk2_host) g2_ptrs.push_back(k.data());
        const uint64_t *const *tabs[2] = {glk_ptrs.data(), g2_ptrs.data()};
        const uint32_t elts[2] = {elt1, elt2};
        std::vector<uint64_t> ref = h1, got(2 * ln);
        orc_hoisting(tool, ref.data(), elts, 2, tabs, ORC_CKKS);
        h.store_to_host(got.data());
        REQUIRE(got == ref);
        bool threw = false;
        try { hoisting_inplace(context, h, both, {3}); } catch (const std::logic_error &) { threw = true; }
        REQUIRE(threw);  // "Galois key not present in hoisting" is a logic_error in the reference
    }

    // pre-condition checks of the reference
    REQUIRE(throws_invalid([&] { relinearize_inplace(context, ct1, rlk); }));            // size must be 3
    {
        PhantomCiphertext c = ct1;
        c.set_ntt_form(false);
        REQUIRE(throws_invalid([&] { multiply_inplace(context, c, ct2); }));              // must be in NTT form
        PhantomCiphertext low = mod_switch_to_next(context, ct1);
        REQUIRE(low.chain_index() == 2 && low.coeff_modulus_size() == size_q - 1);
        REQUIRE(throws_invalid([&] { multiply_inplace(context, low, ct2); }));            // chain index mismatch
        PhantomCiphertext last = ct1;
        while (last.coeff_modulus_size() > 1) mod_switch_to_next_inplace(context, last);
        REQUIRE(throws_invalid([&] { rescale_to_next_inplace(context, last); }));         // no next parameters
    }
    // key generation with caller-supplied randomness (src/secretkey.cu:297-341) + on-disk formats
    // (include/secretkey.h:129-163, include/ciphertext.h:173-214)
    {
        std::vector<uint64_t> sk(size_qp * n), e(dnum * size_qp * n), a;
        for (size_t k = 0; k < n; k++) {
            const int v = static_cast<int>(g() % 3) - 1;
            for (size_t j = 0; j < size_qp; j++) sk[j * n + k] = v < 0 ? qp[j] - 1 : static_cast<uint64_t>(v);
        }
        for (size_t d = 0; d < dnum; d++) {
            auto part = uniform(g, qp, n);
            a.insert(a.end(), part.begin(), part.end());
            for (size_t k = 0; k < n; k++) {
                const int v = static_cast<int>(g() % 7) - 3;
                for (size_t j = 0; j < size_qp; j++)
                    e[(d * size_qp + j) * n + k] = v < 0 ? qp[j] - static_cast<uint64_t>(-v) : static_cast<uint64_t>(v);
            }
        }
        orc_nwt_forward(oc, sk.data(), size_qp, 0);
        std::vector<uint64_t> s2(size_q * n), e_ntt = e, ref(dnum * 2 * size_qp * n);
        orc_multiply_rns_poly(oc, sk.data(), sk.data(), s2.data(), size_q, 0);
        for (size_t d = 0; d < dnum; d++) orc_nwt_forward(oc, e_ntt.data() + d * size_qp * n, size_qp, 0);
        orc_gen_kswitch_key(oc, sk.data(), s2.data(), a.data(), e_ntt.data(), ref.data());
        const auto &st = cudaStreamPerThread;
        auto up = [&](const std::vector<uint64_t> &h) {
            auto d = util::make_cuda_auto_ptr<uint64_t>(h.size(), st);
            util::check_hip(hipMemcpyAsync(d.get(), h.data(), h.size() * 8, hipMemcpyHostToDevice, st), "hipMemcpyAsync");
            return d;
        };
        auto d_sk = up(sk), d_s2 = up(s2), d_a = up(a), d_e = up(e);
        PhantomRelinKey gen;
        gen.generate(context, d_sk.get(), d_s2.get(), d_a.get(), d_e.get());
        REQUIRE(gen.generated() && gen.dnum() == dnum);
        std::vector<uint64_t> got(2 * size_qp * n);
        for (size_t d = 0; d < dnum; d++) {
            gen.public_key(d).store_to_host(got.data());
            REQUIRE(std::memcmp(got.data(), ref.data() + d * 2 * size_qp * n, got.size() * 8) == 0);
        }
        std::stringstream file;
        gen.save(file);
        const size_t header = 4 * sizeof(size_t) + sizeof(double) + sizeof(uint64_t) + sizeof(size_t) + 2 * sizeof(bool);
        REQUIRE(file.str().size() == sizeof(size_t) + dnum * (header + 2 * size_qp * n * 8));
        {
            size_t w[5];
            std::memcpy(w, file.str().data(), sizeof(w));   // dnum | chain_index, size, degree, limbs of key 0
            REQUIRE(w[0] == dnum && w[1] == 0 && w[2] == 2 && w[3] == n && w[4] == size_qp);
        }
        PhantomRelinKey back;
        back.load(file);
        for (size_t d = 0; d < dnum; d++) {
            back.public_key(d).store_to_host(got.data());
            REQUIRE(std::memcmp(got.data(), ref.data() + d * 2 * size_qp * n, got.size() * 8) == 0);
        }
        // the reloaded key relinearizes like the oracle's copy of it
        std::vector<const uint64_t *> ptrs;
        for (size_t d = 0; d < dnum; d++) ptrs.push_back(ref.data() + d * 2 * size_qp * n);
        PhantomCiphertext p3 = multiply(context, ct1, ct2);
        relinearize_inplace(context, p3, back);
        std::vector<uint64_t> r2(ref3.begin(), ref3.begin() + 2 * ln), out(2 * ln);
        orc_keyswitch_inplace(tool, r2.data(), ref3.data() + 2 * ln, ptrs.data(), ORC_CKKS);
        p3.store_to_host(out.data());
        REQUIRE(out == r2);
        std::stringstream cfile;
        p3.save(cfile);
        PhantomCiphertext p4;
        p4.load(cfile);
        REQUIRE(p4.chain_index() == 1 && p4.size() == 2 && p4.scale() == p3.scale() && p4.is_ntt_form());
        p4.store_to_host(out.data());
        REQUIRE(out == r2);
        std::stringstream bad("xx");
        REQUIRE(throws_invalid([&] { PhantomRelinKey k; k.load(bad); }));
    }

    // BGV: multiply -> relinearize -> mod_switch_to_next (examples/2_bgv.cu flow) with plain modulus 65537
    {
        EncryptionParameters bp(scheme_type::bgv);
        bp.set_poly_modulus_degree(n);
        bp.set_special_modulus_size(alpha);
        bp.set_coeff_modulus(parms.coeff_modulus());
        bp.set_plain_modulus(Modulus(65537));
        PhantomContext bctx(bp);
        orc_tool *bt = orc_tool_create(oc, size_q);
        REQUIRE(orc_tool_set_plain_modulus(bt, 65537) == 0);
        PhantomRelinKey brlk;
        {
            std::vector<uint64_t> flat;
            for (auto &k : rlk_host) flat.insert(flat.end(), k.begin(), k.end());
            brlk.load_from_host(bctx, flat.data(), dnum);
        }
        PhantomCiphertext b1, b2;
        b1.load_from_host(bctx, 1, 2, h1.data());
        b2.load_from_host(bctx, 1, 2, h2.data());
        b1.set_correction_factor(3);
        b2.set_correction_factor(5);
        PhantomCiphertext bprod = multiply_and_relin(bctx, b1, b2, brlk);
        REQUIRE(bprod.size() == 2 && bprod.correction_factor() == 15);
        std::vector<uint64_t> r2(ref3.begin(), ref3.begin() + 2 * ln), got(2 * ln);
        orc_keyswitch_inplace(bt, r2.data(), ref3.data() + 2 * ln, rlk_ptrs.data(), ORC_BGV);
        bprod.store_to_host(got.data());
        REQUIRE(got == r2);
        PhantomCiphertext bnext = mod_switch_to_next(bctx, bprod);
        REQUIRE(bnext.chain_index() == 2 && bnext.coeff_modulus_size() == size_q - 1 && bnext.is_ntt_form());
        std::vector<uint64_t> src = r2, ref((size_q - 1) * n * 2), out((size_q - 1) * n * 2);
        orc_mod_t_divide_q_last_ntt(bt, src.data(), 2, ref.data());
        bnext.store_to_host(out.data());
        REQUIRE(out == ref);
        const uint64_t inv = orc_invmod(q.back() % 65537, 65537);   // 65537 is prime
        REQUIRE(bnext.correction_factor() == 15 * inv % 65537);
        // balance_correction_factors (src/evaluate.cu:20-77): known answers from an independent restatement
        const uint64_t kat[][6] = {{3ULL, 5ULL, 65537ULL, 15ULL, 5ULL, 3ULL}, {1ULL, 65536ULL, 65537ULL, 65536ULL, 65536ULL, 1ULL},
                                   {12345ULL, 54321ULL, 65537ULL, 44864ULL, 285ULL, 65533ULL}, {7ULL, 7ULL, 786433ULL, 7ULL, 1ULL, 1ULL},
                                   {2ULL, 1032192ULL, 1032193ULL, 2ULL, 1ULL, 1032191ULL}, {40000ULL, 3ULL, 65537ULL, 429ULL, 136ULL, 143ULL}};
        for (auto &k : kat) {
            const auto f = detail::balance_correction_factors(k[0], k[1], k[2]);
            REQUIRE(std::get<0>(f) == k[3] && std::get<1>(f) == k[4] && std::get<2>(f) == k[5]);
        }
        // add / sub of ciphertexts with different correction factors (3 and 5 -> 15: scalars 5 and 3)
        {
            PhantomCiphertext a = b1, d = b1;
            add_inplace(bctx, a, b2);
            sub_inplace(bctx, d, b2);
            REQUIRE(a.correction_factor() == 15 && d.correction_factor() == 15);
            std::vector<uint64_t> s5(size_q, 5), s3(size_q, 3), x(2 * ln), y(2 * ln), ref(2 * ln), got(2 * ln);
            for (int p = 0; p < 2; p++) {
                orc_multiply_scalar_rns_poly(oc, h1.data() + p * ln, s5.data(), x.data() + p * ln, size_q, 0);
                orc_multiply_scalar_rns_poly(oc, h2.data() + p * ln, s3.data(), y.data() + p * ln, size_q, 0);
                orc_add_rns_poly(oc, x.data() + p * ln, y.data() + p * ln, ref.data() + p * ln, size_q, 0);
            }
            a.store_to_host(got.data());
            REQUIRE(got == ref);
            for (int p = 0; p < 2; p++) orc_sub_rns_poly(oc, x.data() + p * ln, y.data() + p * ln, ref.data() + p * ln, size_q, 0);
            d.store_to_host(got.data());
            REQUIRE(got == ref);
            std::vector<PhantomCiphertext> many{b1, b2};
            PhantomCiphertext total;
            add_many(bctx, many, total);      // BGV folds with add_inplace
            REQUIRE(total.correction_factor() == 15);
        }
        // add_plain / sub_plain / multiply_plain: lift t -> {q_i}, then c0 +- correction_factor * pt
        {
            std::vector<uint64_t> m(n);
            for (auto &v : m) v = g() % 65537;
            PhantomPlaintext pt;
            pt.load_from_host(m.data(), 1, n, 0);
            std::vector<uint64_t> lifted(ln), scaled(ln), cf(size_q, 3), ref = h1, got(2 * ln);
            orc_bgv_lift_plain(oc, size_q, m.data(), lifted.data());
            orc_multiply_scalar_rns_poly(oc, lifted.data(), cf.data(), scaled.data(), size_q, 0);
            auto s1 = add_plain(bctx, b1, pt);
            orc_add_rns_poly(oc, h1.data(), scaled.data(), ref.data(), size_q, 0);
            s1.store_to_host(got.data());
            REQUIRE(got == ref);
            auto s2 = sub_plain(bctx, b1, pt);
            orc_sub_rns_poly(oc, h1.data(), scaled.data(), ref.data(), size_q, 0);
            s2.store_to_host(got.data());
            REQUIRE(got == ref);
            auto s3 = multiply_plain(bctx, b1, pt);
            for (int p = 0; p < 2; p++) orc_multiply_rns_poly(oc, h1.data() + p * ln, lifted.data(), ref.data() + p * ln, size_q, 0);
            s3.store_to_host(got.data());
            REQUIRE(got == ref);
        }
        orc_tool_destroy(bt);
    }
    // BFV: multiply (BEHZ, evaluate.cu:447-548) -> relinearize, coefficient-form ciphertexts (examples/1_bfv.cu flow)
    {
        EncryptionParameters fp(scheme_type::bfv);
        fp.set_poly_modulus_degree(n);
        fp.set_special_modulus_size(alpha);
        fp.set_coeff_modulus(parms.coeff_modulus());
        fp.set_plain_modulus(Modulus(65537));
        fp.set_mul_tech(mul_tech_type::behz);
        PhantomContext fctx(fp);
        orc_behz *ob = orc_behz_create(oc, 65537);
        REQUIRE(ob != nullptr);
        PhantomRelinKey frlk;
        {
            std::vector<uint64_t> flat;
            for (auto &k : rlk_host) flat.insert(flat.end(), k.begin(), k.end());
            frlk.load_from_host(fctx, flat.data(), dnum);
        }
        PhantomCiphertext f1, f2;
        f1.load_from_host(fctx, 1, 2, h1.data());
        f2.load_from_host(fctx, 1, 2, h2.data());
        f1.set_ntt_form(false);
        f2.set_ntt_form(false);
        PhantomCiphertext fprod = multiply(fctx, f1, f2);
        REQUIRE(fprod.size() == 3 && !fprod.is_ntt_form());
        std::vector<uint64_t> r3(3 * ln), got3(3 * ln);
        orc_bfv_multiply_behz(ob, h1.data(), h2.data(), r3.data());
        fprod.store_to_host(got3.data());
        REQUIRE(got3 == r3);
        relinearize_inplace(fctx, fprod, frlk);
        std::vector<uint64_t> r2(r3.begin(), r3.begin() + 2 * ln), got2(2 * ln);
        orc_keyswitch_inplace(tool, r2.data(), r3.data() + 2 * ln, rlk_ptrs.data(), ORC_BFV);
        fprod.store_to_host(got2.data());
        REQUIRE(got2 == r2);
        PhantomCiphertext fsq = f1;
        multiply_inplace(fctx, fsq, fsq);                      // squaring path
        orc_bfv_multiply_behz(ob, h1.data(), h1.data(), r3.data());
        fsq.store_to_host(got3.data());
        REQUIRE(got3 == r3);
        REQUIRE(throws_invalid([&] { PhantomCiphertext c = f1; c.set_ntt_form(true); multiply_inplace(fctx, c, f2); }));
        // add_plain / sub_plain (scaling variant) and multiply_plain (centred lift through the NTT)
        {
            std::vector<uint64_t> m(n);
            for (auto &v : m) v = g() % 65537;
            PhantomPlaintext pt;
            pt.load_from_host(m.data(), 1, n, 0);
            std::vector<uint64_t> ref = h1, got(2 * ln);
            auto s1 = add_plain(fctx, f1, pt);
            orc_bfv_add_plain(oc, size_q, ref.data(), m.data(), 65537, 0);
            s1.store_to_host(got.data());
            REQUIRE(got == ref);
            auto s2 = sub_plain(fctx, f1, pt);
            ref = h1;
            orc_bfv_add_plain(oc, size_q, ref.data(), m.data(), 65537, 1);
            s2.store_to_host(got.data());
            REQUIRE(got == ref);
            auto s3 = multiply_plain(fctx, f1, pt);
            ref = h1;
            orc_bfv_multiply_plain(oc, size_q, ref.data(), 2, m.data(), 65537);
            s3.store_to_host(got.data());
            REQUIRE(got == ref && !s3.is_ntt_form());
            REQUIRE(throws_invalid([&] { PhantomCiphertext c = f1; c.set_ntt_form(true); add_plain_inplace(fctx, c, pt); }));
            // BFV mod_switch_to: divide-and-round chain
            PhantomCiphertext two = mod_switch_to(fctx, f1, 3);
            REQUIRE(two.chain_index() == 3 && two.coeff_modulus_size() == size_q - 2);
        }
        orc_behz_destroy(ob);
        // the reference's default mul_tech (hps) on a second context
        EncryptionParameters hp = fp;
        hp.set_mul_tech(mul_tech_type::hps);
        PhantomContext hctx(hp);
        orc_hps *oh = orc_hps_create(oc, 65537);
        REQUIRE(oh != nullptr);
        PhantomCiphertext g1, g2;
        g1.load_from_host(hctx, 1, 2, h1.data());
        g2.load_from_host(hctx, 1, 2, h2.data());
        g1.set_ntt_form(false);
        g2.set_ntt_form(false);
        PhantomCiphertext hprod = multiply(hctx, g1, g2);
        orc_bfv_multiply_hps(oh, h1.data(), h2.data(), r3.data());
        hprod.store_to_host(got3.data());
        REQUIRE(got3 == r3);
        orc_hps_destroy(oh);
        // mul_tech hps_overq on a third context
        EncryptionParameters qp_ = fp;
        qp_.set_mul_tech(mul_tech_type::hps_overq);
        PhantomContext qctx(qp_);
        orc_hpsq *oq = orc_hpsq_create(oc, 65537);
        REQUIRE(oq != nullptr);
        PhantomCiphertext k1, k2;
        k1.load_from_host(qctx, 1, 2, h1.data());
        k2.load_from_host(qctx, 1, 2, h2.data());
        k1.set_ntt_form(false);
        k2.set_ntt_form(false);
        PhantomCiphertext qprod = multiply(qctx, k1, k2);
        orc_bfv_multiply_hps_overq(oq, h1.data(), h2.data(), r3.data());
        qprod.store_to_host(got3.data());
        REQUIRE(got3 == r3);
        orc_hpsq_destroy(oq);
        // mul_tech hps_overq_leveled: levels from the ciphertext's depth (FindLevelsToDrop), multiply / relinearize /
        // multiply_and_relin with and without dropped levels
        EncryptionParameters lp = fp;
        lp.set_mul_tech(mul_tech_type::hps_overq_leveled);
        PhantomContext lctx(lp);
        {
            // known answers from an independent restatement of src/evaluate.cu:551-647 (n 4096, t 65537, |P| 2, 3 digits, 60-bit primes)
            const size_t mult[9] = {0, 0, 0, 1, 1, 2, 2, 3, 3}, ks[9] = {0, 0, 0, 1, 1, 2, 2, 2, 3};
            for (size_t d = 0; d < 9; d++) {
                REQUIRE(detail::find_levels_to_drop(lctx, d, 60.0, false, false) == mult[d]);
                REQUIRE(detail::find_levels_to_drop(lctx, d, 60.0, true, false) == ks[d]);
                REQUIRE(detail::find_levels_to_drop(lctx, d, 60.0, false, true) == ks[d]);
            }
            REQUIRE(detail::dcrt_bits(lctx) == 60.0);
            REQUIRE(throws_invalid([&] { (void)detail::find_levels_to_drop(fctx, 1, 60.0, false, false); }));
        }
        PhantomRelinKey lrlk;
        {
            std::vector<uint64_t> flat;
            for (auto &k : rlk_host) flat.insert(flat.end(), k.begin(), k.end());
            lrlk.load_from_host(lctx, flat.data(), dnum);
        }
        PhantomCiphertext l1, l2;
        l1.load_from_host(lctx, 1, 2, h1.data());
        l2.load_from_host(lctx, 1, 2, h2.data());
        l1.set_ntt_form(false);
        l2.set_ntt_form(false);
        {   // fresh ciphertexts: nothing dropped, the result is plain hps_overq
            orc_hpsq *top = orc_hpsq_create(oc, 65537);
            PhantomCiphertext lprod = multiply(lctx, l1, l2);
            orc_bfv_multiply_hps_overq(top, h1.data(), h2.data(), r3.data());
            lprod.store_to_host(got3.data());
            REQUIRE(got3 == r3 && lprod.GetNoiseScaleDeg() == 2);
            orc_hpsq_destroy(top);
        }
        {   // depth 3 (noise scale degree 4): one level dropped in the multiply, in the key switch and in the fused form
            orc_hpsq *lev = orc_hpsq_create_level(oc, 65537, size_q - 1);
            orc_tool *lt = orc_tool_create(oc, size_q - 1);
            PhantomCiphertext d1 = l1, d2 = l2;
            d1.SetNoiseScaleDeg(4);
            d2.SetNoiseScaleDeg(2);
            PhantomCiphertext lprod = multiply(lctx, d1, d2);
            orc_bfv_multiply_hps_overq(lev, h1.data(), h2.data(), r3.data());
            lprod.store_to_host(got3.data());
            REQUIRE(got3 == r3 && lprod.GetNoiseScaleDeg() == 5);
            relinearize_inplace(lctx, lprod, lrlk);           // degree 5 -> depth 4 -> one level
            std::vector<uint64_t> r2(r3.begin(), r3.begin() + 2 * ln), got2(2 * ln);
            orc_keyswitch_bfv_leveled(lt, lev, r2.data(), r3.data() + 2 * ln, rlk_ptrs.data());
            lprod.store_to_host(got2.data());
            REQUIRE(got2 == r2 && lprod.size() == 2);
            PhantomCiphertext fused = multiply_and_relin(lctx, d1, d2, lrlk);
            orc_bfv_mul_relin_hps_overq_leveled(lt, lev, h1.data(), h2.data(), rlk_ptrs.data(), r2.data());
            fused.store_to_host(got2.data());
            REQUIRE(got2 == r2 && fused.size() == 2 && fused.GetNoiseScaleDeg() == 5);
            // hoisted rotation with a dropped level: scale to Ql, hoist there, expand (evaluate.cu:1732-1862)
            PhantomGaloisKey lglk;
            {
                PhantomRelinKey k1;
                std::vector<uint64_t> flat;
                for (auto &k : glk_host) flat.insert(flat.end(), k.begin(), k.end());
                k1.load_from_host(lctx, flat.data(), dnum);
                lglk.add(elt1, std::move(k1));
            }
            PhantomCiphertext rot = d1;      // degree 4 -> depth 3 -> one level for a key switch too
            hoisting_inplace(lctx, rot, lglk, {1});
            const size_t lq = size_q - 1;
            std::vector<uint64_t> lowv(2 * lq * n), want(2 * ln), gotr(2 * ln);
            for (int p = 0; p < 2; p++) orc_hps_scale_q_ql(lev, h1.data() + p * ln, lowv.data() + p * lq * n);
            const uint64_t *const *gl[1] = {glk_ptrs.data()};
            orc_hoisting(lt, lowv.data(), &elt1, 1, gl, ORC_BFV);
            for (int p = 0; p < 2; p++) orc_hps_expand_ql_q(lev, lowv.data() + p * lq * n, want.data() + p * ln);
            rot.store_to_host(gotr.data());
            REQUIRE(gotr == want);
            orc_tool_destroy(lt);
            orc_hpsq_destroy(lev);
        }
    }
    phantom::util::check_hip(hipDeviceSynchronize(), "sync");
    orc_tool_destroy(tool);
    orc_ctx_destroy(oc);
    std::printf("HOST_API_OK\n");
    return 0;
}
