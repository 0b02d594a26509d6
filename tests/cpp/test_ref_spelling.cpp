// test_ref_spelling.cpp -- a translation unit written against the REFERENCE's spelling of the launcher level
// (include/ntt.cuh:178-226, include/rns.cuh:156-205, include/evaluate.cuh:25-32, include/host/modulus.h:275): it includes
// the reference's header names from include/phantom/, drives NTT / mod-up / inner product / mod-down / rescale through
// DNTTTable and DRNSTool exactly as src/eval_key_switch.cu:95-182 and src/evaluate.cu:1376-1427 do, and compares every
// stage bit for bit with the CPU oracle.  Needs a GPU; built and run by tests/test_gpu_host_api.py.
#include "context.cuh"
#include "ciphertext.h"
#include "secretkey.h"
#include "evaluate.cuh"
#include "ntt.cuh"
#include "rns.cuh"
#include "rns_bconv.cuh"
#include "polymath.cuh"
#include "host/modulus.h"

#include <cstdio>
#include <random>
#include <string>

extern "C" {
#include "../../oracle/oracle.h"
}

using namespace phantom;
using namespace phantom::arith;
using namespace phantom::util;

#define REQUIRE(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

static std::vector<uint64_t> uniform(std::mt19937_64 &g, const std::vector<uint64_t> &primes, size_t n) {
    std::vector<uint64_t> v(primes.size() * n);
    for (size_t i = 0; i < primes.size(); i++)
        for (size_t k = 0; k < n; k++) v[i * n + k] = g() % primes[i];
    return v;
}
static cuda_auto_ptr<uint64_t> upload(const std::vector<uint64_t> &h, const cudaStream_t &s) {
    auto d = make_cuda_auto_ptr<uint64_t>(h.size(), s);
    hipMemcpyAsync(d.get(), h.data(), h.size() * 8, hipMemcpyHostToDevice, s);
    return d;
}
static std::vector<uint64_t> download(const uint64_t *d, size_t count, const cudaStream_t &s) {
    std::vector<uint64_t> h(count);
    hipMemcpyAsync(h.data(), d, count * 8, hipMemcpyDeviceToHost, s);
    hipStreamSynchronize(s);
    return h;
}

int main(int argc, char **argv) {
    if (argc > 1 && std::string(argv[1]) == "defaults") {   // host only: dump CoeffModulus::BFVDefault / MaxBitCount (no GPU needed)
        for (auto lv : {sec_level_type::tc128, sec_level_type::tc192, sec_level_type::tc256})
            for (size_t deg = 1024; deg <= 65536; deg *= 2) {
                std::printf("%d %zu %d", static_cast<int>(lv), deg, CoeffModulus::MaxBitCount(deg, lv));
                for (auto &m : CoeffModulus::BFVDefault(deg, lv)) std::printf(" 0x%llx", (unsigned long long)m.value());
                std::printf("\n");
            }
        return 0;
    }
    // CoeffModulus::BFVDefault: the reference's literal tables (src/host/globals.cu:71 and neighbours)
    auto d4096 = CoeffModulus::BFVDefault(4096);
    REQUIRE(d4096.size() == 3 && d4096[0].value() == 0xffffee001ULL && d4096[1].value() == 0xffffc4001ULL && d4096[2].value() == 0x1ffffe0001ULL);
    REQUIRE(CoeffModulus::BFVDefault(8192, sec_level_type::tc192).size() == 4 && CoeffModulus::MaxBitCount(16384) == 438);
    bool threw = false;
    try { (void)CoeffModulus::BFVDefault(3000); } catch (const std::invalid_argument &) { threw = true; }
    REQUIRE(threw);

    const size_t n = 8192, alpha = 3;
    const int log_n = 13;
    EncryptionParameters parms(scheme_type::ckks);
    parms.set_poly_modulus_degree(n);
    parms.set_special_modulus_size(alpha);
    parms.set_coeff_modulus(CoeffModulus::Create(n, {60, 50, 50, 50, 50, 50, 50, 50, 50, 60, 60, 60}));
    PhantomContext context(parms);
    const auto &s = cudaStreamPerThread;
    const size_t size_QP = 12, size_Q = 9, size_Ql = 9, size_QlP = size_Ql + alpha;
    std::vector<uint64_t> qp;
    for (auto &m : parms.coeff_modulus()) qp.push_back(m.value());
    const std::vector<uint64_t> q(qp.begin(), qp.begin() + size_Q);
    REQUIRE(context.gpu_rns_tables().n() == n && context.gpu_rns_tables().size() == size_QP);
    REQUIRE(context.gpu_rns_tables().modulus(2).value() == qp[2]);

    orc_ctx *oc = orc_ctx_create(log_n, qp.data(), size_QP, alpha);
    orc_tool *tool = orc_tool_create(oc, size_Ql);
    std::mt19937_64 g(0x5EED0000 + 77);

    // 1. the launchers of ntt.cuh on a [Q][N] polynomial, spelled as in the reference
    auto h = uniform(g, q, n);
    auto d = upload(h, s);
    nwt_2d_radix8_forward_inplace(d.get(), context.gpu_rns_tables(), size_Q, 0, s);
    auto want = h;
    orc_nwt_forward(oc, want.data(), size_Q, 0);
    REQUIRE(download(d.get(), h.size(), s) == want);
    nwt_2d_radix8_backward_inplace(d.get(), context.gpu_rns_tables(), size_Q, 0, s);
    REQUIRE(download(d.get(), h.size(), s) == h);

    // 2. key switch by hand, as src/eval_key_switch.cu:95-182 spells it
    auto &rns_tool = context.get_context_data(1).gpu_rns_tool();
    const size_t beta = rns_tool.v_base_part_Ql_to_compl_part_QlP_conv_size();
    REQUIRE(beta == 3);
    std::vector<std::vector<uint64_t>> evk_host;
    std::vector<uint64_t> flat;
    for (size_t dgt = 0; dgt < size_Q / alpha; dgt++) {
        std::vector<uint64_t> k;
        for (int half = 0; half < 2; half++) { auto part = uniform(g, qp, n); k.insert(k.end(), part.begin(), part.end()); }
        flat.insert(flat.end(), k.begin(), k.end());
        evk_host.push_back(std::move(k));
    }
    PhantomRelinKey relin_keys;
    relin_keys.load_from_host(context, flat.data(), size_Q / alpha);
    std::vector<const uint64_t *> evk_ptrs;
    for (auto &k : evk_host) evk_ptrs.push_back(k.data());

    auto c2_host = uniform(g, q, n);
    auto c2 = upload(c2_host, s);
    auto t_mod_up = make_cuda_auto_ptr<uint64_t>(beta * size_QlP * n, s);
    rns_tool.modup(t_mod_up.get(), c2.get(), context.gpu_rns_tables(), scheme_type::ckks, s);
    std::vector<uint64_t> o_mod_up(beta * size_QlP * n);
    orc_modup(tool, o_mod_up.data(), c2_host.data(), ORC_CKKS);
    REQUIRE(download(t_mod_up.get(), o_mod_up.size(), s) == o_mod_up);

    auto cx = make_cuda_auto_ptr<uint64_t>(2 * size_QlP * n, s);
    const DModulus *modulus_QP = nullptr;   // the library keeps its own device table
    key_switch_inner_prod(cx.get(), t_mod_up.get(), relin_keys.public_keys_ptr(), rns_tool, modulus_QP, 1 << 8, s);
    std::vector<uint64_t> o_cx(2 * size_QlP * n);
    orc_key_switch_inner_prod(tool, o_cx.data(), o_mod_up.data(), evk_ptrs.data());
    REQUIRE(download(cx.get(), o_cx.size(), s) == o_cx);

    auto ct = make_cuda_auto_ptr<uint64_t>(2 * size_Ql * n, s);
    std::vector<uint64_t> o_ct(2 * size_Ql * n);
    for (size_t i = 0; i < 2; i++) {
        rns_tool.moddown_from_NTT(ct.get() + i * size_Ql * n, cx.get() + i * size_QlP * n, context.gpu_rns_tables(), scheme_type::ckks, s);
        orc_moddown_from_ntt(tool, o_ct.data() + i * size_Ql * n, o_cx.data() + i * size_QlP * n, ORC_CKKS);
    }
    REQUIRE(download(ct.get(), o_ct.size(), s) == o_ct);

    // 3. rescale as src/evaluate.cu:1376-1427 spells it
    auto dst = make_cuda_auto_ptr<uint64_t>(2 * (size_Ql - 1) * n, s);
    rns_tool.divide_and_round_q_last_ntt(ct.get(), 2, context.gpu_rns_tables(), dst.get(), s);
    std::vector<uint64_t> o_dst(2 * (size_Ql - 1) * n);
    orc_rescale_ntt(tool, o_ct.data(), 2, o_dst.data());
    REQUIRE(download(dst.get(), o_dst.size(), s) == o_dst);

    // 4. a residue-wise kernel through its polymath.cuh name
    auto a = uniform(g, q, n), b = uniform(g, q, n);
    auto da = upload(a, s), db = upload(b, s);
    multiply_rns_poly(context.gpu_rns_tables(), da.get(), db.get(), da.get(), size_Q, 0, s);
    std::vector<uint64_t> o_prod(a.size());
    orc_multiply_rns_poly(oc, a.data(), b.data(), o_prod.data(), size_Q, 0);
    REQUIRE(download(da.get(), a.size(), s) == o_prod);

    // 5. r06: the launchers of rns.cuh / rns_bconv.cuh that had no entry of their own before -- DRNSTool::moddown with a BFV
    //    input in coefficient form (include/rns.cuh:159-160, caller shape of src/evaluate.cu:1005-1012), bConv_BEHZ_var1
    //    (include/rns_bconv.cuh:64) and exact_convert_array (:68), spelled as the reference spells them
    {
        std::vector<uint64_t> qlp(qp.begin(), qp.begin() + size_Ql);
        qlp.insert(qlp.end(), qp.begin() + size_Q, qp.end());
        auto cx_h = uniform(g, qlp, n);
        auto cx_d = upload(cx_h, s);
        auto ct_d = make_cuda_auto_ptr<uint64_t>(size_Ql * n, s);
        rns_tool.moddown(ct_d.get(), cx_d.get(), context.gpu_rns_tables(), scheme_type::bfv, s);
        std::vector<uint64_t> o_down(size_Ql * n);
        orc_moddown(tool, o_down.data(), cx_h.data(), ORC_BFV);
        REQUIRE(download(ct_d.get(), o_down.size(), s) == o_down);

        const std::vector<uint32_t> ibase{0, 1, 2, 3}, obase{9, 10, 11};
        std::vector<uint64_t> ip, op;
        for (auto r : ibase) ip.push_back(qp[r]);
        for (auto r : obase) op.push_back(qp[r]);
        DBaseConverter conv(context.amd(), ibase, obase);
        auto src_h = uniform(g, ip, n);
        auto src_d = upload(src_h, s);
        auto dst_d = make_cuda_auto_ptr<uint64_t>(op.size() * n, s);
        conv.bConv_BEHZ_var1(dst_d.get(), src_d.get(), n, s);
        std::vector<uint64_t> o_var1(op.size() * n);
        orc_bconv_behz_var1(ip.data(), ip.size(), op.data(), op.size(), src_h.data(), o_var1.data(), n);
        REQUIRE(download(dst_d.get(), o_var1.size(), s) == o_var1);

        const uint64_t t = 1032193;
        DBaseConverter to_t(context.amd(), ibase, t);            // base_q_to_t_conv_ (src/rns.cu:283-284)
        auto one_d = make_cuda_auto_ptr<uint64_t>(n, s);
        to_t.exact_convert_array(one_d.get(), src_d.get(), n, s);
        std::vector<uint64_t> o_exact(n);
        orc_exact_convert_array(ip.data(), ip.size(), t, src_h.data(), o_exact.data(), n);
        REQUIRE(download(one_d.get(), n, s) == o_exact);
        threw = false;
        try { conv.exact_convert_array(one_d.get(), src_d.get(), n, s); } catch (const std::invalid_argument &) { threw = true; }
        REQUIRE(threw);                                          // "out base in exact_convert_array must be one." (rns_bconv.cu:423-425)
    }

    orc_tool_destroy(tool);
    orc_ctx_destroy(oc);
    std::printf("REF_SPELLING_OK\n");
    return 0;
}
