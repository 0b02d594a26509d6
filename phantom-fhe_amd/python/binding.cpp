// binding.cpp -- pybind11 module `pyPhantom` over the C++ host mirror (host/phantom.h).
//
// Mirrors the names of the reference's python/src/binding.cu:8-166 for everything that is on the
// accelerated path: enums scheme_type / mul_tech_type, classes modulus, params, context, ciphertext,
// relin_key, galois_key, and the functions create_coeff_modulus, get_elt_from_step(s), negate, add, sub,
// add_plain, add_many, sub_plain, multiply, multiply_and_relin, multiply_plain, relinearize, rescale_to_next,
// mod_switch_to_next / mod_switch_to (ciphertext and plaintext overloads), apply_galois, rotate, hoisting; class plaintext.  Like the reference, every function returns by value.  Key generation, encryption,
// decryption and the encoders are out of scope, so ciphertexts and keys enter as
// numpy uint64 arrays (`ciphertext.load`, `relin_key.load`) and leave with `ciphertext.to_numpy()`.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <phantom.h>

#include <fstream>
#include <sstream>

namespace py = pybind11;
using namespace phantom;
using namespace phantom::arith;
using u64_array = py::array_t<uint64_t, py::array::c_style | py::array::forcecast>;

PYBIND11_MODULE(pyPhantom, m) {
    m.doc() = "MI355X-native PhantomFHE hot path (libphantom_amd.so) -- reference-compatible surface";

    py::enum_<scheme_type>(m, "scheme_type")
        .value("none", scheme_type::none).value("bfv", scheme_type::bfv)
        .value("ckks", scheme_type::ckks).value("bgv", scheme_type::bgv);
    py::enum_<mul_tech_type>(m, "mul_tech_type")
        .value("none", mul_tech_type::none).value("behz", mul_tech_type::behz).value("hps", mul_tech_type::hps)
        .value("hps_overq", mul_tech_type::hps_overq).value("hps_overq_leveled", mul_tech_type::hps_overq_leveled);

    py::class_<Modulus>(m, "modulus").def(py::init<uint64_t>()).def("value", &Modulus::value).def("bit_count", &Modulus::bit_count);
    py::enum_<sec_level_type>(m, "sec_level_type")
        .value("none", sec_level_type::none).value("tc128", sec_level_type::tc128)
        .value("tc192", sec_level_type::tc192).value("tc256", sec_level_type::tc256);
    m.def("create_coeff_modulus", &CoeffModulus::Create);
    m.def("create_plain_modulus", &PlainModulus::Batching);
    py::class_<util::cuda_stream_wrapper>(m, "cuda_stream").def(py::init<>());
    m.def("get_elt_from_step", &util::get_elt_from_step);
    m.def("get_elts_from_steps", &util::get_elts_from_steps);

    py::class_<EncryptionParameters>(m, "params")
        .def(py::init<scheme_type>())
        .def("set_mul_tech", &EncryptionParameters::set_mul_tech)
        .def("set_poly_modulus_degree", &EncryptionParameters::set_poly_modulus_degree)
        .def("set_special_modulus_size", &EncryptionParameters::set_special_modulus_size)
        .def("set_galois_elts", &EncryptionParameters::set_galois_elts)
        .def("set_coeff_modulus", &EncryptionParameters::set_coeff_modulus)
        .def("set_plain_modulus", &EncryptionParameters::set_plain_modulus);

    py::class_<PhantomContext>(m, "context")
        .def(py::init<const EncryptionParameters &>())
        .def("total_parm_size", &PhantomContext::total_parm_size)
        .def("get_first_index", &PhantomContext::get_first_index)
        .def("coeff_modulus_size", [](const PhantomContext &c, size_t chain_index) {
            return c.get_context_data(chain_index).parms().coeff_modulus().size();
        });

    py::class_<PhantomCiphertext>(m, "ciphertext")
        .def(py::init<>())
        .def("set_scale", &PhantomCiphertext::set_scale)
        .def("scale", &PhantomCiphertext::scale)
        .def("size", &PhantomCiphertext::size)
        .def("chain_index", &PhantomCiphertext::chain_index)
        .def("coeff_modulus_size", &PhantomCiphertext::coeff_modulus_size)
        .def("set_ntt_form", &PhantomCiphertext::set_ntt_form)
        .def("set_correction_factor", &PhantomCiphertext::set_correction_factor)
        .def("correction_factor", &PhantomCiphertext::correction_factor)
        .def("is_ntt_form", &PhantomCiphertext::is_ntt_form)
        .def("load", [](PhantomCiphertext &ct, const PhantomContext &c, size_t chain_index, u64_array data) {
            if (data.ndim() != 3) throw std::invalid_argument("expected [poly][limb][coeff]");
            ct.load_from_host(c, chain_index, static_cast<size_t>(data.shape(0)), data.data());
            (void)hipStreamSynchronize(cudaStreamPerThread);
        })
        .def("to_numpy", [](const PhantomCiphertext &ct) {
            u64_array out({ct.size(), ct.coeff_modulus_size(), ct.poly_modulus_degree()});
            ct.store_to_host(out.mutable_data());
            return out;
        })
        // the reference's on-disk format (include/ciphertext.h:173-214)
        .def("save", [](const PhantomCiphertext &ct, const std::string &path) {
            std::ofstream f(path, std::ios::binary);
            if (!f) throw std::runtime_error("cannot open " + path);
            ct.save(f);
        })
        .def("load_file", [](PhantomCiphertext &ct, const std::string &path) {
            std::ifstream f(path, std::ios::binary);
            if (!f) throw std::runtime_error("cannot open " + path);
            ct.load(f);
        });

    py::class_<PhantomPlaintext>(m, "plaintext")
        .def(py::init<>())
        .def("chain_index", [](const PhantomPlaintext &p) { return p.chain_index(); })
        .def("scale", [](const PhantomPlaintext &p) { return p.scale(); })
        .def("set_scale", &PhantomPlaintext::set_scale)
        .def("coeff_modulus_size", [](const PhantomPlaintext &p) { return p.coeff_modulus_size(); })
        // the encoders are out of scope: a BFV / BGV plaintext is [1][N] coefficients modulo t, a CKKS one [limb][N] in NTT form
        .def("load", [](PhantomPlaintext &p, u64_array data, size_t chain_index, double scale) {
            if (data.ndim() != 2) throw std::invalid_argument("expected [limb][coeff]");
            p.load_from_host(data.data(), static_cast<size_t>(data.shape(0)), static_cast<size_t>(data.shape(1)), chain_index, scale);
        }, py::arg("data"), py::arg("chain_index") = 0, py::arg("scale") = 1.0)
        .def("to_numpy", [](const PhantomPlaintext &p) {
            u64_array out({p.coeff_modulus_size(), p.poly_modulus_degree()});
            p.store_to_host(out.mutable_data());
            return out;
        });

    py::class_<PhantomRelinKey>(m, "relin_key")
        .def(py::init<>())
        .def("load", [](PhantomRelinKey &k, const PhantomContext &c, u64_array evk) {
            if (evk.ndim() != 4) throw std::invalid_argument("expected [dnum][2][QP][N]");
            k.load_from_host(c, evk.data(), static_cast<size_t>(evk.shape(0)));
        })
        // generate_one_kswitch_key (src/secretkey.cu:297-341) with caller-supplied randomness: sk [QP][N] and
        // new_key [Q][N] in NTT form, a [dnum][QP][N] uniform, e [dnum][QP][N] noise in coefficient form
        .def("generate", [](PhantomRelinKey &k, const PhantomContext &c, u64_array sk, u64_array new_key, u64_array a, u64_array e) {
            const auto &s = cudaStreamPerThread;
            auto up = [&](const u64_array &h) {
                auto d = util::make_cuda_auto_ptr<uint64_t>(static_cast<size_t>(h.size()), s);
                util::check_hip(hipMemcpyAsync(d.get(), h.data(), static_cast<size_t>(h.size()) * 8, hipMemcpyHostToDevice, s), "hipMemcpyAsync");
                return d;
            };
            auto d_sk = up(sk), d_nk = up(new_key), d_a = up(a), d_e = up(e);
            k.generate(c, d_sk.get(), d_nk.get(), d_a.get(), d_e.get(), s);
            util::check_hip(hipStreamSynchronize(s), "hipStreamSynchronize");
        })
        .def("to_numpy", [](const PhantomRelinKey &k) {
            const auto &pk0 = k.public_key(0);
            u64_array out({k.dnum(), size_t(2), pk0.coeff_modulus_size(), pk0.poly_modulus_degree()});
            const size_t words = 2 * pk0.coeff_modulus_size() * pk0.poly_modulus_degree();
            for (size_t d = 0; d < k.dnum(); d++) k.public_key(d).store_to_host(out.mutable_data() + d * words);
            return out;
        })
        // the reference's on-disk format (include/secretkey.h:129-163)
        .def("save", [](const PhantomRelinKey &k, const std::string &path) {
            std::ofstream f(path, std::ios::binary);
            if (!f) throw std::runtime_error("cannot open " + path);
            k.save(f);
        })
        .def("load_file", [](PhantomRelinKey &k, const std::string &path) {
            std::ifstream f(path, std::ios::binary);
            if (!f) throw std::runtime_error("cannot open " + path);
            k.load(f);
        });
    py::class_<PhantomGaloisKey>(m, "galois_key")
        .def(py::init<>())
        .def("load", [](PhantomGaloisKey &g, const PhantomContext &c, uint32_t galois_elt, u64_array evk) {
            PhantomRelinKey k;
            k.load_from_host(c, evk.data(), static_cast<size_t>(evk.shape(0)));
            g.add(galois_elt, std::move(k));
        });

    m.def("negate", &negate);
    m.def("add", &add);
    m.def("add_plain", &add_plain);
    m.def("sub_plain", &sub_plain);
    m.def("multiply_plain", &multiply_plain);
    m.def("add_many", [](const PhantomContext &c, const std::vector<PhantomCiphertext> &cts) {
        PhantomCiphertext dest;
        add_many(c, cts, dest);
        return dest;
    });
    m.def("mod_switch_to", py::overload_cast<const PhantomContext &, const PhantomCiphertext &, size_t>(&mod_switch_to));
    m.def("mod_switch_to", py::overload_cast<const PhantomContext &, const PhantomPlaintext &, size_t>(&mod_switch_to));
    m.def("mod_switch_to_next", py::overload_cast<const PhantomContext &, const PhantomPlaintext &>(&mod_switch_to_next));
    m.def("sub", &sub, py::arg(), py::arg(), py::arg(), py::arg("negate") = false);
    m.def("multiply", &multiply);
    m.def("multiply_and_relin", &multiply_and_relin);
    m.def("relinearize", &relinearize);
    m.def("rescale_to_next", &rescale_to_next);
    // extensions (no reference name): relinearize + rescale_to_next as one call, same ciphertext (pha_keyswitch_rescale)
    m.def("relinearize_rescale", &relinearize_rescale);
    m.def("multiply_relin_rescale", &multiply_relin_rescale);
    m.def("mod_switch_to_next", py::overload_cast<const PhantomContext &, const PhantomCiphertext &>(&mod_switch_to_next));
    m.def("apply_galois", &apply_galois);
    m.def("rotate", &rotate);
    m.def("hoisting", &hoisting);
}
