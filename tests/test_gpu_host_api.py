"""C++ host mirror (phantom-fhe_amd/host/phantom.h and the reference-named headers under include/phantom/): compiles
everywhere (CPU check), and on the GPU box runs
  * tests/cpp/test_host_api.cpp: multiply / relinearize / rescale / rotate through the reference's evaluate.* names,
  * tests/cpp/test_ref_spelling.cpp: a translation unit written against the reference's launcher-level spelling
    (#include "ntt.cuh" / "rns.cuh" / "evaluate.cuh"; nwt_2d_radix8_forward_inplace(p, ctx.gpu_rns_tables(), ...),
    rns_tool.modup(...), phantom::key_switch_inner_prod(...)),
both comparing with the oracle bit for bit."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_H = [os.path.join(ROOT, "phantom-fhe_amd", "host", "phantom.h"), os.path.join(ROOT, "phantom-fhe_amd", "host", "default_coeff_modulus.inc"),
          os.path.join(ROOT, "include", "phantom_amd.h")]


def _build(name, extra_includes=()):
    import phantom_fhe_amd as P
    from oracle import oracle as O
    O.build()
    src = os.path.join(ROOT, "tests", "cpp", name + ".cpp")
    exe = os.path.join(ROOT, "tests", "cpp", name)
    libdir = os.path.dirname(P.LIB_PATH)
    newest = max(os.path.getmtime(p) for p in [src] + HOST_H)
    if not os.path.exists(exe) or os.path.getmtime(exe) < newest:
        inc = []
        for d in (os.path.join(ROOT, "include"),) + tuple(extra_includes):
            inc += ["-I", d]
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17"] + inc + [src, "-o", exe,
                               "-L", libdir, "-lphantom_amd", "-L", os.path.join(ROOT, "oracle"), "-loracle",
                               f"-Wl,-rpath,{libdir}", f"-Wl,-rpath,{os.path.join(ROOT, 'oracle')}"])
    return exe


def _host_api():
    return _build("test_host_api", (os.path.join(ROOT, "phantom-fhe_amd", "host"),))


def _ref_spelling():   # -I include/phantom: the reference's own include names resolve
    return _build("test_ref_spelling", (os.path.join(ROOT, "include", "phantom"),))


def test_host_mirror_compiles():
    assert os.path.exists(_host_api()) and os.path.exists(_ref_spelling())


def test_reference_header_names_resolve_both_ways(tmp_path):
    """#include "evaluate.cuh" with -I include/phantom and #include <phantom/evaluate.cuh> with -I include."""
    names = ["phantom.h", "context.cuh", "ciphertext.h", "plaintext.h", "secretkey.h", "evaluate.cuh", "ntt.cuh", "rns.cuh",
             "rns_bconv.cuh", "polymath.cuh", "cuda_wrapper.cuh", "galois.cuh", "host/modulus.h", "host/encryptionparams.h"]
    a = tmp_path / "a.cpp"
    a.write_text("".join(f'#include "{n}"\n' for n in names) + "int main() { return sizeof(PhantomContext) && sizeof(phantom::DRNSTool) && sizeof(DNTTTable) ? 0 : 1; }\n")
    b = tmp_path / "b.cpp"
    b.write_text("".join(f"#include <phantom/{n}>\n" for n in names) + "int main() { return 0; }\n")
    for src, inc in ((a, os.path.join(ROOT, "include", "phantom")), (b, os.path.join(ROOT, "include"))):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), "-I", inc, str(src)])


def test_bfv_default_tables_equal_the_reference_literals():
    """CoeffModulus::BFVDefault / MaxBitCount of the host mirror (src/host/modulus.cu:57-80) against the fixture extracted
    from the reference's tables (tests/golden/make_default_moduli.py); host code only, no GPU."""
    out = subprocess.run([_ref_spelling(), "defaults"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stdout + out.stderr
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "default_coeff_modulus.json")))
    rows = [l.split() for l in out.stdout.splitlines()]
    assert len(rows) == 21
    for lv, deg, bits, *primes in rows:
        assert int(bits) == want["max_bit_count"][lv][deg]
        assert [int(p, 16) for p in primes] == [int(p, 16) for p in want["coeff_modulus"][lv][deg]]


@pytest.mark.gpu
def test_host_mirror_matches_oracle(gpu):
    out = subprocess.run([_host_api()], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "HOST_API_OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_reference_spelling_matches_oracle(gpu):
    out = subprocess.run([_ref_spelling()], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "REF_SPELLING_OK" in out.stdout, out.stdout + out.stderr
