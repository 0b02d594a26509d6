"""C++ host mirror (phantom-fhe_amd/host/phantom.h): compiles everywhere (CPU check), and on the GPU box
runs tests/cpp/test_host_api.cpp, which drives multiply / relinearize / rescale / rotate through the
reference's evaluate.* names and compares with the oracle bit for bit."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_host_api.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "test_host_api")


def _build():
    import phantom_fhe_amd as P
    from oracle import oracle as O
    O.build()
    libdir = os.path.dirname(P.LIB_PATH)
    newest = max(os.path.getmtime(p) for p in (SRC, os.path.join(ROOT, "phantom-fhe_amd", "host", "phantom.h"),
                                                os.path.join(ROOT, "include", "phantom_amd.h")))
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < newest:
        subprocess.check_call([
            "/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"),
            "-I", os.path.join(ROOT, "phantom-fhe_amd", "host"), SRC, "-o", EXE,
            "-L", libdir, "-lphantom_amd", "-L", os.path.join(ROOT, "oracle"), "-loracle",
            f"-Wl,-rpath,{libdir}", f"-Wl,-rpath,{os.path.join(ROOT, 'oracle')}"])
    return EXE


def test_host_mirror_compiles():
    assert os.path.exists(_build())


@pytest.mark.gpu
def test_host_mirror_matches_oracle(gpu):
    exe = _build()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "HOST_API_OK" in out.stdout, out.stdout + out.stderr
