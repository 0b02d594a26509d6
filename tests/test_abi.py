"""The C-ABI library loads and exports exactly what include/phantom_amd.h declares (no compute, no GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(headers=("phantom_amd.h", "phantom_amd_bench.h")):
    """Every pha_* function declared by include/*.h: the drop-in boundary and the bench-only hooks."""
    names = set()
    for h in headers:
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(pha_[a-zA-Z0-9_]+)\s*\(", text))
    return sorted(names)


def test_every_public_header_is_covered():
    assert sorted(f for f in os.listdir(os.path.join(ROOT, "include")) if f.endswith(".h")) == ["phantom_amd.h", "phantom_amd_bench.h"]


def test_product_library_has_no_tuning_knob_and_experiments_library_has():
    import phantom_fhe_amd as P
    if not os.path.exists(P.LIB_PATH) or not os.path.exists(P.EXP_LIB_PATH):
        import __graft_entry__ as g
        g.build()
    assert not hasattr(ctypes.CDLL(P.LIB_PATH), "pha_set_tuning")
    exp = ctypes.CDLL(P.EXP_LIB_PATH)
    assert hasattr(exp, "pha_set_tuning")
    for name in _declared():
        assert hasattr(exp, name), name
    assert "pha_repeat_forward_ntt_batched" not in _declared(("phantom_amd.h",))   # bench hooks are not in the boundary header


def test_library_exports_every_declared_symbol():
    import phantom_fhe_amd as P
    if not os.path.exists(P.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(P.LIB_PATH)
    names = _declared()
    assert len(names) >= 35
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/phantom_amd.h but not exported"
    assert sorted(P.EXPORTED) == names     # the Python binding covers the whole header


def test_host_only_entry_points():
    import phantom_fhe_amd as P
    from oracle import oracle as O
    bits = [60] + [50] * 44 + [60] * 15           # examples/3_ckks.cu:729-739
    assert np.array_equal(P.coeff_modulus_create(1 << 16, bits), O.coeff_modulus_create(1 << 16, bits))
    with pytest.raises(ValueError):
        P.coeff_modulus_create(1000, [30])          # not a power of two
    with pytest.raises(ArithmeticError):
        P.coeff_modulus_create(1 << 16, [18] * 3)   # not enough primes -> logic_error, as the reference


def test_no_cpu_fallback():
    """The product must fail loudly without a HIP device; nothing routes through the oracle."""
    import torch
    import phantom_fhe_amd as P
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        P.PhantomContext(12, [0xffffee001], 0)
    pkg = os.path.join(ROOT, "phantom-fhe_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src
