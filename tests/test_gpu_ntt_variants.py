"""Every NTT geometry that was built and measured but is not the product's plan (16 coefficients per thread, 512-thread
contiguous passes, on-the-fly twiddles everywhere, the one-workgroup N = 2^14 plan, both passes in one launch with the L2
hand-off, both block orders, integer-only butterflies ...) lives in the TEST-ONLY library libphantom_amd_exp.so (all sources
compiled with -DPHA_EXPERIMENTS).  This test re-runs the NTT parity file in a process that loads that library, where the
`ntt_variant` fixture sweeps all 17 variants against the oracle.  The product library is never touched by that process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_all_variants_in_the_experiments_library():
    import phantom_fhe_amd as P
    assert os.path.exists(P.EXP_LIB_PATH), "libphantom_amd_exp.so is not built (make -C phantom-fhe_amd/csrc)"
    env = dict(os.environ, PHA_LIB_OVERRIDE=P.EXP_LIB_PATH)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_ntt.py"), "-x", "-q", "-m", "gpu",
                        "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = "\n".join(r.stdout.splitlines()[-15:])
    assert r.returncode == 0, tail + "\n" + r.stderr[-2000:]
    assert " passed" in tail and "failed" not in tail, tail


def test_fused_conversion_and_strided_pass_in_the_experiments_library():
    """r05: modup_conv_s1_kernel (the mod-up's base conversion as the load of the forward transform's strided pass, N = 2^16).  The
    product takes it for launches of >= 1024 workgroups (batches of >= 6 ciphertexts at beta = 3); the experiments library takes it
    for EVERY N = 2^16 mod-up, so the key-switch parity tests of the C3 set -- per stage, whole key switch, key switch + rescale,
    batches, every level with a short last digit that those tests hold, and the alpha = 12 set (the 16-input instantiation, BGV) -- reach it
    with one or two ciphertexts."""
    import phantom_fhe_amd as P
    assert os.path.exists(P.EXP_LIB_PATH), "libphantom_amd_exp.so is not built (make -C phantom-fhe_amd/csrc)"
    env = dict(os.environ, PHA_LIB_OVERRIDE=P.EXP_LIB_PATH)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_rns.py"), "-x", "-q", "-m", "gpu",
                        "-k", "c3_ckks16 or hyb16_a12 or hoist or galois", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = "\n".join(r.stdout.splitlines()[-15:])
    assert r.returncode == 0, tail + "\n" + r.stderr[-2000:]
    assert " passed" in tail and "failed" not in tail, tail
