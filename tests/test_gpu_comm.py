"""pha_broadcast_keys (include/phantom_amd.h): the one-time RCCL broadcast of evaluation / Galois keys, called from
C / ctypes with a raw ncclComm_t.

* one rank through the REAL RCCL on the GPU box (the copy PyTorch already holds): ncclGetUniqueId + ncclCommInitRank through
  ctypes, then the library's entry point;
* two ranks on the one device of the box: RCCL refuses that, so the library is pointed (PHA_RCCL_LIB) at the host-staged stand-in
  tests/cpp/fake_rccl.cpp, and rank 1 must end up with rank 0's key words."""
import ctypes as C
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class NcclUniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


@pytest.mark.timeout(300)
def test_broadcast_keys_one_rank_through_rccl(gpu):
    import torch
    import phantom_fhe_amd as P
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
    uid = NcclUniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, NcclUniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        count = C.c_int()
        assert rccl.ncclCommCount(comm, C.byref(count)) == 0 and count.value == 1
        ctx = P.PhantomContext(12, [0xffffee001, 0xffffc4001, 0x1ffffe0001], 1, device=gpu)
        rng = np.random.default_rng(7)
        host = [rng.integers(0, 1 << 60, (2, 3, 4096), dtype=np.uint64) for _ in range(2)]
        keys = [P.to_device(h, gpu) for h in host]
        ctx.broadcast_keys(keys, 0, comm.value)
        torch.cuda.synchronize()
        for k, h in zip(keys, host):
            assert np.array_equal(P.to_host(k), h)
        with pytest.raises(ValueError):
            ctx.broadcast_keys(keys, 0, 0)            # null communicator
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)


_RANK = textwrap.dedent("""
    import ctypes as C, os, sys
    import numpy as np
    sys.path[:0] = [{root!r}, os.path.join({root!r}, "phantom-fhe_amd"), os.path.join({root!r}, "tests")]
    import torch
    import phantom_fhe_amd as P
    rank = int(sys.argv[1])
    class FakeComm(C.Structure):
        _fields_ = [("rank", C.c_int), ("nranks", C.c_int), ("seq", C.c_ulong)]
    comm = FakeComm(rank, 2, 0)
    dev = torch.device("cuda:0")
    ctx = P.PhantomContext(12, [0xffffee001, 0xffffc4001, 0x1ffffe0001], 1, device=dev)
    rng = np.random.default_rng(11)
    want = [rng.integers(0, 1 << 60, (2, 3, 4096), dtype=np.uint64) for _ in range(3)]
    keys = [P.to_device(w if rank == 0 else np.zeros_like(w), dev) for w in want]
    ctx.broadcast_keys(keys, 0, C.addressof(comm))
    ctx.broadcast_keys(keys[:1], 0, C.addressof(comm))     # a second broadcast on the same communicator
    torch.cuda.synchronize()
    assert all(np.array_equal(P.to_host(k), w) for k, w in zip(keys, want)), "rank %d: keys differ" % rank
    print("RANK_OK", rank)
""")


@pytest.mark.timeout(600)
def test_broadcast_keys_two_ranks_over_the_stand_in(gpu, tmp_path):
    import phantom_fhe_amd as P  # noqa: F401
    lib = tmp_path / "libfake_rccl.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-shared", "-fPIC", os.path.join(ROOT, "tests", "cpp", "fake_rccl.cpp"),
                           "-o", str(lib)])
    script = tmp_path / "rank.py"
    script.write_text(_RANK.format(root=ROOT))
    env = dict(os.environ, PHA_RCCL_LIB=str(lib), PHA_FAKE_RCCL_DIR=str(tmp_path))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in (1, 0)]                       # the receiver first: it must wait for the root
    outs = [p.communicate(timeout=500) for p in procs]
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 0 and "RANK_OK" in out, out + err
