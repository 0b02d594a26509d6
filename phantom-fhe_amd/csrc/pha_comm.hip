// pha_comm.hip -- one-time RCCL broadcast of evaluation / Galois keys over xGMI for C / C++ callers (keys are
// generated on one GPU and replicated; there is NO collective on the data path).  The reference has no multi-GPU code at all; a
// sharded job built on its C++ API (include/secretkey.h:102-220: PhantomRelinKey / PhantomGaloisKey own [dnum] buffers of
// [2][#QP][N] words) needs exactly this call between key generation and the first key switch.
//
// RCCL is resolved at run time (dlopen / dlsym), not at link time: a process that already holds a RCCL (PyTorch ships its own
// copy) keeps using THAT copy, and single-GPU users of libphantom_amd.so do not need librccl at all.  PHA_RCCL_LIB names another
// library with the same ncclBroadcast / ncclGroupStart / ncclGroupEnd / ncclGetErrorString entry points (tests: a host-staged stand-in
// for two ranks on one device, which RCCL itself refuses).
#include "../../include/phantom_amd.h"
#include "pha_internal.h"

#include <dlfcn.h>

#include <cstdlib>

namespace {

typedef int (*bcast_fn)(const void *, void *, size_t, int, int, void *, hipStream_t);   // ncclBroadcast
typedef int (*void_fn)();                                                                // ncclGroupStart / ncclGroupEnd
typedef const char *(*errstr_fn)(int);

struct Rccl {
    bcast_fn broadcast = nullptr;
    void_fn group_start = nullptr, group_end = nullptr;
    errstr_fn error_string = nullptr;
    std::string origin;
};

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {std::getenv("PHA_RCCL_LIB"), "librccl.so", "librccl.so.1"};
        void *h = nullptr;
        // two passes: first ANY copy this process already holds under one of the names (PyTorch's bundled librccl.so has the
        // soname librccl.so.1 and lives outside the default search path) -- the caller's ncclComm_t was created by that copy and
        // must go back into it; only when none is resident is a library loaded.  PHA_RCCL_LIB, when set, wins in both passes.
        for (int pass = 0; pass < 2 && !h; pass++)
            for (const char *name : names) {
                if (!name || !*name) continue;
                h = dlopen(name, pass == 0 ? (RTLD_NOW | RTLD_NOLOAD) : (RTLD_NOW | RTLD_GLOBAL));
                if (h) {
                    r.origin = name;
                    break;
                }
                if (pass == 0 && name == names[0]) {   // an explicit override is loaded rather than passed over for a resident RCCL
                    h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                    if (h) {
                        r.origin = name;
                        break;
                    }
                }
            }
        if (!h) return;
        r.broadcast = reinterpret_cast<bcast_fn>(dlsym(h, "ncclBroadcast"));
        r.group_start = reinterpret_cast<void_fn>(dlsym(h, "ncclGroupStart"));
        r.group_end = reinterpret_cast<void_fn>(dlsym(h, "ncclGroupEnd"));
        r.error_string = reinterpret_cast<errstr_fn>(dlsym(h, "ncclGetErrorString"));
    });
    return r;
}

void check_nccl(Rccl &r, int code, const char *what) {
    if (code == 0) return;
    throw std::runtime_error(std::string("RCCL error in ") + what + ": " + (r.error_string ? r.error_string(code) : "code " + std::to_string(code)));
}

}  // namespace

extern "C" {

int pha_broadcast_keys(pha_context_t ctx, uint64_t *const *keys, size_t n_keys, size_t words_per_key, int root, void *nccl_comm,
                       void *stream) {
    PHA_CTX_BEGIN(ctx)
    if (!keys && n_keys) throw std::invalid_argument("null key table");
    if (!nccl_comm) throw std::invalid_argument("null communicator");
    if (root < 0) throw std::invalid_argument("root out of range");
    Rccl &r = rccl();
    if (!r.broadcast || !r.group_start || !r.group_end)
        throw std::runtime_error("RCCL is not available (librccl.so not found; set PHA_RCCL_LIB)");
    // all keys of the set as ONE group: RCCL fuses them into as few xGMI transfers as it likes (180 MiB per key at C3)
    check_nccl(r, r.group_start(), "ncclGroupStart");
    int rc = 0;
    for (size_t i = 0; i < n_keys && rc == 0; i++) {
        if (!keys[i]) {
            (void)r.group_end();
            throw std::invalid_argument("null key buffer");
        }
        rc = r.broadcast(keys[i], keys[i], words_per_key, /* ncclUint64 */ 5, root, nccl_comm, pha::as_stream(stream));
    }
    const int rc_end = r.group_end();
    check_nccl(r, rc, "ncclBroadcast");
    check_nccl(r, rc_end, "ncclGroupEnd");
    PHA_API_END
}

}  // extern "C"
