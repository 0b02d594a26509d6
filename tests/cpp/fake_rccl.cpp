// fake_rccl.cpp -- TEST-ONLY stand-in for librccl (PHA_RCCL_LIB): RCCL refuses two ranks on one device, and the GPU box has
// one GPU, so the two-rank path of pha_broadcast_keys is exercised with this host-staged broadcast: the root copies the buffer to
// a file under PHA_FAKE_RCCL_DIR and publishes it by rename; the other ranks poll for the file and copy it to their device buffer.
// The "communicator" is a pointer to {rank, nranks}; sequence numbers keep successive broadcasts apart.  Not part of the product.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

struct FakeComm {
    int rank, nranks;
    unsigned long seq;
};

extern "C" {

int ncclGroupStart() { return 0; }
int ncclGroupEnd() { return 0; }
const char *ncclGetErrorString(int code) { return code == 0 ? "no error" : "fake rccl error"; }

int ncclBroadcast(const void *send, void *recv, size_t count, int datatype, int root, void *comm_, hipStream_t stream) {
    if (datatype != 5) return 4;   // only ncclUint64
    FakeComm *comm = static_cast<FakeComm *>(comm_);
    const char *dir = std::getenv("PHA_FAKE_RCCL_DIR");
    if (!dir || !comm || root >= comm->nranks) return 4;
    const std::string path = std::string(dir) + "/bcast_" + std::to_string(comm->seq++);
    const size_t bytes = count * 8;
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    std::vector<char> host(bytes);
    if (comm->rank == root) {
        if (hipMemcpy(host.data(), send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        const std::string tmp = path + ".tmp";
        FILE *f = std::fopen(tmp.c_str(), "wb");
        if (!f || std::fwrite(host.data(), 1, bytes, f) != bytes) return 2;
        std::fclose(f);
        if (std::rename(tmp.c_str(), path.c_str()) != 0) return 2;
        if (recv != send && hipMemcpy(recv, send, bytes, hipMemcpyDeviceToDevice) != hipSuccess) return 1;
        return 0;
    }
    for (int spin = 0; spin < 60000; spin++) {   // up to a minute
        FILE *f = std::fopen(path.c_str(), "rb");
        if (f) {
            const size_t got = std::fread(host.data(), 1, bytes, f);
            std::fclose(f);
            if (got != bytes) return 2;
            return hipMemcpy(recv, host.data(), bytes, hipMemcpyHostToDevice) == hipSuccess ? 0 : 1;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    return 6;
}

}  // extern "C"
