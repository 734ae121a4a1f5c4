// dpe_comm.hip -- the exchange step of the multi-GPU path behind the C-ABI (one process per GPU): all-reduce(MAX) of the
// packed arg-max keys and all-gather of the stage-1 banks, so that a C++ host (the dsp::Flow mirror, or the reference's own
// flow) shards the manifold grid without Python.
//
// Two transports:
//   RCCL      ncclAllReduce / ncclAllGather over xGMI.  librccl is opened with dlopen at the first use (a process that
//             already carries an RCCL -- PyTorch ships its own copy under the same SONAME -- keeps using that one), the
//             unique id travels through a file: rank 0 writes <rendezvous>/nccl_id, the others wait for it.
//   host files  every rank copies its buffer to the host, publishes it as a file under <rendezvous>, reads the others' and
//             reduces on the host.  For functional tests of the sharded path with several ranks on ONE GPU (RCCL refuses
//             two ranks per device); never a performance path.
// The reference has no counterpart (single GPU); SURVEY.md 8(e) defines the exchange.
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and enums only: the library itself is bound at run time
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <string>
#include <thread>

#include "dpe_common.h"

namespace {

struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi g_rccl;

int load_rccl()
{
    if (g_rccl.lib) return 0;
    void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) {
        dpe::set_error("[dpe_comm] cannot load librccl: %s", dlerror());
        return -1;
    }
#define DPE_SYM(field, name)                                                     \
    *(void **)(&g_rccl.field) = dlsym(lib, name);                               \
    if (!g_rccl.field) {                                                        \
        dpe::set_error("[dpe_comm] librccl lacks %s", name);                    \
        return -1;                                                              \
    }
    DPE_SYM(GetUniqueId, "ncclGetUniqueId")
    DPE_SYM(CommInitRank, "ncclCommInitRank")
    DPE_SYM(CommDestroy, "ncclCommDestroy")
    DPE_SYM(AllReduce, "ncclAllReduce")
    DPE_SYM(AllGather, "ncclAllGather")
    DPE_SYM(GetErrorString, "ncclGetErrorString")
#undef DPE_SYM
    g_rccl.lib = lib;
    return 0;
}

bool wait_for_file(const std::string &path, size_t bytes, double timeoutS)
{
    const auto t0 = std::chrono::steady_clock::now();
    struct stat st;
    while (true) {
        if (stat(path.c_str(), &st) == 0 && (size_t)st.st_size >= bytes) return true;
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeoutS) return false;
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
}

int write_atomic(const std::string &path, const void *data, size_t bytes)
{
    const std::string tmp = path + ".tmp";
    FILE *f = std::fopen(tmp.c_str(), "wb");
    if (!f) return -1;
    const size_t n = std::fwrite(data, 1, bytes, f);
    std::fclose(f);
    if (n != bytes) return -1;
    return std::rename(tmp.c_str(), path.c_str());
}

int read_file(const std::string &path, void *data, size_t bytes)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return -1;
    const size_t n = std::fread(data, 1, bytes, f);
    std::fclose(f);
    return n == bytes ? 0 : -1;
}

}  // namespace

struct dpe_comm {
    int rank = 0, nRanks = 1, backend = DPE_COMM_RCCL;
    ncclComm_t nccl = nullptr;
    bool ownsNccl = false;
    std::string dir;
    unsigned long long seq = 0;
    std::vector<unsigned char> host;   // staging of the host-file transport
};

// host-file transport: publish this rank's buffer as <dir>/<tag>.<seq>.<rank>, collect everybody's
static int hostfile_exchange(dpe_comm *c, const char *tag, const void *mine, size_t bytes, std::vector<unsigned char> &all)
{
    const unsigned long long seq = c->seq++;
    auto name = [&](int r, unsigned long long s) { return c->dir + "/" + tag + "." + std::to_string(s) + "." + std::to_string(r); };
    DPE_REQUIRE(write_atomic(name(c->rank, seq), mine, bytes) == 0, "[dpe_comm] cannot write %s", name(c->rank, seq).c_str());
    all.resize(bytes * c->nRanks);
    for (int r = 0; r < c->nRanks; ++r) {
        DPE_REQUIRE(wait_for_file(name(r, seq), bytes, 120.0), "[dpe_comm] rank %d: no data from rank %d (%s) after 120 s", c->rank, r,
                    name(r, seq).c_str());
        DPE_REQUIRE(read_file(name(r, seq), all.data() + bytes * r, bytes) == 0, "[dpe_comm] cannot read %s", name(r, seq).c_str());
    }
    if (seq >= 2) (void)std::remove(name(c->rank, seq - 2).c_str());   // everybody has passed exchange seq-2 by now (it needed seq-1 of all)
    return 0;
}

extern "C" {

int dpe_comm_create(int32_t rank, int32_t nRanks, const char *rendezvousPath, int32_t backend, dpe_comm **out)
{
    DPE_REQUIRE(out && nRanks >= 1 && rank >= 0 && rank < nRanks, "[dpe_comm] create: bad rank %d of %d", rank, nRanks);
    DPE_REQUIRE(backend == DPE_COMM_RCCL || backend == DPE_COMM_HOSTFILES, "[dpe_comm] create: unknown backend %d", backend);
    DPE_REQUIRE(nRanks == 1 || (rendezvousPath && *rendezvousPath), "[dpe_comm] create: rendezvous directory missing");
    dpe_comm *c = new dpe_comm();
    c->rank = rank; c->nRanks = nRanks; c->backend = backend;
    c->dir = rendezvousPath ? rendezvousPath : "";
    if (backend == DPE_COMM_RCCL) {
        if (load_rccl()) { delete c; return -1; }
        ncclUniqueId id;
        const std::string idFile = c->dir + "/nccl_id";
        if (rank == 0) {
            const ncclResult_t r = g_rccl.GetUniqueId(&id);
            if (r != ncclSuccess) { dpe::set_error("[dpe_comm] ncclGetUniqueId: %s", g_rccl.GetErrorString(r)); delete c; return -1; }
            if (nRanks > 1 && write_atomic(idFile, &id, sizeof(id)) != 0) { dpe::set_error("[dpe_comm] cannot write %s", idFile.c_str()); delete c; return -1; }
        } else {
            if (!wait_for_file(idFile, sizeof(id), 120.0) || read_file(idFile, &id, sizeof(id)) != 0) {
                dpe::set_error("[dpe_comm] rank %d: no unique id at %s after 120 s", rank, idFile.c_str());
                delete c;
                return -1;
            }
        }
        const ncclResult_t r = g_rccl.CommInitRank(&c->nccl, nRanks, id, rank);
        if (r != ncclSuccess) { dpe::set_error("[dpe_comm] ncclCommInitRank: %s", g_rccl.GetErrorString(r)); delete c; return -1; }
        c->ownsNccl = true;
    }
    *out = c;
    return 0;
}

int dpe_comm_wrap_nccl(void *ncclComm, int32_t rank, int32_t nRanks, dpe_comm **out)
{
    DPE_REQUIRE(ncclComm && out && nRanks >= 1 && rank >= 0 && rank < nRanks, "[dpe_comm] wrap_nccl: bad argument");
    if (load_rccl()) return -1;
    dpe_comm *c = new dpe_comm();
    c->rank = rank; c->nRanks = nRanks; c->backend = DPE_COMM_RCCL;
    c->nccl = (ncclComm_t)ncclComm;
    c->ownsNccl = false;   // the caller's communicator: not destroyed here
    *out = c;
    return 0;
}

int dpe_comm_destroy(dpe_comm *c)
{
    if (!c) return 0;
    if (c->nccl && c->ownsNccl) (void)g_rccl.CommDestroy(c->nccl);
    delete c;
    return 0;
}

int dpe_comm_rank(const dpe_comm *c, int32_t *rank, int32_t *nRanks)
{
    DPE_REQUIRE(c, "[dpe_comm] rank: null handle");
    if (rank) *rank = c->rank;
    if (nRanks) *nRanks = c->nRanks;
    return 0;
}

int dpe_comm_allreduce_max_u64(dpe_comm *c, uint64_t *data_dev, int64_t count, dpe_stream_t stream_)
{
    DPE_REQUIRE(c && data_dev && count > 0, "[dpe_comm] allreduce: bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    if (c->backend == DPE_COMM_RCCL) {
        const ncclResult_t r = g_rccl.AllReduce(data_dev, data_dev, (size_t)count, ncclUint64, ncclMax, c->nccl, stream);
        DPE_REQUIRE(r == ncclSuccess, "[dpe_comm] ncclAllReduce: %s", g_rccl.GetErrorString(r));
        return 0;
    }
    std::vector<uint64_t> mine((size_t)count);
    DPE_CHECK_HIP(hipMemcpyAsync(mine.data(), data_dev, sizeof(uint64_t) * count, hipMemcpyDeviceToHost, stream));
    DPE_CHECK_HIP(hipStreamSynchronize(stream));
    if (hostfile_exchange(c, "allreduce", mine.data(), sizeof(uint64_t) * count, c->host)) return -1;
    const uint64_t *all = reinterpret_cast<const uint64_t *>(c->host.data());
    for (int r = 0; r < c->nRanks; ++r)
        for (int64_t i = 0; i < count; ++i)
            if (all[(size_t)r * count + i] > mine[i]) mine[i] = all[(size_t)r * count + i];
    DPE_CHECK_HIP(hipMemcpy(data_dev, mine.data(), sizeof(uint64_t) * count, hipMemcpyHostToDevice));
    return 0;
}

int dpe_comm_allgather(dpe_comm *c, const void *send_dev, void *recv_dev, int64_t bytesPerRank, dpe_stream_t stream_)
{
    DPE_REQUIRE(c && send_dev && recv_dev && bytesPerRank > 0, "[dpe_comm] allgather: bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    if (c->backend == DPE_COMM_RCCL) {
        const ncclResult_t r = g_rccl.AllGather(send_dev, recv_dev, (size_t)bytesPerRank, ncclUint8, c->nccl, stream);
        DPE_REQUIRE(r == ncclSuccess, "[dpe_comm] ncclAllGather: %s", g_rccl.GetErrorString(r));
        return 0;
    }
    std::vector<unsigned char> mine((size_t)bytesPerRank);
    DPE_CHECK_HIP(hipMemcpyAsync(mine.data(), send_dev, (size_t)bytesPerRank, hipMemcpyDeviceToHost, stream));
    DPE_CHECK_HIP(hipStreamSynchronize(stream));
    if (hostfile_exchange(c, "allgather", mine.data(), (size_t)bytesPerRank, c->host)) return -1;
    DPE_CHECK_HIP(hipMemcpy(recv_dev, c->host.data(), (size_t)bytesPerRank * c->nRanks, hipMemcpyHostToDevice));
    return 0;
}

}  // extern "C"
