// dpe_comm.hip -- the exchange step of the multi-GPU path behind the C-ABI (one process per GPU): all-reduce(MAX) of the
// packed arg-max keys and all-gather of the stage-1 banks, so that a C++ host (the dsp::Flow mirror, or the reference's own
// flow) shards the manifold grid without Python.
//
// Two transports:
//   RCCL      ncclAllReduce / ncclAllGather over xGMI.  librccl is opened with dlopen at the first use (a process that
//             already carries an RCCL -- PyTorch ships its own copy under the same SONAME -- keeps using that one); the
//             unique id travels through the rendezvous directory.
//   host files  every rank copies its buffer to the host, publishes it as a file under <rendezvous>, reads the others' and
//             reduces on the host.  For functional tests of the sharded path with several ranks on ONE GPU (RCCL refuses
//             two ranks per device); never a performance path.
// Rendezvous (both transports, nRanks > 1): a directory may be REUSED by later runs, so nothing in it is trusted by name
// alone.  Rank r > 0 publishes a fresh random token in join.<r>; rank 0 draws a run nonce, publishes
// {nonce, tokens of all ranks, ncclUniqueId} in `rendezvous` and re-publishes whenever a join file changes; rank r accepts
// a `rendezvous` file only if it echoes ITS token, then answers with ack.<r> = {nonce, token}; rank 0 proceeds once every
// ack carries its nonce and then removes the handshake files.  A file left behind by an earlier run can therefore delay
// nobody and is never believed.  Every payload of the host-file transport starts with {nonce, sequence number} and a reader
// waits until it sees both.
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

int write_atomic(const std::string &path, const void *data, size_t bytes)
{
    const std::string tmp = path + ".tmp." + std::to_string((long long)getpid());
    FILE *f = std::fopen(tmp.c_str(), "wb");
    if (!f) return -1;
    const size_t n = std::fwrite(data, 1, bytes, f);
    std::fclose(f);
    if (n != bytes) return -1;
    return std::rename(tmp.c_str(), path.c_str());
}

// 0: read `bytes` bytes; -1: missing or shorter (a writer renames complete files into place, so short = not ours yet)
int read_file(const std::string &path, void *data, size_t bytes)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return -1;
    const size_t n = std::fread(data, 1, bytes, f);
    std::fclose(f);
    return n == bytes ? 0 : -1;
}

double seconds_since(const std::chrono::steady_clock::time_point &t0)
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

void nap() { std::this_thread::sleep_for(std::chrono::microseconds(200)); }

// fresh 64-bit token: /dev/urandom, else clock + pid (never 0: 0 marks "no token yet")
unsigned long long fresh_token()
{
    unsigned long long t = 0;
    if (FILE *f = std::fopen("/dev/urandom", "rb")) {
        if (std::fread(&t, sizeof(t), 1, f) != 1) t = 0;
        std::fclose(f);
    }
    if (!t) t = (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count() * 6364136223846793005ull + (unsigned long long)getpid();
    return t ? t : 1ull;
}

constexpr unsigned long long kMagic = 0x4450455f52445a31ull;   // "DPE_RDZ1"
constexpr int kMaxRanks = 64;
struct RendezvousFile {
    unsigned long long magic, nonce;
    long long nRanks;
    unsigned long long token[kMaxRanks];
    unsigned char id[128];   // ncclUniqueId (RCCL transport)
};
static_assert(sizeof(ncclUniqueId) <= 128, "the unique id travels in a 128-byte field");
struct AckFile { unsigned long long nonce, token; };
struct PayloadHeader { unsigned long long nonce, seq; };

}  // namespace

struct dpe_comm {
    int rank = 0, nRanks = 1, backend = DPE_COMM_RCCL;
    ncclComm_t nccl = nullptr;
    bool ownsNccl = false;
    std::string dir;
    unsigned long long nonce = 0;       // run identity agreed in the rendezvous (0: single rank, no files)
    unsigned long long seq = 0;
    std::vector<unsigned char> host;   // staging of the host-file transport
    std::vector<std::string> mine;     // payload files this rank still has in the directory
    // One communicator, several streams (the lanes of a dpe_pipe alternate): two collectives of the same communicator must never be in
    // flight at once, so every RCCL call first makes its stream wait for the previous call's event when that one went to ANOTHER stream.
    // Every rank alternates its lanes in the same order, so the chain is the same on every rank.
    hipEvent_t lastDone = nullptr;
    hipStream_t lastStream = nullptr;
    bool haveLast = false;
};

namespace {
int chain_before(dpe_comm *c, hipStream_t stream)
{
    if (c->haveLast && c->lastStream != stream) DPE_CHECK_HIP(hipStreamWaitEvent(stream, c->lastDone, 0));
    return 0;
}
int chain_after(dpe_comm *c, hipStream_t stream)
{
    if (!c->lastDone) DPE_CHECK_HIP(hipEventCreateWithFlags(&c->lastDone, hipEventDisableTiming));
    DPE_CHECK_HIP(hipEventRecord(c->lastDone, stream));
    c->lastStream = stream;
    c->haveLast = true;
    return 0;
}
}  // namespace

namespace {

// Agrees on the run nonce (and, for RCCL, carries the unique id) through `dir`; see the file header.  timeoutS per phase.
int rendezvous(dpe_comm *c, unsigned char *id /* [128] in (rank 0) / out */, double timeoutS)
{
    const std::string rdz = c->dir + "/rendezvous";
    auto joinName = [&](int r) { return c->dir + "/join." + std::to_string(r); };
    auto ackName = [&](int r) { return c->dir + "/ack." + std::to_string(r); };
    const auto t0 = std::chrono::steady_clock::now();
    if (c->rank == 0) {
        (void)std::remove(rdz.c_str());   // whatever an earlier run left here is not ours
        RendezvousFile f{};
        f.magic = kMagic; f.nonce = fresh_token(); f.nRanks = c->nRanks;
        memcpy(f.id, id, sizeof(f.id));
        bool published = false;
        while (true) {
            bool changed = false, haveAll = true;
            for (int r = 1; r < c->nRanks; ++r) {
                unsigned long long t = 0;
                if (read_file(joinName(r), &t, sizeof(t)) != 0 || t == 0) { haveAll = false; continue; }
                if (t != f.token[r]) { f.token[r] = t; changed = true; }
            }
            if (haveAll && (changed || !published)) {
                DPE_REQUIRE(write_atomic(rdz, &f, sizeof(f)) == 0, "[dpe_comm] cannot write %s", rdz.c_str());
                published = true;
            }
            bool acked = published;
            for (int r = 1; r < c->nRanks && acked; ++r) {
                AckFile a{};
                acked = read_file(ackName(r), &a, sizeof(a)) == 0 && a.nonce == f.nonce && a.token == f.token[r];
            }
            if (acked) break;
            DPE_REQUIRE(seconds_since(t0) < timeoutS, "[dpe_comm] rank 0: %s after %.0f s in %s",
                        haveAll ? "not every rank acknowledged the rendezvous" : "not every rank joined", timeoutS, c->dir.c_str());
            nap();
        }
        c->nonce = f.nonce;
        // everybody holds the nonce (and the id): the handshake files are done with
        (void)std::remove(rdz.c_str());
        for (int r = 1; r < c->nRanks; ++r) { (void)std::remove(joinName(r).c_str()); (void)std::remove(ackName(r).c_str()); }
        return 0;
    }
    const unsigned long long token = fresh_token();
    (void)std::remove(ackName(c->rank).c_str());
    DPE_REQUIRE(write_atomic(joinName(c->rank), &token, sizeof(token)) == 0, "[dpe_comm] cannot write %s", joinName(c->rank).c_str());
    RendezvousFile f{};
    while (true) {
        if (read_file(rdz, &f, sizeof(f)) == 0 && f.magic == kMagic && f.nRanks == c->nRanks && f.token[c->rank] == token) break;
        DPE_REQUIRE(seconds_since(t0) < timeoutS, "[dpe_comm] rank %d: no rendezvous for this run at %s after %.0f s", c->rank, rdz.c_str(), timeoutS);
        nap();
    }
    c->nonce = f.nonce;
    memcpy(id, f.id, sizeof(f.id));
    const AckFile a{f.nonce, token};
    DPE_REQUIRE(write_atomic(ackName(c->rank), &a, sizeof(a)) == 0, "[dpe_comm] cannot write %s", ackName(c->rank).c_str());
    return 0;
}

}  // namespace

// host-file transport: publish this rank's buffer as <dir>/<tag>.<seq>.<rank> = {nonce, seq, payload}, collect everybody's.
// A file of an earlier run under the same name carries another nonce: the reader keeps waiting for ours.
static int hostfile_exchange(dpe_comm *c, const char *tag, const void *mine, size_t bytes, std::vector<unsigned char> &all)
{
    const unsigned long long seq = c->seq++;
    auto name = [&](int r, unsigned long long s) { return c->dir + "/" + tag + "." + std::to_string(s) + "." + std::to_string(r); };
    std::vector<unsigned char> buf(sizeof(PayloadHeader) + bytes);
    const PayloadHeader hdr{c->nonce, seq};
    memcpy(buf.data(), &hdr, sizeof(hdr));
    memcpy(buf.data() + sizeof(hdr), mine, bytes);
    DPE_REQUIRE(write_atomic(name(c->rank, seq), buf.data(), buf.size()) == 0, "[dpe_comm] cannot write %s", name(c->rank, seq).c_str());
    c->mine.push_back(name(c->rank, seq));
    all.resize(bytes * c->nRanks);
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < c->nRanks; ++r) {
        while (true) {
            PayloadHeader h{};
            if (read_file(name(r, seq), buf.data(), buf.size()) == 0 && (memcpy(&h, buf.data(), sizeof(h)), h.nonce == c->nonce && h.seq == seq)) break;
            DPE_REQUIRE(seconds_since(t0) < 120.0, "[dpe_comm] rank %d: no data of this run from rank %d (%s) after 120 s", c->rank, r,
                        name(r, seq).c_str());
            nap();
        }
        memcpy(all.data() + bytes * r, buf.data() + sizeof(PayloadHeader), bytes);
    }
    // everybody has passed exchange seq-2 by now (it needed seq-1 of all): this rank's files from then are unread
    while (c->mine.size() > 2) {
        (void)std::remove(c->mine.front().c_str());
        c->mine.erase(c->mine.begin());
    }
    return 0;
}

extern "C" {

int dpe_comm_create(int32_t rank, int32_t nRanks, const char *rendezvousPath, int32_t backend, dpe_comm **out)
{
    DPE_REQUIRE(out && nRanks >= 1 && rank >= 0 && rank < nRanks, "[dpe_comm] create: bad rank %d of %d", rank, nRanks);
    DPE_REQUIRE(nRanks <= kMaxRanks, "[dpe_comm] create: %d ranks (at most %d)", nRanks, kMaxRanks);
    DPE_REQUIRE(backend == DPE_COMM_RCCL || backend == DPE_COMM_HOSTFILES, "[dpe_comm] create: unknown backend %d", backend);
    DPE_REQUIRE(nRanks == 1 || (rendezvousPath && *rendezvousPath), "[dpe_comm] create: rendezvous directory missing");
    dpe_comm *c = new dpe_comm();
    c->rank = rank; c->nRanks = nRanks; c->backend = backend;
    c->dir = rendezvousPath ? rendezvousPath : "";
    unsigned char idBytes[128] = {};
    if (backend == DPE_COMM_RCCL) {
        if (load_rccl()) { delete c; return -1; }
        if (rank == 0) {
            ncclUniqueId id;
            const ncclResult_t r = g_rccl.GetUniqueId(&id);
            if (r != ncclSuccess) { dpe::set_error("[dpe_comm] ncclGetUniqueId: %s", g_rccl.GetErrorString(r)); delete c; return -1; }
            memcpy(idBytes, &id, sizeof(id));
        }
    }
    if (nRanks > 1) {
        double timeoutS = 120.0;
        if (const char *e = getenv("DPE_COMM_TIMEOUT_S")) timeoutS = atof(e) > 0 ? atof(e) : timeoutS;
        if (rendezvous(c, idBytes, timeoutS)) { delete c; return -1; }
    }
    if (backend == DPE_COMM_RCCL) {
        ncclUniqueId id;
        memcpy(&id, idBytes, sizeof(id));
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
    if (c->lastDone) (void)hipEventDestroy(c->lastDone);
    // the last payload files of this rank: a peer may still be reading the newest one (it is stamped with this run's
    // nonce, so a later run never mistakes it for its own), everything older is unread
    while (c->mine.size() > 1) {
        (void)std::remove(c->mine.front().c_str());
        c->mine.erase(c->mine.begin());
    }
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
        if (chain_before(c, stream)) return -1;
        const ncclResult_t r = g_rccl.AllReduce(data_dev, data_dev, (size_t)count, ncclUint64, ncclMax, c->nccl, stream);
        DPE_REQUIRE(r == ncclSuccess, "[dpe_comm] ncclAllReduce: %s", g_rccl.GetErrorString(r));
        return chain_after(c, stream);
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
        if (chain_before(c, stream)) return -1;
        const ncclResult_t r = g_rccl.AllGather(send_dev, recv_dev, (size_t)bytesPerRank, ncclUint8, c->nccl, stream);
        DPE_REQUIRE(r == ncclSuccess, "[dpe_comm] ncclAllGather: %s", g_rccl.GetErrorString(r));
        return chain_after(c, stream);
    }
    std::vector<unsigned char> mine((size_t)bytesPerRank);
    DPE_CHECK_HIP(hipMemcpyAsync(mine.data(), send_dev, (size_t)bytesPerRank, hipMemcpyDeviceToHost, stream));
    DPE_CHECK_HIP(hipStreamSynchronize(stream));
    if (hostfile_exchange(c, "allgather", mine.data(), (size_t)bytesPerRank, c->host)) return -1;
    DPE_CHECK_HIP(hipMemcpy(recv_dev, c->host.data(), (size_t)bytesPerRank * c->nRanks, hipMemcpyHostToDevice));
    return 0;
}

}  // extern "C"
