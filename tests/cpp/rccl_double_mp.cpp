// rccl_double_mp.cpp -- TEST INFRASTRUCTURE: the multi-PROCESS sibling of rccl_double.cpp.  Same purpose -- the real RCCL refuses two ranks on one
// device and the test boxes have ONE MI355X -- but here a rank is a process, which is what `bench.py --gpus N` launches (torch.distributed.run): with
// this library named to the bench (ONEPIECE_BENCH_RCCL_LIBRARY) its N > 1 path -- RcclCommunicator: ncclGetUniqueId on rank 0, the 128 bytes carried over
// the process group, ncclCommInitRank; then op_volume_merge_rccl on that communicator -- runs end to end with all ranks on device 0.
//
// Mechanics: the unique id names a directory of files under /tmp (not /dev/shm: containers cap it at 64 MB and a rank stages hundreds of MB); the header
// file holds a process-shared pthread barrier and per-rank message directories, every rank has a data file it grows on demand and the others map read-only.
// Semantics kept (as in rccl_double.cpp): collectives are called by every rank in the same order with matching counts; the result is in recvbuff when
// the call returns (the double synchronises the stream it is given and copies through the host); sums are formed in rank order; ncclSend / ncclRecv only
// inside ncclGroupStart / ncclGroupEnd, and every rank closes a group for each group any rank closes.  Nothing here models RCCL's performance.
//
// Build: hipcc -O2 -fPIC -shared tests/cpp/rccl_double_mp.cpp -o tests/cpp/librccl_double_mp.so -lpthread
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr int kMaxRanks = 16, kMaxMsgs = 64;
struct Msg { int dst; size_t offset, bytes; };
struct Header {
    std::atomic<int> magic;     // 0x52434c44 once rank 0 has initialised the barrier
    int n;
    pthread_barrier_t barrier;
    size_t bytes[kMaxRanks];    // what each rank staged in the current collective
    size_t file_size[kMaxRanks];
    int n_msgs[kMaxRanks];      // messages each rank posted in the current group
    Msg msgs[kMaxRanks][kMaxMsgs];
};

struct Comm {
    std::string dir;
    int rank = 0, n = 0;
    Header* h = nullptr;
    int fd[kMaxRanks];
    char* map[kMaxRanks];
    size_t mapped[kMaxRanks];
    Comm() { for (int i = 0; i < kMaxRanks; ++i) { fd[i] = -1; map[i] = nullptr; mapped[i] = 0; } }
};

struct P2P { bool send; void* buf; size_t bytes; int peer; Comm* c; hipStream_t stream; };
thread_local int tl_depth = 0;
thread_local std::vector<P2P> tl_ops;
Comm*& last_comm() { static thread_local Comm* c = nullptr; return c; }

size_t type_size(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}
template <class T>
void fold(T* acc, const T* x, size_t n, ncclRedOp_t op) {
    switch (op) {
        case ncclSum: for (size_t i = 0; i < n; ++i) acc[i] = acc[i] + x[i]; break;
        case ncclMax: for (size_t i = 0; i < n; ++i) acc[i] = x[i] > acc[i] ? x[i] : acc[i]; break;
        case ncclMin: for (size_t i = 0; i < n; ++i) acc[i] = x[i] < acc[i] ? x[i] : acc[i]; break;
        default: break;
    }
}
bool fold_any(void* acc, const void* x, size_t n, ncclDataType_t t, ncclRedOp_t op) {
    if (op != ncclSum && op != ncclMax && op != ncclMin) return false;
    switch (t) {
        case ncclInt32: fold((int32_t*)acc, (const int32_t*)x, n, op); return true;
        case ncclInt64: fold((int64_t*)acc, (const int64_t*)x, n, op); return true;
        case ncclFloat32: fold((float*)acc, (const float*)x, n, op); return true;
        case ncclFloat64: fold((double*)acc, (const double*)x, n, op); return true;
        default: return false;
    }
}

std::string data_path(const Comm* c, int r) { return c->dir + "/rank" + std::to_string(r); }

// this rank's data file holds at least `bytes` (grown with ftruncate, re-mapped read-write)
bool ensure_own(Comm* c, size_t bytes) {
    const int r = c->rank;
    if (c->mapped[r] >= bytes && c->map[r]) return true;
    size_t want = c->mapped[r] ? c->mapped[r] : (size_t)1 << 20;
    while (want < bytes) want *= 2;
    if (c->fd[r] < 0) c->fd[r] = open(data_path(c, r).c_str(), O_RDWR | O_CREAT, 0600);
    if (c->fd[r] < 0 || ftruncate(c->fd[r], (off_t)want) != 0) return false;
    if (c->map[r]) munmap(c->map[r], c->mapped[r]);
    c->map[r] = (char*)mmap(nullptr, want, PROT_READ | PROT_WRITE, MAP_SHARED, c->fd[r], 0);
    if (c->map[r] == MAP_FAILED) { c->map[r] = nullptr; c->mapped[r] = 0; return false; }
    c->mapped[r] = want;
    c->h->file_size[r] = want;
    return true;
}
// rank r's data file, at least as large as its owner says it is (call after a barrier that follows the owner's writes)
const char* peer_view(Comm* c, int r) {
    if (r == c->rank) return c->map[r];
    const size_t size = c->h->file_size[r];
    if (c->map[r] && c->mapped[r] >= size) return c->map[r];
    if (c->fd[r] < 0) c->fd[r] = open(data_path(c, r).c_str(), O_RDONLY);
    if (c->fd[r] < 0) return nullptr;
    if (c->map[r]) munmap(c->map[r], c->mapped[r]);
    c->map[r] = (char*)mmap(nullptr, size, PROT_READ, MAP_SHARED, c->fd[r], 0);
    if (c->map[r] == MAP_FAILED) { c->map[r] = nullptr; c->mapped[r] = 0; return nullptr; }
    c->mapped[r] = size;
    return c->map[r];
}
void meet(Comm* c) { pthread_barrier_wait(&c->h->barrier); }

// every rank: wait for the stream, copy `bytes` of sendbuff into its data file, meet the others
ncclResult_t stage_in(Comm* c, const void* sendbuff, size_t bytes, hipStream_t stream) {
    last_comm() = c;
    ncclResult_t rc = ncclSuccess;
    if (hipStreamSynchronize(stream) != hipSuccess) rc = ncclUnhandledCudaError;
    if (!ensure_own(c, bytes ? bytes : 1)) rc = ncclSystemError;
    c->h->bytes[c->rank] = bytes;
    if (rc == ncclSuccess && bytes && hipMemcpy(c->map[c->rank], sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = ncclUnhandledCudaError;
    meet(c); // (also on failure: the other ranks must not be left waiting)
    return rc;
}

} // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    std::memset(id, 0, sizeof(*id));
    std::random_device rd;
    std::snprintf(id->internal, sizeof(id->internal), "/tmp/rccl_double_mp_%08x%08x_%d", (unsigned)rd(), (unsigned)rd(), (int)getpid());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks || id.internal[0] != '/') return ncclInvalidArgument;
    Comm* c = new Comm();
    c->dir = id.internal; c->rank = rank; c->n = nranks;
    const std::string hp = c->dir + "/header";
    int fd = -1;
    if (rank == 0) {
        if (mkdir(c->dir.c_str(), 0700) != 0) { delete c; return ncclSystemError; }
        fd = open((hp + ".tmp").c_str(), O_RDWR | O_CREAT | O_EXCL, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)sizeof(Header)) != 0) { delete c; return ncclSystemError; }
    } else {
        for (int tries = 0; tries < 60000 && fd < 0; ++tries) { // rank 0 renames the header into place once it is initialised
            fd = open(hp.c_str(), O_RDWR);
            if (fd < 0) std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        if (fd < 0) { delete c; return ncclSystemError; }
    }
    c->h = (Header*)mmap(nullptr, sizeof(Header), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->h == MAP_FAILED) { delete c; return ncclSystemError; }
    if (rank == 0) {
        std::memset((void*)c->h, 0, sizeof(Header));
        c->h->n = nranks;
        pthread_barrierattr_t a;
        pthread_barrierattr_init(&a);
        pthread_barrierattr_setpshared(&a, PTHREAD_PROCESS_SHARED);
        pthread_barrier_init(&c->h->barrier, &a, (unsigned)nranks);
        pthread_barrierattr_destroy(&a);
        c->h->magic.store(0x52434c44);
        if (rename((hp + ".tmp").c_str(), hp.c_str()) != 0) { delete c; return ncclSystemError; }
    } else {
        for (int tries = 0; tries < 60000 && c->h->magic.load() != 0x52434c44; ++tries) std::this_thread::sleep_for(std::chrono::milliseconds(1));
        if (c->h->magic.load() != 0x52434c44 || c->h->n != nranks) { delete c; return ncclSystemError; }
    }
    if (!ensure_own(c, 1)) { delete c; return ncclSystemError; }
    meet(c); // every rank's data file exists
    *comm = (ncclComm_t)c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm* c = (Comm*)comm;
    if (!c) return ncclInvalidArgument;
    meet(c); // nobody reads a peer's file any more
    for (int r = 0; r < c->n; ++r) {
        if (c->map[r]) munmap(c->map[r], c->mapped[r]);
        if (c->fd[r] >= 0) close(c->fd[r]);
    }
    unlink(data_path(c, c->rank).c_str());
    meet(c); // every data file is gone
    if (c->rank == 0) { unlink((c->dir + "/header").c_str()); rmdir(c->dir.c_str()); }
    munmap((void*)c->h, sizeof(Header));
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) {
    if (!comm || !count) return ncclInvalidArgument;
    *count = ((Comm*)comm)->n;
    return ncclSuccess;
}
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* rank) {
    if (!comm || !rank) return ncclInvalidArgument;
    *rank = ((Comm*)comm)->rank;
    return ncclSuccess;
}
const char* ncclGetErrorString(ncclResult_t r) {
    return r == ncclSuccess ? "no error" : (r == ncclInvalidArgument ? "invalid argument (rccl double, multi-process)" : "error (rccl double, multi-process)");
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream) {
    Comm* c = (Comm*)comm;
    const size_t ts = type_size(datatype);
    if (!c || !ts) return ncclInvalidArgument;
    const size_t bytes = sendcount * ts;
    ncclResult_t rc = stage_in(c, sendbuff, bytes, stream);
    for (int r = 0; r < c->n && rc == ncclSuccess; ++r) {
        if (c->h->bytes[r] != bytes) { rc = ncclInvalidArgument; break; } // mismatched counts: a protocol error of the caller
        const char* src = bytes ? peer_view(c, r) : nullptr;
        if (bytes && (!src || hipMemcpy((char*)recvbuff + (size_t)r * bytes, src, bytes, hipMemcpyHostToDevice) != hipSuccess)) rc = ncclUnhandledCudaError;
    }
    meet(c); // the files may be overwritten by the next collective from here on
    return rc;
}

static ncclResult_t reduce_impl(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, int root, ncclComm_t comm, hipStream_t stream) {
    Comm* c = (Comm*)comm;
    const size_t ts = type_size(datatype);
    if (!c || !ts) return ncclInvalidArgument;
    const size_t bytes = count * ts;
    ncclResult_t rc = stage_in(c, sendbuff, bytes, stream);
    if (rc == ncclSuccess && (root < 0 || c->rank == root)) { // root < 0: all-reduce, every rank folds for itself
        std::vector<char> acc(bytes);
        for (int r = 0; r < c->n; ++r) {
            if (c->h->bytes[r] != bytes) { rc = ncclInvalidArgument; break; }
            const char* src = bytes ? peer_view(c, r) : nullptr;
            if (bytes && !src) { rc = ncclSystemError; break; }
            if (r == 0) { if (bytes) std::memcpy(acc.data(), src, bytes); }
            else if (!fold_any(acc.data(), src, count, datatype, op)) { rc = ncclInvalidArgument; break; }
        }
        if (rc == ncclSuccess && bytes && hipMemcpy(recvbuff, acc.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) rc = ncclUnhandledCudaError;
    }
    meet(c);
    return rc;
}
ncclResult_t ncclReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, int root, ncclComm_t comm, hipStream_t stream) {
    Comm* c = (Comm*)comm;
    if (!c || root < 0 || root >= c->n) return ncclInvalidArgument;
    return reduce_impl(sendbuff, recvbuff, count, datatype, op, root, comm, stream);
}
ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
    return reduce_impl(sendbuff, recvbuff, count, datatype, op, -1, comm, stream);
}

ncclResult_t ncclGroupStart(void) { ++tl_depth; return ncclSuccess; }
static ncclResult_t post(bool send, void* buf, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    Comm* c = (Comm*)comm;
    const size_t ts = type_size(datatype);
    if (!c || !ts || peer < 0 || peer >= c->n || peer == c->rank) return ncclInvalidArgument;
    if (tl_depth <= 0) return ncclInvalidUsage; // the double serves point-to-point calls at ncclGroupEnd, where all ranks meet
    tl_ops.push_back(P2P{send, buf, count * ts, peer, c, stream});
    return ncclSuccess;
}
ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    return post(true, const_cast<void*>(sendbuff), count, datatype, peer, comm, stream);
}
ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    return post(false, recvbuff, count, datatype, peer, comm, stream);
}

ncclResult_t ncclGroupEnd(void) {
    if (tl_depth <= 0) return ncclInvalidUsage;
    if (--tl_depth > 0) return ncclSuccess;
    std::vector<P2P> ops;
    ops.swap(tl_ops);
    Comm* c = ops.empty() ? last_comm() : ops[0].c; // (an empty group: the communicator this rank used last)
    if (!c) return ncclInvalidUsage;
    last_comm() = c;
    ncclResult_t rc = ncclSuccess;
    // the sends of the group, back to back in this rank's data file, with a directory in the header
    size_t total = 0;
    int n_send = 0;
    for (const P2P& o : ops) if (o.send && o.c == c) { total += o.bytes; ++n_send; }
    if (n_send > kMaxMsgs) rc = ncclInvalidUsage;
    if (!ensure_own(c, total ? total : 1)) rc = ncclSystemError;
    size_t off = 0;
    int k = 0;
    for (const P2P& o : ops) {
        if (o.c != c) { rc = ncclInvalidArgument; continue; }
        if (hipStreamSynchronize(o.stream) != hipSuccess) rc = ncclUnhandledCudaError;
        if (!o.send || rc != ncclSuccess) continue;
        if (o.bytes && hipMemcpy(c->map[c->rank] + off, o.buf, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = ncclUnhandledCudaError;
        c->h->msgs[c->rank][k++] = Msg{o.peer, off, o.bytes};
        off += o.bytes;
    }
    c->h->n_msgs[c->rank] = rc == ncclSuccess ? k : 0;
    meet(c); // every send of the group is in its file
    std::vector<int> next((size_t)c->n, 0); // per peer: how many of its messages to this rank have been consumed
    for (const P2P& o : ops) {
        if (o.send || o.c != c) continue;
        const int p = o.peer;
        int seen = 0, found = -1;
        for (int m = 0; m < c->h->n_msgs[p]; ++m)
            if (c->h->msgs[p][m].dst == c->rank && seen++ == next[(size_t)p]) { found = m; break; }
        ++next[(size_t)p];
        if (found < 0 || c->h->msgs[p][found].bytes != o.bytes) { rc = ncclInvalidArgument; continue; } // unmatched receive / mismatched count
        const char* src = o.bytes ? peer_view(c, p) : nullptr;
        if (o.bytes && (!src || hipMemcpy(o.buf, src + c->h->msgs[p][found].offset, o.bytes, hipMemcpyHostToDevice) != hipSuccess)) rc = ncclUnhandledCudaError;
    }
    for (int p = 0; p < c->n; ++p) { // something was sent to this rank that it did not receive
        int to_me = 0;
        for (int m = 0; m < c->h->n_msgs[p]; ++m) to_me += c->h->msgs[p][m].dst == c->rank;
        if (p != c->rank && to_me != next[(size_t)p]) rc = ncclInvalidArgument;
    }
    meet(c); // every receive is served: the senders may overwrite their files
    return rc;
}

} // extern "C"
