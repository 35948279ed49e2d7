// rccl_double.cpp -- TEST INFRASTRUCTURE: a stand-in for the few RCCL entry points op_volume_merge_rccl binds at run time
// (onepiece_amd/csrc/merge_rccl.hip: ncclCommCount, ncclCommUserRank, ncclAllGather, ncclAllReduce, ncclReduce, ncclSend, ncclRecv,
// ncclGroupStart, ncclGroupEnd, ncclGetErrorString) plus ncclCommInitAll / ncclCommDestroy for the driver that creates the communicators.
//
// Why it exists: the real RCCL refuses two ranks on one device, and the boxes the GPU tests run on have ONE MI355X -- so the
// multi-rank control flow of op_volume_merge_rccl (padded key all-gather, the ~0 sentinel of the union, the three agreement
// points, the sliced reduce, root-only unpack) would otherwise only ever execute on an 8-GPU node.  With this library
// (named to the library with op_runtime_set_rccl_library before the first merge) N ranks = N host threads of one
// process, all on device 0, exchange their buffers through host memory.  The product never loads it unless that variable
// says so; nothing here is a model of RCCL's performance.
//
// Semantics kept: collectives are called by every rank of the communicator, in the same order, with matching counts; the
// result is in recvbuff when the call returns (the double synchronises the stream it is given, copies through the host,
// and meets the other ranks at a barrier -- a legal, if slow, implementation of stream-ordered completion).  In-place
// operation (sendbuff inside recvbuff) works as in NCCL.  Sums are formed in rank order, so results are deterministic.
// Point-to-point: ncclSend / ncclRecv are only supported INSIDE ncclGroupStart / ncclGroupEnd, and every rank of the communicator
// must close a group (possibly an empty one) for each group any rank closes -- which is how op_volume_merge_rccl uses them: the
// group's sends go to per-pair host mailboxes, all ranks meet, the receives are served in posting order, all ranks meet again.
//
// Build: hipcc -O2 -fPIC -shared tests/cpp/rccl_double.cpp -o tests/cpp/librccl_double.so
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <vector>

namespace {

struct Group {
    int n = 0;
    std::mutex mu;
    std::condition_variable cv;
    int waiting = 0;
    uint64_t generation = 0;
    std::vector<std::vector<char>> stage; // one host buffer per rank
    std::vector<size_t> bytes;            // what each rank staged in the current collective
    std::vector<std::vector<std::vector<char>>> mail; // [src * n + dst]: the messages src sent dst in the current group, in posting order
    int live = 0;                         // communicators not destroyed yet
    void barrier() {
        std::unique_lock<std::mutex> lk(mu);
        const uint64_t g = generation;
        if (++waiting == n) { waiting = 0; ++generation; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != g; });
    }
};

struct Comm { Group* g; int rank; };

struct P2P { bool send; void* buf; size_t bytes; int peer; Comm* c; hipStream_t stream; };
thread_local int tl_depth = 0;           // (a rank is a host thread)
thread_local std::vector<P2P> tl_ops;

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

// every rank: wait for the stream, copy `bytes` of sendbuff to its host slot, meet the others
Comm*& last_comm();
ncclResult_t stage_in(Comm* c, const void* sendbuff, size_t bytes, hipStream_t stream) {
    last_comm() = c;
    Group* g = c->g;
    ncclResult_t rc = ncclSuccess;
    if (hipStreamSynchronize(stream) != hipSuccess) rc = ncclUnhandledCudaError;
    std::vector<char>& s = g->stage[(size_t)c->rank];
    if (s.size() < bytes) s.resize(bytes);
    g->bytes[(size_t)c->rank] = bytes;
    if (rc == ncclSuccess && bytes && hipMemcpy(s.data(), sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = ncclUnhandledCudaError;
    g->barrier(); // (also on failure: the other ranks must not be left waiting)
    return rc;
}

Comm*& last_comm() { static thread_local Comm* c = nullptr; return c; } // the communicator of an EMPTY group is not known from its operations: the last one this rank's thread used

} // namespace

extern "C" {

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
    (void)devlist; // every rank lives on the device its volume was created on: the point of the double
    if (!comms || ndev < 1) return ncclInvalidArgument;
    Group* g = new Group();
    g->n = ndev; g->live = ndev;
    g->stage.resize((size_t)ndev); g->bytes.assign((size_t)ndev, 0);
    g->mail.resize((size_t)ndev * ndev);
    for (int r = 0; r < ndev; ++r) comms[r] = (ncclComm_t) new Comm{g, r};
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm* c = (Comm*)comm;
    if (!c) return ncclInvalidArgument;
    Group* g = c->g;
    bool last;
    { std::lock_guard<std::mutex> lk(g->mu); last = --g->live == 0; }
    delete c;
    if (last) delete g;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) {
    if (!comm || !count) return ncclInvalidArgument;
    *count = ((Comm*)comm)->g->n;
    return ncclSuccess;
}

ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* rank) {
    if (!comm || !rank) return ncclInvalidArgument;
    *rank = ((Comm*)comm)->rank;
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : (r == ncclInvalidArgument ? "invalid argument (rccl double)" : "error (rccl double)"); }

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream) {
    Comm* c = (Comm*)comm;
    const size_t ts = type_size(datatype);
    if (!c || !ts) return ncclInvalidArgument;
    Group* g = c->g;
    const size_t bytes = sendcount * ts;
    ncclResult_t rc = stage_in(c, sendbuff, bytes, stream);
    for (int r = 0; r < g->n && rc == ncclSuccess; ++r) {
        if (g->bytes[(size_t)r] != bytes) { rc = ncclInvalidArgument; break; } // mismatched counts: a protocol error of the caller
        if (bytes && hipMemcpy((char*)recvbuff + (size_t)r * bytes, g->stage[(size_t)r].data(), bytes, hipMemcpyHostToDevice) != hipSuccess) rc = ncclUnhandledCudaError;
    }
    g->barrier(); // the slots may be overwritten by the next collective from here on
    return rc;
}

static ncclResult_t reduce_impl(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, int root, ncclComm_t comm, hipStream_t stream) {
    Comm* c = (Comm*)comm;
    const size_t ts = type_size(datatype);
    if (!c || !ts) return ncclInvalidArgument;
    Group* g = c->g;
    const size_t bytes = count * ts;
    ncclResult_t rc = stage_in(c, sendbuff, bytes, stream);
    if (rc == ncclSuccess && (root < 0 || c->rank == root)) { // root < 0: all-reduce, every rank folds for itself
        std::vector<char> acc(bytes);
        for (int r = 0; r < g->n; ++r) {
            if (g->bytes[(size_t)r] != bytes) { rc = ncclInvalidArgument; break; }
            if (r == 0) std::memcpy(acc.data(), g->stage[0].data(), bytes);
            else if (!fold_any(acc.data(), g->stage[(size_t)r].data(), count, datatype, op)) { rc = ncclInvalidArgument; break; }
        }
        if (rc == ncclSuccess && bytes && hipMemcpy(recvbuff, acc.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) rc = ncclUnhandledCudaError;
    }
    g->barrier();
    return rc;
}

ncclResult_t ncclReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, int root, ncclComm_t comm, hipStream_t stream) {
    Comm* c = (Comm*)comm;
    if (!c || root < 0 || root >= c->g->n) return ncclInvalidArgument;
    return reduce_impl(sendbuff, recvbuff, count, datatype, op, root, comm, stream);
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
    return reduce_impl(sendbuff, recvbuff, count, datatype, op, -1, comm, stream);
}

ncclResult_t ncclGroupStart(void) { ++tl_depth; return ncclSuccess; }

static ncclResult_t post(bool send, void* buf, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    Comm* c = (Comm*)comm;
    const size_t ts = type_size(datatype);
    if (!c || !ts || peer < 0 || peer >= c->g->n || peer == c->rank) return ncclInvalidArgument;
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

// The communicator of an EMPTY group is not known from its operations: the double remembers the last communicator a rank's thread used.
ncclResult_t ncclGroupEnd(void) {
    if (tl_depth <= 0) return ncclInvalidUsage;
    if (--tl_depth > 0) return ncclSuccess;
    std::vector<P2P> ops;
    ops.swap(tl_ops);
    Comm* c = ops.empty() ? last_comm() : ops[0].c;
    if (!c) return ncclInvalidUsage;
    Group* g = c->g;
    ncclResult_t rc = ncclSuccess;
    const size_t n = (size_t)g->n, me = (size_t)c->rank;
    for (const P2P& o : ops) {
        if (o.c != c) { rc = ncclInvalidArgument; continue; }
        if (hipStreamSynchronize(o.stream) != hipSuccess) rc = ncclUnhandledCudaError;
        if (!o.send) continue;
        std::vector<char> m(o.bytes);
        if (o.bytes && hipMemcpy(m.data(), o.buf, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = ncclUnhandledCudaError;
        g->mail[me * n + (size_t)o.peer].push_back(std::move(m));
    }
    g->barrier(); // every send of the group is in its mailbox
    std::vector<size_t> next(n, 0);
    for (const P2P& o : ops) {
        if (o.send || o.c != c) continue;
        std::vector<std::vector<char>>& box = g->mail[(size_t)o.peer * n + me];
        const size_t k = next[(size_t)o.peer]++;
        if (k >= box.size() || box[k].size() != o.bytes) { rc = ncclInvalidArgument; continue; } // unmatched receive / mismatched count: a protocol error of the caller
        if (o.bytes && hipMemcpy(o.buf, box[k].data(), o.bytes, hipMemcpyHostToDevice) != hipSuccess) rc = ncclUnhandledCudaError;
    }
    for (size_t p = 0; p < n; ++p)
        if (next[p] != g->mail[p * n + me].size()) rc = ncclInvalidArgument; // something was sent to this rank that it did not receive
    g->barrier(); // every receive is served: the senders may drop their messages
    for (size_t p = 0; p < n; ++p) g->mail[me * n + p].clear();
    return rc;
}

} // extern "C"
