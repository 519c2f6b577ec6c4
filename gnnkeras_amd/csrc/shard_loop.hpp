// The sharded loop driven from native code: every iteration of a rank - own-range partial sums, the halo
// kernel (whole, or in chunk launches), the exchange of the rows just written, the stream dependencies between them - issued by ONE C
// call instead of ~5 .. 12 Python calls per iteration (gnnkeras_amd/distributed.py drives the same entry points one at a time; at 8
// ranks an iteration has ~240 us, and four chunk launches + four grouped send / receive rounds from the interpreter may not fit).
// The exchange goes over the RCCL C API on a communicator of the library's own (gnn_comm_*: the unique id travels through whatever
// the host already has - torch.distributed here), on an exchange stream of its own, ordered against the compute stream with events:
// no host synchronisation anywhere, the convergence gates stay device words as in the Python-driven loop, and per row the arithmetic
// is the same launches in the same order - bit-identical results (tests/test_gpu_round5.py, RCCL at world size 1; the interpreter
// loop stays the default and the reference for the gloo orchestration tests).
//
// RCCL is reached through dlopen / dlsym - librccl.so.1 is whatever the process has loaded already (PyTorch-ROCm ships and loads its
// own copy; two copies of RCCL in one process would each set up their own peer mappings) - so libgnnloop.so keeps no link-time
// dependency on it and single-GPU hosts never touch it.
#pragma once
#include <dlfcn.h>
#include <memory>
#include <string>
#include <rccl/rccl.h>

namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

RcclApi &rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *env = getenv("GNN_RCCL_LIB");
        const char *names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        // the copy the process already has (RTLD_NOLOAD), else the first that loads
        for (const char *n : names) { if (n && !api.handle) api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD); }
        for (const char *n : names) { if (n && !api.handle) api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL); }
        if (!api.handle) { api.error = "librccl.so.1 cannot be loaded (set GNN_RCCL_LIB)"; return; }
#define RCCL_SYM(field, name) api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.handle, name)); \
        if (!api.field && api.error.empty()) api.error = std::string("RCCL symbol missing: ") + name
        RCCL_SYM(GetUniqueId, "ncclGetUniqueId"); RCCL_SYM(CommInitRank, "ncclCommInitRank"); RCCL_SYM(CommDestroy, "ncclCommDestroy");
        RCCL_SYM(AllGather, "ncclAllGather"); RCCL_SYM(Send, "ncclSend"); RCCL_SYM(Recv, "ncclRecv");
        RCCL_SYM(GroupStart, "ncclGroupStart"); RCCL_SYM(GroupEnd, "ncclGroupEnd"); RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef RCCL_SYM
    });
    return api;
}

#define RCCL_OK(call) do { const ncclResult_t r_ = (call); if (r_ != ncclSuccess) return fail("RCCL: %s (%s)", rccl_api().GetErrorString ? rccl_api().GetErrorString(r_) : "error", #call); } while (0)

constexpr int SHARD_MAX_CHUNKS = 16;

struct GnnComm {
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0;
    hipStream_t xstream = nullptr;                 // the exchange runs here, next to the compute stream
    hipEvent_t ev_rows[SHARD_MAX_CHUNKS] = {};     // "these rows are written" (compute -> exchange), one per chunk of an iteration
    hipEvent_t ev_landed = nullptr;                // "the exchange of this iteration has landed" (exchange -> compute)
};

}  // namespace

extern "C" {

int gnn_comm_unique_id(void *out128) {
    RcclApi &api = rccl_api();
    if (!api.error.empty()) return fail("%s", api.error.c_str());
    if (!out128) return fail("out128 is NULL");
    ncclUniqueId id;
    RCCL_OK(api.GetUniqueId(&id));
    memcpy(out128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
}

int gnn_comm_destroy(void *comm);

static int comm_build(GnnComm *c, const void *unique_id128) {
    RcclApi &api = rccl_api();
    ncclUniqueId id;
    memcpy(id.internal, unique_id128, NCCL_UNIQUE_ID_BYTES);
    RCCL_OK(api.CommInitRank(&c->comm, c->nranks, id, c->rank));
    HIP_OK(hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking));
    for (int i = 0; i < SHARD_MAX_CHUNKS; ++i) HIP_OK(hipEventCreateWithFlags(&c->ev_rows[i], hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&c->ev_landed, hipEventDisableTiming));
    return 0;
}

int gnn_comm_create(int32_t nranks, int32_t rank, const void *unique_id128, void **comm_out) {
    RcclApi &api = rccl_api();
    if (!api.error.empty()) return fail("%s", api.error.c_str());
    if (nranks < 1 || rank < 0 || rank >= nranks || !unique_id128 || !comm_out) return fail("gnn_comm_create: bad arguments");
    GnnComm *c = new GnnComm;
    c->nranks = nranks; c->rank = rank;
    if (comm_build(c, unique_id128)) {              // (the message is set) - the communicator, stream and events built so far go with it
        (void)gnn_comm_destroy(c);
        return 1;
    }
    *comm_out = c;
    return 0;
}

int gnn_comm_destroy(void *comm) {
    if (!comm) return 0;
    GnnComm *c = static_cast<GnnComm *>(comm);
    if (c->xstream) (void)hipStreamSynchronize(c->xstream);
    if (c->comm && rccl_api().CommDestroy) (void)rccl_api().CommDestroy(c->comm);
    for (int i = 0; i < SHARD_MAX_CHUNKS; ++i) if (c->ev_rows[i]) (void)hipEventDestroy(c->ev_rows[i]);
    if (c->ev_landed) (void)hipEventDestroy(c->ev_landed);
    if (c->xstream) (void)hipStreamDestroy(c->xstream);
    delete c;
    return 0;
}

// rows [lo, hi) of EVERY slice of `buf` to / from everybody on the exchange stream: the all-gather collective (whole slices only) or
// R - 1 point-to-point pairs in one group (staggered peer order: rank r starts with r + 1, as gnnkeras_amd/distributed.py does)
static int shard_exchange(GnnComm *c, const gnn_shard_loop_args_t &s, float *buf, int lo, int hi) {
    if (!c) return 0;
    RcclApi &api = rccl_api();
    const size_t n = (size_t)s.rows_per_slice * s.SP;
    if (s.transport == 0 && lo == 0 && hi == s.rows_per_slice) {
        RCCL_OK(api.AllGather(buf + (size_t)c->rank * n, buf, n, ncclFloat, c->comm, c->xstream));
        return 0;
    }
    const size_t a = (size_t)lo * s.SP, cnt = (size_t)(hi - lo) * s.SP;
    RCCL_OK(api.GroupStart());
    for (int off = 1; off < c->nranks; ++off) {
        const int to = (c->rank + off) % c->nranks, frm = (c->rank - off + c->nranks) % c->nranks;
        RCCL_OK(api.Send(buf + (size_t)c->rank * n + a, cnt, ncclFloat, to, c->comm, c->xstream));
        RCCL_OK(api.Recv(buf + (size_t)frm * n + a, cnt, ncclFloat, frm, c->comm, c->xstream));
    }
    RCCL_OK(api.GroupEnd());
    return 0;
}

static int shard_loop_run(const gnn_shard_loop_args_t *sa);

int gnn_shard_loop(const gnn_shard_loop_args_t *sa) {
    const int rc = shard_loop_run(sa);
    if (rc && sa && sa->loop && sa->comm) {
        // a launch failed with exchanges already queued on the exchange stream: the compute stream waits for them, so that whatever the
        // caller queues next (a recovery, a free) is ordered behind every transfer that still touches the state buffers
        GnnComm *c = static_cast<GnnComm *>(sa->comm);
        if (hipEventRecord(c->ev_landed, c->xstream) == hipSuccess) (void)hipStreamWaitEvent((hipStream_t)sa->loop->stream, c->ev_landed, 0);
    }
    return rc;
}

static int shard_loop_run(const gnn_shard_loop_args_t *sa) {
    if (!sa || !sa->loop) return fail("gnn_shard_loop: args / loop is NULL");
    const gnn_shard_loop_args_t &s = *sa;
    const gnn_loop_args_t &a = *s.loop;
    GnnComm *c = static_cast<GnnComm *>(s.comm);
    if (!s.buf[0] || !s.buf[1]) return fail("gnn_shard_loop: state buffers are NULL");
    if (s.world_size < 1 || s.rank < 0 || s.rank >= s.world_size || s.rows_per_slice < 1 || s.chunk < 0 || s.chunk >= s.rows_per_slice || s.SP < 1)
        return fail("gnn_shard_loop: bad slice geometry");
    if (c && (c->nranks != s.world_size || c->rank != s.rank)) return fail("gnn_shard_loop: the communicator has %d ranks (this is %d), the plan %d (%d)", c->nranks, c->rank, s.world_size, s.rank);
    if (s.world_size > 1 && !c && !s.emulated) return fail("gnn_shard_loop: more than one rank needs a communicator (gnn_comm_create)");
    const bool split = s.adjacency_own && s.adjacency_halo && s.agg_partial;
    int n_chunks = std::max(1, (int)s.n_chunks);
    if (n_chunks > 1 && (!split || !s.node_iota || !s.chunk_begin)) return fail("gnn_shard_loop: chunk launches need the own-range / halo split, node_iota and chunk_begin");
    if (n_chunks > SHARD_MAX_CHUNKS) return fail("gnn_shard_loop: at most %d chunks", SHARD_MAX_CHUNKS);
    const int K = a.max_iteration;
    if (s.first_iteration < 0 || s.first_iteration > K) return fail("gnn_shard_loop: first_iteration %d out of [0, %d]", s.first_iteration, K);
    const int it_end = s.n_iterations >= 0 ? std::min<int>(s.first_iteration + s.n_iterations, K) : K;
    hipStream_t st = (hipStream_t)a.stream;
    const int row_base = s.row_base;
    auto gate_of = [&](int it) { return reinterpret_cast<const int32_t *>(s.buf[it & 1] + (size_t)s.chunk * s.SP); };           // the flag word of slice 0 of the buffer the iteration reads
    auto flag_out_of = [&](int it) { return reinterpret_cast<int32_t *>(s.buf[(it + 1) & 1] + ((size_t)row_base + s.chunk) * s.SP); };
    const int gate_stride = s.rows_per_slice * s.SP;
    const bool comm_on = c != nullptr;              // (a one-rank communicator still goes through RCCL: the all-gather of one slice)
    if (split && s.first_iteration == 0 && it_end > 0) TRY(gnn_shard_partial(&a, s.adjacency_own, s.buf[0], s.agg_partial));      // state_0 is complete on every rank
    for (int it = s.first_iteration; it < it_end; ++it) {
        float *src = s.buf[it & 1], *dst = s.buf[(it + 1) & 1];
        if (!split) {
            TRY(gnn_shard_iteration(&a, src, dst, row_base, gate_of(it), s.world_size, gate_stride, flag_out_of(it), it));
            if (comm_on) {
                HIP_OK(hipEventRecord(c->ev_rows[0], st));
                HIP_OK(hipStreamWaitEvent(c->xstream, c->ev_rows[0], 0));
                TRY(shard_exchange(c, s, dst, 0, s.rows_per_slice));
                HIP_OK(hipEventRecord(c->ev_landed, c->xstream));
                HIP_OK(hipStreamWaitEvent(st, c->ev_landed, 0));
            }
            continue;
        }
        if (n_chunks == 1) {
            TRY(gnn_shard_iteration_split(&a, s.adjacency_halo, s.agg_partial, src, dst, row_base, gate_of(it), s.world_size, gate_stride, flag_out_of(it), it));
            if (comm_on) {
                HIP_OK(hipEventRecord(c->ev_rows[0], st));
                HIP_OK(hipStreamWaitEvent(c->xstream, c->ev_rows[0], 0));
                TRY(shard_exchange(c, s, dst, 0, s.rows_per_slice));
            }
        } else {
            for (int ci = 0; ci < n_chunks; ++ci) {        // chunk ci's rows are on the links while chunk ci + 1 is computed
                const int lo = std::min<int>(s.chunk_begin[ci], a.n_nodes), hi = std::min<int>(s.chunk_begin[ci + 1], a.n_nodes);
                TRY(gnn_shard_iteration_split_rows(&a, s.adjacency_halo, s.agg_partial, src, dst, row_base, gate_of(it), s.world_size, gate_stride,
                                                   flag_out_of(it), it, s.node_iota + std::min(lo, hi), std::max(hi - lo, 0), ci == 0));
                if (comm_on) {
                    HIP_OK(hipEventRecord(c->ev_rows[ci], st));
                    HIP_OK(hipStreamWaitEvent(c->xstream, c->ev_rows[ci], 0));
                    // (every rank sends the NOMINAL range of the chunk; the last one runs through the padding rows and the flag row)
                    TRY(shard_exchange(c, s, dst, s.chunk_begin[ci], ci == n_chunks - 1 ? s.rows_per_slice : s.chunk_begin[ci + 1]));
                }
            }
        }
        if (it + 1 < K) TRY(gnn_shard_partial(&a, s.adjacency_own, dst, s.agg_partial));       // reads only the rows this rank has just written
        if (comm_on) {
            HIP_OK(hipEventRecord(c->ev_landed, c->xstream));
            HIP_OK(hipStreamWaitEvent(st, c->ev_landed, 0));
        }
    }
    return 0;
}

}  // extern "C"
