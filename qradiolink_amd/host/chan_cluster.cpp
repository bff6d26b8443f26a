// chan_cluster.cpp — see chan_cluster.h.  Host C++ only: the device work is behind the C ABI (qrl_chan_*), the collective is RCCL.
#include "chan_cluster.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace qrl_host {

namespace {
void chk(int status, const char* what)
{
    if (status != QRL_OK) throw std::runtime_error(std::string(what) + ": " + qrl_strerror(status) + " (" + qrl_last_error() + ")");
}
void hchk(hipError_t e, const char* what)
{
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
void nchk(ncclResult_t r, const char* what)
{
    if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(r));
}
}  // namespace

void self_exchange::all_to_all(const void* send, void* recv, size_t bytes_per_peer, void* stream)
{
    hchk(hipMemcpyAsync(recv, send, bytes_per_peer, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)), "self_exchange: hipMemcpyAsync");
}

void rccl_exchange::unique_id(unsigned char out[kIdBytes])
{
    static_assert(sizeof(ncclUniqueId) <= kIdBytes, "ncclUniqueId larger than the id buffer");
    ncclUniqueId id;
    nchk(ncclGetUniqueId(&id), "ncclGetUniqueId");
    std::memset(out, 0, kIdBytes);
    std::memcpy(out, &id, sizeof id);
}
rccl_exchange::rccl_exchange(int world, int rank, const unsigned char id[kIdBytes]) : d_world(world), d_rank(rank)
{
    if (world < 1 || rank < 0 || rank >= world) throw std::invalid_argument("rccl_exchange: bad world / rank");
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof uid);
    ncclComm_t comm = nullptr;
    nchk(ncclCommInitRank(&comm, world, uid, rank), "ncclCommInitRank");
    d_comm = comm;
}
rccl_exchange::~rccl_exchange()
{
    if (d_comm) (void)ncclCommDestroy(static_cast<ncclComm_t>(d_comm));
}
void rccl_exchange::all_to_all(const void* send, void* recv, size_t bytes_per_peer, void* stream)
{
    // equal blocks to and from every peer: one ncclAllToAll (xGMI is point-to-point: world - 1 links carry bytes_per_peer each)
    if (bytes_per_peer % 4) throw std::invalid_argument("rccl_exchange: blocks are cf32 items");
    nchk(ncclAllToAll(send, recv, bytes_per_peer / 4, ncclFloat32, static_cast<ncclComm_t>(d_comm), static_cast<hipStream_t>(stream)), "ncclAllToAll");
}

void callback_exchange::all_to_all(const void* send, void* recv, size_t bytes_per_peer, void* stream)
{
    if (!d_fn || d_fn(d_user, send, recv, bytes_per_peer, stream) != 0) throw std::runtime_error("callback_exchange: the transport callback failed");
}

local_group::local_group(int world) : d_world(world), d_m(nullptr)
{
    if (world < 1) throw std::invalid_argument("local_group: world must be >= 1");
    d_m = new member[world];
    for (int r = 0; r < world; ++r) {
        hipEvent_t e;
        hchk(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate"); d_m[r].ev_ready = e;
        hchk(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate"); d_m[r].ev_done = e;
    }
}
local_group::~local_group()
{
    for (int r = 0; r < d_world; ++r) {
        if (d_m[r].ev_ready) (void)hipEventDestroy(static_cast<hipEvent_t>(d_m[r].ev_ready));
        if (d_m[r].ev_done) (void)hipEventDestroy(static_cast<hipEvent_t>(d_m[r].ev_done));
    }
    delete[] d_m;
}
void local_group::post(int rank, const void* send, void* recv, size_t bytes_per_peer, void* stream)
{
    if (rank < 0 || rank >= d_world) throw std::invalid_argument("local_group: bad rank");
    member& me = d_m[rank];
    if (me.posted) throw std::logic_error("local_group: a member posted twice in one round");
    if (d_posted && bytes_per_peer != d_nbytes) throw std::invalid_argument("local_group: the members of a round must post equal block sizes");
    d_nbytes = bytes_per_peer;
    me.send = send; me.recv = recv; me.stream = stream; me.posted = true;
    hchk(hipEventRecord(static_cast<hipEvent_t>(me.ev_ready), static_cast<hipStream_t>(stream)), "hipEventRecord");   // send_rank is ready (its stream waited for the channelizer)
    if (++d_posted < d_world) return;
    // the round is complete: every member's receive side on its own stream
    for (int r = 0; r < d_world; ++r) {
        hipStream_t sr = static_cast<hipStream_t>(d_m[r].stream);
        for (int s = 0; s < d_world; ++s)
            if (s != r) hchk(hipStreamWaitEvent(sr, static_cast<hipEvent_t>(d_m[s].ev_ready), 0), "hipStreamWaitEvent");
        if (!d_skip)
            for (int s = 0; s < d_world; ++s)
                hchk(hipMemcpyAsync(static_cast<char*>(d_m[r].recv) + (size_t)s * d_nbytes, static_cast<const char*>(d_m[s].send) + (size_t)r * d_nbytes, d_nbytes,
                                    hipMemcpyDeviceToDevice, sr), "local_group: hipMemcpyAsync");
        hchk(hipEventRecord(static_cast<hipEvent_t>(d_m[r].ev_done), sr), "hipEventRecord");
        d_bytes += (uint64_t)d_world * d_nbytes;
    }
    // ... and every member's send side: its buffer is free when every rank has copied its block out of it
    for (int s = 0; s < d_world; ++s) {
        hipStream_t ss = static_cast<hipStream_t>(d_m[s].stream);
        for (int r = 0; r < d_world; ++r)
            if (r != s) hchk(hipStreamWaitEvent(ss, static_cast<hipEvent_t>(d_m[r].ev_done), 0), "hipStreamWaitEvent");
        d_m[s].posted = false;
    }
    d_posted = 0;
}

chan_cluster::chan_cluster(qrl_ctx* ctx, chan_exchange& ex, int num_channels, int streams_local, size_t max_chunk)
    : d_ex(ex), d_M(num_channels), d_bl(streams_local), d_per(num_channels / (ex.world() > 0 ? ex.world() : 1)), d_n1max(max_chunk / (size_t)num_channels)
{
    const int W = ex.world();
    if (W < 1 || num_channels < 2 || num_channels % W || streams_local < 1 || d_n1max < 1)
        throw std::invalid_argument("chan_cluster: the number of ranks must divide the channels; streams_local >= 1; max_chunk >= num_channels");
    const char* keep = std::getenv("QRL_CLUSTER_COPY_AT_ONE_RANK");
    d_inplace = W == 1 && !(keep && keep[0] == '1');
    try {
        // Priorities, and not only for the hardware queues (streams of one priority share a few of them): when a per-channel kernel ends, the next one
        // on its stream and a channelizer two steps ahead become ready together, and whichever is dispatched first fills the CUs -- the persistent
        // channelizer workgroups must NOT be that one (the per-channel kernels are the step's critical path; the channelizer has two steps of slack).
        // Channelizer: lowest priority; exchange and per-channel handle: normal (they are serial anyway); the handle's symbol synchroniser: highest.
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        hipStream_t fs;
        hchk(hipStreamCreateWithPriority(&fs, hipStreamNonBlocking, prio_lo), "hipStreamCreate");
        d_fs = fs;
        qrl_chan_config c{};
        c.num_channels = num_channels; c.batch = streams_local; c.max_chunk = max_chunk; c.form = 0; c.hip_stream = d_fs;
        chk(qrl_chan_create(ctx, &c, &d_front), "qrl_chan_create (channelizer)");
        qrl_chan_config t{};
        t.num_channels = 1; t.batch = streams_local * W * d_per; t.max_chunk = d_n1max; t.form = 3;
        chk(qrl_chan_create(ctx, &t, &d_tail), "qrl_chan_create (per-channel chains)");
        const size_t items = (size_t)W * d_bl * d_per * d_n1max;
        for (int k = 0; k < kSlots; ++k) {
            hchk(hipMalloc(reinterpret_cast<void**>(&d_send[k]), items * 2 * sizeof(float)), "hipMalloc");
            if (!d_inplace) hchk(hipMalloc(reinterpret_cast<void**>(&d_recv[k]), items * 2 * sizeof(float)), "hipMalloc");
        }
        hipStream_t xs;
        hchk(hipStreamCreateWithFlags(&xs, hipStreamNonBlocking), "hipStreamCreate");   // normal priority: shares queues with the per-channel handle's stream at worst, and the exchange of a step and its per-channel kernels are serial
        d_xs = xs;
        for (int k = 0; k < kSlots; ++k) {
            hipEvent_t e;
            hchk(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate"); d_ev_sent[k] = e;
            hchk(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate"); d_ev_read[k] = e;
        }
    } catch (...) {
        release();   // a constructor that throws has no destructor run
        throw;
    }
}
void chan_cluster::release()
{
    if (d_front) qrl_chan_destroy(d_front);
    if (d_tail) qrl_chan_destroy(d_tail);
    d_front = d_tail = nullptr;
    if (d_xs) { (void)hipStreamSynchronize(static_cast<hipStream_t>(d_xs)); (void)hipStreamDestroy(static_cast<hipStream_t>(d_xs)); d_xs = nullptr; }
    if (d_fs) { (void)hipStreamSynchronize(static_cast<hipStream_t>(d_fs)); (void)hipStreamDestroy(static_cast<hipStream_t>(d_fs)); d_fs = nullptr; }
    for (int k = 0; k < kSlots; ++k) {
        if (d_ev_sent[k]) (void)hipEventDestroy(static_cast<hipEvent_t>(d_ev_sent[k]));
        if (d_ev_read[k]) (void)hipEventDestroy(static_cast<hipEvent_t>(d_ev_read[k]));
        d_ev_sent[k] = d_ev_read[k] = nullptr;
        if (d_send[k]) (void)hipFree(d_send[k]);
        if (d_recv[k]) (void)hipFree(d_recv[k]);
        d_send[k] = d_recv[k] = nullptr;
    }
}
chan_cluster::~chan_cluster() { release(); }
// Ordering of step k (slot = k mod 3), all on the device:
//   channelizer k   waits for exchange k - 3 (ev_sent[slot]: the last reader of send[slot])            -- NOT for exchanges k - 1, k - 2: it runs under them
//   exchange k      waits for channelizer k (qrl_chan_stream_wait) and for the per-channel kernels of step k - 3 (ev_read[slot]: since round 5
//                   they read recv[slot] in place -- there is no copy into a ring any more -- so they are its last readers)
//   per-channel k   waits for exchange k (qrl_chan_wait_for)
void chan_cluster::channelize(const float* iq, size_t stride, size_t n)
{
    if (n % (size_t)d_M || n / (size_t)d_M > d_n1max) throw std::invalid_argument("chan_cluster: n must be a multiple of num_channels and <= max_chunk");
    d_cur = (int)(d_k++ % (unsigned)kSlots);
    d_n1 = n / (size_t)d_M;
    if (d_inplace) {   // one rank: the per-channel kernels of step k - 3 were the readers of send[cur]
        if (d_read_valid[d_cur])
            hchk(hipStreamWaitEvent(static_cast<hipStream_t>(qrl_chan_stream(d_front)), static_cast<hipEvent_t>(d_ev_read[d_cur]), 0), "hipStreamWaitEvent");
    } else if (d_sent_valid[d_cur])
        hchk(hipStreamWaitEvent(static_cast<hipStream_t>(qrl_chan_stream(d_front)), static_cast<hipEvent_t>(d_ev_sent[d_cur]), 0), "hipStreamWaitEvent");
    chk(qrl_chan_channelize(d_front, iq, stride, n, d_send[d_cur], d_n1max, d_ex.world()), "qrl_chan_channelize");
}
void chan_cluster::exchange_begin()
{
    // ONE RANK: every channel is this rank's own, the by-destination layout the channelizer wrote IS what the per-channel handle reads -- the
    // all-to-all of one rank would be a 1 GB device copy per step (0.96 ms as rcclGenericKernel at the bench shape) that moves nothing anywhere.
    // QRL_CLUSTER_COPY_AT_ONE_RANK=1 keeps the collective (what a one-GPU box can exercise of the RCCL transport; bench.py --cluster-copy).
    if (d_inplace) return;
    chk(qrl_chan_stream_wait(d_front, d_xs), "qrl_chan_stream_wait");    // the collective runs behind the channelizer ...
    if (d_read_valid[d_cur])                                              // ... and behind the last readers of recv[cur]
        hchk(hipStreamWaitEvent(static_cast<hipStream_t>(d_xs), static_cast<hipEvent_t>(d_ev_read[d_cur]), 0), "hipStreamWaitEvent");
    d_ex.all_to_all(d_send[d_cur], d_recv[d_cur], (size_t)d_bl * d_per * d_n1max * 2 * sizeof(float), d_xs);
}
void chan_cluster::exchange_end()
{
    if (d_inplace) return;
    hchk(hipEventRecord(static_cast<hipEvent_t>(d_ev_sent[d_cur]), static_cast<hipStream_t>(d_xs)), "hipEventRecord");
    d_sent_valid[d_cur] = true;
}
void chan_cluster::process_channels(int16_t* out, size_t out_cap, uint32_t* counts)
{
    if (d_inplace) chk(qrl_chan_stream_wait(d_front, qrl_chan_stream(d_tail)), "qrl_chan_stream_wait");   // behind the channelizer
    else chk(qrl_chan_wait_for(d_tail, d_xs), "qrl_chan_wait_for");      // the per-channel chains run behind the collective
    chk(qrl_chan_process_channels(d_tail, d_inplace ? d_send[d_cur] : d_recv[d_cur], d_n1max, d_n1, out, out_cap, counts), "qrl_chan_process_channels");
    // (the handle's own stream: the kernels that read recv[cur]; its symbol-sync stream reads the handle's rings only)
    hchk(hipEventRecord(static_cast<hipEvent_t>(d_ev_read[d_cur]), static_cast<hipStream_t>(qrl_chan_stream(d_tail))), "hipEventRecord");
    d_read_valid[d_cur] = true;
}
void chan_cluster::step(const float* iq, size_t stride, size_t n, int16_t* out, size_t out_cap, uint32_t* counts)
{
    channelize(iq, stride, n);
    exchange();
    process_channels(out, out_cap, counts);
}
void chan_cluster::sync()
{
    chk(qrl_chan_sync(d_front), "qrl_chan_sync");
    hchk(hipStreamSynchronize(static_cast<hipStream_t>(d_xs)), "hipStreamSynchronize");
    chk(qrl_chan_sync(d_tail), "qrl_chan_sync");
}

}  // namespace qrl_host

// ---- C ABI ------------------------------------------------------------------------------------------------------------------------
struct qrl_exchange { qrl_host::chan_exchange* ex; };
struct qrl_cluster { qrl_host::chan_cluster* cl; };
struct qrl_exchange_group { qrl_host::local_group* g; };
static thread_local std::string g_cluster_error;
template <class F> static int guarded(F&& f)
{
    try { f(); return QRL_OK; }
    catch (const std::invalid_argument& e) { g_cluster_error = e.what(); return QRL_ERR_ARG; }
    catch (const std::exception& e) { g_cluster_error = e.what(); return QRL_ERR_HIP; }
}
extern "C" {
const char* qrl_cluster_last_error(void) { return g_cluster_error.c_str(); }
int qrl_exchange_unique_id(unsigned char* out128) { return out128 ? guarded([&] { qrl_host::rccl_exchange::unique_id(out128); }) : QRL_ERR_ARG; }
int qrl_exchange_create_rccl(int world, int rank, const unsigned char* id128, qrl_exchange** out)
{
    if (!id128 || !out) return QRL_ERR_ARG;
    return guarded([&] { *out = new qrl_exchange{new qrl_host::rccl_exchange(world, rank, id128)}; });
}
int qrl_exchange_create_self(qrl_exchange** out) { return out ? guarded([&] { *out = new qrl_exchange{new qrl_host::self_exchange}; }) : QRL_ERR_ARG; }
int qrl_exchange_create_callback(int world, int rank, qrl_host::chan_exchange_fn fn, void* user, qrl_exchange** out)
{
    if (!fn || !out || world < 1 || rank < 0 || rank >= world) return QRL_ERR_ARG;
    return guarded([&] { *out = new qrl_exchange{new qrl_host::callback_exchange(world, rank, fn, user)}; });
}
int qrl_exchange_all_to_all(qrl_exchange* ex, const void* send, void* recv, size_t bytes_per_peer, void* hip_stream)
{
    if (!ex || !send || !recv) return QRL_ERR_ARG;
    return guarded([&] { ex->ex->all_to_all(send, recv, bytes_per_peer, hip_stream); });
}
void qrl_exchange_destroy(qrl_exchange* ex) { if (ex) { delete ex->ex; delete ex; } }
int qrl_cluster_create(qrl_ctx* ctx, qrl_exchange* ex, int num_channels, int streams_local, size_t max_chunk, qrl_cluster** out)
{
    if (!ctx || !ex || !out) return QRL_ERR_ARG;
    return guarded([&] { *out = new qrl_cluster{new qrl_host::chan_cluster(ctx, *ex->ex, num_channels, streams_local, max_chunk)}; });
}
void qrl_cluster_destroy(qrl_cluster* c) { if (c) { delete c->cl; delete c; } }
qrl_chan* qrl_cluster_front(qrl_cluster* c) { return c ? c->cl->front() : nullptr; }
qrl_chan* qrl_cluster_tail(qrl_cluster* c) { return c ? c->cl->tail() : nullptr; }
int qrl_cluster_rows(qrl_cluster* c) { return c ? c->cl->rows() : 0; }
int qrl_cluster_step(qrl_cluster* c, const float* iq, size_t stride, size_t n, int16_t* out, size_t out_cap, uint32_t* counts)
{
    return c ? guarded([&] { c->cl->step(iq, stride, n, out, out_cap, counts); }) : QRL_ERR_ARG;
}
int qrl_cluster_channelize(qrl_cluster* c, const float* iq, size_t stride, size_t n) { return c ? guarded([&] { c->cl->channelize(iq, stride, n); }) : QRL_ERR_ARG; }
int qrl_cluster_exchange(qrl_cluster* c) { return c ? guarded([&] { c->cl->exchange(); }) : QRL_ERR_ARG; }
int qrl_cluster_exchange_begin(qrl_cluster* c) { return c ? guarded([&] { c->cl->exchange_begin(); }) : QRL_ERR_ARG; }
int qrl_cluster_exchange_end(qrl_cluster* c) { return c ? guarded([&] { c->cl->exchange_end(); }) : QRL_ERR_ARG; }
int qrl_exchange_group_create(int world, qrl_exchange_group** out) { return out ? guarded([&] { *out = new qrl_exchange_group{new qrl_host::local_group(world)}; }) : QRL_ERR_ARG; }
int qrl_exchange_group_member(qrl_exchange_group* g, int rank, qrl_exchange** out)
{
    if (!g || !out || rank < 0 || rank >= g->g->world()) return QRL_ERR_ARG;
    return guarded([&] { *out = new qrl_exchange{new qrl_host::local_group_exchange(*g->g, rank)}; });
}
unsigned long long qrl_exchange_group_bytes_moved(const qrl_exchange_group* g) { return g ? (unsigned long long)g->g->bytes_moved() : 0ull; }
int qrl_exchange_group_skip_copies(qrl_exchange_group* g, int skip) { if (!g) return QRL_ERR_ARG; g->g->skip_copies(skip != 0); return QRL_OK; }
void qrl_exchange_group_destroy(qrl_exchange_group* g) { if (g) { delete g->g; delete g; } }
int qrl_cluster_process_channels(qrl_cluster* c, int16_t* out, size_t out_cap, uint32_t* counts)
{
    return c ? guarded([&] { c->cl->process_channels(out, out_cap, counts); }) : QRL_ERR_ARG;
}
int qrl_cluster_sync(qrl_cluster* c) { return c ? guarded([&] { c->cl->sync(); }) : QRL_ERR_ARG; }
}
