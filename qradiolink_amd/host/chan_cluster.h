// chan_cluster.h — the multi-GPU host path of the multi-carrier MMDVM receiver (BASELINE config 4), in C++ above the C ABI.
//
// Reference: one process, one channelizer feeding per-channel chains (src/gr/gr_demod_mmdvm_multi2.cpp:98-135), every channel with
// its own sink socket (src/gr/gr_mmdvm_sink.cpp:77-173).  Here (PFB form) one process per GPU:
//     every rank channelizes ITS B / N wideband streams            qrl_chan_channelize   (all 64 channels, rows grouped by destination)
//     ONE all-to-all per step moves each channel's 25 ksps samples to the rank that owns the channel      chan_exchange::all_to_all
//     the owner runs the per-channel chains of its 64 / N channels of EVERY stream                        qrl_chan_process_channels
// ordered on the device (events per buffer slot + qrl_chan_stream_wait / qrl_chan_wait_for around the exchange stream), three send / receive
// buffer sets so that the channelizers of steps k + 1 and k + 2 run while step k's collective and per-channel chains are still in flight; the per-channel
// kernel reads the receive buffer in place (round 5: no copy into the handle's rings).  No host synchronisation.
//
// The transport is an interface: rccl_exchange (ncclAllToAll over xGMI, the production transport), self_exchange (one rank: a device
// copy), callback_exchange (a C function: the CPU tests run torch.distributed / gloo behind it, the single-device emulation a
// permutation copy).  bench.py --config c4 --gpus N, tests/test_sharding.py and tests/test_gpu_sharding.py all go through
// chan_exchange::all_to_all -- one code path.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

#include "../../include/qrl_hip.h"

namespace qrl_host {

class chan_exchange {
public:
    virtual ~chan_exchange() {}
    virtual int world() const = 0;
    virtual int rank() const = 0;
    // send = world blocks of bytes_per_peer (block d goes to rank d), recv = world blocks (block s came from rank s); both device
    // memory for the device transports; enqueued on `stream` (a hipStream_t), no host synchronisation
    virtual void all_to_all(const void* send, void* recv, size_t bytes_per_peer, void* stream) = 0;
};

// one rank: recv = send (hipMemcpyAsync device to device on the stream)
class self_exchange : public chan_exchange {
public:
    int world() const override { return 1; }
    int rank() const override { return 0; }
    void all_to_all(const void* send, void* recv, size_t bytes_per_peer, void* stream) override;
};

// RCCL: ncclAllToAll on the given stream.  unique_id = the 128 bytes of ncclGetUniqueId made by rank 0 (rccl_exchange::unique_id) and
// handed to every rank by whatever launcher started the processes (bench.py: a torch.distributed broadcast).
class rccl_exchange : public chan_exchange {
public:
    static constexpr size_t kIdBytes = 128;
    static void unique_id(unsigned char out[kIdBytes]);
    rccl_exchange(int world, int rank, const unsigned char id[kIdBytes]);   // the calling thread's current HIP device
    ~rccl_exchange() override;
    int world() const override { return d_world; }
    int rank() const override { return d_rank; }
    void all_to_all(const void* send, void* recv, size_t bytes_per_peer, void* stream) override;
private:
    int d_world, d_rank; void* d_comm = nullptr;   // ncclComm_t
};

// a C function does the exchange: int fn(user, send, recv, bytes_per_peer, stream), 0 = ok
typedef int (*chan_exchange_fn)(void* user, const void* send, void* recv, size_t bytes_per_peer, void* stream);
class callback_exchange : public chan_exchange {
public:
    callback_exchange(int world, int rank, chan_exchange_fn fn, void* user) : d_world(world), d_rank(rank), d_fn(fn), d_user(user) {}
    int world() const override { return d_world; }
    int rank() const override { return d_rank; }
    void all_to_all(const void* send, void* recv, size_t bytes_per_peer, void* stream) override;
private:
    int d_world, d_rank; chan_exchange_fn d_fn; void* d_user;
};

// N ranks of ONE process on ONE device (round 6: the 8-rank SHAPE on a one-GPU box, tools/c4_emulated_ranks.py and tests/test_gpu_sharding.py): a real
// all-to-all among the members' buffers with an all-to-all's dependency structure, device copies instead of xGMI.  Every member's all_to_all() records
// "my send buffer is ready" on its stream and registers its buffers; the LAST member of a round then enqueues, on every member r's stream: wait for every
// source's send buffer, copy block r of every rank's send buffer into block s of recv_r, record "recv_r written"; and on every member s's stream a wait for
// all of those -- the last readers of send_s.  Members must call in rounds (each member once per round, any order): the emulation's host loop does
// exchange_begin() for every rank and only then exchange_end() (chan_cluster), so that a rank's "exchange done" event is recorded behind the copies.
class local_group {
public:
    explicit local_group(int world);
    ~local_group();
    int world() const { return d_world; }
    uint64_t bytes_moved() const { return d_bytes; }    // device-to-device bytes enqueued so far (world^2 x bytes_per_peer per round)
    void skip_copies(bool v) { d_skip = v; }            // timing experiment: the dependency structure without the data movement
private:
    friend class local_group_exchange;
    void post(int rank, const void* send, void* recv, size_t bytes_per_peer, void* stream);
    struct member { const void* send = nullptr; void* recv = nullptr; void* stream = nullptr; void* ev_ready = nullptr; void* ev_done = nullptr; bool posted = false; };
    int d_world; member* d_m; int d_posted = 0; size_t d_nbytes = 0; uint64_t d_bytes = 0; bool d_skip = false;
};
class local_group_exchange : public chan_exchange {
public:
    local_group_exchange(local_group& g, int rank) : d_g(g), d_rank(rank) {}
    int world() const override { return d_g.world(); }
    int rank() const override { return d_rank; }
    void all_to_all(const void* send, void* recv, size_t bytes_per_peer, void* stream) override { d_g.post(d_rank, send, recv, bytes_per_peer, stream); }
private:
    local_group& d_g; int d_rank;
};

// One rank of the channel-sharded receiver: `streams_local` wideband streams in, the rank's 64 / world channels of ALL
// streams_local * world streams out (row = (source rank * streams_local + stream) * channels_per_rank + local channel).
class chan_cluster {
public:
    chan_cluster(qrl_ctx* ctx, chan_exchange& ex, int num_channels, int streams_local, size_t max_chunk);
    ~chan_cluster();
    // one step: n wideband samples (multiple of num_channels) of every local stream, device cf32 iq[b * stride + i];
    // out[row * out_cap + k] device int16 @ 24 ksps, counts[row]
    void step(const float* iq, size_t stride, size_t n, int16_t* out, size_t out_cap, uint32_t* counts);
    // the three phases of step() on their own (a single-process emulation of N ranks runs every rank's phase before the next)
    void channelize(const float* iq, size_t stride, size_t n);
    void exchange() { exchange_begin(); exchange_end(); }
    // the two halves of exchange(): begin = order the exchange stream behind the channelizer and the last readers of the receive buffer, hand the buffers to
    // the transport; end = record "exchange done" on the exchange stream.  One process that emulates several ranks (local_group) calls begin for every
    // rank before the first end: the group's copies are enqueued by the last begin.
    void exchange_begin();
    void exchange_end();
    void process_channels(int16_t* out, size_t out_cap, uint32_t* counts);
    void sync();
    qrl_chan* front() const { return d_front; }     // the PFB handle (form 0): options, profiling
    qrl_chan* tail() const { return d_tail; }       // the per-channel handle (form 3): RSSI / 4FSK outputs are set on this one
    int rows() const { return d_bl * d_ex.world() * d_per; }
    int channels_per_rank() const { return d_per; }
    size_t bytes_per_link_per_step(size_t n) const { return (size_t)d_bl * d_per * (n / d_M) * 8; }
private:
    chan_exchange& d_ex;
    qrl_chan *d_front = nullptr, *d_tail = nullptr;
    int d_M, d_bl, d_per; size_t d_n1max, d_n1 = 0;
    static constexpr int kSlots = 3;                // buffer sets: the channelizer may run two steps ahead of the per-channel kernels
    float *d_send[kSlots] = {nullptr, nullptr, nullptr}, *d_recv[kSlots] = {nullptr, nullptr, nullptr};
    void* d_xs = nullptr;                           // hipStream_t of the exchange
    void* d_fs = nullptr;                           // hipStream_t of the channelizer handle (lowest priority)
    // hipEvent_t per buffer slot: exchange k done (send[slot] read, recv[slot] written); the per-channel kernels of step k queued so far done (recv[slot] read)
    void* d_ev_sent[kSlots] = {nullptr, nullptr, nullptr}; void* d_ev_read[kSlots] = {nullptr, nullptr, nullptr};
    bool d_sent_valid[kSlots] = {false, false, false}, d_read_valid[kSlots] = {false, false, false};
    unsigned d_k = 0; int d_cur = 0;
    bool d_inplace = false;                         // one rank: no exchange, the per-channel handle reads send[] in place
    void release();
};

}  // namespace qrl_host

// ---- C ABI of the same objects (ctypes: bench.py, tests) -----------------------------------------------------------------------
extern "C" {
typedef struct qrl_exchange qrl_exchange;
typedef struct qrl_cluster qrl_cluster;
int qrl_exchange_unique_id(unsigned char* out128);
int qrl_exchange_create_rccl(int world, int rank, const unsigned char* id128, qrl_exchange** out);
int qrl_exchange_create_self(qrl_exchange** out);
int qrl_exchange_create_callback(int world, int rank, qrl_host::chan_exchange_fn fn, void* user, qrl_exchange** out);
int qrl_exchange_all_to_all(qrl_exchange* ex, const void* send, void* recv, size_t bytes_per_peer, void* hip_stream);
void qrl_exchange_destroy(qrl_exchange* ex);
int qrl_cluster_create(qrl_ctx* ctx, qrl_exchange* ex, int num_channels, int streams_local, size_t max_chunk, qrl_cluster** out);
void qrl_cluster_destroy(qrl_cluster* c);
qrl_chan* qrl_cluster_front(qrl_cluster* c);
qrl_chan* qrl_cluster_tail(qrl_cluster* c);
int qrl_cluster_rows(qrl_cluster* c);
int qrl_cluster_step(qrl_cluster* c, const float* iq, size_t stride, size_t n, int16_t* out, size_t out_cap, uint32_t* counts);
int qrl_cluster_channelize(qrl_cluster* c, const float* iq, size_t stride, size_t n);
int qrl_cluster_exchange(qrl_cluster* c);
int qrl_cluster_exchange_begin(qrl_cluster* c);
int qrl_cluster_exchange_end(qrl_cluster* c);
typedef struct qrl_exchange_group qrl_exchange_group;
int qrl_exchange_group_create(int world, qrl_exchange_group** out);                               /* qrl_host::local_group */
int qrl_exchange_group_member(qrl_exchange_group* g, int rank, qrl_exchange** out);              /* a local_group_exchange of that group */
unsigned long long qrl_exchange_group_bytes_moved(const qrl_exchange_group* g);
int qrl_exchange_group_skip_copies(qrl_exchange_group* g, int skip);
void qrl_exchange_group_destroy(qrl_exchange_group* g);
int qrl_cluster_process_channels(qrl_cluster* c, int16_t* out, size_t out_cap, uint32_t* counts);
int qrl_cluster_sync(qrl_cluster* c);
const char* qrl_cluster_last_error(void);
}
