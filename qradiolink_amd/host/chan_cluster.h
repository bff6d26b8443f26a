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
    void exchange();
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
int qrl_cluster_process_channels(qrl_cluster* c, int16_t* out, size_t out_cap, uint32_t* counts);
int qrl_cluster_sync(qrl_cluster* c);
const char* qrl_cluster_last_error(void);
}
