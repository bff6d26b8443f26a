// m17_frame_decoder_hip.h — M17FrameDecoder (reference src/M17/M17/M17FrameDecoder.hpp:51-146, .cpp:36-160) for N radios over the C ABI:
// the per-frame FEC (decorrelator, de-interleaver, sync classification, Viterbi, Golay) runs on the device for a whole batch of frames
// (qrl_m17_decode_frames); what stays on the host is the per-stream state the reference keeps between frames -- the latest LSF, the
// latest stream frame and the reassembly of the LSF from six LICH segments with its CRC check (.cpp:130-147).
#pragma once
#include <array>
#include <cstdint>
#include <vector>

#include "gr_hip_blocks.h"

namespace qrl_host {

enum class M17FrameType : uint8_t { PREAMBLE = 0, LINK_SETUP = 1, STREAM = 2, PACKET = 3, UNKNOWN = 4 };

class m17_frame_decoder_hip {
public:
    m17_frame_decoder_hip(qrl_runtime& rt, int streams, size_t max_frames = 4096);
    ~m17_frame_decoder_hip();
    void reset(int stream = 0);                                                  // M17FrameDecoder::reset
    // n frames of 48 bytes (sync word included), frame i belongs to radio stream_of[i] (nullptr: all to stream 0); frames of one
    // stream are applied in the order given.  Returns the frame types (M17FrameDecoder::decodeFrame's return value).
    std::vector<M17FrameType> decodeFrames(const uint8_t* frames, const int* stream_of, size_t n);
    M17FrameType decodeFrame(const std::array<uint8_t, 48>& frame, int stream = 0) { return decodeFrames(frame.data(), &stream, 1)[0]; }
    const std::array<uint8_t, 30>& getLsf(int stream = 0) const { return d_st[stream].lsf; }                  // raw M17LinkSetupFrame bytes
    const std::array<uint8_t, 18>& getStreamFrame(int stream = 0) const { return d_st[stream].stream; }       // raw M17StreamFrame bytes
    static uint16_t crc16(const uint8_t* p, size_t n);                           // M17LinkSetupFrame::crc16 (polynomial 0x5935)

private:
    struct state { std::array<uint8_t, 30> lsf{}, lsf_from_lich{}; std::array<uint8_t, 18> stream{}; uint8_t segment_map = 0; };
    qrl_runtime& d_rt;
    std::vector<state> d_st;
    size_t d_cap;
    uint8_t *d_frames = nullptr, *d_records = nullptr;   // device
    std::vector<uint8_t> d_host;
};

}  // namespace qrl_host
