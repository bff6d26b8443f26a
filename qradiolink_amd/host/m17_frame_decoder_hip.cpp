// m17_frame_decoder_hip.cpp — see the header.  Reference behaviour kept, including its edges: a LICH segment number above 5 marks its
// bit in the (8-bit) segment map like any other, so the map can then no longer reach 0x3F until reset(); the reference would also
// copy such a segment past the end of its 30-byte LSF image -- that write is not reproduced.
#include "m17_frame_decoder_hip.h"

#include <cstring>
#include <stdexcept>
#include <string>

#include <hip/hip_runtime.h>

namespace qrl_host {

static void hchk(hipError_t e, const char* what)
{
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

m17_frame_decoder_hip::m17_frame_decoder_hip(qrl_runtime& rt, int streams, size_t max_frames)
    : d_rt(rt), d_st((size_t)streams), d_cap(max_frames)
{
    if (streams < 1 || max_frames < 1) throw std::invalid_argument("m17_frame_decoder_hip: streams >= 1, max_frames >= 1");
    hchk(hipMalloc(reinterpret_cast<void**>(&d_frames), d_cap * 48), "hipMalloc");
    hchk(hipMalloc(reinterpret_cast<void**>(&d_records), d_cap * QRL_M17_RECORD_BYTES), "hipMalloc");
    d_host.resize(d_cap * QRL_M17_RECORD_BYTES);
}
m17_frame_decoder_hip::~m17_frame_decoder_hip()
{
    if (d_frames) (void)hipFree(d_frames);
    if (d_records) (void)hipFree(d_records);
}
void m17_frame_decoder_hip::reset(int stream) { d_st[(size_t)stream] = state(); }

uint16_t m17_frame_decoder_hip::crc16(const uint8_t* p, size_t n)
{
    uint16_t crc = 0xFFFF;
    for (size_t i = 0; i < n; ++i) {
        crc ^= (uint16_t)(p[i] << 8);
        for (int k = 0; k < 8; ++k) crc = (crc & 0x8000) ? (uint16_t)((crc << 1) ^ 0x5935) : (uint16_t)(crc << 1);
    }
    return crc;
}

std::vector<M17FrameType> m17_frame_decoder_hip::decodeFrames(const uint8_t* frames, const int* stream_of, size_t n)
{
    std::vector<M17FrameType> types(n);
    for (size_t done = 0; done < n;) {
        const size_t m = std::min(n - done, d_cap);
        hchk(hipMemcpy(d_frames, frames + done * 48, m * 48, hipMemcpyHostToDevice), "H2D");
        if (qrl_m17_decode_frames(d_rt.ctx(), nullptr, d_frames, m, d_records) != QRL_OK) throw std::runtime_error(std::string("qrl_m17_decode_frames: ") + qrl_last_error());
        hchk(hipMemcpy(d_host.data(), d_records, m * QRL_M17_RECORD_BYTES, hipMemcpyDeviceToHost), "D2H");   // (synchronises the default stream)
        for (size_t i = 0; i < m; ++i) {
            const uint8_t* r = d_host.data() + i * QRL_M17_RECORD_BYTES;
            state& st = d_st[(size_t)(stream_of ? stream_of[done + i] : 0)];
            types[done + i] = static_cast<M17FrameType>(r[0]);
            if (r[0] == 1) std::memcpy(st.lsf.data(), r + 2, 30);                      // decodeLSF, M17FrameDecoder.cpp:111-117
            if (r[0] == 2) {                                                           // decodeStream, :119-160
                if (r[1]) {
                    const uint8_t seg = r[37];
                    if (seg <= 5) std::memcpy(st.lsf_from_lich.data() + 5 * seg, r + 32, 5);
                    st.segment_map = (uint8_t)(st.segment_map | (1u << (seg & 7u)));   // a corrupt record byte must not shift out of range (the reference indexes a 6-entry map with the same 3-bit field)
                    if (st.segment_map == 0x3F) {
                        const uint16_t crc = crc16(st.lsf_from_lich.data(), 28);
                        if (st.lsf_from_lich[28] == (crc >> 8) && st.lsf_from_lich[29] == (crc & 0xFF)) st.lsf = st.lsf_from_lich;
                        st.segment_map = 0;
                        st.lsf_from_lich.fill(0);
                    }
                }
                std::memcpy(st.stream.data(), r + 2, 18);
            }
        }
        done += m;
    }
    return types;
}

}  // namespace qrl_host
