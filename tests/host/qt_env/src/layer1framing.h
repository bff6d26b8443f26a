// TEST INFRASTRUCTURE: stand-in for the reference's src/layer1framing.h in the test build of qradiolink_amd/host/qt/gr_modem.* (a maintainer's
// build takes the real header from the QRadioLink tree).  Only the frame-type constants the class declaration uses as default arguments: they
// are the sync words of the air interface (the same values as qrl_host::frame_type, qradiolink_amd/host/gr_modem_hip.h).
#pragma once
enum frame_type {
    FrameTypeNone = 0x00, FrameTypeVoice = 0xED89, FrameTypeVoice2 = 0xED89, FrameTypeVoice1 = 0xB5, FrameTypeText = 0x89EDAA, FrameTypeIP = 0xDE98AA,
    FrameTypeVideo = 0x98DEAA, FrameTypeSync = 0xCC, FrameTypeCallsign = 0x8CC8DD, FrameTypeProto = 0xED77AA, FrameTypeEnd = 0x4C8A2B,
    FrameTypeM17Stream = 0xFF5D, FrameTypeM17LSF = 0x55F7, FrameTypeM17EOT = 0x555D555D,
};
