// qt_stub shadow of src/DMR/dmrcontrol.h: the DMR protocol stack is out of scope; gr_modem only forwards to it
#pragma once
#include <vector>
#include "src/DMR/dmrframe.h"
namespace DMR_MODE { enum DMR_MODE { DMR_MODE_REPEATER = 0, DMR_MODE_DMO = 1, DMR_MODE_TRUNKED = 2 }; }
class DMRControl {
public:
    void stopVoiceTX() {}
    void initVoiceTX() {}
    bool getVoiceHeader(std::vector<DMRFrame>&) { return false; }
    bool getTxAudio(DMRFrame&) { return false; }
    bool getTXStatus() { return false; }
    bool getStartCSBK(std::vector<DMRFrame>&) { return false; }
    uint8_t addTxAudio(unsigned char*) { return 0; }
    void addFrames(std::vector<DMRFrame>& f) { frames_in += (int)f.size(); }
    int frames_in = 0;
};
