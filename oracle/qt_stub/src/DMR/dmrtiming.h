// qt_stub shadow of src/DMR/dmrtiming.h
#pragma once
#include <QObject>
#include "src/settings.h"
class DMRTiming : public QObject {
public:
    explicit DMRTiming(const Settings*) {}
    void set_tx_time(bool) {}
    bool timing_recent(int = 0) { return false; }
    void set_slot_times(uint64_t) {}
};
