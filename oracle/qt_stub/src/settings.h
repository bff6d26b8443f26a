// qt_stub shadow of src/settings.h: the fields gr_modem reads (reference src/settings.h:97-136)
#pragma once
#include <QString>
class Settings {
public:
    int burst_ip_modem = 0, tx_band_limits = 0, burst_delay_msec = 60, m17_can_tx = 0, m17_can_rx = 0, m17_decode_all_can = 1,
        m17_destination_type = 0, dmr_mode = 0, dmr_timeslot = 1;
    QString m17_src, m17_dest;
};
