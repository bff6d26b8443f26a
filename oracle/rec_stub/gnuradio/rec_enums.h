// rec_stub — TEST INFRASTRUCTURE: the enumerations and small value types the hier blocks name (values as in GNU Radio 3.10 where the
// log prints them; only their identity matters for the comparison)
#pragma once
#include <gnuradio/recording.h>
namespace gr {
namespace fft { namespace window { enum win_type { WIN_NONE = -1, WIN_HAMMING = 0, WIN_HANN = 1, WIN_BLACKMAN = 2, WIN_RECTANGULAR = 3, WIN_KAISER = 4,
                                                   WIN_BLACKMAN_hARRIS = 5, WIN_BLACKMAN_HARRIS = 5, WIN_BARTLETT = 6, WIN_FLATTOP = 7 }; } }
namespace digital {
enum ted_type { TED_NONE = -1, TED_MUELLER_AND_MULLER = 0, TED_MOD_MUELLER_AND_MULLER = 1, TED_ZERO_CROSSING = 2, TED_GARDNER = 4, TED_EARLY_LATE = 5,
                TED_DANDREA_AND_MENGALI_GEN_MSK = 6, TED_MENGALI_AND_DANDREA_GMSK = 7, TED_SIGNAL_TIMES_SLOPE_ML = 8, TED_SIGNUM_TIMES_SLOPE_ML = 9 };
enum ir_type { IR_NONE = -1, IR_MMSE_8TAP = 0, IR_PFB_NO_MF = 1, IR_PFB_MF = 2 };
}
namespace analog { enum gr_waveform_t { GR_CONST_WAVE = 100, GR_SIN_WAVE, GR_COS_WAVE, GR_SQR_WAVE, GR_TRI_WAVE, GR_SAW_WAVE }; }
}  // namespace gr
