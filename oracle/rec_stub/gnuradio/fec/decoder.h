// rec_stub: see cc_decoder.h
#pragma once
#include <gnuradio/fec/cc_decoder.h>
