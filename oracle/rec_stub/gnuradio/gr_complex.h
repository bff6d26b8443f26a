// rec_stub: see gnuradio/recording.h
#pragma once
#include <gnuradio/recording.h>
#include <gnuradio/rec_enums.h>
