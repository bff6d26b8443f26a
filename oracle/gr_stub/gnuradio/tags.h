// gr_stub: see block.h
#pragma once
#include <gnuradio/block.h>
