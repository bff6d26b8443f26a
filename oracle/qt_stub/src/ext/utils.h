// qt_stub shadow of src/ext/utils.h (gr_modem.cpp uses nothing of it)
#pragma once
#include "qt_stub.h"
