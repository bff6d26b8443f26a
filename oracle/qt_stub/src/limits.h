// qt_stub shadow of src/limits.h
#pragma once
#include <cstdint>
class Limits { public: bool checkLimit(int64_t) { return true; } };
