// qt_stub shadow of src/logger.h
#pragma once
#include <QString>
class Logger {
public:
    enum { LogLevelInfo, LogLevelDebug, LogLevelWarning, LogLevelCritical, LogLevelFatal };
    void log(int, QString) {}
};
