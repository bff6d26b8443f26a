// qt_stub — TEST INFRASTRUCTURE.  The sliver of Qt that lets the reference's OWN src/gr_modem.cpp (the gr_modem class: toggleRxMode /
// toggleTxMode tables, demodulate, synchronize, findSync, processReceivedData, frame, transmit, start / endTransmission, ...) compile
// unmodified into oracle/_ref/libqrl_ref.so.  Signals become plain member functions (the shim defines them and records their
// arguments), the project classes gr_modem only talks to (gr_demod_base, gr_mod_base, Settings, Logger, Limits, DMRControl,
// DMRTiming) are shadowed by the stubs under qt_stub/src/.  Written for this repository; no Qt or reference code.
#pragma once
#include <cstdint>
#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#define Q_OBJECT
#define signals public
#define slots
#define emit
#define SIGNAL(x) #x
#define SLOT(x) #x

typedef uint8_t quint8; typedef uint16_t quint16; typedef uint32_t quint32; typedef uint64_t quint64;
typedef int8_t qint8; typedef int16_t qint16; typedef int32_t qint32; typedef int64_t qint64;
typedef unsigned int uint;

class QObject {
public:
    explicit QObject(QObject* = nullptr) {}
    virtual ~QObject() {}
    static bool connect(const void*, const char*, const void*, const char*) { return true; }
    static bool disconnect(const void*, const char*, const void*, const char*) { return true; }
};

struct QChar { explicit QChar(char c) : v((unsigned char)c) {} unsigned unicode() const { return v; } unsigned v; };
struct QRegExp { explicit QRegExp(const char* p) : pattern(p) {} std::string pattern; };

class QString {
public:
    QString() {}
    QString(const char* s) : d(s ? s : "") {}
    QString(const std::string& s) : d(s) {}
    static QString fromStdString(const std::string& s) { return QString(s); }
    static QString fromLocal8Bit(const char* s, int n = -1) { return n < 0 ? QString(s) : QString(std::string(s, (size_t)n)); }   // n bytes, embedded zeros kept
    static QString fromUtf8(const char* s, int n = -1) { return fromLocal8Bit(s, n); }
    std::string toStdString() const { return d; }
    int size() const { return (int)d.size(); }
    int length() const { return (int)d.size(); }
    QString mid(int pos, int n = -1) const { return pos >= (int)d.size() ? QString() : QString(d.substr((size_t)pos, n < 0 ? std::string::npos : (size_t)n)); }
    bool operator==(const QString& o) const { return d == o.d; }
    bool operator!=(const QString& o) const { return d != o.d; }
    bool operator!=(const char* o) const { return d != o; }
    bool operator==(const char* o) const { return d == o; }
    QString left(int n) const { return QString(d.substr(0, (size_t)std::min<int>(n, (int)d.size()))); }
    // the one pattern gr_modem uses: everything that is not a letter a-z A-Z, '/', a digit or white space goes (bytes are taken as
    // Latin-1 here; Qt decodes UTF-8 first: a byte >= 0x80 becomes a character outside the class either way, except exotic Unicode
    // digits / spaces)
    QString& remove(const QRegExp&)
    {
        std::string r;
        for (unsigned char c : d)
            if ((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '/' || (c >= '0' && c <= '9') || c == ' ' || (c >= 9 && c <= 13)) r.push_back((char)c);
        d = r;
        return *this;
    }
    QString& append(const QString& o) { d += o.d; return *this; }
    std::string d;
};
class QStringList : public std::vector<QString> {
public:
    void append(const QString& s) { push_back(s); }
    int length() const { return (int)size(); }
};
class QByteArray {
public:
    QByteArray() {}
    QByteArray(const char* p, int n) : d(p, (size_t)n) {}
    char* data() { return d.empty() ? nullptr : &d[0]; }
    const char* data() const { return d.data(); }
    const char* constData() const { return d.data(); }
    int size() const { return (int)d.size(); }
    int length() const { return (int)d.size(); }
    QByteArray mid(int pos, int n = -1) const
    {
        QByteArray r;
        if (pos < (int)d.size()) r.d = d.substr((size_t)pos, n < 0 ? std::string::npos : (size_t)n);
        return r;
    }
    std::string d;
};
template <class K, class V> class QMap : public std::map<K, V> {};
class QMutex { public: void lock() { m.lock(); } void unlock() { m.unlock(); } std::mutex m; };
struct QDebugSink { template <class T> QDebugSink& operator<<(const T&) { return *this; } };
inline QDebugSink qDebug() { return QDebugSink(); }

#include <QVector>
