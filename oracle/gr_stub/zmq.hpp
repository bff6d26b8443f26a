// gr_stub — TEST INFRASTRUCTURE: the slice of cppzmq the reference's gr_mmdvm_sink / gr_mmdvm_source use, as in-memory mailboxes:
// send() appends to `sent`, recv() pops from `inbox` (empty message when there is none).  No ZeroMQ, no sockets.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <deque>
#include <optional>
#include <string>
#include <vector>

#define ZMQ_PUSH 8
#define ZMQ_REQ 3

namespace zmq {

class context_t {
public:
    context_t() {}
    explicit context_t(int) {}
};
class message_t {
public:
    message_t() {}
    explicit message_t(size_t n) : d(n) {}
    void* data() { return d.data(); }
    size_t size() const { return d.size(); }
    std::vector<uint8_t> d;
};
namespace sockopt { struct sndhwm_t {}; struct linger_t {}; static const sndhwm_t sndhwm{}; static const linger_t linger{}; }
enum class send_flags { none = 0, dontwait = 1 };
typedef std::optional<size_t> recv_result_t;
typedef std::optional<size_t> send_result_t;
class socket_t {
public:
    socket_t() {}
    socket_t(context_t&, int) {}
    template <class O> void set(O, int) {}
    void bind(const std::string& a) { address = a; }
    void connect(const std::string& a) { address = a; }
    send_result_t send(message_t& m, send_flags) { sent.push_back(m.d); return m.d.size(); }
    recv_result_t recv(message_t& m)
    {
        if (inbox.empty()) { m.d.clear(); return std::nullopt; }
        m.d = inbox.front();
        inbox.pop_front();
        return m.d.size();
    }
    std::string address;
    std::deque<std::vector<uint8_t>> sent, inbox;
};

}  // namespace zmq
