// gr_stub — TEST INFRASTRUCTURE: the three pmt calls the reference's rssi_tag_block makes (see gnuradio/block.h in this directory)
#pragma once
#include <cstdint>
#include <memory>
#include <string>
namespace pmt {
struct pmt_value;
typedef std::shared_ptr<pmt_value> pmt_t;
struct pmt_value { std::string sym; float f = 0.0f; uint64_t u = 0; double d = 0.0; pmt_t t0, t1; };
inline pmt_t string_to_symbol(const std::string& s) { auto p = std::make_shared<pmt_value>(); p->sym = s; return p; }
inline pmt_t from_float(float v) { auto p = std::make_shared<pmt_value>(); p->f = v; return p; }
inline float to_float(const pmt_t& p) { return p->f; }
inline pmt_t from_uint64(uint64_t v) { auto p = std::make_shared<pmt_value>(); p->u = v; return p; }
inline uint64_t to_uint64(const pmt_t& p) { return p->u; }
inline pmt_t from_double(double v) { auto p = std::make_shared<pmt_value>(); p->d = v; return p; }
inline double to_double(const pmt_t& p) { return p->d; }
inline pmt_t from_long(long v) { auto p = std::make_shared<pmt_value>(); p->u = (uint64_t)v; return p; }
inline pmt_t make_tuple(const pmt_t& a, const pmt_t& b) { auto p = std::make_shared<pmt_value>(); p->t0 = a; p->t1 = b; return p; }
inline pmt_t tuple_ref(const pmt_t& p, int k) { return k == 0 ? p->t0 : p->t1; }
}  // namespace pmt
