// gr_stub — TEST INFRASTRUCTURE: the three pmt calls the reference's rssi_tag_block makes (see gnuradio/block.h in this directory)
#pragma once
#include <memory>
#include <string>
namespace pmt {
struct pmt_value { std::string sym; float f = 0.0f; uint64_t u = 0; };
typedef std::shared_ptr<pmt_value> pmt_t;
inline pmt_t string_to_symbol(const std::string& s) { auto p = std::make_shared<pmt_value>(); p->sym = s; return p; }
inline pmt_t from_float(float v) { auto p = std::make_shared<pmt_value>(); p->f = v; return p; }
inline float to_float(const pmt_t& p) { return p->f; }
inline pmt_t from_uint64(uint64_t v) { auto p = std::make_shared<pmt_value>(); p->u = v; return p; }
inline uint64_t to_uint64(const pmt_t& p) { return p->u; }
}  // namespace pmt
