// rec_stub — TEST INFRASTRUCTURE: gr::fec::code::cc_decoder / cc_encoder objects and the gr::fec::decoder / encoder deployments
#pragma once
#include <gnuradio/recording.h>
enum cc_mode_t { CC_STREAMING = 0, CC_TERMINATED, CC_TRUNCATED, CC_TAILBITING };
namespace gr {
namespace fec {
class generic_decoder : public gr::basic_block { public: typedef std::shared_ptr<generic_decoder> sptr; };
class generic_encoder : public gr::basic_block { public: typedef std::shared_ptr<generic_encoder> sptr; };
namespace code {
class cc_decoder : public generic_decoder {
public:
    typedef std::shared_ptr<generic_decoder> sptr;
    template <class... A> static sptr make(const A&... a) { sptr p(new cc_decoder); p->id = gr::rec::add("fec::code::cc_decoder", a...); return p; }
};
class cc_encoder : public generic_encoder {
public:
    typedef std::shared_ptr<generic_encoder> sptr;
    template <class... A> static sptr make(const A&... a) { sptr p(new cc_encoder); p->id = gr::rec::add("fec::code::cc_encoder", a...); return p; }
};
}  // namespace code
}  // namespace fec
}  // namespace gr
GR_REC_BLOCK(fec, decoder)
GR_REC_BLOCK(fec, encoder)
