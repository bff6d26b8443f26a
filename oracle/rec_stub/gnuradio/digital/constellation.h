// rec_stub — TEST INFRASTRUCTURE: the constellation objects the hier blocks create (recorded like blocks; base() hands the object on)
#pragma once
#include <gnuradio/recording.h>
namespace gr {
namespace digital {
class constellation : public gr::basic_block { public: typedef std::shared_ptr<constellation> sptr; };
typedef std::shared_ptr<constellation> constellation_sptr;
#define GR_REC_CONSTELLATION(CLS)                                                                                                           \
    class CLS : public constellation, public std::enable_shared_from_this<CLS> {                                                            \
    public:                                                                                                                                 \
        typedef std::shared_ptr<CLS> sptr;                                                                                                  \
        template <class... A> static sptr make(const A&... a) { sptr p(new CLS); p->id = gr::rec::add("digital::" #CLS, a...); return p; }  \
        constellation_sptr base() { return this->shared_from_this(); }                                                                      \
    };
GR_REC_CONSTELLATION(constellation_bpsk)
GR_REC_CONSTELLATION(constellation_qpsk)
GR_REC_CONSTELLATION(constellation_dqpsk)
GR_REC_CONSTELLATION(constellation_rect)
GR_REC_CONSTELLATION(constellation_expl_rect)
GR_REC_CONSTELLATION(constellation_calcdist)
}  // namespace digital
}  // namespace gr
