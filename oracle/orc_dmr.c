/* orc_dmr.c -- CPU oracle (test infrastructure) of the DMR DMO correlator slicer, SURVEY.md 8(a) row a37b.
 * Restates gr_dmr_dmo_sink (reference src/gr/gr_dmr_dmo_sink.cpp:81-204 processSample, :206-322 correlateSync,
 * :324-357 samplesToBits, :360-369 countSyncErrs, :371-380 writeFrame; constants src/DMR/constants.h:8-36,71-88,
 * src/MMDVM/DMRDefines.h:25-28,82-93) and the slot-type decode it calls (src/MMDVM/DMRSlotType2.cpp:215-264: Golay (20,8)
 * through the (19,8) syndrome table).  The block is fed from port 3 of gr_demod_dmr (RRC-filtered discriminator output, 5
 * samples per symbol at 24 ksps, gr_demod_dmr.cpp:94).
 * PINNED: tests/test_ref_blocks.py runs the reference's gr_dmr_dmo_sink.cpp itself (oracle/_ref) and requires identical records.
 * Differences from the reference that are NOT behaviour: the members the reference constructor leaves uninitialised (m_buffer,
 * m_bitBuffer, m_control, m_centre, m_threshold) start at zero here; frames leave as 40-byte records
 * {frame type, FN, colour code, 0, 33 frame bytes, 3 pad bytes} instead of DMRFrame objects (slot number 2, downlink false are constants). */
#include "orc.h"
#include <string.h>

#define DMO_BUF 1440
#define SYM 5
#define FRAME_BYTES 33
#define FRAME_SYMBOLS 132
#define FRAME_SAMPLES 660
#define SYNC_SYMBOLS 24
#define SYNC_SAMPLES 120
#define SLOT_TYPE_SAMPLES 50
#define INFO_SAMPLES 490
#define NOENDPTR 9999

static const int8_t MS_DATA_VALUES[24] = {-3, +3, +3, +3, -3, +3, +3, -3, -3, -3, +3, -3, +3, -3, -3, -3, -3, +3, +3, -3, +3, +3, +3, -3};
static const int8_t MS_VOICE_VALUES[24] = {+3, -3, -3, -3, +3, -3, -3, +3, +3, +3, -3, +3, -3, +3, +3, +3, +3, -3, -3, +3, -3, -3, -3, +3};
static const uint8_t MS_DATA_BYTES[7] = {0x0D, 0x5D, 0x7F, 0x77, 0xFD, 0x75, 0x70};
static const uint8_t MS_VOICE_BYTES[7] = {0x07, 0xF7, 0xD5, 0xDD, 0x57, 0xDF, 0xD0};
static const uint8_t SYNC_BYTES_MASK[7] = {0x0F, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xF0};
static const uint8_t BIT_MASK[8] = {0x80, 0x40, 0x20, 0x10, 0x08, 0x04, 0x02, 0x01};

/* (19,8) shortened Golay code, generator X^11 + X^10 + X^6 + X^5 + X^4 + X^2 + 1 (DMRSlotType2.cpp:215-238) */
uint32_t orc_golay1987_syndrome(uint32_t pattern)
{
    uint32_t aux = 0x40000u;
    if (pattern >= 0x800u) {
        while (pattern & 0xFFFFF800u) {
            while (!(aux & pattern)) aux >>= 1;
            pattern ^= (aux / 0x800u) * 0xC75u;
        }
    }
    return pattern;
}
/* DECODING_TABLE_1987: syndrome -> coset leader.  Construction rule (verified entry by entry against the 2048 literals of
 * DMRSlotType2.cpp:46-205 by tests/test_dmo_sink.py): error patterns of weight 1, 2, .. 5 over the 19 code bits are enumerated
 * in lexicographic order of their bit positions (lowest position outermost) and a pattern is entered only where the entry of its
 * syndrome is still empty.  Weights <= 3 are unique (the code corrects 3 errors); 802 weight-4 and 82 weight-5 leaders are
 * first-come; four syndromes have no pattern of weight <= 5 and stay 0 (left uncorrected), as in the reference's table. */
static void golay_enum(uint32_t* table, int weight, int from, uint32_t e)
{
    if (weight == 0) {
        const uint32_t s = orc_golay1987_syndrome(e);
        if (s && !table[s]) table[s] = e;
        return;
    }
    for (int b = from; b < 19; b++) golay_enum(table, weight - 1, b + 1, e | (1u << b));
}
void orc_golay1987_table(uint32_t* table /* 2048 */)
{
    memset(table, 0, 2048 * sizeof(uint32_t));
    for (int w = 1; w <= 5; w++) golay_enum(table, w, 0, 0);
}
static uint8_t slot_type_decode(const uint32_t* table, const uint8_t* frame)   /* returns (colour code << 4) | data type */
{
    uint8_t st[3];
    st[0] = (uint8_t)(((frame[12] << 2) & 0xFC) | ((frame[13] >> 6) & 0x03));
    st[1] = (uint8_t)(((frame[13] << 2) & 0xC0) | ((frame[19] << 2) & 0x3C) | ((frame[20] >> 6) & 0x03));
    st[2] = (uint8_t)((frame[20] << 2) & 0xF0);
    uint32_t code = ((uint32_t)st[0] << 11) + ((uint32_t)st[1] << 3) + ((uint32_t)st[2] >> 5);
    const uint32_t e = table[orc_golay1987_syndrome(code)];
    if (e) code ^= e;
    return (uint8_t)(code >> 11);
}

static unsigned count_errs(uint32_t v) { return (unsigned)__builtin_popcount(v); }

static void samples_to_bits(const orc_dmo_state* s, unsigned start, unsigned count, uint8_t* buffer, unsigned offset, float centre, float threshold)
{
    for (unsigned i = 0; i < count; i++) {
        const float sample = s->buffer[start] - centre;
        int b0, b1;
        if (sample < -threshold) { b0 = 1; b1 = 1; }
        else if (sample < 0.0f) { b0 = 1; b1 = 0; }
        else if (sample < threshold) { b0 = 0; b1 = 0; }
        else { b0 = 0; b1 = 1; }
        buffer[offset >> 3] = b0 ? (uint8_t)(buffer[offset >> 3] | BIT_MASK[offset & 7]) : (uint8_t)(buffer[offset >> 3] & ~BIT_MASK[offset & 7]);
        offset++;
        buffer[offset >> 3] = b1 ? (uint8_t)(buffer[offset >> 3] | BIT_MASK[offset & 7]) : (uint8_t)(buffer[offset >> 3] & ~BIT_MASK[offset & 7]);
        offset++;
        start += SYM;
        if (start >= DMO_BUF) start -= DMO_BUF;
    }
}

static void dmo_reset(orc_dmo_state* s)
{
    s->syncPtr = 0; s->maxCorr = 0; s->syncCount = 0; s->state = 0; s->startPtr = 0; s->endPtr = NOENDPTR; s->colorCode = 0; s->n = 0;
}
void orc_dmo_init(orc_dmo_state* s)
{
    memset(s, 0, sizeof *s);
    s->endPtr = NOENDPTR;
}

static void correlate_sync(orc_dmo_state* s, int first)
{
    const uint32_t sh = s->bitBuffer[s->bitPtr] & 0x00FFFFFFu;
    const int data = count_errs(sh ^ 0x0076286Eu) <= 2, voice = count_errs(sh ^ 0x0089D791u) <= 2;
    if (!(data || voice)) return;
    unsigned ptr = s->dataPtr + DMO_BUF - SYNC_SAMPLES + SYM;
    if (ptr >= DMO_BUF) ptr -= DMO_BUF;
    float corr = 0.0f, mn = 100.0f, mx = -100.0f;
    unsigned p = ptr;
    for (int i = 0; i < SYNC_SYMBOLS; i++) {
        const float val = s->buffer[p];
        if (val > mx) mx = val;
        if (val < mn) mn = val;
        corr += (float)(data ? MS_DATA_VALUES[i] : MS_VOICE_VALUES[i]) * val;
        p += SYM;
        if (p >= DMO_BUF) p -= DMO_BUF;
    }
    if (!(corr > s->maxCorr)) return;
    const float centre = (mx + mn) / 2.0f;
    const float threshold = (mx - centre) / 2.0f;
    uint8_t sync[7] = {0, 0, 0, 0, 0, 0, 0};
    samples_to_bits(s, ptr, SYNC_SYMBOLS, sync, 4, centre, threshold);
    unsigned errs = 0;
    const uint8_t* want = data ? MS_DATA_BYTES : MS_VOICE_BYTES;
    for (int i = 0; i < 7; i++) errs += count_errs((uint32_t)((sync[i] & SYNC_BYTES_MASK[i]) ^ want[i]));
    if (errs > 3) return;
    if (first) {
        for (int i = 0; i < 4; i++) { s->threshold[i] = threshold; s->centre[i] = centre; }
        s->averagePtr = 0;
    } else {
        s->threshold[s->averagePtr] = threshold;
        s->centre[s->averagePtr] = centre;
        if (++s->averagePtr >= 4) s->averagePtr = 0;
    }
    s->maxCorr = corr;
    s->control = data ? 0x40 : 0x20;
    s->syncPtr = s->dataPtr;
    unsigned sp = s->dataPtr + DMO_BUF - SLOT_TYPE_SAMPLES / 2 - INFO_SAMPLES / 2 - SYNC_SAMPLES;
    if (sp >= DMO_BUF) sp -= DMO_BUF;
    s->startPtr = (uint16_t)sp;
    unsigned ep = s->dataPtr + SLOT_TYPE_SAMPLES / 2 + INFO_SAMPLES / 2 - 1;
    if (ep >= DMO_BUF) ep -= DMO_BUF;
    s->endPtr = (uint16_t)ep;
}

static size_t write_frame(const orc_dmo_state* s, const uint8_t* frame, uint8_t type, uint8_t* out, size_t nout, size_t cap)
{
    if (nout < cap) {
        uint8_t* r = out + 40 * nout;
        r[0] = type; r[1] = s->n; r[2] = s->colorCode; r[3] = 0;
        memcpy(r + 4, frame, FRAME_BYTES);
    }
    return nout + 1;
}

size_t orc_dmo_process(orc_dmo_state* s, const uint32_t* golay, const float* in, size_t n, uint8_t* out, size_t cap)
{
    size_t nout = 0;
    enum { TYPE_DATA = 0, TYPE_VOICE = 1, TYPE_VOICE_SYNC = 2, RECV_NONE = 0, RECV_DATA = 1, RECV_VOICE = 3 };
    for (size_t k = 0; k < n; k++) {
        const float sample = in[k];
        s->buffer[s->dataPtr] = sample;
        s->bitBuffer[s->bitPtr] <<= 1;
        if (sample > 0.0f) s->bitBuffer[s->bitPtr] |= 1u;
        if (s->state == RECV_NONE) correlate_sync(s, 1);
        else {
            unsigned mn = s->syncPtr + DMO_BUF - 1, mx = s->syncPtr + 1;
            if (mn >= DMO_BUF) mn -= DMO_BUF;
            if (mx >= DMO_BUF) mx -= DMO_BUF;
            if (mn < mx) { if (s->dataPtr >= mn && s->dataPtr <= mx) correlate_sync(s, 0); }
            else { if (s->dataPtr >= mn || s->dataPtr <= mx) correlate_sync(s, 0); }
        }
        if (s->dataPtr == s->endPtr) {
            const float centre = (s->centre[0] + s->centre[1] + s->centre[2] + s->centre[3]) / 4.0f;
            const float threshold = (s->threshold[0] + s->threshold[1] + s->threshold[2] + s->threshold[3]) / 4.0f;
            uint8_t frame[FRAME_BYTES];
            memset(frame, 0, sizeof frame);
            unsigned ptr = s->endPtr + DMO_BUF - FRAME_SAMPLES + SYM + 1;
            if (ptr >= DMO_BUF) ptr -= DMO_BUF;
            samples_to_bits(s, ptr, FRAME_SYMBOLS, frame, 0, centre, threshold);
            if (s->control == 0x40) {
                const uint8_t code = slot_type_decode(golay, frame);
                const uint8_t dataType = code & 0x0F;
                s->colorCode = (code >> 4) & 0x0F;
                s->syncCount = 0; s->n = 0;
                switch (dataType) {
                case 0x06: s->state = RECV_DATA; nout = write_frame(s, frame, TYPE_DATA, out, nout, cap); break;            /* DT_DATA_HEADER */
                case 0x07: case 0x08: case 0x0A:                                                                       /* rate 1/2, 3/4, 1 data */
                    if (s->state == RECV_DATA) nout = write_frame(s, frame, TYPE_DATA, out, nout, cap);
                    break;
                case 0x01: case 0x00: s->state = RECV_VOICE; nout = write_frame(s, frame, TYPE_DATA, out, nout, cap); break; /* voice LC / PI header */
                case 0x02:                                                                                             /* terminator with LC */
                    if (s->state == RECV_VOICE) { nout = write_frame(s, frame, TYPE_DATA, out, nout, cap); dmo_reset(s); }
                    break;
                default: nout = write_frame(s, frame, TYPE_DATA, out, nout, cap); dmo_reset(s); break;                   /* CSBK and the rest */
                }
            } else if (s->control == 0x20) {
                s->state = RECV_VOICE; s->syncCount = 0; s->n = 0;
                nout = write_frame(s, frame, TYPE_VOICE_SYNC, out, nout, cap);
            } else {
                if (s->state != RECV_NONE) {
                    s->syncCount++;
                    if (s->syncCount >= 13) dmo_reset(s);
                }
                if (s->state == RECV_VOICE) {
                    if (s->n >= 5) s->n = 0; else ++s->n;
                    nout = write_frame(s, frame, TYPE_VOICE, out, nout, cap);
                } else if (s->state == RECV_DATA) {
                    nout = write_frame(s, frame, TYPE_DATA, out, nout, cap);
                }
            }
            s->maxCorr = 0;
            s->control = 0;
        }
        if (++s->dataPtr >= DMO_BUF) s->dataPtr = 0;
        if (++s->bitPtr >= SYM) s->bitPtr = 0;
    }
    return nout;
}
