/*
 * fo_bits.c -- ORACLE (test infrastructure): MSB-first bit writer/reader and
 * Exp-Golomb codes.  Follows F/rbsp_IO.cpp:100-190 (writer; the reference's
 * 64-bit accumulator is an implementation detail, the emitted bits are plain
 * MSB-first), F/rbsp_IO.cpp:193-325 (reader) and F/expgolomb.cpp.
 */
#include "fo.h"
#include <string.h>

void fo_bw_init(fo_bw *w, uint8_t *buf, size_t cap)
{
    w->buf = buf;
    w->cap = cap;
    w->nbits = 0;
}

void fo_bw_put(fo_bw *w, int n, uint32_t v)
{
    for (int i = n - 1; i >= 0; i--) {
        size_t byte = w->nbits >> 3;
        int bit = 7 - (int)(w->nbits & 7);
        if (byte < w->cap) {
            if (bit == 7) w->buf[byte] = 0;
            w->buf[byte] |= (uint8_t)(((v >> i) & 1u) << bit);
        }
        w->nbits++;
    }
}

/* prefix length of ue(v): F/expgolomb.cpp:8-45 */
static int ue_prefix(unsigned v)
{
    int p = 0;
    unsigned x = v + 1;
    while (x > 1) {
        x >>= 1;
        p++;
    }
    return p;
}

int fo_ue_len(unsigned v) { return 2 * ue_prefix(v) + 1; }

void fo_bw_ue(fo_bw *w, unsigned v)
{
    int p = ue_prefix(v);
    fo_bw_put(w, p, 0);
    fo_bw_put(w, 1, 1);
    if (p) fo_bw_put(w, p, v + 1 - (1u << p));
}

unsigned fo_se_to_ue(int v) { return v <= 0 ? (unsigned)(-v) * 2u : (unsigned)v * 2u - 1u; }

void fo_bw_se(fo_bw *w, int v) { fo_bw_ue(w, fo_se_to_ue(v)); }

/* rbsp_trailing_bits: a 1 then zero-pad to the byte (F/rbsp_encoding.cpp:108-117) */
size_t fo_bw_trailing(fo_bw *w)
{
    fo_bw_put(w, 1, 1);
    while (w->nbits & 7) fo_bw_put(w, 1, 0);
    return w->nbits >> 3;
}

void fo_br_init(fo_br *r, const uint8_t *buf, size_t size)
{
    r->buf = buf;
    r->size = size;
    r->pos = 0;
}

static unsigned rd_byte(const fo_br *r, size_t i) { return i < r->size ? r->buf[i] : 0u; }

unsigned fo_br_bit(fo_br *r)
{
    unsigned b = (rd_byte(r, r->pos >> 3) >> (7 - (r->pos & 7))) & 1u;
    r->pos++;
    return b;
}

unsigned fo_br_bits(fo_br *r, int n)
{
    unsigned v = 0;
    for (int i = 0; i < n; i++) v = (v << 1) | fo_br_bit(r);
    return v;
}

unsigned fo_br_peek24(fo_br *r)
{
    fo_br t = *r;
    return fo_br_bits(&t, 24);
}

/* F/expgolomb.cpp:122-140: leading zeros are searched in a 24-bit window only */
unsigned fo_br_ue(fo_br *r)
{
    unsigned w = fo_br_peek24(r);
    int i;
    for (i = 0; i < 24; i++)
        if (w & (0x800000u >> i)) break;
    r->pos += (size_t)i + 1;
    unsigned s = fo_br_bits(r, i);
    return (1u << i) - 1u + s;
}

int fo_br_se(fo_br *r)
{
    int v = (int)fo_br_ue(r);
    return (v % 2) ? (v + 1) / 2 : -v / 2;
}

/* F/expgolomb.cpp:156-178, quirks kept */
unsigned fo_br_te(fo_br *r)
{
    unsigned zc = 0;
    while (fo_br_bit(r) == 0) zc++;
    if (zc == 0) return 0;
    unsigned x = fo_br_bits(r, (int)zc);
    if (x > 1) return (1u << zc) - 1u + x;
    return !fo_br_bit(r);
}

/* F/rbsp_IO.cpp:193-196: RBSP_current_byte < RBSP_total_size - 1 */
int fo_br_more(fo_br *r) { return (r->pos >> 3) + 1 < r->size; }
