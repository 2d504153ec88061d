// fer_cavlc_dev.h -- device bit writer and CAVLC block coder shared by the entropy pass
// (fer_cavlc.hip) and the intra mode decision, whose coded_mb_size needs exact bit counts
// (F/rbsp_encoding.cpp:330-488).  WRITE = false only counts bits, like the reference's
// residual_block_cavlc_size (F/residual.cpp:673-957).
#pragma once
#include "fer_dev.h"

struct BitW {
    uint32_t *buf;       // stream RBSP words
    size_t cap_words;
    unsigned long long acc;
    int nacc;            // valid low bits of acc
    size_t word;         // next word index
    bool first;
    unsigned bits;       // total bits produced by this writer
};

template <bool WRITE>
__device__ __forceinline__ void bw_init(BitW &w, uint32_t *buf, size_t cap_words, size_t bitpos)
{
    w.buf = buf;
    w.cap_words = cap_words;
    w.acc = 0;
    w.nacc = (int)(bitpos & 31);  // leading bits of the first word belong to the predecessor
    w.word = bitpos >> 5;
    w.first = true;
    w.bits = 0;
}

template <bool WRITE>
__device__ __forceinline__ void bw_put(BitW &w, int n, unsigned v)
{
    w.bits += (unsigned)n;
    if (!WRITE || n == 0) return;
    w.acc = (w.acc << n) | (unsigned long long)v;
    w.nacc += n;
    if (w.nacc >= 32) {
        uint32_t word = (uint32_t)(w.acc >> (w.nacc - 32));
        w.nacc -= 32;
        w.acc &= (1ull << w.nacc) - 1ull;
        if (w.word < w.cap_words) {
            if (w.first)
                atomicOr(&w.buf[w.word], __builtin_bswap32(word));
            else
                w.buf[w.word] = __builtin_bswap32(word);
        }
        w.first = false;
        w.word++;
    }
}

template <bool WRITE>
__device__ __forceinline__ void bw_flush(BitW &w)
{
    if (!WRITE || w.nacc == 0) return;
    uint32_t word = (uint32_t)(w.acc << (32 - w.nacc));
    if (w.word < w.cap_words) atomicOr(&w.buf[w.word], __builtin_bswap32(word));
}

template <bool WRITE>
__device__ __forceinline__ void bw_ue(BitW &w, unsigned v)
{
    int p = 31 - __clz((int)(v + 1));
    bw_put<WRITE>(w, p, 0);
    bw_put<WRITE>(w, 1, 1);
    bw_put<WRITE>(w, p, v + 1 - (1u << p));
}
template <bool WRITE>
__device__ __forceinline__ void bw_se(BitW &w, int v)
{
    bw_ue<WRITE>(w, v <= 0 ? (unsigned)(-v) * 2u : (unsigned)v * 2u - 1u);
}

// nC of F/residual.cpp:424-538.  blk: luma 0..15, chroma 0..3 (+ plane)
__device__ inline int cavlc_nC(const FerDev &d, int s, int mb, bool luma, int blk, int plane)
{
    const int *mbt = d.mb_type + (size_t)s * d.nmb;
    const uint8_t *cbp = d.cbp + (size_t)s * d.nmb * 2;
    const uint8_t *tc = d.tc + (size_t)s * d.nmb * 24;
    int mbA, mbB, bA, bB;
    if (luma) {
        bool edgeA = blk == 0 || blk == 2 || blk == 8 || blk == 10;
        bool edgeB = blk == 0 || blk == 1 || blk == 4 || blk == 5;
        mbA = edgeA ? ((mb % d.mbw == 0) ? -1 : mb - 1) : mb;
        mbB = edgeB ? ((mb < d.mbw) ? -1 : mb - d.mbw) : mb;
        bA = c_nbA[blk];
        bB = c_nbB[blk];
    } else {
        bool edgeA = blk == 0 || blk == 2, edgeB = blk < 2;
        mbA = edgeA ? ((mb % d.mbw == 0) ? -1 : mb - 1) : mb;
        mbB = edgeB ? ((mb < d.mbw) ? -1 : mb - d.mbw) : mb;
        bA = c_nbcA[blk];
        bB = c_nbcB[blk];
    }
    int nA = 0, nB = 0;
    if (mbA >= 0) {
        bool zero = luma ? ((cbp[mbA * 2] & (1 << (bA / 4))) == 0) : ((cbp[mbA * 2 + 1] & 2) == 0);
        if (!(mbt[mbA] == FER_P_SKIP || zero)) nA = luma ? tc[mbA * 24 + bA] : tc[mbA * 24 + 16 + plane * 4 + bA];
    }
    if (mbB >= 0) {
        bool zero = luma ? ((cbp[mbB * 2] & (1 << (bB / 4))) == 0) : ((cbp[mbB * 2 + 1] & 2) == 0);
        if (!(mbt[mbB] == FER_P_SKIP || zero)) nB = luma ? tc[mbB * 24 + bB] : tc[mbB * 24 + 16 + plane * 4 + bB];
    }
    if (mbA >= 0 && mbB >= 0) return (nA + nB + 1) >> 1;
    if (mbA >= 0) return nA;
    if (mbB >= 0) return nB;
    return 0;
}

// residual_block_cavlc_write / _size, F/residual.cpp:374-666 / :673-957.  Everything is derived from the bit mask
// of the non-zero positions; coefficients are read back by index from st[k * stride] (LDS), so there are no
// indexed private arrays (scratch) and no chains of dependent loads.
template <bool WRITE>
__device__ inline void cavlc_block_core(BitW &w, unsigned nz, int maxNumCoeff, int nC, const int16_t *st, int stride);

// coefficients already in LDS, contiguous
template <bool WRITE>
__device__ inline void cavlc_block(BitW &w, const int16_t *coef, int maxNumCoeff, int nC)
{
    unsigned nz = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) nz |= (i < maxNumCoeff && coef[i] != 0 ? 1u : 0u) << i;
    cavlc_block_core<WRITE>(w, nz, maxNumCoeff, nC, coef, 1);
}
// coefficients in global memory (the entropy pass, one thread per macroblock): staged once in LDS,
// one column per thread
template <bool WRITE>
__device__ inline void cavlc_block_staged(BitW &w, const int16_t *__restrict__ coef, int maxNumCoeff, int nC, int16_t *st,
                                          int stride)
{
    unsigned nz = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        int v = i < maxNumCoeff ? coef[i] : 0;
        st[i * stride] = (int16_t)v;
        nz |= (v != 0 ? 1u : 0u) << i;
    }
    cavlc_block_core<WRITE>(w, nz, maxNumCoeff, nC, st, stride);
}

template <bool WRITE>
__device__ inline void cavlc_block_core(BitW &w, unsigned nz, int maxNumCoeff, int nC, const int16_t *st, int stride)
{
    const int TotalCoeff = __popc(nz);
    int TrailingOnes = 0, total_zeros = 0;
    if (TotalCoeff) {
        total_zeros = (32 - __clz((int)nz)) - TotalCoeff;
        unsigned m = nz;
        while (m && TrailingOnes < 3) {  // leading run of +-1 from the high-frequency end
            int i = 31 - __clz((int)m);
            int v = st[i * stride];
            if (v != 1 && v != -1) break;
            TrailingOnes++;
            m &= ~(1u << i);
        }
    }
    int len;
    unsigned code;
    if (nC == -1) {
        len = c_ctdc_len[TrailingOnes][TotalCoeff];
        code = c_ctdc_code[TrailingOnes][TotalCoeff];
    } else if (nC >= 8) {
        len = 6;
        code = TotalCoeff == 0 ? 3u : (unsigned)(((TotalCoeff - 1) << 2) | TrailingOnes);
    } else {
        int cls = nC <= 1 ? 0 : (nC <= 3 ? 1 : 2);
        len = c_ct_len[cls][TrailingOnes][TotalCoeff];
        code = c_ct_code[cls][TrailingOnes][TotalCoeff];
    }
    bw_put<WRITE>(w, len, code);
    if (TotalCoeff == 0) return;
    int suffixLength = (TotalCoeff > 10 && TrailingOnes < 3) ? 1 : 0;
    unsigned m = nz;
    for (int i = 0; i < TotalCoeff; i++) {
        const int pos = 31 - __clz((int)m);
        m &= ~(1u << pos);
        const int lev = st[pos * stride];
        if (i < TrailingOnes) {
            bw_put<WRITE>(w, 1, (unsigned)((1 - lev) >> 1));
        } else {
            int levelCode = lev < 0 ? -(lev * 2) - 1 : (lev * 2) - 2;
            if (i == TrailingOnes && TrailingOnes < 3) levelCode -= 2;
            // level_prefix / level_suffix (closed form of F/residual_tables.cpp:940-1010)
            int prefix, ss;
            unsigned suf;
            if (suffixLength == 0) {
                if (levelCode < 14) {
                    prefix = levelCode;
                    ss = 0;
                    suf = 0;
                } else if (levelCode < 30) {
                    prefix = 14;
                    ss = 4;
                    suf = (unsigned)(levelCode - 14);
                } else {
                    prefix = 15;
                    ss = 12;
                    suf = (unsigned)(levelCode - 30) & 0xfffu;  // (levels beyond the escape range, |level| > 2063: see DESIGN.md section 4)
                }
            } else if (levelCode < (15 << suffixLength)) {
                prefix = levelCode >> suffixLength;
                ss = suffixLength;
                suf = (unsigned)(levelCode & ((1 << suffixLength) - 1));
            } else {
                prefix = 15;
                ss = 12;
                suf = (unsigned)(levelCode - (15 << suffixLength)) & 0xfffu;
            }
            bw_put<WRITE>(w, prefix, 0);
            bw_put<WRITE>(w, 1, 1);
            if (suffixLength > 0 || prefix >= 14) bw_put<WRITE>(w, ss, suf);
            if (suffixLength == 0) suffixLength = 1;
            if (iabs(lev) > (3 << (suffixLength - 1)) && suffixLength < 6) suffixLength++;
        }
    }
    int zerosLeft = 0;
    if (TotalCoeff < maxNumCoeff) {
        if (nC != -1)
            bw_put<WRITE>(w, c_tz_len[TotalCoeff - 1][total_zeros], c_tz_code[TotalCoeff - 1][total_zeros]);
        else
            bw_put<WRITE>(w, c_tzdc_len[TotalCoeff - 1][total_zeros], c_tzdc_code[TotalCoeff - 1][total_zeros]);
        zerosLeft = total_zeros;
    }
    m = nz;
    for (int j = 0; j < TotalCoeff - 1; j++) {
        const int pos = 31 - __clz((int)m);
        m &= ~(1u << pos);
        const int run = pos - (32 - __clz((int)m));  // zeros between this coefficient and the next lower one
        if (zerosLeft > 0) {
            if (zerosLeft > 6) {
                if (run < 7) {
                    bw_put<WRITE>(w, 3, (unsigned)(7 - run));
                } else {
                    bw_put<WRITE>(w, run - 4, 0);
                    bw_put<WRITE>(w, 1, 1);
                }
            } else {
                bw_put<WRITE>(w, c_rb_len[zerosLeft - 1][run], c_rb_code[zerosLeft - 1][run]);
            }
        }
        zerosLeft -= run;
    }
}

