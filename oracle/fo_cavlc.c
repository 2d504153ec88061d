/*
 * fo_cavlc.c -- ORACLE (test infrastructure): rows a12, a13 and the parse half
 * of a19.  CAVLC residual block writer / sizer / parser with the reference's
 * nC derivation, residual_write block order and the coded_mb_size estimator.
 * Reference: F/residual.cpp (UTF-16), F/rbsp_encoding.cpp:330-488,
 * F/residual_tables.cpp:940-1010 (level code tables).
 */
#include "fo.h"
#include <stdlib.h>
#include <string.h>

int fo_pred_class(const fo_ctx *c, int mb_type);

static const int nbr4[16][2] = {{5, 10}, {0, 11}, {7, 0},  {2, 1},  {1, 14}, {4, 15}, {3, 4},   {6, 5},
                                {13, 2}, {8, 3},  {15, 8}, {10, 9}, {9, 6},  {12, 7}, {11, 12}, {14, 13}};
static const int nbrc[4][2] = {{1, 2}, {0, 3}, {3, 0}, {2, 1}};

/* nC of F/residual.cpp:424-538.  kind: 0 I16 DC, 1 I16 AC, 2 LumaLevel, 3 chroma DC, 4 chroma AC */
int fo_cavlc_nC(fo_ctx *c, int kind, int blk, int iCbCr)
{
    if (kind == 3) return -1;
    int luma = kind <= 2;
    int mbA, mbB, bA, bB;
    if (luma) {
        if (kind == 0) blk = 0;
        if (blk == 0 || blk == 2 || blk == 8 || blk == 10) {
            if (c->cur % c->mbw == 0) {
                mbA = -1;
                bA = -1;
            } else {
                mbA = c->cur - 1;
                bA = nbr4[blk][0];
            }
        } else {
            mbA = c->cur;
            bA = nbr4[blk][0];
        }
        if (blk == 0 || blk == 1 || blk == 4 || blk == 5) {
            if (c->cur < c->mbw) {
                mbB = -1;
                bB = -1;
            } else {
                mbB = c->cur - c->mbw;
                bB = nbr4[blk][1];
            }
        } else {
            mbB = c->cur;
            bB = nbr4[blk][1];
        }
    } else {
        if (blk == 0 || blk == 2) {
            if (c->cur % c->mbw == 0) {
                mbA = -1;
                bA = -1;
            } else {
                mbA = c->cur - 1;
                bA = nbrc[blk][0];
            }
        } else {
            mbA = c->cur;
            bA = nbrc[blk][0];
        }
        if (blk < 2) {
            if (c->cur < c->mbw) {
                mbB = -1;
                bB = -1;
            } else {
                mbB = c->cur - c->mbw;
                bB = nbrc[blk][1];
            }
        } else {
            mbB = c->cur;
            bB = nbrc[blk][1];
        }
    }
    int availA = 1, availB = 1, nA = 0, nB = 0;
    if (mbA < 0) {
        availA = 0;
    } else {
        int zero = luma ? ((c->cbp_l[mbA] & (1 << (bA / 4))) == 0) : ((c->cbp_c[mbA] & 2) == 0);
        if (c->mb_type[mbA] == FO_P_SKIP || zero)
            nA = 0;
        else
            nA = luma ? c->tc_l[mbA][bA] : c->tc_c[mbA][iCbCr][bA];
    }
    if (mbB < 0) {
        availB = 0;
    } else {
        int zero = luma ? ((c->cbp_l[mbB] & (1 << (bB / 4))) == 0) : ((c->cbp_c[mbB] & 2) == 0);
        if (c->mb_type[mbB] == FO_P_SKIP || zero)
            nB = 0;
        else
            nB = luma ? c->tc_l[mbB][bB] : c->tc_c[mbB][iCbCr][bB];
    }
    if (availA && availB) return (nA + nB + 1) >> 1;
    if (availA) return nA;
    if (availB) return nB;
    return 0;
}

/* level_prefix / level_suffix of one levelCode at a given suffixLength: the
 * closed form of the table built at F/residual_tables.cpp:940-1010. */
static void level_code_bits(int levelCode, int suffixLength, int *prefix, int *sufSize, unsigned *suffix)
{
    if (suffixLength == 0) {
        if (levelCode < 14) {
            *prefix = levelCode;
            *sufSize = 0;
            *suffix = 0;
        } else if (levelCode < 30) {
            *prefix = 14;
            *sufSize = 4;
            *suffix = (unsigned)(levelCode - 14);
        } else {
            *prefix = 15;
            *sufSize = 12;
            *suffix = (unsigned)(levelCode - 30);
        }
    } else {
        if (levelCode < (15 << suffixLength)) {
            *prefix = levelCode >> suffixLength;
            *sufSize = suffixLength;
            *suffix = (unsigned)(levelCode & ((1 << suffixLength) - 1));
        } else {
            *prefix = 15;
            *sufSize = 12;
            *suffix = (unsigned)(levelCode - (15 << suffixLength));
        }
    }
}

static inline void put(fo_bw *w, int n, unsigned v)
{
    if (w) fo_bw_put(w, n, v);
}

/* F/residual.cpp:374-666 (write) and :673-957 (size): identical control flow,
 * so one routine serves both; w == NULL only counts. */
unsigned fo_cavlc_encode_block(fo_bw *w, const int *coef, int maxNumCoeff, int nC, int *totalcoeff)
{
    int level[16], run[16];
    int TotalCoeff = 0, TrailingOnes = 0, total_zeros = 0, only_ones = 1;
    unsigned bits = 0;
    for (int i = maxNumCoeff - 1; i >= 0; i--) {
        if (coef[i] != 0) {
            run[TotalCoeff] = 0;
            for (int j = i - 1; j >= 0; j--) {
                if (coef[j] == 0)
                    run[TotalCoeff]++;
                else
                    break;
            }
            if ((coef[i] == 1 || coef[i] == -1) && TrailingOnes < 3 && only_ones)
                TrailingOnes++;
            else
                only_ones = 0;
            level[TotalCoeff++] = coef[i];
        } else if (TotalCoeff > 0) {
            total_zeros++;
        }
    }
    if (totalcoeff) *totalcoeff = TotalCoeff;
    int cls = (nC == -1) ? 4 : (nC <= 1 ? 0 : (nC <= 3 ? 1 : (nC < 8 ? 2 : 3)));
    int len;
    unsigned code;
    fo_coeff_token(cls, TotalCoeff, TrailingOnes, &len, &code);
    put(w, len, code);
    bits += (unsigned)len;
    if (TotalCoeff == 0) return bits;

    int suffixLength = (TotalCoeff > 10 && TrailingOnes < 3) ? 1 : 0;
    for (int i = 0; i < TotalCoeff; i++) {
        if (i < TrailingOnes) {
            put(w, 1, (unsigned)((1 - level[i]) >> 1));
            bits++;
        } else {
            int levelCode = level[i] < 0 ? -(level[i] * 2) - 1 : (level[i] * 2) - 2;
            if (i == TrailingOnes && TrailingOnes < 3) levelCode -= 2;
            int prefix, ss;
            unsigned suf;
            level_code_bits(levelCode, suffixLength, &prefix, &ss, &suf);
            put(w, prefix, 0);
            put(w, 1, 1);
            bits += (unsigned)prefix + 1;
            if (suffixLength > 0 || prefix >= 14) {
                put(w, ss, suf);
                bits += (unsigned)ss;
            }
            if (suffixLength == 0) suffixLength = 1;
            int a = level[i] < 0 ? -level[i] : level[i];
            if (a > (3 << (suffixLength - 1)) && suffixLength < 6) suffixLength++;
        }
    }
    int zerosLeft = 0;
    if (TotalCoeff < maxNumCoeff) { /* endIdx - startIdx + 1 == maxNumCoeff on this path */
        if (nC != -1) {
            put(w, fo_tz_len[TotalCoeff - 1][total_zeros], fo_tz_code[TotalCoeff - 1][total_zeros]);
            bits += fo_tz_len[TotalCoeff - 1][total_zeros];
        } else {
            put(w, fo_tzdc_len[TotalCoeff - 1][total_zeros], fo_tzdc_code[TotalCoeff - 1][total_zeros]);
            bits += fo_tzdc_len[TotalCoeff - 1][total_zeros];
        }
        zerosLeft = total_zeros;
    }
    for (int j = 0; j < TotalCoeff - 1; j++) {
        if (zerosLeft > 0) {
            if (zerosLeft > 6) {
                if (run[j] < 7) {
                    put(w, 3, (unsigned)(7 - run[j]));
                    bits += 3;
                } else {
                    put(w, run[j] - 4, 0);
                    put(w, 1, 1);
                    bits += (unsigned)(run[j] - 4 + 1);
                }
            } else {
                put(w, fo_rb_len[zerosLeft - 1][run[j]], fo_rb_code[zerosLeft - 1][run[j]]);
                bits += fo_rb_len[zerosLeft - 1][run[j]];
            }
        }
        zerosLeft -= run[j];
    }
    return bits;
}

/* one block in MB context: derives nC, records TotalCoeff (the side effect
 * at F/residual.cpp:530-541 / :838-849) */
unsigned fo_cavlc_block(fo_ctx *c, fo_bw *w, const int *coef, int maxNumCoeff, int kind, int blk, int iCbCr)
{
    int nC = fo_cavlc_nC(c, kind, blk, iCbCr);
    int tc;
    unsigned bits = fo_cavlc_encode_block(w, coef, maxNumCoeff, nC, &tc);
    if (kind <= 2)
        c->tc_l[c->cur][kind == 0 ? 0 : blk] = tc;
    else if (kind == 4)
        c->tc_c[c->cur][iCbCr][blk] = tc;
    return bits;
}

/* residual block order of F/residual.cpp:300-372 (write) == F/rbsp_encoding.cpp:433-485 (size) */
static unsigned residual_blocks(fo_ctx *c, fo_bw *w)
{
    unsigned bits = 0;
    int i16 = fo_pred_class(c, c->cur_mb_type) == 1;
    if (i16) bits += fo_cavlc_block(c, w, c->lv.DC16, 16, 0, 0, 0);
    for (int i8 = 0; i8 < 4; i8++)
        for (int i4 = 0; i4 < 4; i4++)
            if (c->cbpL & (1 << i8)) {
                int blk = i8 * 4 + i4;
                if (i16)
                    bits += fo_cavlc_block(c, w, c->lv.AC16[blk], 15, 1, blk, 0);
                else
                    bits += fo_cavlc_block(c, w, c->lv.Lumalevel[blk], 16, 2, blk, 0);
            }
    for (int k = 0; k < 2; k++)
        if (c->cbpC & 3) bits += fo_cavlc_block(c, w, c->lv.CDC[k], 4, 3, 0, k);
    for (int k = 0; k < 2; k++)
        for (int b = 0; b < 4; b++)
            if (c->cbpC & 2) bits += fo_cavlc_block(c, w, c->lv.CAC[k][b], 15, 4, b, k);
    return bits;
}

void fo_residual_write(fo_ctx *c, fo_bw *w) { residual_blocks(c, w); }

/* F/rbsp_encoding.cpp:330-488 */
unsigned fo_coded_mb_size(fo_ctx *c, int mode16, int predL[16][16], int predCb[8][8], int predCr[8][8])
{
    unsigned total = 0;
    if (c->slice_type != 2) {
        fo_quantizationTransform(c, predL, predCb, predCr, 0);
        fo_setCodedBlockPattern(c);
    } else if (mode16 == -1) {
        c->cur_mb_type = FO_I_4x4;
        fo_quantizationTransform(c, predL, predCb, predCr, 0);
        fo_setCodedBlockPattern(c);
    } else {
        c->cur_mb_type = mode16 + 1;
        fo_quantizationTransform(c, predL, predCb, predCr, 0);
        fo_setCodedBlockPattern(c);
        c->cur_mb_type += c->cbpC << 2;
        if (c->cbpL == 15) c->cur_mb_type += 12;
    }
    int t = c->cur_mb_type;
    int pc = fo_pred_class(c, t);
    total += (unsigned)fo_ue_len((unsigned)t);
    if (pc == 2 && (t == FO_P_8x8 || t == FO_P_8x8ref0)) {
        for (int i = 0; i < 4; i++) total += (unsigned)fo_ue_len((unsigned)c->sub_mb_type[i]);
        for (int i = 0; i < 4; i++) {
            total += (unsigned)fo_ue_len(fo_se_to_ue(c->mvd[i][0][0]));
            total += (unsigned)fo_ue_len(fo_se_to_ue(c->mvd[i][0][1]));
        }
    }
    if (pc == 0 || pc == 1) {
        if (pc == 0)
            for (int b = 0; b < 16; b++) total += c->prev_flag[b] ? 1u : 4u;
        total += (unsigned)fo_ue_len((unsigned)c->chroma_mode);
    } else {
        int np = (t == FO_P_L0_16x16 || t == FO_P_SKIP) ? 1 : ((t == FO_P_16x8 || t == FO_P_8x16) ? 2 : 4);
        for (int i = 0; i < np; i++) {
            total += (unsigned)fo_ue_len(fo_se_to_ue(c->mvd[i][0][0]));
            total += (unsigned)fo_ue_len(fo_se_to_ue(c->mvd[i][0][1]));
        }
    }
    if (pc != 1) {
        int cbp = (c->cbpC << 4) | c->cbpL;
        total += (unsigned)fo_ue_len((unsigned)(pc == 0 ? fo_cbp_intra_to_code[cbp] : fo_cbp_inter_to_code[cbp]));
    }
    if (c->cbpL > 0 || c->cbpC > 0 || pc == 1) {
        total += 1;
        total += residual_blocks(c, NULL);
    }
    return total;
}

/* ------------------------------------------------------------------ parse */

static int vlc_match(fo_br *r, int len, unsigned code)
{
    if (len <= 0) return 0;
    fo_br t = *r;
    return fo_br_bits(&t, len) == code;
}

static int read_coeff_token(fo_br *r, int cls, int *tc, int *t1)
{
    if (cls == 3) {
        unsigned v = fo_br_bits(r, 6);
        if (v == 3) {
            *tc = 0;
            *t1 = 0;
        } else {
            *tc = (int)(v >> 2) + 1;
            *t1 = (int)(v & 3);
        }
        return 1;
    }
    int maxtc = (cls == 4) ? 4 : 16;
    for (int T = 0; T <= maxtc; T++)
        for (int o = 0; o <= 3 && o <= T; o++) {
            int len;
            unsigned code;
            fo_coeff_token(cls, T, o, &len, &code);
            if (vlc_match(r, len, code)) {
                r->pos += (size_t)len;
                *tc = T;
                *t1 = o;
                return 1;
            }
        }
    *tc = 0;
    *t1 = 0;
    return 0;
}

/* F/residual.cpp:1069-1386.  Returns 0 on a malformed block. */
static int parse_block(fo_ctx *c, fo_br *r, int *coef, int maxNumCoeff, int kind, int blk, int iCbCr)
{
    for (int i = 0; i < maxNumCoeff; i++) coef[i] = 0;
    int nC = fo_cavlc_nC(c, kind, blk, iCbCr);
    int cls = (nC == -1) ? 4 : (nC <= 1 ? 0 : (nC <= 3 ? 1 : (nC < 8 ? 2 : 3)));
    int TotalCoeff, TrailingOnes;
    if (!read_coeff_token(r, cls, &TotalCoeff, &TrailingOnes)) return 0;
    if (kind <= 2)
        c->tc_l[c->cur][kind == 0 ? 0 : blk] = TotalCoeff;
    else if (kind == 4)
        c->tc_c[c->cur][iCbCr][blk] = TotalCoeff;
    if (TotalCoeff == 0) return 1;
    int level[16], run[16];
    int suffixLength = (TotalCoeff > 10 && TrailingOnes < 3) ? 1 : 0;
    for (int i = 0; i < TotalCoeff; i++) {
        if (i < TrailingOnes) {
            level[i] = 1 - 2 * (int)fo_br_bit(r);
        } else {
            int level_prefix = 0;
            while (fo_br_bit(r) == 0) {
                level_prefix++;
                if (level_prefix > 32) return 0;
            }
            int size;
            if (level_prefix == 14 && suffixLength == 0)
                size = 4;
            else if (level_prefix >= 15)
                size = level_prefix - 3;
            else
                size = suffixLength;
            unsigned suffix = (size > 0 || level_prefix >= 14) ? fo_br_bits(r, size) : 0;
            /* inputstream_to_levelcode, F/residual_tables.cpp:979-997 */
            int levelCode = ((level_prefix < 15 ? level_prefix : 15) << suffixLength);
            if (size > 0 || level_prefix >= 14) levelCode += (int)suffix;
            if (level_prefix >= 15 && suffixLength == 0) levelCode += 15;
            if (i == TrailingOnes && TrailingOnes < 3) levelCode += 2;
            if ((levelCode & 1) == 0)
                level[i] = (levelCode + 2) >> 1;
            else
                level[i] = (-levelCode - 1) >> 1;
            if (suffixLength == 0) suffixLength = 1;
            int a = level[i] < 0 ? -level[i] : level[i];
            if (a > (3 << (suffixLength - 1)) && suffixLength < 6) suffixLength++;
        }
    }
    int zerosLeft = 0;
    if (TotalCoeff < maxNumCoeff) {
        int found = 0;
        int maxtz = (nC == -1) ? 3 : 15;
        for (int tz = 0; tz <= maxtz && !found; tz++) {
            int len = (nC == -1) ? fo_tzdc_len[TotalCoeff - 1][tz] : fo_tz_len[TotalCoeff - 1][tz];
            unsigned code = (nC == -1) ? fo_tzdc_code[TotalCoeff - 1][tz] : fo_tz_code[TotalCoeff - 1][tz];
            if (vlc_match(r, len, code)) {
                r->pos += (size_t)len;
                zerosLeft = tz;
                found = 1;
            }
        }
        if (!found) return 0;
    }
    for (int j = 0; j < TotalCoeff - 1; j++) {
        if (zerosLeft > 0) {
            int rb = 0;
            if (zerosLeft > 6) {
                rb = 7 - (int)fo_br_bits(r, 3);
                if (rb == 7)
                    while (fo_br_bit(r) == 0) {
                        rb++;
                        if (rb > 64) return 0;
                    }
            } else {
                int found = 0;
                for (int k = 0; k <= zerosLeft && !found; k++) {
                    if (vlc_match(r, fo_rb_len[zerosLeft - 1][k], fo_rb_code[zerosLeft - 1][k])) {
                        r->pos += fo_rb_len[zerosLeft - 1][k];
                        rb = k;
                        found = 1;
                    }
                }
                if (!found) return 0;
            }
            run[j] = rb;
        } else {
            run[j] = 0;
        }
        zerosLeft -= run[j];
    }
    run[TotalCoeff - 1] = zerosLeft;
    int coeffNum = -1;
    for (int i = TotalCoeff - 1; i >= 0; i--) {
        coeffNum += run[i] + 1;
        if (coeffNum >= 0 && coeffNum < 16) coef[coeffNum] = level[i];
    }
    return 1;
}

/* residual(0,15): F/residual.cpp:959-1067 */
int fo_residual_parse(fo_ctx *c, fo_br *r)
{
    int ok = 1;
    int i16 = fo_pred_class(c, c->cur_mb_type) == 1;
    if (i16) ok &= parse_block(c, r, c->lv.DC16, 16, 0, 0, 0);
    for (int i8 = 0; i8 < 4; i8++)
        for (int i4 = 0; i4 < 4; i4++) {
            int blk = i8 * 4 + i4;
            if (c->cbpL & (1 << i8)) {
                if (i16)
                    ok &= parse_block(c, r, c->lv.AC16[blk], 15, 1, blk, 0);
                else
                    ok &= parse_block(c, r, c->lv.Lumalevel[blk], 16, 2, blk, 0);
            } else if (i16) {
                c->tc_l[c->cur][blk] = 0;
                for (int i = 0; i < 15; i++) c->lv.AC16[blk][i] = 0;
            } else {
                c->tc_l[c->cur][blk] = 0;
                for (int i = 0; i < 16; i++) c->lv.Lumalevel[blk][i] = 0;
            }
        }
    for (int k = 0; k < 2; k++) {
        if (c->cbpC & 3)
            ok &= parse_block(c, r, c->lv.CDC[k], 4, 3, 0, k);
        else
            for (int i = 0; i < 4; i++) c->lv.CDC[k][i] = 0;
    }
    for (int k = 0; k < 2; k++)
        for (int b = 0; b < 4; b++) {
            if (c->cbpC & 2) {
                ok &= parse_block(c, r, c->lv.CAC[k][b], 15, 4, b, k);
            } else {
                c->tc_c[c->cur][k][b] = 0;
                for (int i = 0; i < 15; i++) c->lv.CAC[k][b][i] = 0;
            }
        }
    return ok;
}
