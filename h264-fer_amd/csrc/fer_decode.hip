// fer_decode.hip -- the decode twin of the hot path (row a19 of SURVEY.md 8a, BASELINE configs[4]):
// the macroblock loop of RBSP_decode (F/rbsp_decoding.cpp:77-351) on the GPU.
//
//   k_dec_parse   slice_data parsing is bit-serial inside a slice (one slice per picture), so one
//                 WAVEFRONT walks one picture and many pictures (streams) are parsed side by side.
//                 All lanes execute the same, lane-independent code, which the compiler keeps in
//                 scalar registers; it emits, per macroblock, mb_type, QP, final motion vectors
//                 (DeriveMVs, F/mode_pred.cpp:428), Intra4x4 modes, CBP, TotalCoeff and the
//                 coefficient levels -- CAVLC of F/residual.cpp:959-1386.
//   k_dec_inter   every inter / P_Skip macroblock in parallel: motion compensation
//                 (F/mocomp.cpp), dequantisation + inverse transform, clipped reconstruction.
//   k_dec_intra   intra macroblocks along the anti-diagonal wavefront x + 2y (they read
//                 reconstructed neighbours): intraPrediction (F/intra.cpp:770) + residual.
//
// Supported syntax: what the reference encoder emits plus intra macroblocks inside P slices and
// non-zero mb_qp_delta.  Sub-macroblock types other than P_L0_8x8 set FER_ERR_DEC_UNSUPPORTED.
// Decoder quirks of the reference that its output depends on are kept: mb_qp_delta persists when
// absent (F/rbsp_decoding.cpp:298,322), chroma AC levels persist into cbp == 0 macroblocks
// (F/residual.cpp:28-49), the more_rbsp_data heuristic (F/rbsp_IO.cpp:193).
#include "fer_internal.h"
#include "fer_intra_dev.h"
#include "fer_mvpred.h"

struct DecBits {
    const uint8_t *buf;
    unsigned size;  // bytes
    unsigned pos;   // bit position
};

__device__ __forceinline__ unsigned db_peek(const DecBits &b, int n)  // n <= 25
{
    unsigned byte = b.pos >> 3;
    unsigned v = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) v = (v << 8) | (byte + i < b.size ? (unsigned)b.buf[byte + i] : 0u);
    return (v << (b.pos & 7)) >> (32 - n);
}
__device__ __forceinline__ unsigned db_bits(DecBits &b, int n)
{
    if (n == 0) return 0;
    unsigned v = db_peek(b, n);
    b.pos += n;
    return v;
}
__device__ __forceinline__ unsigned db_bit(DecBits &b) { return db_bits(b, 1); }
__device__ __forceinline__ unsigned db_ue(DecBits &b)  // F/expgolomb.cpp:122-140
{
    unsigned w = db_peek(b, 24);
    int z = w ? __clz((int)w) - 8 : 24;
    b.pos += z + 1;
    unsigned s = db_bits(b, z);
    return (1u << z) - 1u + s;
}
__device__ __forceinline__ int db_se(DecBits &b)
{
    int v = (int)db_ue(b);
    return (v & 1) ? (v + 1) / 2 : -v / 2;
}
__device__ __forceinline__ bool db_more(const DecBits &b) { return (b.pos >> 3) + 1 < b.size; }

// residual_block_cavlc, F/residual.cpp:1069-1386.  coef: int16 destination (maxNumCoeff entries,
// already zero).  Returns TotalCoeff, or -1 on a malformed block.
__device__ int dec_block(DecBits &b, int16_t *coef, int maxNumCoeff, int nC)
{
    int TotalCoeff = -1, TrailingOnes = 0;
    if (nC >= 8) {
        unsigned v = db_bits(b, 6);
        if (v == 3) {
            TotalCoeff = 0;
        } else {
            TotalCoeff = (int)(v >> 2) + 1;
            TrailingOnes = (int)(v & 3);
        }
    } else {
        unsigned w = db_peek(b, 16);
        int maxtc = nC == -1 ? 4 : 16;
        for (int T = 0; T <= maxtc && TotalCoeff < 0; T++)
            for (int o = 0; o <= 3 && o <= T; o++) {
                int len;
                unsigned code;
                if (nC == -1) {
                    len = c_ctdc_len[o][T];
                    code = c_ctdc_code[o][T];
                } else {
                    int cls = nC <= 1 ? 0 : (nC <= 3 ? 1 : 2);
                    len = c_ct_len[cls][o][T];
                    code = c_ct_code[cls][o][T];
                }
                if (len > 0 && (w >> (16 - len)) == code) {
                    TotalCoeff = T;
                    TrailingOnes = o;
                    b.pos += len;
                    break;
                }
            }
        if (TotalCoeff < 0) return -1;
    }
    if (TotalCoeff == 0) return 0;
    if (TotalCoeff > maxNumCoeff) return -1;
    int level[16], run[16];
    int suffixLength = (TotalCoeff > 10 && TrailingOnes < 3) ? 1 : 0;
    for (int i = 0; i < TotalCoeff; i++) {
        if (i < TrailingOnes) {
            level[i] = 1 - 2 * (int)db_bit(b);
        } else {
            int prefix = 0;
            while (db_bit(b) == 0) {
                prefix++;
                if (prefix > 32) return -1;
            }
            int size = (prefix == 14 && suffixLength == 0) ? 4 : (prefix >= 15 ? prefix - 3 : suffixLength);
            unsigned suffix = (size > 0 || prefix >= 14) ? db_bits(b, size) : 0;
            int levelCode = (min(prefix, 15) << suffixLength);
            if (size > 0 || prefix >= 14) levelCode += (int)suffix;
            if (prefix >= 15 && suffixLength == 0) levelCode += 15;
            if (i == TrailingOnes && TrailingOnes < 3) levelCode += 2;
            level[i] = (levelCode & 1) == 0 ? (levelCode + 2) >> 1 : (-levelCode - 1) >> 1;
            if (suffixLength == 0) suffixLength = 1;
            if (iabs(level[i]) > (3 << (suffixLength - 1)) && suffixLength < 6) suffixLength++;
        }
    }
    int zerosLeft = 0;
    if (TotalCoeff < maxNumCoeff) {
        unsigned w = db_peek(b, 9);
        int found = -1;
        int maxtz = nC == -1 ? 3 : 15;
        for (int tz = 0; tz <= maxtz; tz++) {
            int len = nC == -1 ? c_tzdc_len[TotalCoeff - 1][tz] : c_tz_len[TotalCoeff - 1][tz];
            unsigned code = nC == -1 ? c_tzdc_code[TotalCoeff - 1][tz] : c_tz_code[TotalCoeff - 1][tz];
            if (len > 0 && (w >> (9 - len)) == code) {
                found = tz;
                b.pos += len;
                break;
            }
        }
        if (found < 0) return -1;
        zerosLeft = found;
    }
    for (int j = 0; j < TotalCoeff - 1; j++) {
        int rb = 0;
        if (zerosLeft > 0) {
            if (zerosLeft > 6) {
                rb = 7 - (int)db_bits(b, 3);
                if (rb == 7)
                    while (db_bit(b) == 0) {
                        rb++;
                        if (rb > 64) return -1;
                    }
            } else {
                unsigned w = db_peek(b, 3);
                int found = -1;
                for (int k = 0; k <= zerosLeft; k++) {
                    int len = c_rb_len[zerosLeft - 1][k];
                    if (len > 0 && (w >> (3 - len)) == c_rb_code[zerosLeft - 1][k]) {
                        found = k;
                        b.pos += len;
                        break;
                    }
                }
                if (found < 0) return -1;
                rb = found;
            }
        }
        run[j] = rb;
        zerosLeft -= rb;
    }
    run[TotalCoeff - 1] = zerosLeft;
    int coeffNum = -1;
    for (int i = TotalCoeff - 1; i >= 0; i--) {
        coeffNum += run[i] + 1;
        if (coeffNum >= 0 && coeffNum < maxNumCoeff) coef[coeffNum] = (int16_t)level[i];
    }
    return TotalCoeff;
}

// nC from global side info + the current MB's counts kept in LDS (wave-uniform)
__device__ int dec_nC(const FerDev &d, int s, int mb, bool luma, int blk, int plane, const uint8_t *tcur, int cbpL,
                      int cbpC)
{
    const int *mbt = d.mb_type + (size_t)s * d.nmb;
    const uint8_t *cbp = d.cbp + (size_t)s * d.nmb * 2;
    const uint8_t *tc = d.tc + (size_t)s * d.nmb * 24;
    bool edgeA, edgeB;
    int bA, bB;
    if (luma) {
        edgeA = blk == 0 || blk == 2 || blk == 8 || blk == 10;
        edgeB = blk == 0 || blk == 1 || blk == 4 || blk == 5;
        bA = c_nbA[blk];
        bB = c_nbB[blk];
    } else {
        edgeA = blk == 0 || blk == 2;
        edgeB = blk < 2;
        bA = c_nbcA[blk];
        bB = c_nbcB[blk];
    }
    bool availA = true, availB = true;
    int nA = 0, nB = 0;
    if (edgeA) {
        if (mb % d.mbw == 0)
            availA = false;
        else {
            int m = mb - 1;
            bool zero = luma ? ((cbp[m * 2] & (1 << (bA / 4))) == 0) : ((cbp[m * 2 + 1] & 2) == 0);
            if (!(mbt[m] == FER_P_SKIP || zero)) nA = luma ? tc[m * 24 + bA] : tc[m * 24 + 16 + plane * 4 + bA];
        }
    } else {
        bool zero = luma ? ((cbpL & (1 << (bA / 4))) == 0) : ((cbpC & 2) == 0);
        if (!zero) nA = luma ? tcur[bA] : tcur[16 + plane * 4 + bA];
    }
    if (edgeB) {
        if (mb < d.mbw)
            availB = false;
        else {
            int m = mb - d.mbw;
            bool zero = luma ? ((cbp[m * 2] & (1 << (bB / 4))) == 0) : ((cbp[m * 2 + 1] & 2) == 0);
            if (!(mbt[m] == FER_P_SKIP || zero)) nB = luma ? tc[m * 24 + bB] : tc[m * 24 + 16 + plane * 4 + bB];
        }
    } else {
        bool zero = luma ? ((cbpL & (1 << (bB / 4))) == 0) : ((cbpC & 2) == 0);
        if (!zero) nB = luma ? tcur[bB] : tcur[16 + plane * 4 + bB];
    }
    if (availA && availB) return (nA + nB + 1) >> 1;
    if (availA) return nA;
    if (availB) return nB;
    return 0;
}

// P macroblock vectors: PredictMV + DeriveMVs (F/mode_pred.cpp:381-482), quadrant storage
__device__ void dec_derive_mvs(const FerDev &d, short *mvs, const int *mbt, int mb, int type, const int mvd[4][2])
{
    MvCtx c;
    c.mv = mvs;
    c.mb_type = mbt;
    c.mbw = d.mbw;
    c.cur = mb;
    c.type = type;
    c.coh = false;
    short *o = mvs + (size_t)mb * 8;
    if (type == FER_P_SKIP) {
        int mx = 0, my = 0;
        if (!(mb < d.mbw || mb % d.mbw == 0)) {
            int up = mb - d.mbw, lf = mb - 1;
            bool iu = mbt[up] >= 5 && mbt[up] <= 30, il = mbt[lf] >= 5 && mbt[lf] <= 30;
            bool zu = !iu && (mvs[(up * 4 + 2) * 2] | mvs[(up * 4 + 2) * 2 + 1]) == 0;
            bool zl = !il && (mvs[(lf * 4 + 1) * 2] | mvs[(lf * 4 + 1) * 2 + 1]) == 0;
            if (!(zu || zl)) predict_luma(c, 0, mx, my);
        }
        for (int q = 0; q < 4; q++) {
            o[q * 2] = (short)mx;
            o[q * 2 + 1] = (short)my;
        }
        return;
    }
    int np = type == 0 ? 1 : (type <= 2 ? 2 : 4);
    for (int i = 0; i < np; i++) {
        int px, py;
        predict_luma(c, i, px, py);
        px += mvd[i][0];
        py += mvd[i][1];
        // quadrants covered by partition i
        int q0, q1, q2 = -1, q3 = -1;
        if (np == 1) {
            q0 = 0;
            q1 = 1;
            q2 = 2;
            q3 = 3;
        } else if (type == FER_P_16x8) {
            q0 = i * 2;
            q1 = i * 2 + 1;
        } else if (type == FER_P_8x16) {
            q0 = i;
            q1 = i + 2;
        } else {
            q0 = q1 = i;
        }
        int qs[4] = {q0, q1, q2, q3};
        for (int k = 0; k < 4; k++)
            if (qs[k] >= 0) {
                o[qs[k] * 2] = (short)px;
                o[qs[k] * 2 + 1] = (short)py;
            }
    }
}

__global__ __launch_bounds__(64) void k_dec_parse(FerDev d, const uint8_t *rbsp, size_t rbsp_stride,
                                                  const uint32_t *info /* [S][4]: bytes, first bit, slice_type%5, SliceQPy */)
{
    __shared__ uint8_t tcur[24];
    __shared__ int16_t cac[2][4][16];  // ChromaACLevel persists across macroblocks (reference quirk)
    const int lane = threadIdx.x;
    const int s = blockIdx.x;
    DecBits b;
    b.buf = rbsp + (size_t)s * rbsp_stride;
    b.size = info[s * 4];
    b.pos = info[s * 4 + 1];
    const int stype = (int)info[s * 4 + 2];
    if (b.size == 0) {  // no picture for this stream at this step
        if (lane == 0) d.dec_state[(size_t)s * 4 + 1] = 0;
        return;
    }
    int QPy = (int)info[s * 4 + 3];
    int *mbt = d.mb_type + (size_t)s * d.nmb;
    short *mvs = d.mv + (size_t)s * d.nmb * 8;
    int *st = d.dec_state + (size_t)s * 4;  // [0] = mb_qp_delta carried across slices
    int mb_qp_delta = st[0];
    for (int i = lane; i < 2 * 4 * 16; i += 64) (&cac[0][0][0])[i] = d.dec_cac[(size_t)s * 128 + i];
    __syncthreads();
    int cur = 0;
    bool more = true;
    int mvd[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    while (more && cur < d.nmb) {
        if (stype != 2) {
            int run = (int)db_ue(b);
            for (int i = 0; i < run && cur < d.nmb; i++) {
                size_t mbi = (size_t)s * d.nmb + cur;
                mbt[cur] = FER_P_SKIP;
                for (int k = 0; k < 4; k++) mvd[k][0] = mvd[k][1] = 0;  // ClearMVD in PredictMV
                dec_derive_mvs(d, mvs, mbt, cur, FER_P_SKIP, mvd);
                QPy = (QPy + mb_qp_delta + 52) % 52;
                d.dec_qp[mbi] = (uint8_t)QPy;
                cur++;
            }
            if (cur != 0 || run > 0) more = db_more(b);
        }
        if (!(more && cur < d.nmb)) break;
        const size_t mbi = (size_t)s * d.nmb + cur;
        int16_t *lv = d.levels + mbi * FER_LEVELS;
        for (int i = lane; i < FER_LEVELS; i += 64) lv[i] = 0;
        int t = (int)db_ue(b);
        if (t > 31 || (stype == 2 && t > 24)) {
            atomicOr(&d.status[s], FER_ERR_DEC_SYNTAX);
            break;
        }
        mbt[cur] = t;
        const int k = stype == 2 ? t : t - 5;  // index into the I macroblock table
        const bool i4 = stype == 2 ? t == 0 : t == 5;
        const bool i16 = stype == 2 ? (t >= 1 && t <= 24) : (t >= 6 && t <= 29);
        const bool inter = !i4 && !i16;
        if (t == 25 && stype == 2) {
            atomicOr(&d.status[s], FER_ERR_DEC_UNSUPPORTED);  // I_PCM
            break;
        }
        int chroma_mode = 0;
        if (inter) {
            if (t == 3 || t == 4) {
                int sub[4];
                for (int i = 0; i < 4; i++) sub[i] = (int)db_ue(b);
                if (sub[0] | sub[1] | sub[2] | sub[3]) {
                    atomicOr(&d.status[s], FER_ERR_DEC_UNSUPPORTED);  // sub-8x8 partitions
                    break;
                }
                for (int i = 0; i < 4; i++) {
                    mvd[i][0] = db_se(b);
                    mvd[i][1] = db_se(b);
                }
            } else {
                int np = t == 0 ? 1 : 2;
                for (int i = 0; i < np; i++) {
                    mvd[i][0] = db_se(b);
                    mvd[i][1] = db_se(b);
                }
            }
        } else {
            if (i4)
                for (int blk = 0; blk < 16; blk++) {
                    int f = (int)db_bit(b);
                    int rem = f ? 0 : (int)db_bits(b, 3);
                    d.i4flag[mbi * 16 + blk] = (uint8_t)((f << 3) | rem);
                }
            chroma_mode = (int)db_ue(b);
            if (chroma_mode > 3) {
                atomicOr(&d.status[s], FER_ERR_DEC_SYNTAX);
                break;
            }
        }
        int cbpL, cbpC;
        if (!i16) {
            unsigned code = db_ue(b);
            if (code > 47) {
                atomicOr(&d.status[s], FER_ERR_DEC_SYNTAX);
                break;
            }
            int cbp = i4 ? c_code_cbp_intra[code] : c_code_cbp_inter[code];
            cbpL = cbp & 15;
            cbpC = cbp >> 4;
        } else {
            cbpC = ((k - 1) / 4) % 3;
            cbpL = k >= 13 ? 15 : 0;
        }
        d.cbp[mbi * 2] = (uint8_t)cbpL;
        d.cbp[mbi * 2 + 1] = (uint8_t)cbpC;
        d.chroma_mode[mbi] = (uint8_t)chroma_mode;
        for (int i = lane; i < 24; i += 64) tcur[i] = 0;
        __syncthreads();
        bool bad = false;
        if (cbpL > 0 || cbpC > 0 || i16) {
            mb_qp_delta = db_se(b);
            if (mb_qp_delta < -26 || mb_qp_delta > 25) bad = true;
            // residual(0,15), F/residual.cpp:959-1067
            if (i16 && !bad) {
                int n = dec_block(b, lv + FER_LV_DC16, 16, dec_nC(d, s, cur, true, 0, 0, tcur, cbpL, cbpC));
                bad |= n < 0;
                if (n >= 0) tcur[0] = (uint8_t)n;
                __syncthreads();
            }
            for (int i8 = 0; i8 < 4 && !bad; i8++)
                if (cbpL & (1 << i8))
                    for (int i4x = 0; i4x < 4 && !bad; i4x++) {
                        int blk = i8 * 4 + i4x;
                        int n = dec_block(b, lv + blk * 16, i16 ? 15 : 16, dec_nC(d, s, cur, true, blk, 0, tcur, cbpL, cbpC));
                        bad |= n < 0;
                        if (n >= 0) tcur[blk] = (uint8_t)n;
                        __syncthreads();
                    }
            for (int pl = 0; pl < 2 && !bad; pl++)
                if (cbpC & 3) bad |= dec_block(b, lv + FER_LV_CDC + pl * 4, 4, -1) < 0;
            for (int pl = 0; pl < 2 && !bad; pl++)
                for (int cb = 0; cb < 4 && !bad; cb++) {
                    if (cbpC & 2) {
                        for (int i = lane; i < 16; i += 64) cac[pl][cb][i] = 0;
                        __syncthreads();
                        int n = dec_block(b, &cac[pl][cb][0], 15, dec_nC(d, s, cur, false, cb, pl, tcur, cbpL, cbpC));
                        bad |= n < 0;
                        if (n >= 0) tcur[16 + pl * 4 + cb] = (uint8_t)n;
                        __syncthreads();
                    } else {
                        for (int i = lane; i < 16; i += 64) cac[pl][cb][i] = 0;
                        __syncthreads();
                    }
                }
        }
        if (bad) {
            atomicOr(&d.status[s], FER_ERR_DEC_SYNTAX);
            break;
        }
        // chroma AC of this macroblock = the persistent ChromaACLevel (stale when cbp == 0)
        for (int i = lane; i < 120; i += 64) lv[FER_LV_CAC + i] = cac[i / 60][(i % 60) / 15][i % 15];
        for (int i = lane; i < 24; i += 64) d.tc[mbi * 24 + i] = tcur[i];
        QPy = (QPy + mb_qp_delta + 52) % 52;
        d.dec_qp[mbi] = (uint8_t)QPy;
        if (inter) {
            dec_derive_mvs(d, mvs, mbt, cur, t, mvd);
        } else if (i4) {
            // getIntra4x4PredMode, F/intra.cpp:77-136
            for (int blk = 0; blk < 16; blk++) {
                bool edgeA = blk == 0 || blk == 2 || blk == 8 || blk == 10;
                bool edgeB = blk == 0 || blk == 1 || blk == 4 || blk == 5;
                bool okA = !(edgeA && cur % d.mbw == 0), okB = !(edgeB && cur < d.mbw);
                int mA = 2, mB = 2;
                if (okA && okB && !d.dec_constrained_intra) {
                    int ma = edgeA ? cur - 1 : cur, mb2 = edgeB ? cur - d.mbw : cur;
                    int ta = mbt[ma], tb = mbt[mb2];
                    bool a4 = stype == 2 ? ta == 0 : ta == 5, b4 = stype == 2 ? tb == 0 : tb == 5;
                    mA = a4 ? d.i4mode[((size_t)s * d.nmb + ma) * 16 + c_nbA[blk]] : 2;
                    mB = b4 ? d.i4mode[((size_t)s * d.nmb + mb2) * 16 + c_nbB[blk]] : 2;
                }
                int pm = mA <= mB ? mA : mB;
                int f = d.i4flag[mbi * 16 + blk];
                int mode = (f & 8) ? pm : ((f & 7) < pm ? (f & 7) : (f & 7) + 1);
                d.i4mode[mbi * 16 + blk] = (uint8_t)mode;
            }
        }
        more = db_more(b);
        cur++;
    }
    __syncthreads();
    for (int i = lane; i < 128; i += 64) d.dec_cac[(size_t)s * 128 + i] = (&cac[0][0][0])[i];
    if (lane == 0) {
        st[0] = mb_qp_delta;
        st[1] = cur;  // macroblocks reached (the rest of the picture keeps the previous content)
    }
}

// ---- reconstruction of one 4x4 block owned by a lane
__device__ __forceinline__ void recon_block(const int16_t *__restrict__ lvl, int n, int dc, bool keepDC, int qP,
                                            const int p[16], uint8_t *dst, int stride)
{
    int c[16], r[16];
#pragma unroll
    for (int i = 0; i < 16; i++) c[i] = 0;
    if (keepDC) {
        c[0] = dc;
        for (int k = 1; k < 16; k++) c[c_zz[k]] = lvl[k - 1];
    } else {
        for (int k = 0; k < n; k++) c[c_zz[k]] = lvl[k];
    }
    inv4x4(c, r, qP, keepDC);
#pragma unroll
    for (int i = 0; i < 16; i++) dst[(size_t)(i >> 2) * stride + (i & 3)] = (uint8_t)clip255(p[i] + r[i]);
}

__device__ __forceinline__ int dec_qpc(const FerDev &d, int QPy)
{
    const int qpc[52] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17,
                         18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31, 32, 32, 33,
                         34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39};
    return qpc[iclamp(QPy + d.dec_chroma_qp_offset, 0, 51)];
}

// chroma residual of lanes 16..23 (dc Hadamard shared through shuffles), F/inttransform.cpp:237-320
__device__ __forceinline__ void recon_chroma(const FerDev &d, const int16_t *lv, int lane, int qpc_, const uint8_t *pred,
                                             uint8_t *C0, uint8_t *C1, int Wc, int xp, int yp)
{
    const bool isC = lane >= 16 && lane < 24;
    const int pl = (lane - 16) >> 2, cb = (lane - 16) & 3;
    int cq[4], dq[4];
    int base = lane >= 20 ? 1 : 0;
#pragma unroll
    for (int i = 0; i < 4; i++) cq[i] = lv[FER_LV_CDC + base * 4 + i];
    inv_dc_chroma(cq, dq, qpc_);
    if (isC) {
        int x0 = (cb & 1) * 4, y0 = (cb >> 1) * 4;
        int p[16];
#pragma unroll
        for (int i = 0; i < 16; i++) p[i] = pred[pl * 64 + (y0 + (i >> 2)) * 8 + x0 + (i & 3)];
        uint8_t *dst = (pl ? C1 : C0) + (size_t)(yp / 2 + y0) * Wc + xp / 2 + x0;
        recon_block(lv + FER_LV_CAC + (pl * 4 + cb) * 15, 15, dq[cb], true, qpc_, p, dst, Wc);
    }
}

__global__ __launch_bounds__(64) void k_dec_inter(FerDev d)
{
    __shared__ uint8_t pL[16][16], pC[2 * 64];
    const int lane = threadIdx.x;
    const int s = blockIdx.y, mb = blockIdx.x;
    const size_t mbi = (size_t)s * d.nmb + mb;
    if (mb >= d.dec_state[s * 4 + 1]) return;  // not reached by the parser
    const int t = d.mb_type[mbi];
    const bool inter = t <= 4 || t == FER_P_SKIP;
    if (d.hdr[s * 4 + 3] != 0 || !inter) return;
    const int W = d.W, H = d.H, Wc = d.Wc, Hc = d.Hc;
    uint8_t *Y = d.curY + (size_t)s * d.ysz;
    uint8_t *C0 = d.curCb + (size_t)s * d.csz, *C1 = d.curCr + (size_t)s * d.csz;
    const uint8_t *RY = d.refY + (size_t)s * d.ysz;
    const uint8_t *RC0 = d.refCb + (size_t)s * d.csz, *RC1 = d.refCr + (size_t)s * d.csz;
    const short *mv = d.mv + mbi * 8;
    const int xp = (mb % d.mbw) << 4, yp = (mb / d.mbw) << 4;
    {
        int lx = (lane & 3) * 4, ly = lane >> 2;
        int q = (ly >> 3) * 2 + (lx >> 3);
#pragma unroll
        for (int k = 0; k < 4; k++) pL[ly][lx + k] = (uint8_t)mc_luma(RY, W, H, xp, yp, lx + k, ly, mv[q * 2], mv[q * 2 + 1]);
        int cx = lane & 7, cy = lane >> 3;
        int qc = (cy >> 2) * 2 + (cx >> 2);
        pC[cy * 8 + cx] = (uint8_t)mc_chroma(RC0, Wc, Hc, xp / 2, yp / 2, cx, cy, mv[qc * 2], mv[qc * 2 + 1]);
        pC[64 + cy * 8 + cx] = (uint8_t)mc_chroma(RC1, Wc, Hc, xp / 2, yp / 2, cx, cy, mv[qc * 2], mv[qc * 2 + 1]);
    }
    __syncthreads();
    const int16_t *lv = d.levels + mbi * FER_LEVELS;
    const int QPy = d.dec_qp[mbi];
    if (t == FER_P_SKIP) {  // all levels zero: reconstruction == prediction
        int lx = (lane & 3) * 4, ly = lane >> 2;
#pragma unroll
        for (int k = 0; k < 4; k++) Y[(size_t)(yp + ly) * W + xp + lx + k] = pL[ly][lx + k];
        int cx = lane & 7, cy = lane >> 3;
        C0[(size_t)(yp / 2 + cy) * Wc + xp / 2 + cx] = pC[cy * 8 + cx];
        C1[(size_t)(yp / 2 + cy) * Wc + xp / 2 + cx] = pC[64 + cy * 8 + cx];
        return;
    }
    if (lane < 16) {
        int x0 = c_bx[lane], y0 = c_by[lane];
        int p[16];
#pragma unroll
        for (int i = 0; i < 16; i++) p[i] = pL[y0 + (i >> 2)][x0 + (i & 3)];
        recon_block(lv + lane * 16, 16, 0, false, QPy, p, Y + (size_t)(yp + y0) * W + xp + x0, W);
    }
    recon_chroma(d, lv, lane, dec_qpc(d, QPy), pC, C0, C1, Wc, xp, yp);
}

__global__ __launch_bounds__(64) void k_dec_intra(FerDev d, int diag)
{
    __shared__ IntraLds L;
    const int lane = threadIdx.x;
    const int s = blockIdx.y;
    int y_lo = diag - (d.mbw - 1);
    y_lo = y_lo > 0 ? (y_lo + 1) >> 1 : 0;
    const int mby = y_lo + blockIdx.x, mbx = diag - 2 * mby;
    if (mby >= d.mbh || mbx < 0 || mbx >= d.mbw) return;
    const int mb = mby * d.mbw + mbx;
    const size_t mbi = (size_t)s * d.nmb + mb;
    if (mb >= d.dec_state[s * 4 + 1]) return;
    const int stype = (int)d.hdr[s * 4 + 3];
    const int t = d.mb_type[mbi];
    const bool i4 = stype == 2 ? t == 0 : t == 5;
    const bool i16 = stype == 2 ? (t >= 1 && t <= 24) : (t >= 6 && t <= 29);
    if (!i4 && !i16) return;
    const int W = d.W, Wc = d.Wc;
    uint8_t *Y = d.curY + (size_t)s * d.ysz;
    uint8_t *Cp[2] = {d.curCb + (size_t)s * d.csz, d.curCr + (size_t)s * d.csz};
    const int xp = mbx << 4, yp = mby << 4;
    const bool availL = mbx > 0, availT = mby > 0, lastcol = mbx == d.mbw - 1;
    for (int i = lane; i < 17 * 21; i += 64) {
        int r = i / 21, c = i % 21;
        int gx = xp + c - 1, gy = yp + r - 1;
        int v = -1;
        if (gx >= 0 && gy >= 0 && gx < W && (r == 0 || c == 0)) v = Y[(size_t)gy * W + gx];
        L.fr[r][c] = (int16_t)v;
    }
    for (int i = lane; i < 2 * 9 * 9; i += 64) {
        int pl = i / 81, r = (i % 81) / 9, c = i % 9;
        int gx = xp / 2 + c - 1, gy = yp / 2 + r - 1;
        L.cfr[pl][r][c] = (int16_t)((gx >= 0 && gy >= 0 && (r == 0 || c == 0)) ? Cp[pl][(size_t)gy * Wc + gx] : -1);
    }
    __syncthreads();
    const int16_t *lv = d.levels + mbi * FER_LEVELS;
    const int QPy = d.dec_qp[mbi];
    const int chroma_mode = d.chroma_mode[mbi];
    // chroma prediction -> L.predC (same code path as the encoder's phase 2a)
    {
        int cx = lane & 7, cy = lane >> 3;
        for (int pl = 0; pl < 2; pl++) L.predC[pl][cy][cx] = (uint8_t)pred_chroma_px(L.cfr[pl], chroma_mode, cx, cy, availL, availT);
    }
    if (i16) {
        P16 q16;
        pred16_params(L, availL, availT, q16);
        int k = stype == 2 ? t : t - 5;
        int mode = (k - 1) & 3;
        if (lane == 0) {
            int c[16], dq[16];
            for (int i = 0; i < 16; i++) c[c_zz[i]] = lv[FER_LV_DC16 + i];
            inv_dc_luma(c, dq, QPy);
            for (int i = 0; i < 16; i++) L.dcdeq[i] = dq[i];
        }
        __syncthreads();
        if (lane < 16) {
            int x0 = c_bx[lane], y0 = c_by[lane];
            int p[16];
            for (int i = 0; i < 16; i++) p[i] = pred16_px(L, q16, mode, x0 + (i & 3), y0 + (i >> 2));
            recon_block(lv + lane * 16, 15, L.dcdeq[(y0 >> 2) * 4 + (x0 >> 2)], true, QPy, p,
                        Y + (size_t)(yp + y0) * W + xp + x0, W);
        }
    } else {
        __syncthreads();
        if (lane == 0) {
            for (int blk = 0; blk < 16; blk++) {
                int p[13], o[16], c[16], r[16];
                fetch4(L, blk, lastcol, p);
                pred4x4(d.i4mode[mbi * 16 + blk], p, o);
                for (int i = 0; i < 16; i++) c[i] = 0;
                for (int kk = 0; kk < 16; kk++) c[c_zz[kk]] = lv[blk * 16 + kk];
                inv4x4(c, r, QPy, false);
                int x0 = c_bx[blk], y0 = c_by[blk];
                for (int i = 0; i < 16; i++) {
                    int v = clip255(o[i] + r[i]);
                    L.fr[1 + y0 + (i >> 2)][1 + x0 + (i & 3)] = (int16_t)v;
                    Y[(size_t)(yp + y0 + (i >> 2)) * W + xp + x0 + (i & 3)] = (uint8_t)v;
                }
            }
        }
    }
    __syncthreads();
    recon_chroma(d, lv, lane, dec_qpc(d, QPy), &L.predC[0][0][0], Cp[0], Cp[1], Wc, xp, yp);
}

void fer_launch_decode(const FerDev &d, const uint8_t *rbsp, size_t stride, const uint32_t *info, bool anyP, bool anyIntra,
                       hipStream_t st)
{
    hipLaunchKernelGGL(k_dec_parse, dim3(d.S), dim3(64), 0, st, d, rbsp, stride, info);
    if (anyP) hipLaunchKernelGGL(k_dec_inter, dim3(d.nmb, d.S), dim3(64), 0, st, d);
    if (anyIntra) {
        int ndiag = d.mbw + 2 * (d.mbh - 1);
        int maxk = min(d.mbh, (d.mbw + 1) / 2);
        for (int dg = 0; dg < ndiag; dg++) hipLaunchKernelGGL(k_dec_intra, dim3(maxk, d.S), dim3(64), 0, st, d, dg);
    }
}
