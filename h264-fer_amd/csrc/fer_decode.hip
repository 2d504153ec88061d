// fer_decode.hip -- the decode twin of the hot path (row a19 of SURVEY.md 8a, BASELINE configs[4]):
// the macroblock loop of RBSP_decode (F/rbsp_decoding.cpp:77-351) on the GPU.
//
//   k_dec_parse   slice_data parsing is bit-serial inside a slice (one slice per picture), so one
//                 WAVEFRONT walks one picture and the parallelism is pictures: every picture of a
//                 window of all streams is parsed by one launch (k_dec_carry / k_dec_patch then
//                 resolve what a picture inherits from its predecessor).  The code is
//                 lane-independent and kept on the scalar unit; it emits, per macroblock, mb_type,
//                 QP, final motion vectors (DeriveMVs, F/mode_pred.cpp:428), Intra4x4 modes, CBP,
//                 TotalCoeff and the coefficient levels -- CAVLC of F/residual.cpp:959-1386.
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

// Bit reader over one slice: a 64-bit window of the stream starting at the 32-bit aligned position wbase (big-endian),
// refilled 32 bits at a time.  The stream itself comes in through a 512-byte ring in LDS, one coalesced 256-byte load of
// the whole wavefront per 2 048 bits: a refill is then a DS read, not a trip to memory (a dword requested from memory
// one refill ahead was waited for on the spot all the same -- the copy into the loop-carried register needs the data).
// Position, window and everything derived from them are wave-uniform (scalar registers).
struct DecBits {
    const uint8_t *buf;  // 4-byte aligned; the slice buffers are padded by 16 bytes
    unsigned size;  // bytes
    unsigned pos;   // bit position
    unsigned long long win;  // stream bits [wbase, wbase + 64)
    unsigned wbase;          // multiple of 32, pos - wbase < 32
    uint32_t *ring;          // [128] dwords of this wavefront in LDS: stream bytes [256 c, 256 c + 256) in half c & 1
};

// chunk c of the stream into its half of the ring (addresses clamped into the padded buffer)
__device__ __forceinline__ void db_chunk(const DecBits &b, unsigned c)
{
    const unsigned lane = threadIdx.x & 63;
    const unsigned a = min(c * 256u + lane * 4u, (b.size + 3u) & ~3u);
    const uint32_t v = *(const uint32_t *)(b.buf + a);
    b.ring[((c & 1u) << 6) + lane] = v;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// the dword at byte offset `byte` (a multiple of 4) as big-endian stream bits, zero past the end of the slice
__device__ __forceinline__ unsigned db_dword(const DecBits &b, unsigned byte)
{
    unsigned v = __builtin_bswap32((unsigned)__builtin_amdgcn_readfirstlane((int)b.ring[(byte >> 2) & 127u]));
    if (byte + 4 > b.size) v = byte >= b.size ? 0u : (v & ~(0xffffffffu >> (8 * (b.size - byte))));
    return v;
}
__device__ __forceinline__ void db_open(DecBits &b, const uint8_t *buf, unsigned size, unsigned pos, uint32_t *ring)
{
    b.buf = buf;
    b.size = size;
    b.pos = pos;
    b.ring = ring;
    b.wbase = pos & ~31u;
    const unsigned by = b.wbase >> 3;
    db_chunk(b, by >> 8);
    if (((by + 4) >> 8) != (by >> 8)) db_chunk(b, (by + 4) >> 8);
    b.win = ((unsigned long long)db_dword(b, by) << 32) | db_dword(b, by + 4);
}
__device__ __forceinline__ void db_skip(DecBits &b, unsigned n)  // n <= 32
{
    b.pos += n;
    if (__builtin_expect(b.pos - b.wbase >= 32, 0)) {  // (a lone wavefront pays ~20 cycles per taken branch: the common case falls through)
        const unsigned by = (b.wbase >> 3) + 8;
        if (__builtin_expect((by & 255u) == 0, 0)) db_chunk(b, by >> 8);
        b.win = (b.win << 32) | db_dword(b, by);
        b.wbase += 32;
    }
}
__device__ __forceinline__ unsigned db_peek(const DecBits &b, int n)  // 1 <= n <= 32
{
    return (unsigned)((b.win << (b.pos - b.wbase)) >> (64 - n));
}
__device__ __forceinline__ unsigned db_bits(DecBits &b, int n)
{
    if (n == 0) return 0;
    unsigned v = db_peek(b, n);
    db_skip(b, (unsigned)n);
    return v;
}
__device__ __forceinline__ unsigned db_bit(DecBits &b) { return db_bits(b, 1); }
__device__ __forceinline__ unsigned db_ue(DecBits &b)  // F/expgolomb.cpp:122-140
{
    unsigned w = db_peek(b, 24);
    int z = w ? __clz((int)w) - 8 : 24;
    db_skip(b, (unsigned)(z + 1));
    unsigned s = db_bits(b, z);
    return (1u << z) - 1u + s;
}
__device__ __forceinline__ int db_se(DecBits &b)
{
    int v = (int)db_ue(b);
    return (v & 1) ? (v + 1) / 2 : -v / 2;
}
// te(v) as the reference reads it, whatever the range (F/expgolomb.cpp:156-178): an Exp-Golomb prefix and suffix;
// a code number above 2 is returned as is, anything else (other than the single bit "1" = 0) is followed by one
// more bit whose inverse is the value.  The bits consumed are what matters here -- the value is never used.
__device__ __forceinline__ unsigned db_te(DecBits &b)
{
    unsigned w = db_peek(b, 24);
    int z = w ? __clz((int)w) - 8 : 24;
    db_skip(b, (unsigned)(z + 1));
    if (z == 0) return 0;
    unsigned x = db_bits(b, z);
    if (x > 1) return (1u << z) - 1u + x;
    return db_bit(b) ? 0u : 1u;
}
__device__ __forceinline__ bool db_more(const DecBits &b) { return (b.pos >> 3) + 1 < b.size; }

// ---- decode tables: the code tables of F/residual_tables.cpp inverted once per process into direct
// look-ups by the next bits of the stream (coeff_token: 16 bits, total_zeros: 9, run_before: 3).
// entry = len << 8 | value, 0 = no code matches; coeff_token value = TotalCoeff << 2 | TrailingOnes.
#define DEC_CT_SUBS 48
// One coefficient level per look-up (level_prefix + level_suffix, F/residual.cpp:1183-1262) by the next DEC_LEV_BITS bits
// of the stream and the decoder state: states 0..6 = suffixLength, 7 / 8 = suffixLength 0 / 1 at the first level after
// fewer than three trailing ones (levelCode + 2).  entry = bits | next suffixLength << 4 | level << 7 (signed 9 bits),
// 0 = the code does not fit the window (long prefixes, escapes): the arithmetic path decodes it.
#define DEC_LEV_BITS 11
#define DEC_LEV_STATES 9
struct DecLuts {
    uint16_t ct[4][65536];   // class 0..2 by nC, 3 = chroma DC: the full 16-bit look-up (build step only)
    // what the parse kernel keeps in LDS: coeff_token in two levels of 8 bits
    uint16_t ct1[4][256];    // len <= 8: the entry; longer codes: 0x8000 | sub-table
    uint16_t ct2[DEC_CT_SUBS][256];
    uint16_t tz[15][512];    // by TotalCoeff - 1
    uint16_t tzdc[3][8];
    uint16_t rb[6][8];       // by zerosLeft - 1 (zerosLeft <= 6)
    uint16_t lev[DEC_LEV_STATES][1 << DEC_LEV_BITS];
    int nsub;
};
struct DecLutsLds {  // the same tables, resident in LDS for one parse wavefront
    uint16_t ct1[4][256];
    uint16_t ct2[DEC_CT_SUBS][256];
    uint16_t tz[15][512];
    uint16_t tzdc[3][8];
    uint16_t rb[6][8];
    uint16_t lev[DEC_LEV_STATES][1 << DEC_LEV_BITS];
};
static DecLuts *g_dec_luts = nullptr;

// second build step: split the 16-bit coeff_token table by its first 8 bits
__global__ void k_dec_split_luts(DecLuts *L)
{
    const int cls = blockIdx.x, p = threadIdx.x;  // 4 x 256
    const uint16_t e0 = L->ct[cls][p << 8];
    bool uniform = true;  // a code of <= 8 bits fills all 256 completions of its prefix
    for (int x = 1; x < 256; x++) uniform &= L->ct[cls][(p << 8) | x] == e0;
    if (uniform && (e0 == 0 || (e0 >> 8) <= 8)) {
        L->ct1[cls][p] = e0;
        return;
    }
    int id = atomicAdd(&L->nsub, 1);
    if (id >= DEC_CT_SUBS) {
        L->ct1[cls][p] = 0;
        return;
    }
    L->ct1[cls][p] = (uint16_t)(0x8000 | id);
    for (int x = 0; x < 256; x++) L->ct2[id][x] = L->ct[cls][(p << 8) | x];
}

__global__ void k_dec_build_luts(DecLuts *L)
{
    const unsigned v = blockIdx.x * blockDim.x + threadIdx.x;  // 16-bit window
    if (v < 65536) {
        for (int cls = 0; cls < 4; cls++) {
            uint16_t e = 0;
            const int maxtc = cls == 3 ? 4 : 16;
            for (int T = 0; T <= maxtc && !e; T++)
                for (int o = 0; o <= 3 && o <= T; o++) {
                    int len = cls == 3 ? c_ctdc_len[o][T] : c_ct_len[cls][o][T];
                    unsigned code = cls == 3 ? c_ctdc_code[o][T] : c_ct_code[cls][o][T];
                    if (len > 0 && (v >> (16 - len)) == code) {
                        e = (uint16_t)((len << 8) | (T << 2) | o);
                        break;
                    }
                }
            L->ct[cls][v] = e;
        }
    }
    if (v < (1u << DEC_LEV_BITS))
        for (int st = 0; st < DEC_LEV_STATES; st++) {
            const int sl = st < 7 ? st : st - 7;
            uint16_t e = 0;
            if (v != 0) {
                const int prefix = __clz((int)v) - (32 - DEC_LEV_BITS);  // < 14: no escape, the suffix has sl bits
                const int total = prefix + 1 + sl;
                if (total <= DEC_LEV_BITS) {
                    const int suffix = (int)((v >> (DEC_LEV_BITS - total)) & ((1u << sl) - 1u));
                    int levelCode = (prefix << sl) + suffix;
                    if (st >= 7) levelCode += 2;
                    const int lev = (levelCode & 1) == 0 ? (levelCode + 2) >> 1 : (-levelCode - 1) >> 1;
                    int nsl = sl == 0 ? 1 : sl;
                    if (iabs(lev) > (3 << (nsl - 1)) && nsl < 6) nsl++;
                    e = (uint16_t)(total | (nsl << 4) | ((lev & 0x1ff) << 7));
                }
            }
            L->lev[st][v] = e;
        }
    if (v < 512)
        for (int tcm1 = 0; tcm1 < 15; tcm1++) {
            uint16_t e = 0;
            for (int tz = 0; tz <= 15; tz++) {
                int len = c_tz_len[tcm1][tz];
                if (len > 0 && (v >> (9 - len)) == c_tz_code[tcm1][tz]) {
                    e = (uint16_t)((len << 8) | tz);
                    break;
                }
            }
            L->tz[tcm1][v] = e;
        }
    if (v < 8) {
        for (int tcm1 = 0; tcm1 < 3; tcm1++) {
            uint16_t e = 0;
            for (int tz = 0; tz <= 3; tz++) {
                int len = c_tzdc_len[tcm1][tz];
                if (len > 0 && (v >> (3 - len)) == c_tzdc_code[tcm1][tz]) {
                    e = (uint16_t)((len << 8) | tz);
                    break;
                }
            }
            L->tzdc[tcm1][v] = e;
        }
        for (int zl = 0; zl < 6; zl++) {
            uint16_t e = 0;
            for (int k = 0; k <= zl + 1; k++) {
                int len = c_rb_len[zl][k];
                if (len > 0 && (v >> (3 - len)) == c_rb_code[zl][k]) {
                    e = (uint16_t)((len << 8) | k);
                    break;
                }
            }
            L->rb[zl][v] = e;
        }
    }
}

// residual_block_cavlc, F/residual.cpp:1069-1386.  coef: int16 destination (maxNumCoeff entries, already zero).
// The levels of the block live in one vector register (lane i = level i), the runs in a scalar 64-bit word.  Returns TotalCoeff, or -1 on a malformed block.
__device__ int dec_block(DecBits &b, const DecLutsLds *L, int16_t *coef, int maxNumCoeff, int nC)
{
    int TotalCoeff, TrailingOnes = 0;
    if (nC >= 8) {
        unsigned v = db_bits(b, 6);
        if (v == 3) {
            TotalCoeff = 0;
        } else {
            TotalCoeff = (int)(v >> 2) + 1;
            TrailingOnes = (int)(v & 3);
        }
    } else {
        const int cls = nC == -1 ? 3 : (nC <= 1 ? 0 : (nC <= 3 ? 1 : 2));
        const unsigned w16 = db_peek(b, 16);
        unsigned e = (unsigned)__builtin_amdgcn_readfirstlane((int)L->ct1[cls][w16 >> 8]);
        if (e & 0x8000) e = (unsigned)__builtin_amdgcn_readfirstlane((int)L->ct2[e & 0x7fff][w16 & 0xff]);
        if (e == 0) return -1;
        db_skip(b, e >> 8);
        TotalCoeff = (int)((e >> 2) & 31);
        TrailingOnes = (int)(e & 3);
    }
    if (TotalCoeff == 0) return 0;
    if (TotalCoeff > maxNumCoeff) return -1;
    const int lane = (int)(threadIdx.x & 63);
    int lvv = 0;
    if (TrailingOnes) {  // the sign bits of the trailing ones, first coefficient first
        const unsigned sg = db_bits(b, TrailingOnes);
        lvv = 1 - 2 * (int)((sg >> ((TrailingOnes - 1 - lane) & 31)) & 1u);
    }
    int suffixLength = (TotalCoeff > 10 && TrailingOnes < 3) ? 1 : 0;
    int st = TrailingOnes < 3 ? 7 + suffixLength : suffixLength;
    for (int i = TrailingOnes; i < TotalCoeff; i++) {
        int lev;
        const unsigned e = (unsigned)__builtin_amdgcn_readfirstlane((int)L->lev[st][db_peek(b, DEC_LEV_BITS)]);
        if (__builtin_expect(e != 0, 1)) {
            db_skip(b, e & 15u);
            lev = ((int)(e << 16)) >> 23;
            suffixLength = (int)((e >> 4) & 7u);
        } else {
            const unsigned w = db_peek(b, 32);
            if (w == 0) return -1;  // level_prefix beyond 31
            const int prefix = __clz((int)w);
            db_skip(b, (unsigned)(prefix + 1));
            int size = (prefix == 14 && suffixLength == 0) ? 4 : (prefix >= 15 ? prefix - 3 : suffixLength);
            unsigned suffix = (size > 0 || prefix >= 14) ? db_bits(b, size) : 0;
            int levelCode = (min(prefix, 15) << suffixLength);
            if (size > 0 || prefix >= 14) levelCode += (int)suffix;
            if (prefix >= 15 && suffixLength == 0) levelCode += 15;
            if (st >= 7) levelCode += 2;
            lev = (levelCode & 1) == 0 ? (levelCode + 2) >> 1 : (-levelCode - 1) >> 1;
            if (suffixLength == 0) suffixLength = 1;
            if (iabs(lev) > (3 << (suffixLength - 1)) && suffixLength < 6) suffixLength++;
        }
        st = suffixLength;
        lvv = lane == i ? lev : lvv;
    }
    int zerosLeft = 0;
    if (TotalCoeff < maxNumCoeff) {
        const unsigned e = (unsigned)__builtin_amdgcn_readfirstlane(
            (int)(nC == -1 ? L->tzdc[TotalCoeff - 1][db_peek(b, 3)] : L->tz[TotalCoeff - 1][db_peek(b, 9)]));
        if (e == 0) return -1;
        db_skip(b, e >> 8);
        zerosLeft = (int)(e & 0xff);
    }
    unsigned long long runs = 0;  // run_before of coefficient j in bits 4j..4j+3 (scalar registers, no LDS round trips)
    int j = 0;
    for (; j < TotalCoeff - 1 && zerosLeft > 0; j++) {  // (no zeros left: every further run is 0)
        int rb;
        if (zerosLeft > 6) {
            rb = 7 - (int)db_bits(b, 3);
            if (rb == 7) {
                const unsigned w = db_peek(b, 32);
                if (w == 0) return -1;
                const int z = __clz((int)w);
                db_skip(b, (unsigned)(z + 1));
                rb += z;
            }
        } else {
            const unsigned e = (unsigned)__builtin_amdgcn_readfirstlane((int)L->rb[zerosLeft - 1][db_peek(b, 3)]);
            if (e == 0) return -1;
            db_skip(b, e >> 8);
            rb = (int)(e & 0xff);
        }
        runs |= (unsigned long long)(rb & 15) << (4 * j);
        zerosLeft -= rb;
    }
    runs |= (unsigned long long)(max(zerosLeft, 0) & 15) << (4 * (TotalCoeff - 1));
    {  // placement, one coefficient per lane: position of coefficient i = sum over k >= i of (run_k + 1), minus 1
        const int i = lane;
        if (i < TotalCoeff) {
            unsigned long long x = runs >> (4 * i);
            unsigned long long sm = (x & 0x0f0f0f0f0f0f0f0full) + ((x >> 4) & 0x0f0f0f0f0f0f0f0full);
            unsigned s32 = (unsigned)sm + (unsigned)(sm >> 32);
            int pos = (TotalCoeff - i) - 1 + (int)((s32 * 0x01010101u) >> 24);
            if (pos < maxNumCoeff) coef[pos] = (int16_t)lvv;
        }
    }
    return TotalCoeff;
}

// What the total-coefficient prediction needs from a decoded neighbour, kept in LDS: one entry per macroblock
// column holds the macroblock above until the current one replaces it (so entry x - 1 is the left neighbour).
struct DecNb {
    uint8_t tc[24];
    uint8_t cbpL, cbpC, skip, i4;  // i4: an Intra4x4 macroblock (its bottom-row prediction modes are in the i4row entry)
};
// nC of F/residual.cpp:424-538 (wave-uniform)
__device__ int dec_nC(const DecNb *row, int x, int y, bool luma, int blk, int plane, const uint8_t *tcur, int cbpL, int cbpC)
{
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
        if (x == 0)
            availA = false;
        else {
            const DecNb &m = row[x - 1];
            bool zero = luma ? ((m.cbpL & (1 << (bA / 4))) == 0) : ((m.cbpC & 2) == 0);
            if (!(m.skip || zero)) nA = luma ? m.tc[bA] : m.tc[16 + plane * 4 + bA];
        }
    } else {
        bool zero = luma ? ((cbpL & (1 << (bA / 4))) == 0) : ((cbpC & 2) == 0);
        if (!zero) nA = luma ? tcur[bA] : tcur[16 + plane * 4 + bA];
    }
    if (edgeB) {
        if (y == 0)
            availB = false;
        else {
            const DecNb &m = row[x];
            bool zero = luma ? ((m.cbpL & (1 << (bB / 4))) == 0) : ((m.cbpC & 2) == 0);
            if (!(m.skip || zero)) nB = luma ? m.tc[bB] : m.tc[16 + plane * 4 + bB];
        }
    } else {
        bool zero = luma ? ((cbpL & (1 << (bB / 4))) == 0) : ((cbpC & 2) == 0);
        if (!zero) nB = luma ? tcur[bB] : tcur[16 + plane * 4 + bB];
    }
    int r = 0;
    if (availA && availB)
        r = (nA + nB + 1) >> 1;
    else if (availA)
        r = nA;
    else if (availB)
        r = nB;
    return __builtin_amdgcn_readfirstlane(r);
}

// P macroblock vectors: PredictMV + DeriveMVs (F/mode_pred.cpp:381-482), quadrant storage
__device__ void dec_derive_mvs(const FerDev &d, short *mvs, const int *mbt, int mb, int type, const int mvd[4][2], const int sub[4])
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
        if (np == 4)
            predict_luma_quadrant(c, i, sub, px, py);
        else
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

// One wavefront per (stream, picture of the window).  What a picture inherits from its predecessor -- the
// mb_qp_delta that persists when absent and the ChromaACLevel block that persists into macroblocks without
// residual (reference quirks) -- changes no bit position, so every picture is parsed as if it inherited zeros;
// it reports how many macroblocks ran on the inherited delta and flags those that show the inherited block,
// k_dec_carry hands the true values down the pictures of a stream and k_dec_patch applies them.
// PW pictures (wavefronts) per workgroup share the decode tables in LDS; the launch takes the largest PW whose tables,
// per-wavefront scratch and neighbour rows fit the CU's 160 KB (16 at 1080p: the scalar chains of 16 pictures hide
// each other's table round trips, 8 left the scalar unit half idle).
#define DEC_WSYNC()                                        \
    do {                                                   \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                   \
    } while (0)
template <int DEC_PW>
__global__ __launch_bounds__(64 * DEC_PW) void k_dec_parse(FerDev d, DecBatch B, const DecLuts *__restrict__ luts)
{
    __shared__ uint8_t tcur_w[DEC_PW][24];
    __shared__ uint32_t ring_w[DEC_PW][128];  // the bit reader's window on the stream
    __shared__ int16_t cac_w[DEC_PW][2][4][16];  // ChromaACLevel persists across macroblocks (reference quirk)
    __shared__ __attribute__((aligned(16))) int16_t mblv_w[DEC_PW][FER_LEVELS];  // levels of the macroblock being parsed
    __shared__ DecLutsLds lut;
    extern __shared__ __attribute__((aligned(16))) uint8_t dyn_lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    DecNb *row = (DecNb *)dyn_lds + (size_t)wv * d.mbw;  // [mbw] per wavefront
    // Intra4x4PredMode of blocks 10, 11, 14, 15 of the macroblock above, 4 bits each: getIntra4x4PredMode reads its neighbours
    // from here and from scalar registers, not from the mode array in memory (16 dependent round trips per macroblock)
    uint16_t *i4row = (uint16_t *)((DecNb *)dyn_lds + (size_t)DEC_PW * d.mbw) + (size_t)wv * d.mbw;
    unsigned long long left_modes = 0;  // the modes of the last Intra4x4 macroblock parsed (the left neighbour when its i4 flag says so)
    uint8_t *tcur = tcur_w[wv];
    int16_t(*cac)[4][16] = cac_w[wv];
    int16_t *mblv = mblv_w[wv];
    {  // decode tables into LDS, once per workgroup
        const uint32_t *src = (const uint32_t *)&luts->ct1[0][0];
        uint32_t *dst = (uint32_t *)&lut;
        for (int i = threadIdx.x; i < (int)(sizeof(DecLutsLds) / 4); i += 64 * DEC_PW) dst[i] = src[i];
        __syncthreads();
    }
    // Pictures are handed out through a counter (B.state[2] of the window's first picture, cleared by the launch): the
    // I pictures of the window's first step -- the longest scalar chains of the launch -- go out first, one to each
    // workgroup in turn (so to different CUs), and a wavefront that has finished a short P picture takes the next one
    // instead of idling beside an I picture until the workgroup ends.
    for (;;) {
        size_t pic;
        {
            int tk = 0;
            if (lane == 0) tk = atomicAdd(&B.state[2], 1);
            tk = __builtin_amdgcn_readfirstlane(tk);
            if (tk >= (int)gridDim.x * DEC_PW) break;
            pic = (size_t)(tk % DEC_PW) * gridDim.x + (size_t)(tk / DEC_PW);  // ticket -> picture: DEC_PW consecutive tickets are gridDim.x pictures apart
        }
        if (pic >= (size_t)B.TW * d.S) continue;
        const int tpic = (int)(pic / d.S), s = (int)(pic % d.S);
        const uint32_t *info = B.info + pic * 6;
        // this picture's slice of the side information (the kernel argument d itself is never written: a modified copy of
        // that 900-byte structure would live in scratch memory)
        const size_t po = (size_t)tpic * d.S * d.nmb;
        int *const p_mb_type = B.mb_type + po;
        short *const p_mv = B.mv + po * 8;
        uint8_t *const p_cbp = B.cbp + po * 2, *const p_tc = B.tc + po * 24, *const p_i4mode = B.i4mode + po * 16;
        uint8_t *const p_i4flag = B.i4flag + po * 16, *const p_chroma_mode = B.chroma_mode + po, *const p_dec_qp = B.dec_qp + po;
        int16_t *const p_levels = B.levels + po * FER_LEVELS;
        uint8_t *carry = B.carry + pic * d.nmb;
        int *st = B.state + pic * 4, *summ = B.summ + pic * 4;
        // the slice parameters are the same in every lane: scalar registers, so that everything derived from them
        // (bit position, window, QP) stays on the scalar unit
        const unsigned i_size = (unsigned)__builtin_amdgcn_readfirstlane((int)info[0]);
        const unsigned i_pos = (unsigned)__builtin_amdgcn_readfirstlane((int)info[1]);
        const int i_slice = __builtin_amdgcn_readfirstlane((int)info[2]);
        const int stype = i_slice & 255;
        // ref_idx_l0 is parsed and dropped (every prediction uses the one stored picture): in sub_mb_pred the reference
        // reads it when the slice carried num_ref_idx_active_override_flag, in mb_pred when the active count left behind
        // by the last override is > 1 (F/rbsp_decoding.cpp:156-161, :217-221)
        const bool ref_sub = ((i_slice >> 8) & 1) != 0, ref_mb = (i_slice >> 16) > 0;
        const unsigned i_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)info[4]), i_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)info[5]);
        if (i_size == 0) {  // no picture for this stream at this step
            if (lane == 0) {
                st[1] = 0;
                summ[0] = summ[3] = 0;
            }
            continue;
        }
        const bool constrained_intra = __builtin_amdgcn_readfirstlane((int)B.hdr[pic * 4 + 1]) != 0;
        DecBits b;
        db_open(b, B.rbsp + (((size_t)i_hi << 32) | i_lo), i_size, i_pos, ring_w[wv]);
        int QPy = __builtin_amdgcn_readfirstlane((int)info[3]);
        int *mbt = p_mb_type + (size_t)s * d.nmb;
        short *mvs = p_mv + (size_t)s * d.nmb * 8;
        int mb_qp_delta = 0;  // the inherited value is added by k_dec_patch
        bool delta_known = false, cac_known = false;
        int n_inherit = 0;    // macroblocks whose QP step used the inherited delta
        for (int i = lane; i < 2 * 4 * 16; i += 64) (&cac[0][0][0])[i] = 0;
        DEC_WSYNC();
        int cur = 0;
        bool more = true;
        int mvd[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
        int sub[4] = {0, 0, 0, 0};
#ifdef FER_PROBE
        long long tacc[4] = {0, 0, 0, 0}, tmark = wall_clock64();
#define DP_MARK(k)                        \
        {                                     \
            long long now_ = wall_clock64();  \
            tacc[k] += now_ - tmark;          \
            tmark = now_;                     \
        }
#else
#define DP_MARK(k)
#endif
        while (more && cur < d.nmb) {
            DP_MARK(3)
            if (stype != 2) {
                int run = (int)db_ue(b);
                for (int i = 0; i < run && cur < d.nmb; i++) {
                    size_t mbi = (size_t)s * d.nmb + cur;
                    mbt[cur] = FER_P_SKIP;
                    if (lane == 0) {
                        row[cur % d.mbw].skip = 1;
                        row[cur % d.mbw].i4 = 0;
                    }
                    for (int k = 0; k < 4; k++) mvd[k][0] = mvd[k][1] = 0;  // ClearMVD in PredictMV
                    dec_derive_mvs(d, mvs, mbt, cur, FER_P_SKIP, mvd, sub);
                    QPy = (QPy + mb_qp_delta + 52) % 52;
                    n_inherit += delta_known ? 0 : 1;
                    p_dec_qp[mbi] = (uint8_t)QPy;
                    if (lane == 0) carry[cur] = 0;
                    cur++;
                }
                if (cur != 0 || run > 0) more = db_more(b);
            }
            if (!(more && cur < d.nmb)) break;
            const size_t mbi = (size_t)s * d.nmb + cur;
            const int mbx = cur % d.mbw, mby = cur / d.mbw;
            int16_t *lv = mblv;
            for (int i = lane; i < FER_LEVELS / 2; i += 64) ((uint32_t *)mblv)[i] = 0;
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
            if ((t == 25 && stype == 2) || (t == 30 && stype != 2)) {
                atomicOr(&d.status[s], FER_ERR_DEC_UNSUPPORTED);  // I_PCM
                break;
            }
            int chroma_mode = 0;
            unsigned long long i4flags = 0;  // prev_intra4x4_pred_mode_flag << 3 | rem_intra4x4_pred_mode, 4 bits per block
            bool left_i4 = false, above_i4 = false;
            unsigned above_modes = 0;
            if (inter) {
                if (t == 3 || t == 4) {
                    bool badsub = false;
                    for (int i = 0; i < 4; i++) {
                        sub[i] = (int)db_ue(b);
                        badsub |= sub[i] > 3;
                    }
                    if (badsub) {
                        atomicOr(&d.status[s], FER_ERR_DEC_SYNTAX);
                        break;
                    }
                    if (ref_sub && t != FER_P_8x8ref0)
                        for (int i = 0; i < 4; i++) db_te(b);
                    for (int i = 0; i < 4; i++) {
                        // sub-partitions: 8x8 1, 8x4 2, 4x8 2, 4x4 4 vector differences, of which DeriveMVs keeps the first
                        const int nsub = sub[i] == 0 ? 1 : (sub[i] == 3 ? 4 : 2);
                        for (int j = 0; j < nsub; j++) {
                            int dx = db_se(b), dy = db_se(b);
                            if (j == 0) {
                                mvd[i][0] = dx;
                                mvd[i][1] = dy;
                            }
                        }
                    }
                } else {
                    int np = t == 0 ? 1 : 2;
                    if (ref_mb)
                        for (int i = 0; i < np; i++) db_te(b);
                    for (int i = 0; i < np; i++) {
                        mvd[i][0] = db_se(b);
                        mvd[i][1] = db_se(b);
                    }
                }
            } else {
                if (i4) {
                    for (int blk = 0; blk < 16; blk++) {
                        int f = (int)db_bit(b);
                        int rem = f ? 0 : (int)db_bits(b, 3);
                        i4flags |= (unsigned long long)((f << 3) | rem) << (4 * blk);
                    }
                    if (lane < 16) p_i4flag[mbi * 16 + lane] = (uint8_t)((i4flags >> (4 * lane)) & 15);
                    // the neighbours' side of getIntra4x4PredMode, before this macroblock takes the place of the one above
                    left_i4 = mbx > 0 && __builtin_amdgcn_readfirstlane((int)row[mbx - 1].i4) != 0;
                    above_i4 = mby > 0 && __builtin_amdgcn_readfirstlane((int)row[mbx].i4) != 0;
                    above_modes = (unsigned)__builtin_amdgcn_readfirstlane((int)i4row[mbx]);
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
            p_cbp[mbi * 2] = (uint8_t)cbpL;
            p_cbp[mbi * 2 + 1] = (uint8_t)cbpC;
            p_chroma_mode[mbi] = (uint8_t)chroma_mode;
            for (int i = lane; i < 24; i += 64) tcur[i] = 0;
            DEC_WSYNC();
            DP_MARK(0)
            bool bad = false;
            if (cbpL > 0 || cbpC > 0 || i16) {
                mb_qp_delta = db_se(b);
                delta_known = true;
                if (mb_qp_delta < -26 || mb_qp_delta > 25) bad = true;
                // residual(0,15), F/residual.cpp:959-1067
                if (i16 && !bad) {
                    int n = dec_block(b, &lut, lv + FER_LV_DC16, 16, dec_nC(row, mbx, mby, true, 0, 0, tcur, cbpL, cbpC));
                    bad |= n < 0;
                    if (n >= 0) tcur[0] = (uint8_t)n;
                    DEC_WSYNC();
                }
                for (int i8 = 0; i8 < 4 && !bad; i8++)
                    if (cbpL & (1 << i8))
                        for (int i4x = 0; i4x < 4 && !bad; i4x++) {
                            int blk = i8 * 4 + i4x;
                            int n = dec_block(b, &lut, lv + blk * 16, i16 ? 15 : 16, dec_nC(row, mbx, mby, true, blk, 0, tcur, cbpL, cbpC));
                            bad |= n < 0;
                            if (n >= 0) tcur[blk] = (uint8_t)n;
                            DEC_WSYNC();
                        }
                for (int pl = 0; pl < 2 && !bad; pl++)
                    if (cbpC & 3) bad |= dec_block(b, &lut, lv + FER_LV_CDC + pl * 4, 4, -1) < 0;
                for (int pl = 0; pl < 2 && !bad; pl++)
                    for (int cb = 0; cb < 4 && !bad; cb++) {
                        if (cbpC & 2) {
                            for (int i = lane; i < 16; i += 64) cac[pl][cb][i] = 0;
                            DEC_WSYNC();
                            int n = dec_block(b, &lut, &cac[pl][cb][0], 15, dec_nC(row, mbx, mby, false, cb, pl, tcur, cbpL, cbpC));
                            bad |= n < 0;
                            if (n >= 0) tcur[16 + pl * 4 + cb] = (uint8_t)n;
                            DEC_WSYNC();
                        } else {
                            for (int i = lane; i < 16; i += 64) cac[pl][cb][i] = 0;
                            DEC_WSYNC();
                        }
                    }
            }
            DP_MARK(1)
            if (bad) {
                atomicOr(&d.status[s], FER_ERR_DEC_SYNTAX);
                break;
            }
            // chroma AC of this macroblock = the persistent ChromaACLevel (stale when cbp == 0)
            if (cbpL > 0 || cbpC > 0 || i16) cac_known = true;  // every block was parsed or cleared above
            if (lane == 0) carry[cur] = cac_known ? 0 : 1;
            for (int i = lane; i < 120; i += 64) lv[FER_LV_CAC + i] = cac[i / 60][(i % 60) / 15][i % 15];
            DEC_WSYNC();
            {  // the finished macroblock: levels to memory (one coalesced pass), counts into the neighbour row
                uint32_t *g = (uint32_t *)(p_levels + mbi * FER_LEVELS);
                for (int i = lane; i < FER_LEVELS / 2; i += 64) g[i] = ((const uint32_t *)mblv)[i];
                DecNb &me = row[mbx];
                if (lane < 24) {
                    p_tc[mbi * 24 + lane] = tcur[lane];
                    me.tc[lane] = tcur[lane];
                }
                if (lane == 0) {
                    me.cbpL = (uint8_t)cbpL;
                    me.cbpC = (uint8_t)cbpC;
                    me.skip = 0;
                    me.i4 = i4 ? 1 : 0;
                }
            }
            DEC_WSYNC();
            QPy = (QPy + mb_qp_delta + 52) % 52;
            n_inherit += delta_known ? 0 : 1;
            p_dec_qp[mbi] = (uint8_t)QPy;
            if (inter) {
                dec_derive_mvs(d, mvs, mbt, cur, t, mvd, sub);
            } else if (i4) {
                // getIntra4x4PredMode, F/intra.cpp:77-136
                unsigned long long cm = 0;  // this macroblock's modes, 4 bits per block
                for (int blk = 0; blk < 16; blk++) {
                    bool edgeA = blk == 0 || blk == 2 || blk == 8 || blk == 10;
                    bool edgeB = blk == 0 || blk == 1 || blk == 4 || blk == 5;
                    bool okA = !(edgeA && mbx == 0), okB = !(edgeB && mby == 0);
                    int mA = 2, mB = 2;
                    if (okA && okB && !constrained_intra) {
                        // c_nbA / c_nbB (the neighbouring block to the left / above), 4 bits per block: no table load in the chain
                        const int nA = (int)((0xebc9af8d63412705ull >> (4 * blk)) & 15), nB = (int)((0xdc76983254fe10baull >> (4 * blk)) & 15);
                        if (edgeA)
                            mA = left_i4 ? (int)((left_modes >> (4 * nA)) & 15) : 2;
                        else
                            mA = (int)((cm >> (4 * nA)) & 15);
                        if (edgeB)
                            mB = above_i4 ? (int)((above_modes >> (4 * ((nB & 1) | ((nB >> 1) & 2)))) & 15) : 2;  // 10, 11, 14, 15 -> 0 .. 3
                        else
                            mB = (int)((cm >> (4 * nB)) & 15);
                    }
                    int pm = mA <= mB ? mA : mB;
                    int f = (int)((i4flags >> (4 * blk)) & 15);
                    int mode = (f & 8) ? pm : ((f & 7) < pm ? (f & 7) : (f & 7) + 1);
                    cm |= (unsigned long long)mode << (4 * blk);
                }
                if (lane < 16) p_i4mode[mbi * 16 + lane] = (uint8_t)((cm >> (4 * lane)) & 15);
                if (lane == 0) i4row[mbx] = (uint16_t)(((cm >> 40) & 0xff) | (((cm >> 56) & 0xff) << 8));  // blocks 10, 11 | 14, 15
                left_modes = cm;
            }
            more = db_more(b);
            cur++;
            DP_MARK(2)
        }
#ifdef FER_PROBE
        if (s == 0 && lane == 0) {
            for (int k = 0; k < 4; k++) d.timing[48 + k] += tacc[k];
            d.timing[52] += cur;
        }
#endif
#undef DP_MARK
        DEC_WSYNC();
        for (int i = lane; i < 128; i += 64) B.cac_out[pic * 128 + i] = (&cac[0][0][0])[i];
        if (lane == 0) {
            st[1] = cur;  // macroblocks reached (the rest of the picture keeps the previous content)
            summ[0] = delta_known;
            summ[1] = mb_qp_delta;
            summ[2] = n_inherit;
            summ[3] = cac_known;
        }
        DEC_WSYNC();
    }  // next picture
}

// per stream, down the pictures of the window: what each picture inherits (FerDev.dec_state / dec_cac hold the
// state between windows)
__global__ __launch_bounds__(128) void k_dec_carry(FerDev d, DecBatch B)
{
    const int s = blockIdx.x, i = threadIdx.x;  // thread i carries ChromaACLevel entry i
    int delta = d.dec_state[(size_t)s * 4];
    int16_t v = d.dec_cac[(size_t)s * 128 + i];
    for (int t = 0; t < B.TW; t++) {
        const size_t pic = (size_t)t * d.S + s;
        if (B.info[pic * 6] == 0) continue;
        B.cac_in[pic * 128 + i] = v;
        if (i == 0) B.state[pic * 4] = delta;
        if (B.summ[pic * 4 + 0]) delta = B.summ[pic * 4 + 1];
        if (B.summ[pic * 4 + 3]) v = B.cac_out[pic * 128 + i];
    }
    d.dec_cac[(size_t)s * 128 + i] = v;
    if (i == 0) d.dec_state[(size_t)s * 4] = delta;
}

// QP of the macroblocks that ran on the inherited mb_qp_delta, chroma AC of those that show the inherited block
__global__ __launch_bounds__(256) void k_dec_patch(FerDev d, DecBatch B)
{
    const int s = blockIdx.y, t = blockIdx.z;
    const int mb = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t pic = (size_t)t * d.S + s;
    if (mb >= B.state[pic * 4 + 1]) return;
    const size_t mbi = pic * d.nmb + mb;
    const int delta = B.state[pic * 4];
    if (delta != 0) {
        int steps = min(mb + 1, B.summ[pic * 4 + 2]);
        int q = ((int)B.dec_qp[mbi] + steps * delta) % 52;
        B.dec_qp[mbi] = (uint8_t)(q < 0 ? q + 52 : q);
    }
    if (B.carry[mbi]) {
        int16_t *lv = B.levels + mbi * FER_LEVELS + FER_LV_CAC;
        const int16_t *c = B.cac_in + pic * 128;
        for (int i = 0; i < 120; i++) lv[i] = c[(i / 60) * 64 + ((i % 60) / 15) * 16 + i % 15];
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

// chroma_qp_index_offset and constrained_intra_pred_flag belong to the stream's PPS: they travel with the picture in
// hdr[s][0] and hdr[s][1] (hdr[s][3] = slice type)
__device__ __forceinline__ int dec_qpc(const FerDev &d, int s, int QPy)
{
    const int qpc[52] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17,
                         18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31, 32, 32, 33,
                         34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39};
    return qpc[iclamp(QPy + (int)d.hdr[s * 4], 0, 51)];
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
    recon_chroma(d, lv, lane, dec_qpc(d, s, QPy), pC, C0, C1, Wc, xp, yp);
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
        // Intra4x4: the blocks reconstruct one after the other (each predicts from the ones before it); four lanes per
        // block, lane r < 4 = row r (inv_row: the vertical half of the inverse transform crosses the quad with DPP),
        // the levels of the macroblock staged in LDS by one coalesced read
        ((uint32_t *)&L.lv4[0][0])[lane] = ((const uint32_t *)lv)[lane];
        ((uint32_t *)&L.lv4[0][0])[64 + lane] = ((const uint32_t *)lv)[64 + lane];
        __syncthreads();
        const int row = lane & 3;
        RowQ rq;
        rq.row = row;
        {
            const int m = QPy % 6;
            const int s_ee = level_scale(m, 0, 0), s_oo = level_scale(m, 1, 1), s_eo = level_scale(m, 0, 1);
            rq.ls[0] = rq.ls[2] = (row & 1) ? s_eo : s_ee;
            rq.ls[1] = rq.ls[3] = (row & 1) ? s_oo : s_eo;
        }
        const uint32_t zrow = ((const uint32_t *)c_izz)[row];
        for (int blk = 0; blk < 16; blk++) {
            int p[13], o[16], c[4], r[4];
            fetch4(L, blk, lastcol, p);
            pred4x4(d.i4mode[mbi * 16 + blk], p, o);
#pragma unroll
            for (int k = 0; k < 4; k++) c[k] = L.lv4[blk][(zrow >> (8 * k)) & 15u];
            inv_row(rq, c, QPy, false, r);
            const int x0 = c_bx[blk], y0 = c_by[blk];
            if (lane < 4) {
                uint32_t pk = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int po = row == 0 ? o[k] : (row == 1 ? o[4 + k] : (row == 2 ? o[8 + k] : o[12 + k]));
                    const int v = clip255(po + r[k]);
                    L.fr[1 + y0 + row][1 + x0 + k] = (int16_t)v;
                    pk |= (uint32_t)v << (8 * k);
                }
                *(uint32_t *)(Y + (size_t)(yp + y0 + row) * W + xp + x0) = pk;
            }
            __syncthreads();
        }
    }
    __syncthreads();
    recon_chroma(d, lv, lane, dec_qpc(d, s, QPy), &L.predC[0][0][0], Cp[0], Cp[1], Wc, xp, yp);
}

void fer_launch_decode_parse(const FerDev &d, const DecBatch &B, hipStream_t st)
{
    if (!g_dec_luts) {  // once per process: invert the code tables (the buffer lives until exit)
        if (hipMalloc((void **)&g_dec_luts, sizeof(DecLuts)) != hipSuccess) return;
        hipMemsetAsync(g_dec_luts, 0, sizeof(DecLuts), st);
        hipLaunchKernelGGL(k_dec_build_luts, dim3(256), dim3(256), 0, st, g_dec_luts);
        hipLaunchKernelGGL(k_dec_split_luts, dim3(4), dim3(256), 0, st, g_dec_luts);
    }
    // static LDS of k_dec_parse<PW>: the tables + PW * (tcur 24 + stream ring 512 + ChromaACLevel 256 + the macroblock's levels); dynamic: the rows
    const size_t fixed = sizeof(DecLutsLds), per_wave = 24 + 512 + 256 + FER_LEVELS * 2 + (size_t)d.mbw * (sizeof(DecNb) + 2);
    const int npic = d.S * B.TW;
#define DEC_PARSE_LAUNCH(PW)                                                                                                          \
    hipLaunchKernelGGL(k_dec_parse<PW>, dim3((npic + PW - 1) / PW), dim3(64 * PW), (size_t)PW * d.mbw * (sizeof(DecNb) + 2), st, d, B, \
                       g_dec_luts)
    const size_t lds = 160 * 1024 - 512;
    hipMemsetAsync(B.state + 2, 0, sizeof(int), st);
    if (fixed + 16 * per_wave <= lds)
        DEC_PARSE_LAUNCH(16);
    else if (fixed + 8 * per_wave <= lds)
        DEC_PARSE_LAUNCH(8);
    else if (fixed + 4 * per_wave <= lds)
        DEC_PARSE_LAUNCH(4);
    else
        DEC_PARSE_LAUNCH(1);
#undef DEC_PARSE_LAUNCH
    hipLaunchKernelGGL(k_dec_carry, dim3(d.S), dim3(128), 0, st, d, B);
    hipLaunchKernelGGL(k_dec_patch, dim3((d.nmb + 255) / 256, d.S, B.TW), dim3(256), 0, st, d, B);
}

// reconstruction of one picture per stream; dslice = FerDev whose side-information pointers are the picture's slice
void fer_launch_decode_recon(const FerDev &d, bool anyP, bool anyIntra, hipStream_t st)
{
    if (anyP) hipLaunchKernelGGL(k_dec_inter, dim3(d.nmb, d.S), dim3(64), 0, st, d);
    if (anyIntra) {
        int ndiag = d.mbw + 2 * (d.mbh - 1);
        int maxk = min(d.mbh, (d.mbw + 1) / 2);
        for (int dg = 0; dg < ndiag; dg++) hipLaunchKernelGGL(k_dec_intra, dim3(maxk, d.S), dim3(64), 0, st, d, dg);
    }
}
