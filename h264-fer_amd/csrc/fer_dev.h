// fer_dev.h -- device-side layout and shared device helpers of libferhip (gfx950).
//
// One encoder context drives S independent streams ("shards": closed GOPs that the
// reference would encode as separate runs).  Everything a picture needs lives in HBM
// as structure-of-arrays indexed [stream][macroblock]; the reference keeps the same
// information in process globals (F/h264_globals.cpp:174-186, F/mode_pred.cpp:16-17,
// F/residual.cpp:10-17, F/moestimation.cpp:21-27).  F/ = fer_h264/fer_h264/ of the
// reference tree.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifdef FER_PROBE
#define FER_DBGF(d, m) ((d).dbg & (m))
#else
#define FER_DBGF(d, m) 0
#endif

#define FER_P_L0_16x16 0
#define FER_P_16x8 1
#define FER_P_8x16 2
#define FER_P_8x8 3
#define FER_P_8x8ref0 4
#define FER_P_SKIP 31
#define FER_MV_NA ((int)0x80808080u)

#define FER_IP_L 16  // margins of the interpolated planes: left / right / top / bottom
#define FER_IP_R 16
#define FER_IP_T 4
#define FER_IP_B 12
#define FER_BIGS 3            // big bucket slices of a crowded partition that are described by class (more = the general bound)
#define FER_P2_CAP 320        // listed candidates of a crowded partition (slots 64 .. 383 of its stage-2 list)
#define FER_OUTL 512          // outliers of a big bucket that are listed (more = the bucket has no modal class)
#define FER_BRANGE_MIN 1024  // buckets with more positions than this get feature ranges (FerDev.brange)
#define FER_ST2_CAP 384  // stage-2 candidates kept per 8x8 partition
#define FER_BIG_SLICE 1024  // records of one bucket inside a partition's column range beyond which k_me_walk bounds instead of reading
#define FER_LEVELS 400   // int16 per MB: luma 16x16, dc16 16, cdc 2x4, cac 2x4x15
#define FER_LV_DC16 256
#define FER_LV_CDC 272
#define FER_LV_CAC 280

struct FerDev {
    int W, H, Wc, Hc, mbw, mbh, nmb, S;
    int qp, qpc, window, maxdiff_set, basic;
    int16_t lsq[2][6];  // [luma qp / chroma qpc][LevelScale of class (even,even), (odd,odd), mixed; then LevelQuantize likewise]
    int dbg;  // -DFER_PROBE builds only (env FER_DBG): bit mask that skips kernel stages for timing; the shipped
              // library compiles every test of it away (FER_DBGF)
    size_t ysz, csz;
    // pictures: cur = `frame` (source in, reconstruction out), ref = `dpb`
    uint8_t *curY, *curCb, *curCr;
    uint8_t *refY, *refCb, *refCr;
    // a16: 16 quarter-pel planes, 5 box features per plane, positions sorted by 8x8 sum
    // 16 quarter-pel planes of the reference luma, each with a replicated margin (FER_IP_* pixels): a block that
    // hangs over the right / bottom picture edge reads what the reference's per-sample clamp (F/moestimation.cpp:
    // 107-115,189-190) would give without any clamp, and 16-byte row loads around any valid position stay inside
    // the plane.  Pixel (x, y) of plane f of stream s: interp + (s * 16 + f) * iplane + ioff + y * ipitch + x.
    uint8_t *interp;
    int ipitch, ioff;
    size_t iplane;
    uint16_t *feat;      // [S][H][W][16][6]  k0..k4 + pad per (position, frac): one 12-byte load per candidate
    uint16_t *feat0;     // [S][H][W][6]      plane-0 copy for the wide integer search
    uint32_t *sort_pos;  // [S][W*H]  (tx << 16) | ty, ordered by (sum, tx, ty)
    uint32_t *sort_rec;  // [S*W*H][3] {(tx << 16) | ty, kar1 | kar2 << 16, kar3 | kar4 << 16}: what the bucket walk reads
    // Two-level bucket index over the device-wide sorted order (stream-major): kol2[(s*16384 + a)*kt + t] = index of
    // the first sorted record of stream s with sum a and tx >= t << ktw_shift (a tile of columns); the entry at
    // t == kt is the next bucket's first.  koliko[a] of the reference = kol2[(s*16384 + a)*kt] - s*W*H.
    uint32_t *kol2;      // [S*16384*kt + 1]
    int kt, ktw_shift;
    // per bucket (8x8 sum) the ranges of the other four sums over its positions: [S][16384][8] = max of k1..k4, then max of
    // 65535 - k1..k4 (so that one atomicMax and a zero fill serve both ends), for buckets of more than FER_BRANGE_MIN
    // positions; all zero = no ranges.  Lower bounds of the feature distance for crowded partitions (k_me_walk).
    uint32_t *brange;
    // A bucket of more than FER_BRANGE_MIN positions is usually ONE flat area plus a few stray positions of the same
    // sum: bmodal[S][16384][4] = the two feature dwords of the record in the middle of the bucket (its "modal class"),
    // the number of records that differ from it, unused; boutl[S][nlists][FER_OUTL] = the sorted-array indices of those
    // records (any order), list number = the bucket's first place in the stream's sorted order / FER_BRANGE_MIN.  With at most FER_OUTL outliers the class is bounded EXACTLY by its one
    // feature distance and the outliers are listed as candidates of their own (k_me_walk, resolve_crowded).
    uint32_t *bmodal, *boutl;
    int *nbig;           // [S] == serial: the stream's reference picture has big buckets (set by k_sort_index)
    int nlists;          // outlier lists per stream = W*H / FER_BRANGE_MIN + 1
    int *zero_cnt;       // [S] positions of the reference picture whose 8x8 sum is 0 (see "bucket 0" in k_sort_finish)
    // per-MB side information (a20)
    int *mb_type;        // [S][nmb]
    int *prev_mb_type;   // [S][nmb] mb_type_array left by the previous picture
    short *mv;           // [S][nmb][4][2] one MV per 8x8 quadrant
    short *mvd;          // [S][nmb][4][2]
    uint8_t *cbp;        // [S][nmb][2] luma, chroma
    uint8_t *tc;         // [S][nmb][24] TotalCoeff: luma 16, Cb 4, Cr 4
    uint8_t *i4mode;     // [S][nmb][16]
    uint8_t *i4flag;     // [S][nmb][16] bit3 = prev_intra4x4_pred_mode_flag, bits0-2 = rem
    uint8_t *chroma_mode;  // [S][nmb]
    int16_t *levels;     // [S][nmb][FER_LEVELS]
    int *mbsize;         // [S][nmb][2] what coded_mb_size returned for the Intra16x16 and the Intra4x4 alternative (I pictures; test read-back)
    // motion-search precompute (neighbour independent)
    int *suma;           // [S][nmb][4][5]
    int *st3;            // [S][nmb][4][33][3] bx, by, sad
    int *st3n;           // [S][nmb][4]
    int *st2;            // [S][nmb][4][CAP][2] (tmpx & 0xffff) | tmpy << 16, D
    int *st2n;           // [S][nmb][4]
    // speculation on the predictor (k_me_spec -> k_me_resolve): v0 = the stage-3 survivor of smallest SAD, a guess of the
    // partition's final vector that needs no neighbour; from the neighbours' v0 every partition gets a guessed predictor,
    // and the two predictor-dependent searches are run for it in a fully parallel launch.  The chain only has to check
    // the guess against the true predictor and price the stored lists.
    int *v0;             // [S][nmb][4] packed vector
    int4 *spec_hdr;      // [S][nmb][4]: x = guessed integer centre (genx & 0xffff | geny << 16), y = cnt1 | cnt2 << 8 |
                         // lists valid << 16 | P_Skip verdict valid << 17, z = guessed P_Skip vector (partition 0), w = its verdict
    int2 *spec_l1;       // [S][nmb][4][17] stage-1 list: (packed vector, SAD)
    int2 *spec_l2;       // [S][nmb][4][33] stage-2 list
    unsigned long long *spec_stat;  // [8] partitions the chain decided, hits, P_Skip verdicts needed, taken from the guess
    int speculate;       // 0 = the chain searches everything itself (ferhip_tune; results do not depend on it)
    int *chain;          // row ticket of k_me_resolve
    unsigned long long *chain64;  // [S][nmb][4] vector | picture serial << 32 | P_Skip << 63, see k_me_resolve
    long long *timing;   // [64] in-kernel wall-clock sums of one probe wavefront (FER_DBG bit 7)
    int serial;          // serial number of the picture being encoded (never 0)
    int resolve_wgs;     // workgroups of the persistent k_me_resolve launch (ferhip_tune)
    int resolve_group;   // streams per ticket group of k_me_resolve (ferhip_tune)
    // entropy coding
    uint32_t *mb_bits;   // [S][nmb+1] bit sizes, then exclusive offsets
    uint32_t *bits;      // [S][bits_cap_words] RBSP, big-endian bit order
    size_t bits_cap_words;
    uint32_t *hdr;       // [S][4] slice header: bits hi, bits lo, nbits, slice_type(0=P,2=I)
    uint32_t *out_bytes; // [S] RBSP length
    int *status;         // [S] sticky error flags
    unsigned long long *sad; // [S] frame SAD for the IDR decision
    int *stats;          // [S][5] brojTipova
    // decoder (row a19)
    uint8_t *dec_qp;     // [S][nmb] QPy of every macroblock
    int *dec_state;      // [S][4]: [0] mb_qp_delta carried from picture to picture; the reconstruction kernels are given the
                         // per-picture state here ([1] = macroblocks the parser reached)
    int16_t *dec_cac;    // [S][2][4][16] persistent ChromaACLevel
};

// One window of pictures of the decode twin: slice data of TW pictures x S streams is parsed in one launch, so the
// per-picture side information carries a leading [TW] dimension; slice t of every array is exactly what the
// reconstruction kernels see through FerDev ([S][nmb]...).
struct DecBatch {
    int TW;
    int *mb_type;
    short *mv;
    uint8_t *cbp, *tc, *i4mode, *i4flag, *chroma_mode;
    int16_t *levels;
    uint8_t *dec_qp;
    uint8_t *carry;       // [TW][S][nmb] 1 = the macroblock's chroma AC levels are the block carried into its picture
    uint32_t *hdr;        // [TW][S][4] like FerDev.hdr (slice type in [3])
    int *state;           // [TW][S][4]: [0] mb_qp_delta carried into the picture, [1] macroblocks reached
    int *summ;            // [TW][S][4]: parsed a mb_qp_delta?, its last value, macroblocks before the first one, wrote chroma AC?
    int16_t *cac_in;      // [TW][S][128] ChromaACLevel carried into the picture
    int16_t *cac_out;     // [TW][S][128] ... left behind by it
    const uint8_t *rbsp;  // all slices of the window
    const uint32_t *info; // [TW][S][6]: bytes, first bit of slice_data, slice_type % 5, SliceQPy, byte offset lo, hi
};

// bits 0 and 1 (stage-2 list overflow, zero-sum blocks) are no longer raised: both cases are handled exactly
#define FER_ERR_BITS_OVERFLOW 4
#define FER_ERR_DEC_SYNTAX 8
#define FER_ERR_DEC_UNSUPPORTED 16
#define FER_ERR_CHAIN_TIMEOUT 32
#define FER_ERR_CHAIN_UNRESOLVED 64  // a P macroblock reached the residual stage without this picture's vectors

// ---------------------------------------------------------------- tables
// CAVLC tables: H.264 Tables 9-5, 9-7..9-10 as (length, code); zig-zag; block origins.
static __constant__ uint8_t c_ct_len[3][4][17] = {
    {{1, 6, 8, 9, 10, 11, 13, 13, 13, 14, 14, 15, 15, 16, 16, 16, 16},
     {0, 2, 6, 8, 9, 10, 11, 13, 13, 14, 14, 15, 15, 15, 16, 16, 16},
     {0, 0, 3, 7, 8, 9, 10, 11, 13, 13, 14, 14, 15, 15, 16, 16, 16},
     {0, 0, 0, 5, 6, 7, 8, 9, 10, 11, 13, 14, 14, 15, 15, 16, 16}},
    {{2, 6, 6, 7, 8, 8, 9, 11, 11, 12, 12, 12, 13, 13, 13, 14, 14},
     {0, 2, 5, 6, 6, 7, 8, 9, 11, 11, 12, 12, 13, 13, 14, 14, 14},
     {0, 0, 3, 6, 6, 7, 8, 9, 11, 11, 12, 12, 13, 13, 13, 14, 14},
     {0, 0, 0, 4, 4, 5, 6, 6, 7, 9, 11, 11, 12, 13, 13, 13, 14}},
    {{4, 6, 6, 6, 7, 7, 7, 7, 8, 8, 9, 9, 9, 10, 10, 10, 10},
     {0, 4, 5, 5, 5, 5, 6, 6, 7, 8, 8, 9, 9, 9, 10, 10, 10},
     {0, 0, 4, 5, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 10},
     {0, 0, 0, 4, 4, 4, 4, 4, 5, 6, 7, 8, 8, 9, 10, 10, 10}}};
static __constant__ uint8_t c_ct_code[3][4][17] = {
    {{1, 5, 7, 7, 7, 7, 15, 11, 8, 15, 11, 15, 11, 15, 11, 7, 4},
     {0, 1, 4, 6, 6, 6, 6, 14, 10, 14, 10, 14, 10, 1, 14, 10, 6},
     {0, 0, 1, 5, 5, 5, 5, 5, 13, 9, 13, 9, 13, 9, 13, 9, 5},
     {0, 0, 0, 3, 3, 4, 4, 4, 4, 4, 12, 12, 8, 12, 8, 12, 8}},
    {{3, 11, 7, 7, 7, 4, 7, 15, 11, 15, 11, 8, 15, 11, 7, 9, 7},
     {0, 2, 7, 10, 6, 6, 6, 6, 14, 10, 14, 10, 14, 10, 11, 8, 6},
     {0, 0, 3, 9, 5, 5, 5, 5, 13, 9, 13, 9, 13, 9, 6, 10, 5},
     {0, 0, 0, 5, 4, 6, 8, 4, 4, 4, 12, 8, 12, 12, 8, 1, 4}},
    {{15, 15, 11, 8, 15, 11, 9, 8, 15, 11, 15, 11, 8, 13, 9, 5, 1},
     {0, 14, 15, 12, 10, 8, 14, 10, 14, 14, 10, 14, 10, 7, 12, 8, 4},
     {0, 0, 13, 14, 11, 9, 13, 9, 13, 10, 13, 9, 13, 9, 11, 7, 3},
     {0, 0, 0, 12, 11, 10, 9, 8, 13, 12, 12, 12, 8, 12, 10, 6, 2}}};
static __constant__ uint8_t c_ctdc_len[4][5] = {{2, 6, 6, 6, 6}, {0, 1, 6, 7, 8}, {0, 0, 3, 7, 8}, {0, 0, 0, 6, 7}};
static __constant__ uint8_t c_ctdc_code[4][5] = {{1, 7, 4, 3, 2}, {0, 1, 6, 3, 3}, {0, 0, 1, 2, 2}, {0, 0, 0, 5, 0}};
static __constant__ uint8_t c_tz_len[15][16] = {
    {1, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 9}, {3, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 6, 6, 6, 6, 0},
    {4, 3, 3, 3, 4, 4, 3, 3, 4, 5, 5, 6, 5, 6, 0, 0}, {5, 3, 4, 4, 3, 3, 3, 4, 3, 4, 5, 5, 5, 0, 0, 0},
    {4, 4, 4, 3, 3, 3, 3, 3, 4, 5, 4, 5, 0, 0, 0, 0}, {6, 5, 3, 3, 3, 3, 3, 3, 4, 3, 6, 0, 0, 0, 0, 0},
    {6, 5, 3, 3, 3, 2, 3, 4, 3, 6, 0, 0, 0, 0, 0, 0}, {6, 4, 5, 3, 2, 2, 3, 3, 6, 0, 0, 0, 0, 0, 0, 0},
    {6, 6, 4, 2, 2, 3, 2, 5, 0, 0, 0, 0, 0, 0, 0, 0}, {5, 5, 3, 2, 2, 2, 4, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {4, 4, 3, 3, 1, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, {4, 4, 2, 1, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {3, 3, 1, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, {2, 2, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}};
static __constant__ uint8_t c_tz_code[15][16] = {
    {1, 3, 2, 3, 2, 3, 2, 3, 2, 3, 2, 3, 2, 3, 2, 1}, {7, 6, 5, 4, 3, 5, 4, 3, 2, 3, 2, 3, 2, 1, 0, 0},
    {5, 7, 6, 5, 4, 3, 4, 3, 2, 3, 2, 1, 1, 0, 0, 0}, {3, 7, 5, 4, 6, 5, 4, 3, 3, 2, 2, 1, 0, 0, 0, 0},
    {5, 4, 3, 7, 6, 5, 4, 3, 2, 1, 1, 0, 0, 0, 0, 0}, {1, 1, 7, 6, 5, 4, 3, 2, 1, 1, 0, 0, 0, 0, 0, 0},
    {1, 1, 5, 4, 3, 3, 2, 1, 1, 0, 0, 0, 0, 0, 0, 0}, {1, 1, 1, 3, 3, 2, 2, 1, 0, 0, 0, 0, 0, 0, 0, 0},
    {1, 0, 1, 3, 2, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0}, {1, 0, 1, 3, 2, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {0, 1, 1, 2, 1, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, {0, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {0, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, {0, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}};
static __constant__ uint8_t c_tzdc_len[3][4] = {{1, 2, 3, 3}, {1, 2, 2, 0}, {1, 1, 0, 0}};
static __constant__ uint8_t c_tzdc_code[3][4] = {{1, 1, 1, 0}, {1, 1, 0, 0}, {1, 0, 0, 0}};
static __constant__ uint8_t c_rb_len[6][7] = {{1, 1, 0, 0, 0, 0, 0}, {1, 2, 2, 0, 0, 0, 0}, {2, 2, 2, 2, 0, 0, 0},
                                       {2, 2, 2, 3, 3, 0, 0}, {2, 2, 3, 3, 3, 3, 0}, {2, 3, 3, 3, 3, 3, 3}};
static __constant__ uint8_t c_rb_code[6][7] = {{1, 0, 0, 0, 0, 0, 0}, {1, 1, 0, 0, 0, 0, 0}, {3, 2, 1, 0, 0, 0, 0},
                                        {3, 2, 1, 1, 0, 0, 0}, {3, 2, 3, 2, 1, 0, 0}, {3, 0, 1, 3, 2, 5, 4}};
// raster index (y*4+x) of scan position k (F/scaleTransform.cpp:43-47)
static __constant__ uint8_t c_zz[16] = {0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15};
// x,y origin of luma 4x4 block k (F/h264_globals.cpp:209-214)
static __constant__ uint8_t c_bx[16] = {0, 4, 0, 4, 8, 12, 8, 12, 0, 4, 0, 4, 8, 12, 8, 12};
static __constant__ uint8_t c_by[16] = {0, 0, 4, 4, 0, 0, 4, 4, 8, 8, 12, 12, 8, 8, 12, 12};
static __constant__ uint8_t c_cbp_intra_code[48] = {3,  29, 30, 17, 31, 18, 37, 8,  32, 38, 19, 9,  20, 10, 11, 2,
                                             16, 33, 34, 21, 35, 22, 39, 4,  36, 40, 23, 5,  24, 6,  7,  1,
                                             41, 42, 43, 25, 44, 26, 46, 12, 45, 47, 27, 13, 28, 14, 15, 0};
static __constant__ uint8_t c_cbp_inter_code[48] = {0,  2,  3,  7,  4,  8,  17, 13, 5,  18, 9,  14, 10, 15, 16, 11,
                                             1,  32, 33, 36, 34, 37, 44, 40, 35, 45, 38, 41, 39, 42, 43, 19,
                                             6,  24, 25, 20, 26, 21, 46, 28, 27, 47, 22, 29, 23, 30, 31, 12};
// codeNum -> coded_block_pattern (Table 9-4, F/h264_globals.cpp:140-153)
static __constant__ uint8_t c_code_cbp_intra[48] = {47, 31, 15, 0,  23, 27, 29, 30, 7,  11, 13, 14, 39, 43, 45, 46,
                                             16, 3,  5,  10, 12, 19, 21, 26, 28, 35, 37, 42, 44, 1,  2,  4,
                                             8,  17, 18, 20, 24, 6,  9,  22, 25, 32, 33, 34, 36, 40, 38, 41};
static __constant__ uint8_t c_code_cbp_inter[48] = {0,  16, 1,  2,  4,  8,  32, 3,  5,  10, 12, 15, 47, 7,  11, 13,
                                             14, 6,  9,  31, 35, 37, 42, 44, 33, 34, 36, 40, 39, 43, 45, 46,
                                             17, 18, 20, 24, 19, 21, 26, 28, 23, 27, 29, 30, 22, 25, 38, 41};
// neighbour A (left) / B (up) block of luma block k (6.4.10.4), chroma likewise
static __constant__ uint8_t c_nbA[16] = {5, 0, 7, 2, 1, 4, 3, 6, 13, 8, 15, 10, 9, 12, 11, 14};
static __constant__ uint8_t c_nbB[16] = {10, 11, 0, 1, 14, 15, 4, 5, 2, 3, 8, 9, 6, 7, 12, 13};
static __constant__ uint8_t c_nbcA[4] = {1, 0, 3, 2};
static __constant__ uint8_t c_nbcB[4] = {2, 3, 0, 1};

__device__ __forceinline__ int iabs(int a) { return a < 0 ? -a : a; }
__device__ __forceinline__ int clip255(int x) { return x < 0 ? 0 : (x > 255 ? 255 : x); }
__device__ __forceinline__ int iclamp(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

// 16*v[m][class] (F/scaleTransform.cpp:32-40) and round(32768/LevelScale) (F/quantizationTransform.cpp:24-32)
__device__ __forceinline__ int level_scale(int m, int i, int j)
{
    const int v[6][3] = {{10, 16, 13}, {11, 18, 14}, {13, 20, 16}, {14, 23, 18}, {16, 25, 20}, {18, 29, 23}};
    int cls = ((i | j) & 1) == 0 ? 0 : ((i & j & 1) ? 1 : 2);
    return 16 * v[m][cls];
}
// (derived at compile time: an integer division per coefficient in the kernels otherwise)
#define FER_LQ(v) ((65536 + 16 * (v)) / (2 * 16 * (v)))
__device__ __forceinline__ int level_quant(int m, int i, int j)
{
    const int q[6][3] = {{FER_LQ(10), FER_LQ(16), FER_LQ(13)}, {FER_LQ(11), FER_LQ(18), FER_LQ(14)}, {FER_LQ(13), FER_LQ(20), FER_LQ(16)},
                         {FER_LQ(14), FER_LQ(23), FER_LQ(18)}, {FER_LQ(16), FER_LQ(25), FER_LQ(20)}, {FER_LQ(18), FER_LQ(29), FER_LQ(23)}};
    int cls = ((i | j) & 1) == 0 ? 0 : ((i & j & 1) ? 1 : 2);
    return q[m][cls];
}

// ---------------------------------------------------------------- 4x4 transforms (one block per lane)
// a1: F/quantizationTransform.cpp:41-100
__device__ __forceinline__ void fwd4x4(const int r[16], int d[16])
{
    int h[16], f[16];
#pragma unroll
    for (int i = 0; i < 16; i++) h[i] = r[i] == 0 ? 0 : r[i] * 64 - 32;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int a = h[j], b = h[4 + j], c = h[8 + j], e = h[12 + j];
        f[j] = (256 * (a + b + c + e) + 512) >> 10;
        f[4 + j] = (416 * a + 208 * b - 208 * c - 416 * e + 512) >> 10;
        f[8 + j] = (256 * (a - b - c + e) + 512) >> 10;
        f[12 + j] = (208 * a - 416 * b + 416 * c - 208 * e + 512) >> 10;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int a = f[4 * i], b = f[4 * i + 1], c = f[4 * i + 2], e = f[4 * i + 3];
        d[4 * i] = (256 * (a + b + c + e) + 512) >> 10;
        d[4 * i + 1] = (416 * a + 208 * b - 208 * c - 416 * e + 512) >> 10;
        d[4 * i + 2] = (256 * (a - b - c + e) + 512) >> 10;
        d[4 * i + 3] = (208 * a - 416 * b + 416 * c - 208 * e + 512) >> 10;
    }
}

// a2: F/quantizationTransform.cpp:183-223
__device__ __forceinline__ void quant4x4(const int d[16], int c[16], int qP, bool keepDC)
{
    int q6 = qP / 6, m = qP % 6;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        int lq = level_quant(m, i >> 2, i & 3);
        int t = qP < 24 ? ((d[i] * (1 << (4 - q6))) - (1 << (3 - q6))) * lq : (d[i] >> (q6 - 4)) * lq;
        c[i] = (t + 16384) >> 15;
    }
    if (keepDC) c[0] = d[0];
}

// a6 + a7: F/scaleTransform.cpp:308-340, :101-150
__device__ __forceinline__ void inv4x4(const int c[16], int r[16], int qP, bool keepDC)
{
    int d[16], e[16];
    int q6 = qP / 6, m = qP % 6;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        int ls = level_scale(m, i >> 2, i & 3);
        d[i] = qP >= 24 ? (c[i] * ls) * (1 << (q6 - 4)) : (c[i] * ls + (1 << (3 - q6))) >> (4 - q6);
    }
    if (keepDC) d[0] = c[0];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int e0 = d[4 * i] + d[4 * i + 2], e1 = d[4 * i] - d[4 * i + 2];
        int e2 = (d[4 * i + 1] >> 1) - d[4 * i + 3], e3 = d[4 * i + 1] + (d[4 * i + 3] >> 1);
        e[4 * i] = e0 + e3;
        e[4 * i + 1] = e1 + e2;
        e[4 * i + 2] = e1 - e2;
        e[4 * i + 3] = e0 - e3;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int g0 = e[j] + e[8 + j], g1 = e[j] - e[8 + j];
        int g2 = (e[4 + j] >> 1) - e[12 + j], g3 = e[4 + j] + (e[12 + j] >> 1);
        r[j] = (g0 + g3 + 32) >> 6;
        r[4 + j] = (g1 + g2 + 32) >> 6;
        r[8 + j] = (g1 - g2 + 32) >> 6;
        r[12 + j] = (g0 - g3 + 32) >> 6;
    }
}

// a3: F/quantizationTransform.cpp:105-152 + :227-260 (in/out raster 4x4)
__device__ __forceinline__ void fwd_dc_luma(const int f[16], int c[16], int qP)
{
    int e[16], t[16];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int g0 = f[j] + f[12 + j], g1 = f[4 + j] + f[8 + j], g2 = f[4 + j] - f[8 + j], g3 = f[j] - f[12 + j];
        e[j] = g0 + g1;
        e[4 + j] = g3 + g2;
        e[8 + j] = g0 - g1;
        e[12 + j] = g3 - g2;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int d0 = e[4 * i] + e[4 * i + 3], d1 = e[4 * i + 1] + e[4 * i + 2];
        int d2 = e[4 * i + 1] - e[4 * i + 2], d3 = e[4 * i] - e[4 * i + 3];
        t[4 * i] = (d0 + d1 + 8) >> 4;
        t[4 * i + 1] = (d3 + d2 + 8) >> 4;
        t[4 * i + 2] = (d0 - d1 + 8) >> 4;
        t[4 * i + 3] = (d3 - d2 + 8) >> 4;
    }
    int q6 = qP / 6, ql = level_quant(qP % 6, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; i++) {
        int v = qP >= 36 ? (t[i] >> (q6 - 6)) * ql : ((t[i] * (1 << (6 - q6))) - (1 << (5 - q6))) * ql;
        c[i] = (v + 16384) >> 15;
    }
}

// F/scaleTransform.cpp:154-189 + :344-376
__device__ __forceinline__ void inv_dc_luma(const int c[16], int dcY[16], int qP)
{
    int e[16], f[16];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int d0 = c[4 * i] + c[4 * i + 2], d1 = c[4 * i] - c[4 * i + 2];
        int d2 = c[4 * i + 1] - c[4 * i + 3], d3 = c[4 * i + 1] + c[4 * i + 3];
        e[4 * i] = d0 + d3;
        e[4 * i + 1] = d1 + d2;
        e[4 * i + 2] = d1 - d2;
        e[4 * i + 3] = d0 - d3;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int g0 = e[j] + e[8 + j], g1 = e[j] - e[8 + j], g2 = e[4 + j] - e[12 + j], g3 = e[4 + j] + e[12 + j];
        f[j] = g0 + g3;
        f[4 + j] = g1 + g2;
        f[8 + j] = g1 - g2;
        f[12 + j] = g0 - g3;
    }
    int q6 = qP / 6, ls = level_scale(qP % 6, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; i++)
        dcY[i] = qP >= 36 ? (f[i] * ls) * (1 << (q6 - 6)) : (f[i] * ls + (1 << (5 - q6))) >> (6 - q6);
}

// a4: F/quantizationTransform.cpp:157-178 + :264-282; in/out raster 2x2
__device__ __forceinline__ void fwd_dc_chroma(const int f[4], int c[4], int qP)
{
    int d00 = f[0] + f[1], d01 = f[0] - f[1], d10 = f[2] + f[3], d11 = f[2] - f[3];
    int t[4] = {(d00 + d10 + 2) >> 2, (d01 + d11 + 2) >> 2, (d00 - d10 + 2) >> 2, (d01 - d11 + 2) >> 2};
    int q6 = qP / 6, ql = level_quant(qP % 6, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; i++) c[i] = ((((t[i] * 32) >> q6) * ql) + 16384) >> 15;
}

// F/scaleTransform.cpp:247-261 + :408-421
__device__ __forceinline__ void inv_dc_chroma(const int c[4], int dc[4], int qP)
{
    int d00 = c[0] + c[2], d01 = c[1] + c[3], d10 = c[0] - c[2], d11 = c[1] - c[3];
    int f[4] = {d00 + d01, d00 - d01, d10 + d11, d10 - d11};
    int q6 = qP / 6, ls = level_scale(qP % 6, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; i++) dc[i] = ((f[i] * ls) * (1 << q6)) >> 5;
}

// ---------------------------------------------------------------- motion compensation (a17)
__device__ __forceinline__ int tap6(int E, int F, int G, int H, int I, int J)
{
    return clip255((E - 5 * F + 20 * G + 20 * H - 5 * I + J + 16) >> 5);
}
#define FER_MID(a, b) (((a) + (b) + 1) >> 1)

// Quarter-pel sample at integer position (X,Y) of plane R with per-coordinate edge clamping,
// F/mocomp.cpp:11-36,50-78.  The centre sample j filters clipped intermediates (reference quirk).
__device__ __forceinline__ int luma_frac_at(const uint8_t *__restrict__ R, int W, int H, int X, int Y, int frac)
{
    int xs[6], ys[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        xs[k] = iclamp(X + k - 2, 0, W - 1);
        ys[k] = iclamp(Y + k - 2, 0, H - 1) * W;
    }
#define PXL(dx, dy) ((int)R[ys[(dy) + 2] + xs[(dx) + 2]])
    int G = PXL(0, 0);
    if (frac == 0) return G;
    int b = tap6(PXL(-2, 0), PXL(-1, 0), G, PXL(1, 0), PXL(2, 0), PXL(3, 0));
    if (frac == 1) return FER_MID(G, b);
    if (frac == 2) return b;
    if (frac == 3) return FER_MID(b, PXL(1, 0));
    int h = tap6(PXL(0, -2), PXL(0, -1), G, PXL(0, 1), PXL(0, 2), PXL(0, 3));
    if (frac == 4) return FER_MID(G, h);
    if (frac == 8) return h;
    if (frac == 12) return FER_MID(h, PXL(0, 1));
    if (frac == 5) return FER_MID(b, h);
    int m = tap6(PXL(1, -2), PXL(1, -1), PXL(1, 0), PXL(1, 1), PXL(1, 2), PXL(1, 3));
    if (frac == 7) return FER_MID(b, m);
    int s = tap6(PXL(-2, 1), PXL(-1, 1), PXL(0, 1), PXL(1, 1), PXL(2, 1), PXL(3, 1));
    if (frac == 13) return FER_MID(h, s);
    if (frac == 15) return FER_MID(s, m);
    int cc = tap6(PXL(-2, -2), PXL(-2, -1), PXL(-2, 0), PXL(-2, 1), PXL(-2, 2), PXL(-2, 3));
    int dd = tap6(PXL(-1, -2), PXL(-1, -1), PXL(-1, 0), PXL(-1, 1), PXL(-1, 2), PXL(-1, 3));
    int ee = tap6(PXL(2, -2), PXL(2, -1), PXL(2, 0), PXL(2, 1), PXL(2, 2), PXL(2, 3));
    int ff = tap6(PXL(3, -2), PXL(3, -1), PXL(3, 0), PXL(3, 1), PXL(3, 2), PXL(3, 3));
    int j = tap6(cc, dd, h, m, ee, ff);
    if (frac == 10) return j;
    if (frac == 6) return FER_MID(b, j);
    if (frac == 9) return FER_MID(h, j);
    if (frac == 14) return FER_MID(j, s);
    return FER_MID(j, m);  // frac == 11
#undef PXL
}

// luma prediction sample (x,y) of the MB at (xP,yP) for a quadrant MV: F/mocomp.cpp:152-177
__device__ __forceinline__ int mc_luma(const uint8_t *__restrict__ R, int W, int H, int xP, int yP, int x, int y,
                                       int mvx, int mvy)
{
    return luma_frac_at(R, W, H, xP + x + (mvx >> 2), yP + y + (mvy >> 2), (mvy & 3) * 4 + (mvx & 3));
}

// chroma prediction sample (x,y in 0..7) of the MB: bilinear over the 3x3 patch of the 4x4 luma
// sub-block that owns it, F/mocomp.cpp:179-194
__device__ __forceinline__ int mc_chroma(const uint8_t *__restrict__ R, int Wc, int Hc, int xPc, int yPc, int x,
                                         int y, int mvx, int mvy)
{
    int bx = (x >> 1) << 1, by = (y >> 1) << 1;  // origin of the 2x2 chroma block (xAl/2, yAl/2)
    int cx = xPc + bx + (mvx >> 3), cy = yPc + by + (mvy >> 3);
    int ox = x & 1, oy = y & 1;
    int x0 = iclamp(cx + ox, 0, Wc - 1), x1 = iclamp(cx + ox + 1, 0, Wc - 1);
    int y0 = iclamp(cy + oy, 0, Hc - 1) * Wc, y1 = iclamp(cy + oy + 1, 0, Hc - 1) * Wc;
    int xl = mvx & 7, yl = mvy & 7;
    return ((8 - xl) * (8 - yl) * R[y0 + x0] + xl * (8 - yl) * R[y0 + x1] + (8 - xl) * yl * R[y1 + x0] +
            xl * yl * R[y1 + x1] + 32) >>
           6;
}

// four chroma samples (x .. x+3, y) of the block at (xPc, yPc), F/mocomp.cpp:80-110 (as mc_chroma, sample by sample)
__device__ __forceinline__ void mc_chroma_row4(const uint8_t *__restrict__ R, int Wc, int Hc, int xPc, int yPc, int x, int y, int mvx,
                                               int mvy, int out[4])
{
    const int by = (y >> 1) << 1, oy = y & 1;
    const int cy = yPc + by + (mvy >> 3);
    const uint32_t y0 = __umul24((uint32_t)iclamp(cy + oy, 0, Hc - 1), (uint32_t)Wc), y1 = __umul24((uint32_t)iclamp(cy + oy + 1, 0, Hc - 1), (uint32_t)Wc);
    const int X = xPc + x + (mvx >> 3);  // x is a multiple of 4: sample k reads columns X + k and X + k + 1
    int t[5], u[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const uint32_t xc = (uint32_t)iclamp(X + k, 0, Wc - 1);
        t[k] = R[y0 + xc];  // (32-bit offsets from the stream's uniform plane base)
        u[k] = R[y1 + xc];
    }
    const int xl = mvx & 7, yl = mvy & 7;
    const int w00 = (8 - xl) * (8 - yl), w01 = xl * (8 - yl), w10 = (8 - xl) * yl, w11 = xl * yl;
#pragma unroll
    for (int k = 0; k < 4; k++) out[k] = (w00 * t[k] + w01 * t[k + 1] + w10 * u[k] + w11 * u[k + 1] + 32) >> 6;
}

// 4 / 8 consecutive bytes at an arbitrary byte address with aligned dword loads + v_alignbyte
// (the arrays are allocated with 256 bytes of slack, so the trailing dword is always readable)
__device__ __forceinline__ uint32_t load_u8x4(const uint8_t *__restrict__ p)
{
    const uint32_t *a = (const uint32_t *)(p - ((uintptr_t)p & 3));  // pointer arithmetic keeps the global address space
    uint32_t sh = (uint32_t)((uintptr_t)p & 3);
    return __builtin_amdgcn_alignbyte(a[1], a[0], sh);
}
__device__ __forceinline__ void load_u8x8(const uint8_t *__restrict__ p, uint32_t &lo, uint32_t &hi)
{
    const uint32_t *a = (const uint32_t *)(p - ((uintptr_t)p & 3));  // pointer arithmetic keeps the global address space
    uint32_t sh = (uint32_t)((uintptr_t)p & 3);
    uint32_t w0 = a[0], w1 = a[1], w2 = a[2];
    lo = __builtin_amdgcn_alignbyte(w1, w0, sh);
    hi = __builtin_amdgcn_alignbyte(w2, w1, sh);
}

// luma prediction of 4 consecutive samples (x..x+3, y) of the MB at (xP,yP): when every target
// position lies inside the picture the motion-compensated value IS the interpolated plane
// (identical clamped taps), read as packed bytes; otherwise the exact per-sample path.
struct IPlanes {  // the 16 interpolated planes of one stream
    const uint8_t *base;  // pixel (0, 0) of plane 0
    int pitch;
    size_t plane;
};
__device__ __forceinline__ IPlanes ip_stream(const FerDev &d, int s)
{
    IPlanes ip;
    ip.base = d.interp + (size_t)s * 16 * d.iplane + d.ioff;
    ip.pitch = d.ipitch;
    ip.plane = d.iplane;
    return ip;
}
__device__ __forceinline__ void mc_luma4(const uint8_t *__restrict__ R, const IPlanes &ip, int W, int H, int xP, int yP, int x,
                                         int y, int mvx, int mvy, int out[4])
{
    int X = xP + x + (mvx >> 2), Y = yP + y + (mvy >> 2);
    if (X >= 0 && X + 3 < W && Y >= 0 && Y < H) {
        // (32-bit byte offset from the stream's uniform plane base, 24-bit multiplies: a padded plane has < 2^24 samples)
        const uint32_t o = __umul24((uint32_t)((mvy & 3) * 4 + (mvx & 3)), (uint32_t)ip.plane) + __umul24((uint32_t)Y, (uint32_t)ip.pitch) + (uint32_t)X;
        const uint32_t *a = (const uint32_t *)(ip.base + (o & ~3u));  // planes start on 16-byte boundaries
        const uint32_t v = __builtin_amdgcn_alignbyte(a[1], a[0], o & 3u);
        out[0] = v & 0xff;
        out[1] = (v >> 8) & 0xff;
        out[2] = (v >> 16) & 0xff;
        out[3] = v >> 24;
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) out[k] = mc_luma(R, W, H, xP, yP, x + k, y, mvx, mvy);
    }
}

// XCD-aware work mapping: workgroups are dealt round-robin over the 8 XCDs (private L2s), so give every
// XCD one contiguous eighth of the logical index space; neighbouring tiles then share an L2.
__device__ __forceinline__ unsigned xcd_swizzle(unsigned b, unsigned n)
{
    unsigned per = n >> 3;
    if (per == 0 || b >= per * 8) return b;  // tail (n not a multiple of 8) keeps its index
    return (b & 7) * per + (b >> 3);
}

// ---------------------------------------------------------------- wave helpers
// DPP (gfx9 row/wave controls) instead of ds_bpermute: the cross-lane steps sit on the critical
// path of one-wave-per-macroblock code, where LDS-routed shuffles cost ~100 cycles each.
#define FER_DPP(v, ctrl) __builtin_amdgcn_update_dpp(0, (v), (ctrl), 0xf, 0xf, false)
#define DPP_QUAD_XOR1 0xB1   // quad_perm [1,0,3,2]
#define DPP_QUAD_XOR2 0x4E   // quad_perm [2,3,0,1]
#define DPP_ROW_HALF_MIRROR 0x141
#define DPP_ROW_MIRROR 0x140
#define DPP_WAVE_SHR1 0x138  // lane i <- lane i-1 over the whole wavefront

// sum over each row of 16 lanes, result in every lane of the row
// ---------------------------------------------------------------- 4x4 transforms, four lanes per block
// lane = block * 4 + row holds one ROW of its block; the vertical half of every transform crosses the four lanes of a
// quad with DPP quad_perm broadcasts, the horizontal half stays in the lane (k_p_resid, the Intra4x4 chain of k_intra_mb)
// scan position of raster sample (row, x): the inverse of c_zz, one dword of four positions per row
static __constant__ __align__(4) uint8_t c_izz[16] = {0, 1, 5, 6, 2, 4, 7, 12, 3, 8, 11, 13, 9, 10, 14, 15};

#define QB0 0x00  // quad_perm broadcasts of lane 0..3 of the quad
#define QB1 0x55
#define QB2 0xAA
#define QB3 0xFF

struct RowQ {  // per-lane constants of a row of a 4x4 block
    int row;
    int k0, k1, k2, k3;  // vertical forward coefficients of this row (F/quantizationTransform.cpp:41-100)
    int lq[4], ls[4];    // LevelQuantize / LevelScale of (row, x)
};

__device__ __forceinline__ RowQ rowq_make(int row, const int16_t (&t)[6])  // t = FerDev::lsq[luma / chroma]
{
    RowQ q;
    q.row = row;
    const bool r0 = row == 0, r1 = row == 1, r2 = row == 2;
    q.k0 = (r0 || r2) ? 256 : (r1 ? 416 : 208);
    q.k1 = r0 ? 256 : (r1 ? 208 : (r2 ? -256 : -416));
    q.k2 = r0 ? 256 : (r1 ? -208 : (r2 ? -256 : 416));
    q.k3 = (r0 || r2) ? 256 : (r1 ? -416 : -208);
    const int s_ee = t[0], s_oo = t[1], s_eo = t[2], q_ee = t[3], q_oo = t[4], q_eo = t[5];
    const bool odd = row & 1;
    q.ls[0] = q.ls[2] = odd ? s_eo : s_ee;
    q.ls[1] = q.ls[3] = odd ? s_oo : s_eo;
    q.lq[0] = q.lq[2] = odd ? q_eo : q_ee;
    q.lq[1] = q.lq[3] = odd ? q_oo : q_eo;
    return q;
}

// a1 + a2 on the row r[0..3] (residual) of this lane: forward core (vertical across the quad, then horizontal) and
// quantiser; c = levels of (row, x); returns the unquantised DC in dc0 (meaningful in row 0)
__device__ __forceinline__ void fwd_row(const RowQ &q, const int r[4], int qP, bool keepDC, int c[4], int &dc0)
{
    int f[4];
#pragma unroll
    for (int x = 0; x < 4; x++) {
        const int h = r[x] == 0 ? 0 : r[x] * 64 - 32;
        const int a = FER_DPP(h, QB0), b = FER_DPP(h, QB1), cc = FER_DPP(h, QB2), e = FER_DPP(h, QB3);
        f[x] = (q.k0 * a + q.k1 * b + q.k2 * cc + q.k3 * e + 512) >> 10;
    }
    int t[4];
    t[0] = (256 * (f[0] + f[1] + f[2] + f[3]) + 512) >> 10;
    t[1] = (416 * f[0] + 208 * f[1] - 208 * f[2] - 416 * f[3] + 512) >> 10;
    t[2] = (256 * (f[0] - f[1] - f[2] + f[3]) + 512) >> 10;
    t[3] = (208 * f[0] - 416 * f[1] + 416 * f[2] - 208 * f[3] + 512) >> 10;
    const int q6 = qP / 6;
#pragma unroll
    for (int x = 0; x < 4; x++) {
        const int v = qP < 24 ? ((t[x] * (1 << (4 - q6))) - (1 << (3 - q6))) * q.lq[x] : (t[x] >> (q6 - 4)) * q.lq[x];
        c[x] = (v + 16384) >> 15;
    }
    dc0 = t[0];
    if (keepDC && q.row == 0) c[0] = t[0];
}

// a6 + a7: dequantiser, inverse core (horizontal in the lane, then vertical across the quad) -> residual row
__device__ __forceinline__ void inv_row(const RowQ &q, const int c[4], int qP, bool keepDC, int r[4])
{
    const int q6 = qP / 6;
    int dq[4];
#pragma unroll
    for (int x = 0; x < 4; x++)
        dq[x] = qP >= 24 ? (c[x] * q.ls[x]) * (1 << (q6 - 4)) : (c[x] * q.ls[x] + (1 << (3 - q6))) >> (4 - q6);
    if (keepDC && q.row == 0) dq[0] = c[0];
    const int e0 = dq[0] + dq[2], e1 = dq[0] - dq[2], e2 = (dq[1] >> 1) - dq[3], e3 = dq[1] + (dq[3] >> 1);
    const int E[4] = {e0 + e3, e1 + e2, e1 - e2, e0 - e3};
    const bool outer = q.row == 0 || q.row == 3, plus = q.row < 2;
#pragma unroll
    for (int x = 0; x < 4; x++) {
        const int a = FER_DPP(E[x], QB0), b = FER_DPP(E[x], QB1), cc = FER_DPP(E[x], QB2), e = FER_DPP(E[x], QB3);
        const int g0 = a + cc, g1 = a - cc, g2 = (b >> 1) - e, g3 = b + (e >> 1);
        const int u = outer ? g0 : g1, w = outer ? g3 : g2;
        r[x] = ((plus ? u + w : u - w) + 32) >> 6;
    }
}

__device__ __forceinline__ int quad_sum(int v)
{
    v += FER_DPP(v, DPP_QUAD_XOR1);
    v += FER_DPP(v, DPP_QUAD_XOR2);
    return v;
}

__device__ __forceinline__ int row16_sum(int v)
{
    v += FER_DPP(v, DPP_QUAD_XOR1);
    v += FER_DPP(v, DPP_QUAD_XOR2);
    v += FER_DPP(v, DPP_ROW_HALF_MIRROR);
    v += FER_DPP(v, DPP_ROW_MIRROR);
    return v;
}
// sum over each group of 8 lanes, result in every lane of the group
__device__ __forceinline__ int oct_sum(int v)
{
    v += FER_DPP(v, DPP_QUAD_XOR1);
    v += FER_DPP(v, DPP_QUAD_XOR2);
    v += FER_DPP(v, DPP_ROW_HALF_MIRROR);
    return v;
}
__device__ __forceinline__ int wave_sum(int v)
{
    v = row16_sum(v);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
}
__device__ __forceinline__ int wave_min(int v)
{
    v = min(v, FER_DPP(v, DPP_QUAD_XOR1));
    v = min(v, FER_DPP(v, DPP_QUAD_XOR2));
    v = min(v, FER_DPP(v, DPP_ROW_HALF_MIRROR));
    v = min(v, FER_DPP(v, DPP_ROW_MIRROR));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_max(int v)
{
    v = max(v, FER_DPP(v, DPP_QUAD_XOR1));
    v = max(v, FER_DPP(v, DPP_QUAD_XOR2));
    v = max(v, FER_DPP(v, DPP_ROW_HALF_MIRROR));
    v = max(v, FER_DPP(v, DPP_ROW_MIRROR));
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
// value of lane `src` (wave-uniform index) in every lane
__device__ __forceinline__ int lane_bcast(int v, int src)
{
    return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src));
}
