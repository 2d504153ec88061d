/*
 * fo.h -- CPU ORACLE for the fer_h264 per-macroblock hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C, single-threaded, re-entrant
 * restatement of the reference's algorithm (zoltanmaric/h264-fer,
 * fer_h264/fer_h264/ *.cpp, abbreviated F/ below).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call it.
 * The product (h264-fer_amd/csrc, libferhip.so) never does.
 *
 * PARITY STATUS
 *   - bit writer / exp-Golomb / CAVLC tables / level tables: pinned against the
 *     reference's own leaf sources compiled as oracle/_ref/libfer_leaf.so
 *     (F/rbsp_IO.cpp, F/expgolomb.cpp, F/residual_tables.cpp, F/h264_math.cpp).
 *   - decoder path (CAVLC parse, dequant, inverse transforms, intra prediction,
 *     MV prediction, motion compensation): pinned by decoding the reference's
 *     own fixture F/drugi.264 and comparing with the md5 of the reference's
 *     output recorded in SURVEY.md section 4.
 *   - encoder mode decisions (intra mode choice, motion search): PARITY UNPINNED.
 *     The reference's full translation-unit set cannot be built in this image
 *     (F/stdafx.h:9 needs <tchar.h>, a Windows SDK header the image lacks), so
 *     they are pinned only by the encode->decode round trip (encoder recon ==
 *     decoder output) and by careful restatement with file:line citations.
 */
#ifndef FO_H
#define FO_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* mb_type values as stored in mb_type_array (F/h264_globals.h:26-62) */
#define FO_P_L0_16x16 0
#define FO_P_16x8 1
#define FO_P_8x16 2
#define FO_P_8x8 3
#define FO_P_8x8ref0 4
#define FO_I_4x4 0
#define FO_P_SKIP 31
#define FO_MV_NA ((int)0x80808080u) /* F/mode_pred.h:6 */

#define FO_NAL_SLICE 1
#define FO_NAL_IDR 5
#define FO_NAL_SEI 6
#define FO_NAL_SPS 7
#define FO_NAL_PPS 8

/* ---- bit writer (F/rbsp_IO.cpp:100-190) ---- */
typedef struct {
    uint8_t *buf;
    size_t cap;
    size_t nbits; /* bits written so far */
} fo_bw;

void fo_bw_init(fo_bw *w, uint8_t *buf, size_t cap);
void fo_bw_put(fo_bw *w, int n, uint32_t v); /* MSB first, n<=32 */
void fo_bw_ue(fo_bw *w, unsigned v);         /* F/expgolomb.cpp:80 */
void fo_bw_se(fo_bw *w, int v);              /* F/expgolomb.cpp:94 */
int fo_ue_len(unsigned v);                   /* 2*prefix+1, F/rbsp_encoding.cpp:375 */
unsigned fo_se_to_ue(int v);                 /* F/expgolomb.cpp:108 */
size_t fo_bw_trailing(fo_bw *w);             /* F/rbsp_encoding.cpp:108; returns bytes */

/* ---- bit reader (F/rbsp_IO.cpp:193-325) ---- */
typedef struct {
    const uint8_t *buf;
    size_t size; /* bytes */
    size_t pos;  /* bit position */
} fo_br;
void fo_br_init(fo_br *r, const uint8_t *buf, size_t size);
unsigned fo_br_bit(fo_br *r);
unsigned fo_br_bits(fo_br *r, int n);
unsigned fo_br_peek24(fo_br *r);
unsigned fo_br_ue(fo_br *r);
int fo_br_se(fo_br *r);
unsigned fo_br_te(fo_br *r); /* F/expgolomb.cpp:156 (with its quirk) */
int fo_br_more(fo_br *r);    /* F/rbsp_IO.cpp:193 heuristic */

/* ---- tables (fo_tables.c) ---- */
extern const uint8_t fo_ct_len[3][4][17], fo_ct_code[3][4][17]; /* coeff_token nC<8: [class][T1][TC] */
extern const uint8_t fo_ctdc_len[4][5], fo_ctdc_code[4][5];     /* chroma DC coeff_token [T1][TC] */
extern const uint8_t fo_tz_len[15][16], fo_tz_code[15][16];     /* total_zeros 4x4 [TC-1][tz] */
extern const uint8_t fo_tzdc_len[3][4], fo_tzdc_code[3][4];     /* total_zeros chroma DC */
extern const uint8_t fo_rb_len[6][7], fo_rb_code[6][7];         /* run_before zerosLeft 1..6 */
extern const int fo_zigzag[16][2];                              /* {y,x} F/scaleTransform.cpp:43 */
extern const int fo_blk_xy[16][2];                              /* {x,y} F/h264_globals.cpp:209 */
extern const int fo_qpc[52];                                    /* F/inttransform.cpp:8 */
extern const int fo_cbp_intra_to_code[48], fo_cbp_inter_to_code[48];
extern const int fo_code_to_cbp_intra[48], fo_code_to_cbp_inter[48];
int fo_level_scale(int m, int i, int j);    /* 16*v F/scaleTransform.cpp:32 */
int fo_level_quantize(int m, int i, int j); /* F/quantizationTransform.cpp:24 */
/* coeff_token (len,code) for any nC class: cls 0..2 tables, 3 = 6-bit FLC, 4 = chroma DC */
void fo_coeff_token(int cls, int tc, int t1, int *len, unsigned *code);

/* ---- a1..a8 transforms (fo_transform.c) ---- */
void fo_forwardTransform4x4(const int r[4][4], int d[4][4]);
void fo_quantResidual(const int d[4][4], int c[4][4], int qP, int keepDC);
void fo_forwardResidual(int qP, const int in[4][4], int out[4][4], int keepDC);
void fo_forwardDCLumaIntra(int qP, const int dc[4][4], int c[4][4]);
void fo_forwardDCChroma(int qP, const int dc[2][2], int c[2][2]);
void fo_inverseResidual(int qP, const int c[4][4], int r[4][4], int keepDC);
void fo_inverseDCLumaIntra(int qP, const int c[4][4], int dcY[4][4]);
void fo_inverseDCChroma(int qP, const int c[2][2], int dcC[2][2]);
void fo_scan(const int c[4][4], int list[16], int ac); /* transformScan */
void fo_invscan(const int list[16], int c[4][4]);

/* ---- picture / codec context ---- */
typedef struct {
    int Lumalevel[16][16];  /* LumaLevel */
    int DC16[16];           /* Intra16x16DCLevel */
    int AC16[16][16];       /* Intra16x16ACLevel */
    int CDC[2][4];          /* ChromaDCLevel */
    int CAC[2][4][16];      /* ChromaACLevel */
} fo_levels;

typedef struct fo_ctx {
    int W, H, Wc, Hc, mbw, mbh, nmb;
    uint8_t *L, *C[2];   /* `frame`: source in, reconstruction out (in place) */
    uint8_t *dL, *dC[2]; /* `dpb` */
    int have_dpb;
    /* parameters (F/fer_h264.cpp:169-178) */
    int qp, basic, window, maxdiff_set, intra_every;
    int chroma_qp_offset; /* pps.chroma_qp_index_offset */
    /* slice state */
    int slice_type; /* 0 = P, 2 = I */
    int frame_num, poc_lsb, idr_pic_id, first_idr_done;
    int QPy;
    int frames_done; /* currFrameCount analogue */
    /* per-MB persistent side info (a20) */
    int *mb_type;    /* mb_type_array */
    int *cbp_l, *cbp_c;
    int (*tc_l)[16];
    int (*tc_c)[2][4]; /* [mb][iCbCr][blk] */
    int *i4mode;       /* Intra4x4PredMode[(mb<<4)+blk] */
    int (*mvx)[4][4], (*mvy)[4][4];
    int *refidx;
    /* current-MB state (globals in the reference) */
    int cur;           /* CurrMbAddr */
    int cur_mb_type;   /* mb_type */
    int cbpL, cbpC;
    fo_levels lv;
    int prev_flag[16], rem_mode[16], chroma_mode;
    int mvd[4][4][2];
    int sub_mb_type[4];
    int mb_qp_delta;
    int MAXDIFF;
    /* motion estimation structures (a16) */
    uint8_t *interp[16];
    int *kar[5][16]; /* (H+8) x (W+8) */
    int *sorted[5];
    int *sorted_tmp[5];
    int koliko[16385];
    int me_ready;
    /* stats */
    int type_count[5];
    /* decoder extras */
    int num_ref_idx_override, num_ref_idx_l0_active_minus1;
    int log2_max_frame_num, log2_max_poc_lsb, pic_init_qp, deblock_ctl;
    int nal_ref_idc;
    int constrained_intra; /* pps.constrained_intra_pred_flag (decoder) */
    int (*ref_idx_l0)[4];
    /* test hook: MV field of the last picture saved before FillInterpolatedRefFrame clobbers it */
    int (*dbg_mvx)[4][4], (*dbg_mvy)[4][4];
    /* test hook: what coded_mb_size returned for the Intra16x16 / Intra4x4 alternative of every macroblock (I pictures) */
    int (*dbg_mbsize)[2];
} fo_ctx;

fo_ctx *fo_create(int W, int H);
void fo_destroy(fo_ctx *c);
void fo_set_params(fo_ctx *c, int qp, int basic, int window, int maxdiff, int intra_every);

/* ---- encoder (fo_encode.c) ---- */
size_t fo_write_sps(fo_ctx *c, uint8_t *rbsp, size_t cap);
size_t fo_write_pps(fo_ctx *c, uint8_t *rbsp, size_t cap);
int fo_select_nal_type(fo_ctx *c); /* F/ref_frames.cpp:185 */
/* RBSP_encode for a slice NAL: c->L/C hold the source, overwritten by recon. */
size_t fo_encode_slice(fo_ctx *c, int nal_type, uint8_t *rbsp, size_t cap);
size_t fo_write_nal(int nal_ref_idc, int nal_type, const uint8_t *rbsp, size_t n, uint8_t *out); /* F/nal.cpp:261 */
/* whole-stream helper: frames = nframes * (W*H*3/2) bytes of coded-size I420 */
size_t fo_encode_stream(fo_ctx *c, const uint8_t *frames, int nframes, uint8_t *out, size_t cap,
                        uint8_t *recon_out);
void fo_fill_interpolated(fo_ctx *c); /* FillInterpolatedRefFrame */

/* per-MB pieces exposed for KATs */
void fo_quantizationTransform(fo_ctx *c, int predL[16][16], int predCb[8][8], int predCr[8][8], int reconstruct);
void fo_setCodedBlockPattern(fo_ctx *c);
unsigned fo_coded_mb_size(fo_ctx *c, int mode16, int predL[16][16], int predCb[8][8], int predCr[8][8]);
int fo_intraPredictionEncoding(fo_ctx *c, int predL[16][16], int predCr[8][8], int predCb[8][8]);
void fo_interEncoding(fo_ctx *c, int predL[16][16], int predCr[8][8], int predCb[8][8]);
void fo_DeriveMVs(fo_ctx *c);
void fo_Decode(fo_ctx *c, int predL[16][16], int predCr[8][8], int predCb[8][8]);
void fo_mc_sub(fo_ctx *c, int predL[16][16], int predCr[8][8], int predCb[8][8], const uint8_t *rL,
               const uint8_t *rCb, const uint8_t *rCr, int mb, int sub, int part);
void fo_intra4x4_fetch(fo_ctx *c, int blk, int p[14]);
void fo_intra4x4_pred(int mode, const int p[14], int pred[4][4]);
void fo_intra16_fetch(fo_ctx *c, int p[33]);
void fo_intra16_pred(int mode, const int p[33], int pred[16][16]);
void fo_intra_chroma(fo_ctx *c, int predCr[8][8], int predCb[8][8]);
void fo_transformDecoding4x4Luma(fo_ctx *c, int level[16][16], int predL[16][16], int blk, int QPy);
void fo_transformDecoding16x16Luma(fo_ctx *c, int dc[16], int ac[16][16], int predL[16][16], int QPy);
void fo_transformDecodingChroma(fo_ctx *c, int dc[4], int ac[4][16], int predC[8][8], int QPy, int cb);
void fo_transformDecodingPSkip(fo_ctx *c, int predL[16][16], int predCb[8][8], int predCr[8][8], int QPy);

/* CAVLC: kind 0 = Intra16x16DC, 1 = Intra16x16AC, 2 = LumaLevel, 3 = ChromaDC, 4 = ChromaAC */
int fo_cavlc_nC(fo_ctx *c, int kind, int blk, int iCbCr);
unsigned fo_cavlc_block(fo_ctx *c, fo_bw *w, const int *coef, int maxNumCoeff, int kind, int blk, int iCbCr);
void fo_residual_write(fo_ctx *c, fo_bw *w);
/* context-free: encode one block given nC; returns bits; w may be NULL (size only) */
unsigned fo_cavlc_encode_block(fo_bw *w, const int *coef, int maxNumCoeff, int nC, int *totalcoeff);

/* ---- decoder (fo_decode.c) ---- */
typedef void (*fo_frame_cb)(fo_ctx *c, void *user);
/* Annex-B stream in, calls cb after every decoded picture. Returns pictures decoded. */
int fo_decode_stream(const uint8_t *stream, size_t n, fo_frame_cb cb, void *user, fo_ctx **ctx_out);
int fo_decode_slice(fo_ctx *c, int nal_type, int nal_ref_idc, const uint8_t *rbsp, size_t n);

/* ---- synthetic input (fo_gen.c) ---- */
void fo_gen_frame(int W, int H, int t, uint64_t seed, int noise_amp, uint8_t *Y, uint8_t *U, uint8_t *V);

#ifdef __cplusplus
}
#endif
#endif
