/*
 * ferhip.h -- C ABI of libferhip, the MI355X-native drop-in for the per-macroblock hot path
 * of fer_h264 (zoltanmaric/h264-fer).  F/ = fer_h264/fer_h264/ in the reference tree.
 *
 * The reference has no plugin/FFI seam: its NAL/slice driver (F/fer_h264.cpp:55-134) calls
 *     void RBSP_encode(NALunit &nal_unit);      F/rbsp_encoding.h:3, F/rbsp_encoding.cpp:119
 *     int  selectNALUnitType();                 F/ref_frames.h, F/ref_frames.cpp:185
 *     void writeNAL(NALunit nu);                F/nal.h:27, F/nal.cpp:261
 * and communicates through process globals (`frame`, `_qParameter`, `WindowSize`, ...
 * F/h264_globals.h:152-176).  That makes it one stream per process.  This library exports
 *   (1) a context-based ABI in which one context carries what those globals carry for S
 *       independent streams, so that many pictures are in flight on one GPU, and
 *   (2) the legacy global-state entry points (same names, same argument meaning) as thin
 *       shims over a one-stream context: see "legacy seam" below and INTEGRATION.md.
 * All pointers are plain host or device pointers; no C++ or torch types cross the boundary.
 * Every function returns 0 on success or a negative FERHIP_E_* code; nothing falls back to
 * a CPU path.
 */
#ifndef FERHIP_H
#define FERHIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FERHIP_E_ARG -1     /* bad argument */
#define FERHIP_E_HIP -2     /* HIP runtime error (no device, out of memory, launch failure) */
#define FERHIP_E_STATE -3   /* call order violated */
#define FERHIP_E_UNSUP -4   /* parameter combination the GPU path does not implement */
#define FERHIP_E_DEVICE -5  /* sticky device-side error flag, see ferhip_status() */

#define FERHIP_NAL_SLICE 1 /* NAL_UNIT_TYPE_NOT_IDR, F/h264_globals.h:84 */
#define FERHIP_NAL_IDR 5   /* NAL_UNIT_TYPE_IDR */
#define FERHIP_NAL_AUTO 0  /* decide like selectNALUnitType() */

typedef struct ferhip_ctx ferhip_ctx;

/* Parameters of Starter::PostaviParametre (F/fer_h264.cpp:169-178) minus the frame range. */
typedef struct {
    int qp;          /* _qParameter: QPy of every slice, 10..30 in the reference GUI */
    int basic;       /* BasicInterEncoding: 1 = stage 1 of the search only, counters of the discarded exhaustive pass kept */
    int window;      /* WindowSize: +-window/2 integer search, +-window/16 quarter-pel search */
    int maxdiff;     /* MAXDIFF_SET, -1 = adaptive */
    int intra_every; /* IntraEvery */
} ferhip_params;

/* ---- context ---- */
/* width/height: coded picture size, multiples of 16 (the reference crops to that,
 * F/fileIO.cpp:242-243).  nstreams: independent streams encoded side by side. */
/* width, height: multiples of 16; an encoder context takes pictures of up to (width + 32) * (height + 16) < 2^24 samples
 * (3840x2160 is half of that), larger ones return FERHIP_E_UNSUP */
int ferhip_create(ferhip_ctx **out, int width, int height, int nstreams, const ferhip_params *p);
void ferhip_destroy(ferhip_ctx *c);

/* Device (kind 0) / pinned host (kind 1) memory and synchronous copies for hosts without a HIP binding of their
 * own: device pictures for ferhip_set_frames(host = 0), buffers for ferhip_copy_rbsp, pinned sources for
 * ferhip_upload_frames. */
void *ferhip_mem_alloc(size_t bytes, int kind);
void ferhip_mem_free(void *p, int kind);
int ferhip_mem_copy(void *dst, const void *src, size_t bytes);

/* ---- picture input: replaces ReadFromY4M() filling the global `frame` (F/fileIO.cpp:258) ----
 * I420 pictures of coded size, one per stream, stream-major: [nstreams][W*H*3/2].
 * host = 1: src is host memory (copied H2D); host = 0: src is a device pointer (D2D). */
int ferhip_set_frames(ferhip_ctx *c, const void *src, int host);

/* Asynchronous ingest for many streams (the successor of ReadFromY4M's one-picture read, row f3 of SURVEY.md 8f):
 * ferhip_upload_frames starts the H2D copy of the NEXT pictures ([nstreams][W*H*3/2] in pinned host memory) on the
 * context's copy stream and returns (two uploads may be in flight); ferhip_set_frames_uploaded makes the oldest
 * upload the current picture.  Uploading picture t + 1 before encoding picture t overlaps PCIe with the kernels. */
int ferhip_upload_frames(ferhip_ctx *c, const void *pinned_src);
int ferhip_set_frames_uploaded(ferhip_ctx *c);

/* ---- RBSP_encode for slice NAL units (F/rbsp_encoding.cpp:139-323) ----
 * nal_type[s]: FERHIP_NAL_IDR / FERHIP_NAL_SLICE / FERHIP_NAL_AUTO per stream on input, the
 * type actually used on output (NULL = AUTO for all).  After the call the picture buffers
 * hold the reconstruction (like the reference's `frame`) and become the reference picture.
 * rbsp (host): nstreams * rbsp_stride bytes; rbsp_len[s] receives NumBytesInRBSP. */
int ferhip_encode_picture(ferhip_ctx *c, int *nal_type, uint8_t *rbsp, size_t rbsp_stride, uint32_t *rbsp_len);

/* selectNALUnitType() (F/ref_frames.cpp:185-234) for the pictures set by ferhip_set_frames: IDR for the
 * first picture, every IntraEvery-th picture and when the luma SAD against the reference picture
 * exceeds 16 per pixel (evaluated on the device); writes FERHIP_NAL_IDR / FERHIP_NAL_SLICE per stream. */
int ferhip_select_nal_type(ferhip_ctx *c, int *nal_type_out);

/* Same, but leaves the RBSP in device memory (no D2H): *d_rbsp receives the device base,
 * words are big-endian bit order, stream s starts at byte s * *stride. */
int ferhip_encode_picture_dev(ferhip_ctx *c, int *nal_type, const uint8_t **d_rbsp, size_t *stride,
                              const uint32_t **d_rbsp_len);

/* RBSP of the last picture to caller buffers, asynchronously on the context's stream (ordered before the next
 * picture reuses the device buffer): bytes_per_stream bytes per stream to dst + s * dst_stride, the lengths to
 * len_dst[S].  host = 1: dst / len_dst are host memory (pinned for a truly asynchronous copy), else device memory. */
int ferhip_copy_rbsp(ferhip_ctx *c, void *dst, size_t dst_stride, size_t bytes_per_stream, uint32_t *len_dst, int host);
/* waits for everything the context has enqueued */
int ferhip_sync(ferhip_ctx *c);

/* reconstruction of the last encoded picture, [nstreams][W*H*3/2]; host = 1 copies D2H */
int ferhip_get_recon(ferhip_ctx *c, void *dst, int host);

/* SPS / PPS RBSP (F/headers_and_parameter_sets.cpp:305-391,478-513) and NAL framing with
 * emulation prevention (F/nal.cpp:261-299); host-side, byte-serial. */
size_t ferhip_write_sps(ferhip_ctx *c, uint8_t *rbsp, size_t cap);
size_t ferhip_write_pps(ferhip_ctx *c, uint8_t *rbsp, size_t cap);
size_t ferhip_write_nal(int nal_ref_idc, int nal_type, const uint8_t *rbsp, size_t n, uint8_t *out);

/* encode() + NastaviEncode() for S streams of T pictures each (F/fer_h264.cpp:55-134):
 * frames host [T][S][W*H*3/2]; out host [S][out_stride] Annex-B; out_len[S].
 * recon (optional) host [T][S][W*H*3/2]. */
int ferhip_encode_streams(ferhip_ctx *c, const uint8_t *frames, int nframes, uint8_t *out, size_t out_stride,
                          size_t *out_len, uint8_t *recon);

/* statistics of Starter::DohvatiStatistiku: brojTipova[5] per stream, accumulated */
int ferhip_get_stats(ferhip_ctx *c, int *counts5_per_stream);
/* sticky device error flags per stream (bits 0, 1: unused, bit 2: RBSP buffer overflow, bits 3, 4: decoder syntax
 * error / unsupported syntax, bit 5: motion chain timeout, bit 6: a P macroblock without this picture's vectors) */
int ferhip_status(ferhip_ctx *c, int *flags_per_stream);
const char *ferhip_version(void);

/* Live kernel timing: when enabled every kernel group of a picture is bracketed by HIP events
 * on the launch stream.  ferhip_get_profile synchronises and returns accumulated milliseconds
 * and launch counts per phase. */
#define FERHIP_PH_INTERP 0     /* k_interp: the 16 quarter-pel planes */
#define FERHIP_PH_ME_PRE 1     /* k_me_pre: box sums, stage-3 search, its SADs */
#define FERHIP_PH_ME_RESOLVE 2 /* k_me_resolve, one persistent launch per picture */
#define FERHIP_PH_P_RESID 3    /* k_p_resid (partition merge, mvd, snapping, residual) */
#define FERHIP_PH_INTRA 4      /* k_intra_mb, one launch per MB anti-diagonal */
#define FERHIP_PH_CAVLC 5      /* size + scan + emit */
#define FERHIP_PH_FRAME_SAD 6
#define FERHIP_PH_ME_SPEC 7    /* k_me_spec: the predictor-dependent searches for a guessed predictor */
#define FERHIP_PH_SORT 8       /* the two radix passes: k_rs_hist, k_rs_scan, k_rs_scatter, each twice */
#define FERHIP_PH_ME_WALK 9    /* k_me_walk: stage-2 candidate sets */
#define FERHIP_PH_SORT_KEYS 10   /* k_feat0: plane-0 features + the sort's input records */
#define FERHIP_PH_SORT_FINISH 11 /* k_sort_index (+ k_bucket_classes, k_sort_quirk): bucket index of the sorted order */
#define FERHIP_NPHASE 12
int ferhip_profile(ferhip_ctx *c, int enable);
/* launch-shape knobs; results never depend on them.  RESOLVE_WGS = workgroups of the persistent motion-chain launch
 * (default 6144 single-wavefront workgroups; any value >= 1 resolves every row: a workgroup whose own queue is empty takes rows of the others) */
#define FERHIP_TUNE_RESOLVE_WGS 1
#define FERHIP_TUNE_RESOLVE_GROUP 2 /* streams whose rows the motion chain keeps in flight together (cache footprint); clamped to the context's streams */
#define FERHIP_TUNE_OVERLAP_SORT 4  /* 0 (default): the radix sort + bucket index of the reference picture run before k_me_pre; 1 / 2: on a
                                       second stream beside it (measured on MI355X / ROCm 7.2: the two launches do not share the
                                       GPU, the sort simply ends later -- kept as an experiment switch) */
#define FERHIP_TUNE_SPECULATE 3     /* 1 (default): k_me_spec runs the predictor-dependent searches for a guessed predictor and the
                                       chain verifies; 0: the chain searches everything itself */
int ferhip_tune(ferhip_ctx *c, int key, int value);
int ferhip_get_profile(ferhip_ctx *c, double *ms, long *launches, int reset);

/* ---- per-stage entry points (unit-parity surface, SURVEY.md 8b "per-MB") ----
 * They operate on the pictures currently in the context, for all streams. */
/* FillInterpolatedRefFrame(), F/moestimation.h / F/moestimation.cpp:74 */
int ferhip_fill_interpolated(ferhip_ctx *c);
/* motion decision of every MB = interEncoding() over the picture, F/moestimation.cpp:392; the partition merge
 * shares its wavefront with the residual of the macroblock, so the picture buffers hold the reconstruction of
 * the inter macroblocks afterwards */
int ferhip_inter_encoding(ferhip_ctx *c);
/* debug/test read-back of device state; which: see FERHIP_BUF_*; returns bytes copied */
#define FERHIP_BUF_INTERP 1   /* uint8  [S][16][H][W] */
#define FERHIP_BUF_FEAT 2     /* uint16 [S][H][W][16][6] (k0..k4, pad) */
#define FERHIP_BUF_SORTPOS 3  /* uint32 [S][W*H] */
#define FERHIP_BUF_KOLIKO 4   /* int32  [S][16385] */
#define FERHIP_BUF_MBTYPE 5   /* int32  [S][nmb] */
#define FERHIP_BUF_MV 6       /* int16  [S][nmb][4][2] */
#define FERHIP_BUF_MVD 7      /* int16  [S][nmb][4][2] */
#define FERHIP_BUF_LEVELS 8   /* int16  [S][nmb][400] */
#define FERHIP_BUF_CBP 9      /* uint8  [S][nmb][2] */
#define FERHIP_BUF_TC 10      /* uint8  [S][nmb][24] */
#define FERHIP_BUF_I4MODE 11  /* uint8  [S][nmb][16] */
#define FERHIP_BUF_CUR 12     /* uint8  [S][W*H*3/2] current picture buffers */
#define FERHIP_BUF_REF 13     /* uint8  [S][W*H*3/2] reference picture buffers */
#define FERHIP_BUF_TIMING 14  /* int64  [64] in-kernel wall-clock sums (10 ns units) when FER_DBG bit 7 is set */
#define FERHIP_BUF_ST2N 15    /* int32  [S][nmb][4] stage-2 candidates of every 8x8 partition (before the list cap) */
#define FERHIP_BUF_SPEC_STAT 17 /* uint64 [8] since the context was created: partitions the motion chain decided, of which the guessed
                                 predictor was right, P_Skip verdicts needed, of which taken from the guess */
#define FERHIP_BUF_MBSIZE 18   /* int32  [S][nmb][2] coded_mb_size (F/rbsp_encoding.cpp:330) of the Intra16x16 and of the Intra4x4 alternative of every
                                 macroblock of the last I picture */
#define FERHIP_BUF_ST2 16     /* int32  [S][nmb][4][384][2] the candidates (position relative to the block, feature distance); a crowded
                                 partition (count > 384) holds its summary instead: [40] = (last step, distance bound), [41] = (zeros, 0) */
size_t ferhip_read_buffer(ferhip_ctx *c, int which, void *dst, size_t cap);
/* set the reference picture (dpb) directly, [S][W*H*3/2] host */
int ferhip_set_reference(ferhip_ctx *c, const void *src);

/* ---- decode twin (row a19): decode() / RBSP_decode(), F/fer_h264.cpp:26-53, F/rbsp_decoding.cpp:17 ----
 * S Annex-B streams (4-byte start codes, as the reference reads them) of equal picture size are
 * decoded side by side: slice_data parsing runs one wavefront per picture over a window of pictures of
 * every stream at once, reconstruction one wavefront per macroblock, picture by picture.  out (host, may be NULL): [max_pictures][S][W*H*3/2], picture t of
 * stream s at (t*S + s)*W*H*3/2; pictures[s] = number decoded.  Sub-8x8 partitions, ref_idx_l0 and
 * reference list modification are handled the way the reference handles them; syntax the
 * GPU path does not implement (I_PCM, CABAC, High profiles, field coding, slice groups) returns FERHIP_E_UNSUP. */
int ferhip_decode_streams(const uint8_t *const *streams, const size_t *lens, int nstreams, uint8_t *out,
                          int max_pictures, int *pictures, int *width, int *height);
/* frees the window buffers ferhip_decode_streams keeps between calls (tens of GB of HBM for large batches) and the
 * calling thread's host copy of the streams' RBSP (as large as the streams of its last call); FERHIP_E_STATE while a
 * decode is running */
int ferhip_decode_release(void);

/* ---- streaming decoder: RBSP_decode(NALunit) of F/rbsp_decoding.cpp:17 for one stream, NAL unit by NAL unit ----
 * rbsp = the NAL unit's payload without header byte and emulation prevention bytes (what getNAL delivers,
 * F/nal.cpp:68-223).  nal_unit_type 7 (SPS) sizes the decoder, 8 (PPS) is kept, 5 / 1 decode one picture: when
 * `picture` is not NULL it receives W*H*3/2 bytes of I420 and *got_picture = 1.  Other NAL unit types are ignored. */
typedef struct ferhip_dec ferhip_dec;
int ferhip_dec_create(ferhip_dec **out);
int ferhip_dec_nal(ferhip_dec *d, int nal_unit_type, int nal_ref_idc, const uint8_t *rbsp, size_t n, uint8_t *picture,
                   int *got_picture, int *width, int *height);
void ferhip_dec_destroy(ferhip_dec *d);

/* ---- Y4M ingest (row f3): LoadY4MHeader / ReadFromY4M of F/fileIO.cpp:228-346 without the globals ----
 * The picture size comes from the header's " W" / " H" tokens; coded size = cropped to multiples of 16 around the
 * centre.  ferhip_y4m_read fills one coded-size I420 picture (use pinned memory when it feeds ferhip_set_frames);
 * returns 0, or 1 at the end of the stream. */
typedef struct ferhip_y4m ferhip_y4m;
int ferhip_y4m_open(ferhip_y4m **out, const char *path, int *in_width, int *in_height, int *coded_width, int *coded_height);
int ferhip_y4m_read(ferhip_y4m *y, unsigned char *dst);
void ferhip_y4m_close(ferhip_y4m *y);
/* emit, F/fileIO.cpp:100-176: the stream header "YUV4MPEG2 C420jpeg W%d H%d F24:1 Ip A1:1\n" and one picture
 * ("FRAME\n" + I420 when with_frame_line, bare I420 = writeToYUV otherwise); file = a FILE* */
int ferhip_y4m_write_header(void *file, int width, int height);
int ferhip_y4m_write_frame(void *file, const unsigned char *i420, int width, int height, int with_frame_line);

/* ---- block-level KAT surface: the reference's own signatures as batched device calls ----
 * forwardResidual(qP, c, r, Intra, Intra16x16OrChroma), F/quantizationTransform.h:
 * n blocks of 16 int32 (raster) in, 16 int32 out. */
int ferhip_forward_residual(int qP, const int32_t *in, int32_t *out, int keep_dc, size_t nblocks);
/* inverseResidual(bitDepth, qP, c, r, intra16x16OrChroma), F/scaleTransform.h */
int ferhip_inverse_residual(int qP, const int32_t *in, int32_t *out, int keep_dc, size_t nblocks);
/* forwardDCLumaIntra(qP, dcY, c) (F/quantizationTransform.cpp:293) / InverseDCLumaIntra(bitDepth, qP, c, dcY)
 * (F/scaleTransform.h): 16 int32 raster in, 16 out */
int ferhip_forward_dc_luma_intra(int qP, const int32_t *in, int32_t *out, size_t nblocks);
int ferhip_inverse_dc_luma_intra(int qP, const int32_t *in, int32_t *out, size_t nblocks);
/* forwardDCChroma(qP, dcC, c, Intra) (F/quantizationTransform.cpp:303) / InverseDCChroma: the 2x2 block (raster) in
 * slots 0..3 of a 16-int32 record, the rest ignored / zero */
int ferhip_forward_dc_chroma(int qP, const int32_t *in, int32_t *out, size_t nblocks);
int ferhip_inverse_dc_chroma(int qP, const int32_t *in, int32_t *out, size_t nblocks);
/* transformScan(c, list, Intra16x16AC) (F/quantizationTransform.cpp:310; AC variant: 15 entries from index 1,
 * = scanChroma) and transformInverseScan(list, c) (F/scaleTransform.cpp:454) */
int ferhip_transform_scan(const int32_t *in, int32_t *out, int intra16x16_ac, size_t nblocks);
int ferhip_transform_inverse_scan(const int32_t *in, int32_t *out, size_t nblocks);

/* ---- per-macroblock unit-parity surface (SURVEY.md 8b "per-MB"): the reference's macroblock-level functions as batched
 * device calls over the same device functions the encode / decode kernels are built from.  A seam for known-answer tests,
 * not a fast path.  Arrays use the reference's own layouts: samples raster [y][x], levels per luma4x4BlkIdx in scan order
 * (the Intra16x16 AC / chroma AC lists hold 15 entries from index 0). */
#define FERHIP_MBU_QT 0    /* quantizationTransform(predL, predCb, predCr, reconstruct), F/quantizationTransform.cpp:349 */
#define FERHIP_MBU_DEC4 1  /* transformDecoding4x4LumaResidual(LumaLevel, predL, luma4x4BlkIdx, QPy), F/inttransform.cpp:133 */
#define FERHIP_MBU_DEC16 2 /* transformDecodingIntra_16x16Luma(DC, AC, predL, QPy), F/inttransform.cpp:157 */
#define FERHIP_MBU_DECC 3  /* transformDecodingChroma(DC, AC, predC, QPy, Cb) for both planes, F/inttransform.cpp:237 */
#define FERHIP_MBU_SKIP 4  /* transformDecodingP_Skip(predL, predCb, predCr, QPy), F/inttransform.cpp:215 */
typedef struct {
    int32_t op, cls /* MbPartPredMode(mb_type,0): 0 Intra_4x4, 1 Intra_16x16, 2 inter */, qp /* QPy */, qpc /* QPc */, reconstruct, blk;
    int32_t srcY[256], srcCb[64], srcCr[64];    /* the macroblock of `frame` (FERHIP_MBU_QT) */
    int32_t predY[256], predCb[64], predCr[64];
    int32_t lumaLevel[16][16], dc16[16], ac16[16][16], cdc[2][4], cac[2][4][16]; /* levels in (decode-side ops) */
} ferhip_mb_job;
typedef struct {
    int32_t lumaLevel[16][16], dc16[16], ac16[16][16], cdc[2][4], cac[2][4][16]; /* levels out (FERHIP_MBU_QT) */
    int32_t recY[256], recCb[64], recCr[64];    /* samples written into `frame` (0 where the call writes none) */
} ferhip_mb_result;
int ferhip_mb_unit(const ferhip_mb_job *jobs, ferhip_mb_result *results, size_t njobs);
/* residual_block_cavlc_write(coeffLevel, 0, maxNumCoeff - 1, maxNumCoeff) (F/residual.cpp:374) for n blocks with nC given
 * (-1 = chroma DC): bits[n][64] receives each block's code MSB first, nbits[n] its length -- also what
 * residual_block_cavlc_size (F/residual.cpp:673) returns; the call fails if the device's counting form disagrees --,
 * total_coeff[n] the TotalCoeff the block leaves for its neighbours */
int ferhip_cavlc_blocks(const int32_t *coef, const int32_t *nC, const int32_t *max_num_coeff, size_t nblocks, uint8_t *bits,
                        uint32_t *nbits, int32_t *total_coeff);
/* MotionCompensateSubMBPart(predL, predCr, predCb, refPic, mbPartIdx, subMbIdx, subMbPartIdx) (F/mocomp.cpp:152) for n
 * sub-blocks of one reference picture (host I420, width x height): desc[n][5] = macroblock address, subMbIdx, subMbPartIdx,
 * mvx, mvy (quarter samples); predL[n][16] = the 4x4 luma block, predCb / predCr[n][4] = the 2x2 chroma blocks (raster).
 * Chroma is also evaluated through the four-samples-per-row form the residual kernel uses; the call fails if they differ. */
int ferhip_mc_sub_mb_parts(const uint8_t *ref_i420, int width, int height, const int32_t *desc, size_t n, int32_t *predL,
                           int32_t *predCb, int32_t *predCr);

#ifdef __cplusplus
}
#endif
#endif
