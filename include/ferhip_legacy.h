/*
 * ferhip_legacy.h -- the reference's own global-state seam, exported by libferhip with C linkage.
 *
 * A maintainer of fer_h264 who wants the GPU path behind the existing NAL/slice driver
 * (encode()/NastaviEncode(), F/fer_h264.cpp:55-134) links libferhip instead of compiling
 * rbsp_encoding.cpp, intra.cpp, moestimation.cpp, mocomp.cpp, mode_pred.cpp,
 * quantizationTransform.cpp, scaleTransform.cpp, inttransform.cpp, residual.cpp and ref_frames.cpp.
 * Names, argument meaning and side effects are the reference's:
 *
 *   reference (C++ linkage)                          here (C linkage)
 *   void RBSP_encode(NALunit &nal_unit)              void RBSP_encode(NALunit *nal_unit)     F/rbsp_encoding.h:3
 *   int  selectNALUnitType()                         int  selectNALUnitType(void)            F/ref_frames.cpp:185
 *   frame_type frame                                 frame_type frame                        F/h264_globals.h:152-158
 *   int _qParameter, BasicInterEncoding, WindowSize, MAXDIFF_SET, IntraEvery, currFrameCount
 *                                                                                            F/h264_globals.h:168-176,193
 *   int brojTipova[5], vrijeme                                                               F/h264_globals.h:166-167
 *   void RBSP_decode(NALunit nal_unit)               void RBSP_decode(NALunit nal_unit)      F/rbsp_decoding.h:3
 *   LoadY4MHeader / ReadFromY4M / writeToY4M / writeToYUV, FILE *yuvinput, *yuvoutput        F/fileIO.h
 *   forwardResidual, transformScan, forwardDCLumaIntra, forwardDCChroma                      F/quantizationTransform.h
 *   transformInverseScan, inverseResidual, InverseDCLumaIntra, InverseDCChroma               F/scaleTransform.h
 *
 * Contract kept from F/rbsp_encoding.cpp:119-326: nal_unit_type 7 / 8 write SPS / PPS (the SPS
 * call also sizes the encoder from frame.Lwidth x frame.Lheight); 5 / 1 encode the picture in
 * `frame` (tightly packed planes), overwrite `frame` with the reconstruction and make it the
 * reference picture; rbsp_byte is caller-allocated and NumBytesInRBSP is the out-length.
 * Errors cannot be returned through this signature (the reference returns void): on failure
 * NumBytesInRBSP is set to 0 and a message goes to stderr.  One stream per process, like the
 * reference.
 */
#ifndef FERHIP_LEGACY_H
#define FERHIP_LEGACY_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    unsigned char forbidden_zero_bit; /* bool in the reference (F/nal.h:8-14) */
    unsigned int nal_ref_idc, nal_unit_type, NumBytesInRBSP;
    unsigned char *rbsp_byte;
} NALunit;

typedef struct {
    int Lwidth, Lheight;
    int Cwidth, Cheight;
    unsigned char *L, *C[2];
} frame_type;

extern frame_type frame;
extern int _qParameter, BasicInterEncoding, WindowSize, MAXDIFF_SET, IntraEvery, currFrameCount;
extern int brojTipova[5];
extern int vrijeme; /* clock() ticks spent in the last picture */

void RBSP_encode(NALunit *nal_unit);
int selectNALUnitType(void);

/* void RBSP_decode(NALunit nal_unit), F/rbsp_decoding.h:3 (by value, as in the reference): the callee of decode()'s
 * getNAL loop (F/fer_h264.cpp:37-47).  nal_unit_type 7 sizes and allocates `frame`, 8 takes the PPS, 5 / 1 decode one
 * picture into `frame` (it becomes the reference picture) and append it to `yuvoutput` when that file is open. */
void RBSP_decode(NALunit nal_unit);

/* Y4M ingest / emit of F/fileIO.h, through `frame`, `yuvinput` and `yuvoutput` (F/fileIO.cpp:10-11,100-176,228-346) */
#include <stdio.h>
extern FILE *yuvinput, *yuvoutput;
extern int inputWidth, inputHeight;
void LoadY4MHeader(void);
int ReadFromY4M(void);
void writeToY4M(void);
void writeToYUV(void);

/* The block-level entry points of F/quantizationTransform.h and F/scaleTransform.h under their own names (the unit-parity
 * surface of SURVEY 8b): one block per call through the same device functions the encode / decode kernels use -- a seam
 * for the maintainer's unit tests, not a fast path (the batched forms are ferhip_forward_residual ... in ferhip.h).
 * `unsigned char` stands for the reference's bool.  On a device error the output is left unchanged and a message goes
 * to stderr (the reference returns void). */
void forwardResidual(int qP, int c[4][4], int r[4][4], unsigned char Intra, unsigned char Intra16x16OrChroma); /* F/quantizationTransform.cpp:284 */
void transformScan(int c[4][4], int list[16], unsigned char Intra16x16AC);                                     /* :310 */
void forwardDCLumaIntra(int qP, int dcY[4][4], int c[4][4]);                                                   /* :293 */
void forwardDCChroma(int qP, int dcC[2][2], int c[2][2], unsigned char Intra);                                 /* :302 */
void transformInverseScan(int list[16], int c[4][4]);                                                          /* F/scaleTransform.cpp:454 */
void inverseResidual(int bitDepth, int qP, int c[4][4], int r[4][4], unsigned char intra16x16OrChroma);        /* F/scaleTransform.h */
void InverseDCLumaIntra(int bitDepth, int qP, int c[4][4], int dcY[4][4]);
void InverseDCChroma(int bitDepth, int qP, int c[2][2], int dcC[2][2]);

/* ---- per-macroblock entry points (the "per-MB" row of SURVEY.md 8b): one macroblock per call, through the same device
 * functions the kernels use (ferhip_mb_unit, ferhip_cavlc_blocks, ferhip_mc_sub_mb_parts of ferhip.h).  They work on the
 * reference's own globals: the macroblock CurrMbAddr of `frame` (F/h264_globals.h:181), QPy (:179), mb_type (:106), the
 * level arrays of F/residual.h, the vector arrays of F/mode_pred.h. */
extern int CurrMbAddr, QPy, mb_type;
extern int LumaLevel[16][16], Intra16x16DCLevel[16], Intra16x16ACLevel[16][16];
extern int ChromaDCLevel[2][4], ChromaACLevel[2][4][16];
extern int ***mvL0x, ***mvL0y; /* [macroblock][subMbIdx][subMbPartIdx], quarter samples (F/mode_pred.h) */
extern frame_type dpb;         /* the reference picture Decode() predicts from (RefPicList0[0].frame, F/ref_frames.cpp) */
void AllocateMemory(void);     /* F/mode_pred.cpp:22: sizes mvL0x / mvL0y from `frame` */
/* what the reference keeps where a caller of the bare functions cannot reach it:
 * ferhip_legacy_slice_type = shd.slice_type (MbPartPredMode reads it, F/h264_globals.h:123; RBSP_encode sets it);
 * ferhip_legacy_nC = the nC residual_block_cavlc_write / _size use (the reference derives it from file-statics that only
 * its residual_write() sets, F/residual.cpp:433-538; -1 = chroma DC); ferhip_legacy_bits / _nbits = where
 * residual_block_cavlc_write appends (the bit writer F/rbsp_IO.cpp is a leaf file the maintainer keeps) */
extern int ferhip_legacy_slice_type, ferhip_legacy_nC;
extern unsigned char ferhip_legacy_bits[1 << 16];
extern unsigned int ferhip_legacy_nbits;
void quantizationTransform(int predL[16][16], int predCb[8][8], int predCr[8][8], unsigned char reconstruct);              /* F/quantizationTransform.cpp:349 */
void transformDecoding4x4LumaResidual(int LumaLevel[16][16], int predL[16][16], int luma4x4BlkIdx, int QPy);               /* F/inttransform.cpp:133 */
void transformDecodingIntra_16x16Luma(int Intra16x16DCLevel[16], int Intra16x16ACLevel[16][16], int predL[16][16], int QPy); /* :157 */
void transformDecodingP_Skip(int predL[16][16], int predCb[8][8], int predCr[8][8], int QPy);                              /* :215 */
void transformDecodingChroma(int ChromaDCLevel[4], int ChromaACLevel[4][16], int predC[8][8], int QPy, unsigned char Cb);  /* :237 */
void residual_block_cavlc_write(int coeffLevel[16], int startIdx, int endIdx, int maxNumCoeff);                            /* F/residual.cpp:374 */
unsigned int residual_block_cavlc_size(int coeffLevel[16], int startIdx, int endIdx, int maxNumCoeff);                     /* :673 */
void MotionCompensateSubMBPart(int predL[16][16], int predCr[8][8], int predCb[8][8], frame_type *refPic, int mbPartIdx, int subMbIdx,
                               int subMbPartIdx);                                                                         /* F/mocomp.cpp:152 */
void Decode(int predL[16][16], int predCr[8][8], int predCb[8][8]);                                                        /* :200 */
/* coded_mb_size (F/rbsp_encoding.cpp:330) and intraPrediction / intraPredictionEncoding (F/intra.cpp:770,949) have no
 * stand-alone shim: their device form is the body of k_intra_mb, which needs the reconstructed neighbourhood of a whole
 * picture.  Their unit parity is checked on every macroblock of a picture through ferhip_read_buffer(FERHIP_BUF_MBSIZE). */

#ifdef __cplusplus
}
#endif
#endif
