// fer_legacy.hip -- the reference's global-state entry points (include/ferhip_legacy.h) as thin
// shims over a one-stream context of the context-based ABI.  No hot-path arithmetic here.
#include "../../include/ferhip.h"
#include "../../include/ferhip_legacy.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <vector>

extern "C" {
frame_type frame;
int _qParameter = 12;       // F/h264_globals.cpp:217
int BasicInterEncoding = 0; // F/h264_globals.cpp:301-306
int WindowSize = 16;
int MAXDIFF_SET = -1;
int IntraEvery = 30;        // the reference leaves it 0 until PostaviParametre (division by zero at F/ref_frames.cpp:191)
int currFrameCount = 0;
int brojTipova[5];
int vrijeme = 0;
}

static ferhip_ctx *g_ctx = nullptr;
static int g_have_dpb = 0;
static ferhip_dec *g_dec = nullptr;
extern "C" void ferhip_legacy_frame_alloc(void);
extern "C" void ferhip_legacy_frame_drop(void);

extern "C" frame_type dpb;
extern "C" int ferhip_legacy_slice_type;
static std::vector<unsigned char> g_dpb_store;
static void legacy_dpb_store(const unsigned char *pic, size_t ys, size_t cs)
{
    g_dpb_store.assign(pic, pic + ys + 2 * cs);
    dpb.Lwidth = frame.Lwidth;
    dpb.Lheight = frame.Lheight;
    dpb.Cwidth = frame.Cwidth;
    dpb.Cheight = frame.Cheight;
    dpb.L = g_dpb_store.data();
    dpb.C[0] = g_dpb_store.data() + ys;
    dpb.C[1] = g_dpb_store.data() + ys + cs;
}

static int ensure_ctx()
{
    if (g_ctx) return 0;
    ferhip_params p = {_qParameter, BasicInterEncoding, WindowSize, MAXDIFF_SET, IntraEvery > 0 ? IntraEvery : 1};
    int rc = ferhip_create(&g_ctx, frame.Lwidth, frame.Lheight, 1, &p);
    if (rc) fprintf(stderr, "RBSP_encode: ferhip_create(%dx%d) failed (%d); there is no CPU fallback\n", frame.Lwidth, frame.Lheight, rc);
    return rc;
}

extern "C" void RBSP_encode(NALunit *nu)
{
    if (!nu || !nu->rbsp_byte) return;
    nu->NumBytesInRBSP = 0;
    if (ensure_ctx()) return;
    if (nu->nal_unit_type == 7) {
        nu->NumBytesInRBSP = (unsigned)ferhip_write_sps(g_ctx, nu->rbsp_byte, 500000);
        ferhip_legacy_frame_alloc();  // init_h264_structures_encoder(): `frame` exists from the SPS on
        return;
    }
    if (nu->nal_unit_type == 8) {
        nu->NumBytesInRBSP = (unsigned)ferhip_write_pps(g_ctx, nu->rbsp_byte, 500000);
        return;
    }
    if (nu->nal_unit_type != 5 && nu->nal_unit_type != 1) return;
    clock_t t0 = clock();
    size_t ys = (size_t)frame.Lwidth * frame.Lheight, cs = ys / 4;
    std::vector<unsigned char> pic(ys + 2 * cs);
    memcpy(pic.data(), frame.L, ys);
    memcpy(pic.data() + ys, frame.C[0], cs);
    memcpy(pic.data() + ys + cs, frame.C[1], cs);
    int type = (int)nu->nal_unit_type;
    uint32_t len = 0;
    size_t cap = (size_t)(frame.Lwidth / 16) * (frame.Lheight / 16) * 1024 + 4096;  // the library's own RBSP capacity
    std::vector<unsigned char> rbsp(cap);
    int before[5], after[5];
    ferhip_get_stats(g_ctx, before);
    int rc = ferhip_set_frames(g_ctx, pic.data(), 1);
    if (!rc) rc = ferhip_encode_picture(g_ctx, &type, rbsp.data(), cap, &len);
    if (!rc) rc = ferhip_get_recon(g_ctx, pic.data(), 1);
    if (rc) {
        fprintf(stderr, "RBSP_encode: GPU encode failed (%d)\n", rc);
        return;
    }
    memcpy(nu->rbsp_byte, rbsp.data(), len);  // the caller owns rbsp_byte and its size, as in the reference
    nu->NumBytesInRBSP = len;
    memcpy(frame.L, pic.data(), ys);  // `frame` now holds the reconstruction (F/inttransform.cpp:62-131)
    memcpy(frame.C[0], pic.data() + ys, cs);
    memcpy(frame.C[1], pic.data() + ys + cs, cs);
    ferhip_get_stats(g_ctx, after);
    for (int i = 0; i < 5; i++) brojTipova[i] += after[i] - before[i];
    g_have_dpb = 1;
    legacy_dpb_store(pic.data(), ys, cs);  // frameDeepCopy (F/ref_frames.cpp:17): `dpb` for the per-macroblock Decode()
    ferhip_legacy_slice_type = type == 5 ? 2 : 0;
    vrijeme = (int)(clock() - t0);
}

// selectNALUnitType of F/ref_frames.cpp:185-234 on the picture currently in `frame`: the frame SAD
// against the reference picture is evaluated on the device.
extern "C" int selectNALUnitType(void)
{
    if (!g_have_dpb || ensure_ctx()) return 5;
    size_t ys = (size_t)frame.Lwidth * frame.Lheight, cs = ys / 4;
    std::vector<unsigned char> pic(ys + 2 * cs);
    memcpy(pic.data(), frame.L, ys);
    memcpy(pic.data() + ys, frame.C[0], cs);
    memcpy(pic.data() + ys + cs, frame.C[1], cs);
    int type = 5;
    if (ferhip_set_frames(g_ctx, pic.data(), 1) || ferhip_select_nal_type(g_ctx, &type)) return 5;
    return type;
}

// RBSP_decode(NALunit nal_unit), F/rbsp_decoding.h:3, F/rbsp_decoding.cpp:17-367, the callee of decode()'s
// getNAL loop (F/fer_h264.cpp:37-47): nal_unit_type 7 / 8 take the parameter sets (the SPS sizes and allocates
// `frame`), 5 / 1 decode one picture into `frame`, which becomes the reference picture, and append it to
// `yuvoutput` with writeToY4M() when that file is open (F/rbsp_decoding.cpp:364); other types are ignored.
// The macroblock loop runs on the GPU (ferhip_dec_nal); there is no CPU fallback.
extern "C" void RBSP_decode(NALunit nal_unit)
{
    if (!nal_unit.rbsp_byte || nal_unit.NumBytesInRBSP == 0) return;
    if (!g_dec && ferhip_dec_create(&g_dec)) {
        fprintf(stderr, "RBSP_decode: no GPU decoder (there is no CPU fallback)\n");
        return;
    }
    const int type = (int)nal_unit.nal_unit_type;
    int got = 0, W = 0, H = 0;
    std::vector<unsigned char> pic;
    // the decoder's own picture size (from its SPS), not whatever `frame` says now: `frame` is shared with the encoder seam
    // and the file readers, which may have resized it since
    static int dec_W = 0, dec_H = 0;
    if (type == 5 || type == 1) {
        if (dec_W <= 0 || dec_H <= 0) {
            fprintf(stderr, "RBSP_decode: slice before any sequence parameter set\n");
            return;
        }
        pic.resize((size_t)dec_W * dec_H * 3 / 2);
    }
    int rc = ferhip_dec_nal(g_dec, type, (int)nal_unit.nal_ref_idc, nal_unit.rbsp_byte, nal_unit.NumBytesInRBSP,
                            pic.empty() ? nullptr : pic.data(), &got, &W, &H);
    if (rc) {
        fprintf(stderr, "RBSP_decode: NAL unit type %d failed (%d)\n", type, rc);
        return;
    }
    if (type == 7) {  // fill_sps + init_h264_structures: `frame` takes the picture size
        dec_W = W;
        dec_H = H;
        if (frame.L && (frame.Lwidth != W || frame.Lheight != H)) ferhip_legacy_frame_drop();
        frame.Lwidth = W;
        frame.Lheight = H;
        frame.Cwidth = W >> 1;
        frame.Cheight = H >> 1;
        ferhip_legacy_frame_alloc();
    }
    if (got) {
        if (!frame.L || frame.Lwidth != W || frame.Lheight != H) {  // `frame` was resized behind the decoder's back
            if (frame.L) ferhip_legacy_frame_drop();
            frame.Lwidth = W;
            frame.Lheight = H;
            frame.Cwidth = W >> 1;
            frame.Cheight = H >> 1;
            ferhip_legacy_frame_alloc();
        }
        size_t ys = (size_t)W * H, cs = ys / 4;
        memcpy(frame.L, pic.data(), ys);
        memcpy(frame.C[0], pic.data() + ys, cs);
        memcpy(frame.C[1], pic.data() + ys + cs, cs);
        writeToY4M();
    }
}

// ---- block-level entry points under the reference's names (F/quantizationTransform.h, F/scaleTransform.h): one block per
// call through the batched device entry points of ferhip.h
static void legacy_block(const char *name, int rc)
{
    if (rc) fprintf(stderr, "%s: device call failed (%d)\n", name, rc);
}
extern "C" void forwardResidual(int qP, int c[4][4], int r[4][4], unsigned char Intra, unsigned char Intra16x16OrChroma)
{
    (void)Intra;  // unused by the reference as well (F/quantizationTransform.cpp:183-223)
    int32_t o[16];
    int rc = ferhip_forward_residual(qP, &c[0][0], o, Intra16x16OrChroma ? 1 : 0, 1);
    legacy_block("forwardResidual", rc);
    if (!rc) memcpy(&r[0][0], o, sizeof o);
}
extern "C" void transformScan(int c[4][4], int list[16], unsigned char Intra16x16AC)
{
    int32_t o[16];
    int rc = ferhip_transform_scan(&c[0][0], o, Intra16x16AC ? 1 : 0, 1);
    legacy_block("transformScan", rc);
    if (!rc) memcpy(list, o, sizeof(int32_t) * (Intra16x16AC ? 15 : 16));
}
extern "C" void forwardDCLumaIntra(int qP, int dcY[4][4], int c[4][4])
{
    int32_t o[16];
    int rc = ferhip_forward_dc_luma_intra(qP, &dcY[0][0], o, 1);
    legacy_block("forwardDCLumaIntra", rc);
    if (!rc) memcpy(&c[0][0], o, sizeof o);
}
extern "C" void forwardDCChroma(int qP, int dcC[2][2], int c[2][2], unsigned char Intra)
{
    (void)Intra;
    int32_t i[16] = {dcC[0][0], dcC[0][1], dcC[1][0], dcC[1][1]}, o[16];
    int rc = ferhip_forward_dc_chroma(qP, i, o, 1);
    legacy_block("forwardDCChroma", rc);
    if (!rc) memcpy(&c[0][0], o, sizeof(int32_t) * 4);
}
extern "C" void transformInverseScan(int list[16], int c[4][4])
{
    int32_t o[16];
    int rc = ferhip_transform_inverse_scan(list, o, 1);
    legacy_block("transformInverseScan", rc);
    if (!rc) memcpy(&c[0][0], o, sizeof o);
}
extern "C" void inverseResidual(int bitDepth, int qP, int c[4][4], int r[4][4], unsigned char intra16x16OrChroma)
{
    (void)bitDepth;  // 8 everywhere in the reference
    int32_t o[16];
    int rc = ferhip_inverse_residual(qP, &c[0][0], o, intra16x16OrChroma ? 1 : 0, 1);
    legacy_block("inverseResidual", rc);
    if (!rc) memcpy(&r[0][0], o, sizeof o);
}
extern "C" void InverseDCLumaIntra(int bitDepth, int qP, int c[4][4], int dcY[4][4])
{
    (void)bitDepth;
    int32_t o[16];
    int rc = ferhip_inverse_dc_luma_intra(qP, &c[0][0], o, 1);
    legacy_block("InverseDCLumaIntra", rc);
    if (!rc) memcpy(&dcY[0][0], o, sizeof o);
}
extern "C" void InverseDCChroma(int bitDepth, int qP, int c[2][2], int dcC[2][2])
{
    (void)bitDepth;
    int32_t i[16] = {c[0][0], c[0][1], c[1][0], c[1][1]}, o[16];
    int rc = ferhip_inverse_dc_chroma(qP, i, o, 1);
    legacy_block("InverseDCChroma", rc);
    if (!rc) memcpy(&dcC[0][0], o, sizeof(int32_t) * 4);
}

// ---- per-macroblock entry points under the reference's names: thin shims over ferhip_mb_unit /
// ferhip_cavlc_blocks / ferhip_mc_sub_mb_parts (fer_mbunit.hip).  They read and write the globals the reference's
// functions read and write -- CurrMbAddr, QPy, mb_type, the level arrays, `frame`, mvL0x / mvL0y -- one macroblock per
// call: a seam for the maintainer's unit tests.  What the reference keeps in file-statics or in `shd` and a caller of
// the bare function cannot set there is exported under a ferhip_legacy_ name (include/ferhip_legacy.h).
extern "C" {
int CurrMbAddr = 0, QPy = 12, mb_type = 0;
int ferhip_legacy_slice_type = 2;
int ferhip_legacy_nC = 0;
unsigned char ferhip_legacy_bits[1 << 16];
unsigned int ferhip_legacy_nbits = 0;
int LumaLevel[16][16], Intra16x16DCLevel[16], Intra16x16ACLevel[16][16], ChromaDCLevel[2][4], ChromaACLevel[2][4][16];
int ***mvL0x = nullptr, ***mvL0y = nullptr;
frame_type dpb;
int ferhip_chroma_qp(int qpy);
}
static int g_mv_nmb = 0;

static int pred_class_of(int t)  // MbPartPredMode(mb_type, 0), F/h264_globals.h:123 over the tables of F/h264_globals.cpp:25-132
{
    if (ferhip_legacy_slice_type % 5 == 0) {
        if (t == 5) return 0;
        return (t >= 6 && t <= 29) ? 1 : 2;
    }
    if (t == 0) return 0;
    return (t >= 1 && t <= 24) ? 1 : 2;
}
static void mb_origin(int &xP, int &yP)
{
    const int mbw = frame.Lwidth >> 4;
    xP = (CurrMbAddr % mbw) << 4;
    yP = (CurrMbAddr / mbw) << 4;
}
static void job_preds(ferhip_mb_job &J, int predL[16][16], int predCb[8][8], int predCr[8][8])
{
    for (int i = 0; i < 256; i++) J.predY[i] = predL ? predL[i >> 4][i & 15] : 0;
    for (int i = 0; i < 64; i++) {
        J.predCb[i] = predCb ? predCb[i >> 3][i & 7] : 0;
        J.predCr[i] = predCr ? predCr[i >> 3][i & 7] : 0;
    }
}
static void put_luma(const ferhip_mb_result &R, int blk_only)
{
    static const int bx[16] = {0, 4, 0, 4, 8, 12, 8, 12, 0, 4, 0, 4, 8, 12, 8, 12}, by[16] = {0, 0, 4, 4, 0, 0, 4, 4, 8, 8, 12, 12, 8, 8, 12, 12};
    int xP, yP;
    mb_origin(xP, yP);
    for (int y = 0; y < 16; y++)
        for (int x = 0; x < 16; x++) {
            if (blk_only >= 0 && !(x >= bx[blk_only] && x < bx[blk_only] + 4 && y >= by[blk_only] && y < by[blk_only] + 4)) continue;
            frame.L[(size_t)(yP + y) * frame.Lwidth + xP + x] = (unsigned char)R.recY[y * 16 + x];
        }
}
static void put_chroma(const ferhip_mb_result &R, int plane)  // plane 0 = Cb, 1 = Cr
{
    int xP, yP;
    mb_origin(xP, yP);
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++)
            frame.C[plane][(size_t)(yP / 2 + y) * frame.Cwidth + xP / 2 + x] = (unsigned char)(plane ? R.recCr : R.recCb)[y * 8 + x];
}

extern "C" void quantizationTransform(int predL[16][16], int predCb[8][8], int predCr[8][8], unsigned char reconstruct)
{
    if (!frame.L) return;
    ferhip_mb_job *J = new ferhip_mb_job();
    ferhip_mb_result *R = new ferhip_mb_result();
    J->op = FERHIP_MBU_QT;
    J->cls = pred_class_of(mb_type);
    J->qp = QPy;
    J->qpc = ferhip_chroma_qp(QPy);
    J->reconstruct = reconstruct ? 1 : 0;
    int xP, yP;
    mb_origin(xP, yP);
    for (int i = 0; i < 256; i++) J->srcY[i] = frame.L[(size_t)(yP + (i >> 4)) * frame.Lwidth + xP + (i & 15)];
    for (int i = 0; i < 64; i++) {
        J->srcCb[i] = frame.C[0][(size_t)(yP / 2 + (i >> 3)) * frame.Cwidth + xP / 2 + (i & 7)];
        J->srcCr[i] = frame.C[1][(size_t)(yP / 2 + (i >> 3)) * frame.Cwidth + xP / 2 + (i & 7)];
    }
    job_preds(*J, predL, predCb, predCr);
    int rc = ferhip_mb_unit(J, R, 1);
    legacy_block("quantizationTransform", rc);
    if (!rc) {
        if (J->cls == 2) memcpy(LumaLevel, R->lumaLevel, sizeof LumaLevel);
        if (J->cls == 1) {
            memcpy(Intra16x16DCLevel, R->dc16, sizeof Intra16x16DCLevel);
            memcpy(Intra16x16ACLevel, R->ac16, sizeof Intra16x16ACLevel);
        }
        memcpy(ChromaDCLevel, R->cdc, sizeof ChromaDCLevel);
        memcpy(ChromaACLevel, R->cac, sizeof ChromaACLevel);
        if (reconstruct) {
            if (J->cls != 0) put_luma(*R, -1);
            put_chroma(*R, 0);
            put_chroma(*R, 1);
        }
    }
    delete J;
    delete R;
}

static void decode_op(const char *name, int op, int predL[16][16], int predCb[8][8], int predCr[8][8], int qpy, int blk, int (*ll)[16],
                      int *dc, int (*ac)[16], int *cdc, int (*cac)[16], int plane)
{
    if (!frame.L) return;
    ferhip_mb_job *J = new ferhip_mb_job();
    ferhip_mb_result *R = new ferhip_mb_result();
    J->op = op;
    J->qp = qpy;
    J->qpc = ferhip_chroma_qp(qpy);
    J->blk = blk;
    job_preds(*J, predL, predCb, predCr);
    if (ll) memcpy(J->lumaLevel, ll, sizeof J->lumaLevel);
    if (dc) memcpy(J->dc16, dc, sizeof J->dc16);
    if (ac) memcpy(J->ac16, ac, sizeof J->ac16);
    if (cdc) memcpy(J->cdc[plane], cdc, sizeof(int) * 4);
    if (cac) memcpy(J->cac[plane], cac, sizeof(int) * 64);
    int rc = ferhip_mb_unit(J, R, 1);
    legacy_block(name, rc);
    if (!rc) {
        if (op == FERHIP_MBU_DEC4) put_luma(*R, blk);
        if (op == FERHIP_MBU_DEC16 || op == FERHIP_MBU_SKIP) put_luma(*R, -1);
        if (op == FERHIP_MBU_DECC) put_chroma(*R, plane);
        if (op == FERHIP_MBU_SKIP) {
            put_chroma(*R, 0);
            put_chroma(*R, 1);
        }
    }
    delete J;
    delete R;
}
extern "C" void transformDecoding4x4LumaResidual(int LumaLevel_[16][16], int predL[16][16], int luma4x4BlkIdx, int QPy_)
{
    decode_op("transformDecoding4x4LumaResidual", FERHIP_MBU_DEC4, predL, nullptr, nullptr, QPy_, luma4x4BlkIdx, LumaLevel_, nullptr, nullptr, nullptr, nullptr, 0);
}
extern "C" void transformDecodingIntra_16x16Luma(int Intra16x16DCLevel_[16], int Intra16x16ACLevel_[16][16], int predL[16][16], int QPy_)
{
    decode_op("transformDecodingIntra_16x16Luma", FERHIP_MBU_DEC16, predL, nullptr, nullptr, QPy_, 0, nullptr, Intra16x16DCLevel_, Intra16x16ACLevel_, nullptr, nullptr, 0);
}
extern "C" void transformDecodingP_Skip(int predL[16][16], int predCb[8][8], int predCr[8][8], int QPy_)
{
    decode_op("transformDecodingP_Skip", FERHIP_MBU_SKIP, predL, predCb, predCr, QPy_, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
}
extern "C" void transformDecodingChroma(int ChromaDCLevel_[4], int ChromaACLevel_[4][16], int predC[8][8], int QPy_, unsigned char Cb)
{
    const int plane = Cb ? 0 : 1;
    decode_op("transformDecodingChroma", FERHIP_MBU_DECC, nullptr, Cb ? predC : nullptr, Cb ? nullptr : predC, QPy_, 0, nullptr, nullptr, nullptr, ChromaDCLevel_,
              ChromaACLevel_, plane);
}

// residual_block_cavlc_write / _size, F/residual.cpp:374 / :673.  The reference derives nC inside from file-statics
// (invoked_for_*, i8x8, i4x4 ...) that only its own residual_write() sets; here the caller states it in
// ferhip_legacy_nC (-1 = chroma DC).  The written bits are appended to ferhip_legacy_bits (MSB first, ferhip_legacy_nbits
// counts them): the reference's bit writer (F/rbsp_IO.cpp, a leaf file a maintainer keeps compiling) is not part of
// this library.
static unsigned cavlc_one(int coeffLevel[16], int maxNumCoeff, bool write, const char *name)
{
    int32_t nc = ferhip_legacy_nC, mx = maxNumCoeff, tc = 0;
    uint32_t nb = 0;
    uint8_t out[64];
    int rc = ferhip_cavlc_blocks(coeffLevel, &nc, &mx, 1, out, &nb, &tc);
    legacy_block(name, rc);
    if (rc) return 0;
    if (write) {
        for (uint32_t b = 0; b < nb && ferhip_legacy_nbits < sizeof(ferhip_legacy_bits) * 8; b++) {
            const unsigned bit = (out[b >> 3] >> (7 - (b & 7))) & 1u;
            const unsigned pos = ferhip_legacy_nbits++;
            if ((pos & 7) == 0) ferhip_legacy_bits[pos >> 3] = 0;
            ferhip_legacy_bits[pos >> 3] |= (unsigned char)(bit << (7 - (pos & 7)));
        }
    }
    return nb;
}
extern "C" void residual_block_cavlc_write(int coeffLevel[16], int startIdx, int endIdx, int maxNumCoeff)
{
    (void)startIdx;  // 0 and maxNumCoeff - 1 at every call site of the reference
    (void)endIdx;
    cavlc_one(coeffLevel, maxNumCoeff, true, "residual_block_cavlc_write");
}
extern "C" unsigned int residual_block_cavlc_size(int coeffLevel[16], int startIdx, int endIdx, int maxNumCoeff)
{
    (void)startIdx;
    (void)endIdx;
    return cavlc_one(coeffLevel, maxNumCoeff, false, "residual_block_cavlc_size");
}

// AllocateMemory() of F/mode_pred.cpp:22-39: the vector arrays [macroblock][subMbIdx][subMbPartIdx]
extern "C" void AllocateMemory(void)
{
    const int nmb = (frame.Lwidth >> 4) * (frame.Lheight >> 4);
    if (mvL0x && g_mv_nmb == nmb) return;
    auto alloc3 = [&](int ***&p) {
        p = new int **[nmb];
        for (int m = 0; m < nmb; m++) {
            p[m] = new int *[4];
            for (int s2 = 0; s2 < 4; s2++) p[m][s2] = new int[4]();
        }
    };
    alloc3(mvL0x);  // (like the reference, never freed)
    alloc3(mvL0y);
    g_mv_nmb = nmb;
}

static void pack_i420(const frame_type *f, std::vector<unsigned char> &pic)
{
    const size_t ys = (size_t)f->Lwidth * f->Lheight, cs = ys / 4;
    pic.resize(ys + 2 * cs);
    memcpy(pic.data(), f->L, ys);
    memcpy(pic.data() + ys, f->C[0], cs);
    memcpy(pic.data() + ys + cs, f->C[1], cs);
}
static void mc_parts(const char *name, int predL[16][16], int predCr[8][8], int predCb[8][8], frame_type *refPic, int mb, int nparts, const int *subs,
                     const int *parts)
{
    if (!refPic || !refPic->L || !mvL0x) return;
    std::vector<unsigned char> pic;
    pack_i420(refPic, pic);
    std::vector<int32_t> desc(nparts * 5), pl(nparts * 16), pb(nparts * 4), pr(nparts * 4);
    for (int k = 0; k < nparts; k++) {
        desc[k * 5] = mb;
        desc[k * 5 + 1] = subs[k];
        desc[k * 5 + 2] = parts[k];
        desc[k * 5 + 3] = mvL0x[mb][subs[k]][parts[k]];
        desc[k * 5 + 4] = mvL0y[mb][subs[k]][parts[k]];
    }
    int rc = ferhip_mc_sub_mb_parts(pic.data(), refPic->Lwidth, refPic->Lheight, desc.data(), nparts, pl.data(), pb.data(), pr.data());
    legacy_block(name, rc);
    if (rc) return;
    for (int k = 0; k < nparts; k++) {
        const int oy = ((subs[k] & 2) << 2) + ((parts[k] & 2) << 1), ox = ((subs[k] & 1) << 3) + ((parts[k] & 1) << 2);
        for (int i = 0; i < 16; i++) predL[oy + (i >> 2)][ox + (i & 3)] = pl[k * 16 + i];
        for (int i = 0; i < 4; i++) {
            predCb[oy / 2 + (i >> 1)][ox / 2 + (i & 1)] = pb[k * 4 + i];
            predCr[oy / 2 + (i >> 1)][ox / 2 + (i & 1)] = pr[k * 4 + i];
        }
    }
}
// MotionCompensateSubMBPart(predL, predCr, predCb, refPic, mbPartIdx, subMbIdx, subMbPartIdx), F/mocomp.cpp:152-195
// (mbPartIdx is the macroblock address there)
extern "C" void MotionCompensateSubMBPart(int predL[16][16], int predCr[8][8], int predCb[8][8], frame_type *refPic, int mbPartIdx, int subMbIdx,
                                          int subMbPartIdx)
{
    mc_parts("MotionCompensateSubMBPart", predL, predCr, predCb, refPic, mbPartIdx, 1, &subMbIdx, &subMbPartIdx);
}
// Decode(predL, predCr, predCb), F/mocomp.cpp:200-209: the sixteen sub-blocks of macroblock CurrMbAddr against `dpb`
extern "C" void Decode(int predL[16][16], int predCr[8][8], int predCb[8][8])
{
    int subs[16], parts[16];
    for (int k = 0; k < 16; k++) {
        subs[k] = k >> 2;
        parts[k] = k & 3;
    }
    mc_parts("Decode", predL, predCr, predCb, &dpb, CurrMbAddr, 16, subs, parts);
}
