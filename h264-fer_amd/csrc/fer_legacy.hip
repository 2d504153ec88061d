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
    if (type == 5 || type == 1) pic.resize((size_t)frame.Lwidth * frame.Lheight * 3 / 2);
    int rc = ferhip_dec_nal(g_dec, type, (int)nal_unit.nal_ref_idc, nal_unit.rbsp_byte, nal_unit.NumBytesInRBSP,
                            pic.empty() ? nullptr : pic.data(), &got, &W, &H);
    if (rc) {
        fprintf(stderr, "RBSP_decode: NAL unit type %d failed (%d)\n", type, rc);
        return;
    }
    if (type == 7) {  // fill_sps + init_h264_structures: `frame` takes the picture size
        if (frame.L && (frame.Lwidth != W || frame.Lheight != H)) ferhip_legacy_frame_drop();
        frame.Lwidth = W;
        frame.Lheight = H;
        frame.Cwidth = W >> 1;
        frame.Cheight = H >> 1;
        ferhip_legacy_frame_alloc();
    }
    if (got) {
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
