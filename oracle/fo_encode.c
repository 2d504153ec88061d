/*
 * fo_encode.c -- ORACLE (test infrastructure): row a14 and the "next" rows
 * NEXT-1/NEXT-2 of SURVEY.md: context, SPS/PPS/slice header writers, the
 * RBSP_encode macroblock loop, NAL framing, IDR/P decision and a whole-stream
 * helper mirroring encode()/NastaviEncode().
 * Reference: F/rbsp_encoding.cpp:119-326, F/headers_and_parameter_sets.cpp,
 * F/nal.cpp:261-299, F/ref_frames.cpp, F/fer_h264.cpp:55-134.
 */
#include "fo.h"
#include <stdlib.h>
#include <string.h>

int fo_pred_class(const fo_ctx *c, int mb_type);

fo_ctx *fo_create(int W, int H)
{
    fo_ctx *c = (fo_ctx *)calloc(1, sizeof *c);
    c->W = W;
    c->H = H;
    c->Wc = W / 2;
    c->Hc = H / 2;
    c->mbw = W / 16;
    c->mbh = H / 16;
    c->nmb = c->mbw * c->mbh;
    c->L = (uint8_t *)calloc(1, (size_t)W * H);
    c->C[0] = (uint8_t *)calloc(1, (size_t)c->Wc * c->Hc);
    c->C[1] = (uint8_t *)calloc(1, (size_t)c->Wc * c->Hc);
    c->dL = (uint8_t *)calloc(1, (size_t)W * H);
    c->dC[0] = (uint8_t *)calloc(1, (size_t)c->Wc * c->Hc);
    c->dC[1] = (uint8_t *)calloc(1, (size_t)c->Wc * c->Hc);
    c->mb_type = (int *)calloc((size_t)c->nmb, sizeof(int));
    c->cbp_l = (int *)calloc((size_t)c->nmb, sizeof(int));
    c->cbp_c = (int *)calloc((size_t)c->nmb, sizeof(int));
    c->tc_l = calloc((size_t)c->nmb, sizeof *c->tc_l);
    c->tc_c = calloc((size_t)c->nmb, sizeof *c->tc_c);
    c->i4mode = (int *)calloc((size_t)c->nmb * 16, sizeof(int));
    c->mvx = calloc((size_t)c->nmb, sizeof *c->mvx);
    c->mvy = calloc((size_t)c->nmb, sizeof *c->mvy);
    c->refidx = (int *)calloc((size_t)c->nmb, sizeof(int));
    c->ref_idx_l0 = calloc((size_t)c->nmb, sizeof *c->ref_idx_l0);
    c->dbg_mvx = calloc((size_t)c->nmb, sizeof *c->dbg_mvx);
    c->dbg_mbsize = calloc((size_t)c->nmb, sizeof *c->dbg_mbsize);
    c->dbg_mvy = calloc((size_t)c->nmb, sizeof *c->dbg_mvy);
    /* defaults of F/h264_globals.cpp:217,301-306 and the GUI */
    c->qp = 12;
    c->basic = 0;
    c->window = 16;
    c->maxdiff_set = -1;
    c->intra_every = 30;
    c->MAXDIFF = 2;
    c->nal_ref_idc = 1;
    return c;
}

void fo_destroy(fo_ctx *c)
{
    if (!c) return;
    free(c->L);
    free(c->C[0]);
    free(c->C[1]);
    free(c->dL);
    free(c->dC[0]);
    free(c->dC[1]);
    free(c->mb_type);
    free(c->cbp_l);
    free(c->cbp_c);
    free(c->tc_l);
    free(c->tc_c);
    free(c->i4mode);
    free(c->mvx);
    free(c->mvy);
    free(c->refidx);
    free(c->ref_idx_l0);
    free(c->dbg_mvx);
    free(c->dbg_mbsize);
    free(c->dbg_mvy);
    for (int i = 0; i < 16; i++) {
        free(c->interp[i]);
        for (int k = 0; k < 5; k++) free(c->kar[k][i]);
    }
    for (int k = 0; k < 5; k++) {
        free(c->sorted[k]);
        free(c->sorted_tmp[k]);
    }
    free(c);
}

/* Starter::PostaviParametre, F/fer_h264.cpp:169-178 */
void fo_set_params(fo_ctx *c, int qp, int basic, int window, int maxdiff, int intra_every)
{
    c->qp = qp;
    c->basic = basic;
    c->window = window;
    c->maxdiff_set = maxdiff;
    c->intra_every = intra_every;
}

/* F/headers_and_parameter_sets.cpp:305-391 */
size_t fo_write_sps(fo_ctx *c, uint8_t *rbsp, size_t cap)
{
    fo_bw w;
    fo_bw_init(&w, rbsp, cap);
    fo_bw_put(&w, 8, 66);
    fo_bw_put(&w, 1, 1);
    fo_bw_put(&w, 1, 1);
    fo_bw_put(&w, 1, 0);
    fo_bw_put(&w, 5, 0);
    fo_bw_put(&w, 8, 41);
    fo_bw_ue(&w, 0);     /* sps id */
    fo_bw_ue(&w, 9 - 4); /* log2_max_frame_num - 4 */
    fo_bw_ue(&w, 0);     /* poc type */
    fo_bw_ue(&w, 10 - 4);
    fo_bw_ue(&w, 1); /* max_num_ref_frames */
    fo_bw_put(&w, 1, 0);
    fo_bw_ue(&w, (unsigned)(c->mbw - 1));
    fo_bw_ue(&w, (unsigned)(c->mbh - 1));
    fo_bw_put(&w, 1, 1); /* frame_mbs_only */
    fo_bw_put(&w, 1, 1); /* direct_8x8_inference */
    fo_bw_put(&w, 1, 0); /* cropping */
    fo_bw_put(&w, 1, 0); /* vui */
    c->log2_max_frame_num = 9;
    c->log2_max_poc_lsb = 10;
    return fo_bw_trailing(&w);
}

/* F/headers_and_parameter_sets.cpp:478-513 */
size_t fo_write_pps(fo_ctx *c, uint8_t *rbsp, size_t cap)
{
    fo_bw w;
    fo_bw_init(&w, rbsp, cap);
    c->pic_init_qp = 14 + c->qp;
    c->chroma_qp_offset = 0;
    fo_bw_ue(&w, 0);
    fo_bw_ue(&w, 0);
    fo_bw_put(&w, 1, 0); /* CAVLC */
    fo_bw_put(&w, 1, 0);
    fo_bw_ue(&w, 0); /* slice groups - 1 */
    fo_bw_ue(&w, 0);
    fo_bw_ue(&w, 0);
    fo_bw_put(&w, 1, 0); /* weighted_pred */
    fo_bw_put(&w, 2, 1); /* QUIRK: num_ref_idx_l1_active written as weighted_bipred_idc (:505) */
    fo_bw_se(&w, c->pic_init_qp - 26);
    fo_bw_se(&w, 0);
    fo_bw_se(&w, 0);
    fo_bw_put(&w, 1, 0);
    fo_bw_put(&w, 1, 0);
    fo_bw_put(&w, 1, 0);
    return fo_bw_trailing(&w);
}

/* F/ref_frames.cpp:185-234 (CPU path).  c->L holds the new source picture. */
int fo_select_nal_type(fo_ctx *c)
{
    if (!c->have_dpb || c->frames_done % c->intra_every == 0) return FO_NAL_IDR;
    unsigned long sad = 0;
    for (size_t i = 0; i < (size_t)c->W * c->H; i++) {
        int d = (int)c->L[i] - (int)c->dL[i];
        sad += (unsigned long)(d < 0 ? -d : d);
    }
    if (sad > ((unsigned long)c->nmb << 12)) return FO_NAL_IDR;
    return FO_NAL_SLICE;
}

/* shd_write, F/headers_and_parameter_sets.cpp:172-239 */
static void slice_header(fo_ctx *c, fo_bw *w, int nal_type)
{
    if (c->slice_type == 2)
        c->poc_lsb = 0;
    else
        c->poc_lsb += 2;
    c->QPy = c->pic_init_qp - 14;
    fo_bw_ue(w, 0);
    fo_bw_ue(w, (unsigned)c->slice_type);
    fo_bw_ue(w, 0);
    fo_bw_put(w, c->log2_max_frame_num, (uint32_t)c->frame_num);
    if (nal_type == FO_NAL_IDR) fo_bw_ue(w, (unsigned)c->idr_pic_id);
    fo_bw_put(w, c->log2_max_poc_lsb, (uint32_t)c->poc_lsb);
    if (c->slice_type == 0) fo_bw_put(w, 1, 0); /* num_ref_idx_active_override_flag */
    if (c->slice_type == 0) fo_bw_put(w, 1, 0); /* ref_pic_list_modification_flag_l0 */
    if (c->nal_ref_idc != 0) {
        if (nal_type == FO_NAL_IDR) {
            fo_bw_put(w, 1, 0);
            fo_bw_put(w, 1, 0);
        } else {
            fo_bw_put(w, 1, 0);
        }
    }
    fo_bw_se(w, -14);
}

static void dpb_copy(fo_ctx *c)
{
    memcpy(c->dL, c->L, (size_t)c->W * c->H);
    memcpy(c->dC[0], c->C[0], (size_t)c->Wc * c->Hc);
    memcpy(c->dC[1], c->C[1], (size_t)c->Wc * c->Hc);
    c->have_dpb = 1;
}

/* RBSP_encode for IDR / non-IDR slices, F/rbsp_encoding.cpp:139-323 */
size_t fo_encode_slice(fo_ctx *c, int nal_type, uint8_t *rbsp, size_t cap)
{
    fo_bw w;
    fo_bw_init(&w, rbsp, cap);
    if (!c->pic_init_qp) c->pic_init_qp = 14 + c->qp;
    if (!c->log2_max_frame_num) {
        c->log2_max_frame_num = 9;
        c->log2_max_poc_lsb = 10;
    }
    if (nal_type == FO_NAL_IDR) {
        c->slice_type = 2;
        if (!c->first_idr_done) {
            c->first_idr_done = 1;
            c->idr_pic_id = 0;
        } else if (c->frame_num == 0) {
            c->idr_pic_id++;
        } else {
            c->idr_pic_id = 0;
        }
        c->frame_num = 0;
    } else {
        c->slice_type = 0;
        c->frame_num++;
    }
    slice_header(c, &w, nal_type);

    int predL[16][16], predCb[8][8], predCr[8][8];
    int mb_skip_run = 0;
    for (c->cur = 0; c->cur < c->nmb; c->cur++) {
        if (c->slice_type != 2) {
            fo_interEncoding(c, predL, predCr, predCb);
            c->mb_type[c->cur] = c->cur_mb_type;
            if (c->cur_mb_type == FO_P_SKIP) {
                mb_skip_run++;
                fo_transformDecodingPSkip(c, predL, predCb, predCr, c->QPy);
                continue;
            }
            fo_bw_ue(&w, (unsigned)mb_skip_run);
            mb_skip_run = 0;
            fo_quantizationTransform(c, predL, predCb, predCr, 1);
            fo_setCodedBlockPattern(c);
        } else {
            int m16 = fo_intraPredictionEncoding(c, predL, predCr, predCb);
            if (m16 == -1) {
                c->cur_mb_type = FO_I_4x4;
                fo_quantizationTransform(c, predL, predCb, predCr, 1);
                fo_setCodedBlockPattern(c);
            } else {
                c->cur_mb_type = m16 + 1;
                fo_quantizationTransform(c, predL, predCb, predCr, 1);
                fo_setCodedBlockPattern(c);
                c->cur_mb_type += c->cbpC << 2;
                if (c->cbpL == 15) c->cur_mb_type += 12;
            }
            c->mb_type[c->cur] = c->cur_mb_type;
        }
        int t = c->cur_mb_type;
        int pc = fo_pred_class(c, t);
        fo_bw_ue(&w, (unsigned)t);
        if (pc == 2 && (t == FO_P_8x8 || t == FO_P_8x8ref0)) {
            for (int i = 0; i < 4; i++) fo_bw_ue(&w, (unsigned)c->sub_mb_type[i]);
            for (int i = 0; i < 4; i++) { /* NumSubMbPart(P_L0_8x8) == 1 */
                fo_bw_se(&w, c->mvd[i][0][0]);
                fo_bw_se(&w, c->mvd[i][0][1]);
            }
        } else if (pc == 0 || pc == 1) {
            if (pc == 0)
                for (int b = 0; b < 16; b++) {
                    fo_bw_put(&w, 1, (uint32_t)c->prev_flag[b]);
                    if (!c->prev_flag[b]) fo_bw_put(&w, 3, (uint32_t)c->rem_mode[b]);
                }
            fo_bw_ue(&w, (unsigned)c->chroma_mode);
        } else {
            int np = (t == FO_P_L0_16x16) ? 1 : 2;
            for (int i = 0; i < np; i++) {
                fo_bw_se(&w, c->mvd[i][0][0]);
                fo_bw_se(&w, c->mvd[i][0][1]);
            }
        }
        if (pc != 1) {
            int cbp = (c->cbpC << 4) | c->cbpL;
            fo_bw_ue(&w, (unsigned)(pc == 0 ? fo_cbp_intra_to_code[cbp] : fo_cbp_inter_to_code[cbp]));
        }
        if (c->cbpL > 0 || c->cbpC > 0 || pc == 1) {
            fo_bw_se(&w, 0); /* mb_qp_delta */
            fo_residual_write(c, &w);
        } else {
            /* clear_residual_structures(), F/residual.cpp:28-49: luma levels and chroma DC only */
            memset(c->lv.Lumalevel, 0, sizeof c->lv.Lumalevel);
            memset(c->lv.DC16, 0, sizeof c->lv.DC16);
            memset(c->lv.AC16, 0, sizeof c->lv.AC16);
            memset(c->lv.CDC, 0, sizeof c->lv.CDC);
        }
    }
    if (mb_skip_run > 0) fo_bw_ue(&w, (unsigned)mb_skip_run);
    size_t n = fo_bw_trailing(&w);
    memcpy(c->dbg_mvx, c->mvx, (size_t)c->nmb * sizeof *c->mvx);
    memcpy(c->dbg_mvy, c->mvy, (size_t)c->nmb * sizeof *c->mvy);
    dpb_copy(c); /* initialisationProcess + modificationProcess -> frameDeepCopy */
    fo_fill_interpolated(c);
    c->frames_done++;
    return n;
}

/* F/nal.cpp:261-299 */
size_t fo_write_nal(int nal_ref_idc, int nal_type, const uint8_t *rbsp, size_t n, uint8_t *out)
{
    size_t pos = 0;
    out[pos++] = 0;
    out[pos++] = 0;
    out[pos++] = 0;
    out[pos++] = 1;
    out[pos++] = (uint8_t)((nal_ref_idc << 5) | (nal_type & 31));
    int zc = 0;
    for (size_t i = 0; i < n; i++) {
        if (zc >= 2 && rbsp[i] <= 3) {
            out[pos++] = 3;
            zc = 0;
        }
        out[pos++] = rbsp[i];
        if (rbsp[i] == 0)
            zc++;
        else
            zc = 0;
    }
    return pos;
}

/* encode() + NastaviEncode(), F/fer_h264.cpp:55-134 (file I/O replaced by buffers).
 * frames: nframes pictures of coded size, I420.  recon_out (optional) receives
 * the reconstruction of every picture. */
size_t fo_encode_stream(fo_ctx *c, const uint8_t *frames, int nframes, uint8_t *out, size_t cap,
                        uint8_t *recon_out)
{
    size_t fsz = (size_t)c->W * c->H * 3 / 2, ysz = (size_t)c->W * c->H, csz = ysz / 4;
    size_t rcap = fsz * 4 + 65536;
    uint8_t *rbsp = (uint8_t *)malloc(rcap);
    size_t pos = 0, n;
    n = fo_write_sps(c, rbsp, rcap);
    pos += fo_write_nal(1, FO_NAL_SPS, rbsp, n, out + pos);
    n = fo_write_pps(c, rbsp, rcap);
    pos += fo_write_nal(1, FO_NAL_PPS, rbsp, n, out + pos);
    for (int f = 0; f < nframes; f++) {
        const uint8_t *src = frames + (size_t)f * fsz;
        memcpy(c->L, src, ysz);
        memcpy(c->C[0], src + ysz, csz);
        memcpy(c->C[1], src + ysz + csz, csz);
        int type = fo_select_nal_type(c);
        n = fo_encode_slice(c, type, rbsp, rcap);
        if (pos + n * 3 / 2 + 16 > cap) break;
        pos += fo_write_nal(1, type, rbsp, n, out + pos);
        if (recon_out) {
            uint8_t *r = recon_out + (size_t)f * fsz;
            memcpy(r, c->L, ysz);
            memcpy(r + ysz, c->C[0], csz);
            memcpy(r + ysz + csz, c->C[1], csz);
        }
    }
    free(rbsp);
    return pos;
}
