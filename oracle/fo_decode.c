/*
 * fo_decode.c -- ORACLE (test infrastructure): row a19 of SURVEY.md 8a, the
 * decode twin of the hot path.  Annex-B scan and emulation-prevention removal
 * (F/nal.cpp:68-258), parameter-set / slice-header parse
 * (F/headers_and_parameter_sets.cpp:245-298,398-537) and the RBSP_decode
 * macroblock loop (F/rbsp_decoding.cpp:17-367).  The reference's decoder
 * quirks are kept (marked QUIRK): they are what its output -- and therefore the
 * md5 this oracle is pinned with -- contains.
 */
#include "fo.h"
#include <stdlib.h>
#include <string.h>

int fo_pred_class(const fo_ctx *c, int mb_type);
int fo_residual_parse(fo_ctx *c, fo_br *r);
void fo_intraPrediction_dec(fo_ctx *c, int predL[16][16], int predCr[8][8], int predCb[8][8]);

typedef struct {
    int have_sps;
    int W, H;
    int log2_max_frame_num, poc_type, log2_max_poc_lsb;
    int pic_init_qp, chroma_qp_offset, deblock_ctl, constrained_intra;
    int modification_flag_l0; /* persists across slices like the global shd */
    int mod_copies;
} dec_hdr;

static void parse_sps(dec_hdr *h, fo_br *r)
{
    fo_br_bits(r, 8);
    fo_br_bits(r, 8);
    fo_br_bits(r, 8);
    fo_br_ue(r);
    h->log2_max_frame_num = (int)fo_br_ue(r) + 4;
    h->poc_type = (int)fo_br_ue(r);
    h->log2_max_poc_lsb = 0;
    if (h->poc_type == 0) {
        h->log2_max_poc_lsb = (int)fo_br_ue(r) + 4;
    } else if (h->poc_type == 1) {
        fo_br_bits(r, 1);
        fo_br_se(r);
        fo_br_se(r);
        int n = (int)fo_br_ue(r);
        for (int i = 0; i < n; i++) fo_br_se(r);
    }
    fo_br_ue(r); /* max_num_ref_frames */
    fo_br_bits(r, 1);
    int wmb = (int)fo_br_ue(r) + 1;
    int hmu = (int)fo_br_ue(r) + 1;
    int fmo = (int)fo_br_bits(r, 1);
    h->W = wmb * 16;
    h->H = (2 - fmo) * hmu * 16;
    h->have_sps = 1;
}

static void parse_pps(dec_hdr *h, fo_br *r)
{
    fo_br_ue(r);
    fo_br_ue(r);
    fo_br_bits(r, 1);
    fo_br_bits(r, 1);
    fo_br_ue(r);
    fo_br_ue(r);
    fo_br_ue(r);
    fo_br_bits(r, 1);
    fo_br_bits(r, 2);
    h->pic_init_qp = fo_br_se(r) + 26;
    fo_br_se(r);
    h->chroma_qp_offset = fo_br_se(r);
    h->deblock_ctl = (int)fo_br_bits(r, 1);
    h->constrained_intra = (int)fo_br_bits(r, 1);
    fo_br_bits(r, 1);
}

static void clear_residual(fo_ctx *c)
{
    /* QUIRK: clear_residual_structures (F/residual.cpp:28-49) leaves ChromaACLevel alone */
    memset(c->lv.Lumalevel, 0, sizeof c->lv.Lumalevel);
    memset(c->lv.DC16, 0, sizeof c->lv.DC16);
    memset(c->lv.AC16, 0, sizeof c->lv.AC16);
    memset(c->lv.CDC, 0, sizeof c->lv.CDC);
}

static int p_num_part(int t) { return (t == 0) ? 1 : ((t == 1 || t == 2) ? 2 : ((t == 3 || t == 4) ? 4 : (t == 5 ? 0 : 0xff))); }
static int num_sub(int s)
{
    static const int n[4] = {1, 2, 2, 4};
    return (s >= 0 && s < 4) ? n[s] : 0;
}

/* RBSP_decode for slice NAL units */
static int decode_slice(fo_ctx *c, dec_hdr *h, int nal_type, int nal_ref_idc, const uint8_t *rbsp, size_t n)
{
    fo_br br, *r = &br;
    fo_br_init(r, rbsp, n);
    fo_br_ue(r); /* first_mb_in_slice */
    int st = (int)fo_br_ue(r);
    fo_br_ue(r);
    c->frame_num = (int)fo_br_bits(r, h->log2_max_frame_num);
    if (nal_type == 5) c->idr_pic_id = (int)fo_br_ue(r);
    c->poc_lsb = (int)fo_br_bits(r, h->log2_max_poc_lsb);
    int s5 = st % 5;
    if (s5 == 0 || s5 == 1 || s5 == 3) {
        c->num_ref_idx_override = (int)fo_br_bits(r, 1);
        if (c->num_ref_idx_override == 1) c->num_ref_idx_l0_active_minus1 = (int)fo_br_ue(r);
    }
    if (s5 != 2 && s5 != 4) { /* ref_pic_list_modification */
        h->modification_flag_l0 = (int)fo_br_bits(r, 1);
        h->mod_copies = 0;
        if (h->modification_flag_l0) {
            unsigned idc;
            do {
                idc = fo_br_ue(r);
                if (idc == 0 || idc == 1 || idc == 2) {
                    fo_br_ue(r);
                    h->mod_copies++;
                }
            } while (idc != 3);
        }
    }
    if (nal_ref_idc != 0) {
        if (nal_type == 5) {
            fo_br_bits(r, 2);
        } else if (fo_br_bits(r, 1)) {
            unsigned op;
            do {
                op = fo_br_ue(r);
                if (op == 1 || op == 3) fo_br_ue(r);
                if (op == 2) fo_br_ue(r);
                if (op == 3 || op == 6) fo_br_ue(r);
                if (op == 4) fo_br_ue(r);
            } while (op != 0);
        }
    }
    int slice_qp = h->pic_init_qp + fo_br_se(r);
    if (h->deblock_ctl == 1) {
        unsigned idc = fo_br_ue(r);
        if (idc != 1) {
            fo_br_se(r);
            fo_br_se(r);
        }
    }
    c->slice_type = s5;
    c->chroma_qp_offset = h->chroma_qp_offset;
    c->constrained_intra = h->constrained_intra;
    c->QPy = slice_qp;

    int MbCount = c->nmb, more = 1;
    int predL[16][16], predCb[8][8], predCr[8][8];
    c->cur = 0;
    while (more && c->cur < MbCount) {
        if (s5 != 2 && s5 != 4) {
            int run = (int)fo_br_ue(r);
            for (int i = 0; i < run; i++) {
                if (c->cur >= MbCount) break;
                c->cur_mb_type = FO_P_SKIP;
                c->mb_type[c->cur] = FO_P_SKIP;
                fo_DeriveMVs(c);
                fo_Decode(c, predL, predCr, predCb);
                c->QPy = (c->QPy + c->mb_qp_delta + 52) % 52; /* QUIRK: stale mb_qp_delta re-applied */
                fo_transformDecodingPSkip(c, predL, predCb, predCr, c->QPy);
                c->cur++;
            }
            if (c->cur != 0 || run > 0) more = fo_br_more(r);
        }
        if (more && c->cur < MbCount) {
            int t = (int)fo_br_ue(r);
            c->cur_mb_type = t;
            c->mb_type[c->cur] = t;
            if (t > 31 || (s5 == 2 && t > 24)) return -1;
            int pc = fo_pred_class(c, t);
            int np = (s5 == 2) ? 0 : p_num_part(t);
            if (pc == 2 && np == 4) {
                for (int i = 0; i < 4; i++) c->sub_mb_type[i] = (int)fo_br_ue(r);
                for (int i = 0; i < 4; i++)
                    if (c->num_ref_idx_override > 0 && t != FO_P_8x8ref0) c->ref_idx_l0[c->cur][i] = (int)fo_br_te(r);
                for (int i = 0; i < 4; i++)
                    for (int j = 0; j < num_sub(c->sub_mb_type[i]); j++) {
                        c->mvd[i][j][0] = fo_br_se(r);
                        c->mvd[i][j][1] = fo_br_se(r);
                    }
            } else if (pc == 0 || pc == 1) {
                if (pc == 0)
                    for (int b = 0; b < 16; b++) {
                        c->prev_flag[b] = (int)fo_br_bit(r);
                        if (!c->prev_flag[b]) c->rem_mode[b] = (int)fo_br_bits(r, 3);
                    }
                c->chroma_mode = (int)fo_br_ue(r);
                if (c->chroma_mode > 3) return -1;
            } else {
                if (np == 0xff) np = 0;
                for (int i = 0; i < np; i++)
                    if (c->num_ref_idx_l0_active_minus1 > 0) c->ref_idx_l0[c->cur][i] = (int)fo_br_te(r);
                for (int i = 0; i < np; i++) {
                    c->mvd[i][0][0] = fo_br_se(r);
                    c->mvd[i][0][1] = fo_br_se(r);
                }
            }
            if (pc != 1) {
                unsigned code = fo_br_ue(r);
                if (code > 47) return -1;
                int cbp = (pc == 0) ? fo_code_to_cbp_intra[code] : fo_code_to_cbp_inter[code];
                c->cbpL = cbp & 15;
                c->cbpC = cbp >> 4;
            } else {
                int k = (s5 == 2) ? t : t - 5; /* columns 5,6 of the mode tables */
                c->cbpC = ((k - 1) / 4) % 3;
                c->cbpL = (k >= 13) ? 15 : 0;
            }
            c->cbp_l[c->cur] = c->cbpL;
            c->cbp_c[c->cur] = c->cbpC;
            if (c->cbpL > 0 || c->cbpC > 0 || pc == 1) {
                c->mb_qp_delta = fo_br_se(r);
                if (c->mb_qp_delta < -26 || c->mb_qp_delta > 25) return -1;
                if (!fo_residual_parse(c, r)) return -1;
            } else {
                clear_residual(c);
            }
            c->QPy = (c->QPy + c->mb_qp_delta + 52) % 52;
            if (pc == 0 || pc == 1) {
                fo_intraPrediction_dec(c, predL, predCr, predCb);
            } else {
                fo_DeriveMVs(c);
                fo_Decode(c, predL, predCr, predCb);
            }
            if (pc == 1)
                fo_transformDecoding16x16Luma(c, c->lv.DC16, c->lv.AC16, predL, c->QPy);
            else if (pc != 0)
                for (int b = 0; b < 16; b++) fo_transformDecoding4x4Luma(c, c->lv.Lumalevel, predL, b, c->QPy);
            fo_transformDecodingChroma(c, c->lv.CDC[0], c->lv.CAC[0], predCb, c->QPy, 1);
            fo_transformDecodingChroma(c, c->lv.CDC[1], c->lv.CAC[1], predCr, c->QPy, 0);
            more = fo_br_more(r);
            c->cur++;
        }
    }
    /* initialisationProcess / modificationProcess -> frameDeepCopy (F/ref_frames.cpp:55-183) */
    if (!h->modification_flag_l0 || h->mod_copies > 0) {
        memcpy(c->dL, c->L, (size_t)c->W * c->H);
        memcpy(c->dC[0], c->C[0], (size_t)c->Wc * c->Hc);
        memcpy(c->dC[1], c->C[1], (size_t)c->Wc * c->Hc);
        c->have_dpb = 1;
    }
    return 0;
}

/* findNALstart / findNALend / parseNAL, F/nal.cpp:68-223 (4-byte start codes only) */
int fo_decode_stream(const uint8_t *s, size_t n, fo_frame_cb cb, void *user, fo_ctx **ctx_out)
{
    dec_hdr h;
    memset(&h, 0, sizeof h);
    fo_ctx *c = NULL;
    uint8_t *rbsp = (uint8_t *)malloc(n + 8);
    size_t pos = 0;
    int pictures = 0;
    for (;;) {
        size_t st = (size_t)-1;
        for (size_t i = pos; i + 3 < n; i++)
            if (s[i] == 0 && s[i + 1] == 0 && s[i + 2] == 0 && s[i + 3] == 1) {
                st = i + 4;
                break;
            }
        if (st == (size_t)-1) break;
        size_t en = n;
        for (size_t i = st; i + 2 < n; i++)
            if (s[i] == 0 && s[i + 1] == 0 && (s[i + 2] == 0 || s[i + 2] == 1)) {
                en = i;
                break;
            }
        pos = en;
        if (en <= st) continue;
        int ref_idc = (s[st] & 0x7f) >> 5, type = s[st] & 0x1f;
        size_t m = 0;
        for (size_t i = st + 1; i < en; i++) {
            if (i + 2 < en && s[i] == 0 && s[i + 1] == 0 && s[i + 2] == 3) {
                rbsp[m++] = s[i];
                rbsp[m++] = s[i + 1];
                i += 2;
            } else {
                rbsp[m++] = s[i];
            }
        }
        if (m == 0) break; /* NumBytesInRBSP == 0 ends decode(), F/fer_h264.cpp:41 */
        fo_br r;
        fo_br_init(&r, rbsp, m);
        if (type == FO_NAL_SPS) {
            parse_sps(&h, &r);
            if (!c) c = fo_create(h.W, h.H);
        } else if (type == FO_NAL_PPS) {
            parse_pps(&h, &r);
        } else if ((type == FO_NAL_IDR || type == FO_NAL_SLICE) && c) {
            if (decode_slice(c, &h, type, ref_idc, rbsp, m) < 0) break;
            pictures++;
            if (cb) cb(c, user);
        }
    }
    free(rbsp);
    if (ctx_out)
        *ctx_out = c;
    else
        fo_destroy(c);
    return pictures;
}

int fo_decode_slice(fo_ctx *c, int nal_type, int nal_ref_idc, const uint8_t *rbsp, size_t n)
{
    dec_hdr h;
    memset(&h, 0, sizeof h);
    h.log2_max_frame_num = c->log2_max_frame_num ? c->log2_max_frame_num : 9;
    h.log2_max_poc_lsb = c->log2_max_poc_lsb ? c->log2_max_poc_lsb : 10;
    h.pic_init_qp = c->pic_init_qp ? c->pic_init_qp : 14 + c->qp;
    h.chroma_qp_offset = c->chroma_qp_offset;
    return decode_slice(c, &h, nal_type, nal_ref_idc, rbsp, n);
}
