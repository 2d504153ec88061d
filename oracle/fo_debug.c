/*
 * fo_debug.c -- ORACLE (test infrastructure): accessors used by tests/ to
 * compare intermediate state (planes, features, sort, MVs, levels) of the HIP
 * path with the CPU restatement through ctypes.
 */
#include "fo.h"
#include <string.h>

uint8_t *fo_dbg_plane(fo_ctx *c, int which) /* 0..2 frame, 3..5 dpb */
{
    switch (which) {
    case 0: return c->L;
    case 1: return c->C[0];
    case 2: return c->C[1];
    case 3: return c->dL;
    case 4: return c->dC[0];
    default: return c->dC[1];
    }
}
uint8_t *fo_dbg_interp(fo_ctx *c, int f) { return c->interp[f]; }
int *fo_dbg_kar(fo_ctx *c, int k, int f) { return c->kar[k][f]; }
int *fo_dbg_sorted(fo_ctx *c, int k) { return c->sorted[k]; }
int *fo_dbg_koliko(fo_ctx *c) { return c->koliko; }
int *fo_dbg_mb_type(fo_ctx *c) { return c->mb_type; }
int *fo_dbg_cbp(fo_ctx *c, int chroma) { return chroma ? c->cbp_c : c->cbp_l; }
int *fo_dbg_mv(fo_ctx *c, int y) { return y ? &c->dbg_mvy[0][0][0] : &c->dbg_mvx[0][0][0]; }
int *fo_dbg_tc_l(fo_ctx *c) { return &c->tc_l[0][0]; }
int *fo_dbg_tc_c(fo_ctx *c) { return &c->tc_c[0][0][0]; }
int *fo_dbg_i4mode(fo_ctx *c) { return c->i4mode; }
int *fo_dbg_type_count(fo_ctx *c) { return c->type_count; } /* brojTipova */
void fo_dbg_set_dpb(fo_ctx *c, const uint8_t *y, const uint8_t *u, const uint8_t *v)
{
    memcpy(c->dL, y, (size_t)c->W * c->H);
    memcpy(c->dC[0], u, (size_t)c->Wc * c->Hc);
    memcpy(c->dC[1], v, (size_t)c->Wc * c->Hc);
    c->have_dpb = 1;
}
void fo_dbg_set_frame(fo_ctx *c, const uint8_t *y, const uint8_t *u, const uint8_t *v)
{
    memcpy(c->L, y, (size_t)c->W * c->H);
    memcpy(c->C[0], u, (size_t)c->Wc * c->Hc);
    memcpy(c->C[1], v, (size_t)c->Wc * c->Hc);
}

/* coded_mb_size of the Intra16x16 / Intra4x4 alternative of every macroblock of the last I picture */
int *fo_dbg_mbsize(fo_ctx *c) { return &c->dbg_mbsize[0][0]; }
/* context-free pieces for the per-macroblock KATs: one macroblock of a W x H context */
void fo_dbg_set_mb(fo_ctx *c, int cur, int mb_type, int slice_type, int qp)
{
    c->cur = cur;
    c->cur_mb_type = mb_type;
    c->mb_type[cur] = mb_type;
    c->slice_type = slice_type;
    c->QPy = qp;
}
int *fo_dbg_levels(fo_ctx *c) { return &c->lv.Lumalevel[0][0]; } /* fo_levels: Lumalevel[16][16], DC16[16], AC16[16][16], CDC[2][4], CAC[2][4][16] */
void fo_dbg_set_mv(fo_ctx *c, int mb, int sub, int part, int mvx, int mvy)
{
    c->mvx[mb][sub][part] = mvx;
    c->mvy[mb][sub][part] = mvy;
}
uint8_t *fo_dbg_dpb(fo_ctx *c, int k) { return k == 0 ? c->dL : c->dC[k - 1]; }
