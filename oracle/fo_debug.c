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
