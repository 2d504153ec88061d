// fer_internal.h -- host-side declarations shared by the translation units of libferhip.
#pragma once
#include "fer_dev.h"

struct FerSortTmp {
    uint4 *rec1;        // [S][n] plane-0 records + position (k0|k1, k2|k3, k4, tx<<16|ty) after the first radix pass
    uint16_t *keyT;     // [S][W][H] sort keys in arrival order
    uint8_t *dig2;      // [S][n] high digit of the keys after the first pass
    uint16_t *skey;     // [S][n] sorted keys
    uint32_t *rec_tmp;  // [S][n][3] plain sorted records of streams that take the reference's mis-filed layout 
    void *tmp;          // digit histograms
    size_t tmp_bytes;
};

size_t fer_sort_tmp_bytes(int n, int S);
void fer_launch_refprep(const FerDev &d, FerSortTmp &t, const int *types, hipStream_t st);
void fer_launch_interp(const FerDev &d, hipStream_t st);
void fer_launch_features(const FerDev &d, hipStream_t st);
void fer_launch_sort(const FerDev &d, FerSortTmp &t, hipStream_t st);
void fer_launch_sort_keys(const FerDev &d, FerSortTmp &t, hipStream_t st);
void fer_launch_sort_radix(const FerDev &d, FerSortTmp &t, hipStream_t st);
void fer_launch_sort_finish(const FerDev &d, FerSortTmp &t, hipStream_t st);
void fer_launch_me_walk(const FerDev &d, hipStream_t st);
void fer_launch_frame_sad(const FerDev &d, hipStream_t st);
void fer_launch_me_pre(const FerDev &d, hipStream_t st);
void fer_launch_me_spec(const FerDev &d, hipStream_t st);
void fer_launch_me_resolve(const FerDev &d, hipStream_t st);
void fer_launch_basic_stat(const FerDev &d, hipStream_t st);
void fer_launch_p_resid(const FerDev &d, hipStream_t st);
void fer_launch_intra(const FerDev &d, hipStream_t st);
void fer_launch_cavlc(const FerDev &d, hipStream_t st);
void fer_launch_block_kat(int qP, const int32_t *in, int32_t *out, int keep_dc, int inverse, size_t n, hipStream_t st);
void fer_launch_decode_parse(const FerDev &d, const DecBatch &B, hipStream_t st);
void fer_launch_decode_recon(const FerDev &dslice, bool anyP, bool anyIntra, hipStream_t st);
