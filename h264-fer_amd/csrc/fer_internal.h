// fer_internal.h -- host-side declarations shared by the translation units of libferhip.
#pragma once
#include "fer_dev.h"

struct FerSortTmp {
    uint32_t *keys_in, *keys_out;
    uint32_t *vals_in, *vals_out;
    void *tmp;
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
void fer_launch_me_resolve(const FerDev &d, hipStream_t st);
void fer_launch_basic_stat(const FerDev &d, hipStream_t st);
void fer_launch_p_resid(const FerDev &d, hipStream_t st);
void fer_launch_intra(const FerDev &d, hipStream_t st);
void fer_launch_cavlc(const FerDev &d, hipStream_t st);
void fer_launch_block_kat(int qP, const int32_t *in, int32_t *out, int keep_dc, int inverse, size_t n, hipStream_t st);
void fer_launch_decode_parse(const FerDev &d, const DecBatch &B, hipStream_t st);
void fer_launch_decode_recon(const FerDev &dslice, bool anyP, bool anyIntra, hipStream_t st);
