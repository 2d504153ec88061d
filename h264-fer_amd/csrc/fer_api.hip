// fer_api.hip -- the C ABI of libferhip (include/ferhip.h): context, HBM layout, picture
// driver (the RBSP_encode sequence of F/rbsp_encoding.cpp:139-323 expressed as kernel
// launches), host-side slice / parameter-set headers (F/headers_and_parameter_sets.cpp) and
// NAL framing (F/nal.cpp:261-299).  Host code is C-style C++; nothing here runs the hot path
// on the CPU.
#include "../../include/ferhip.h"
#include "fer_internal.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <utility>
#include <chrono>
#include <mutex>
#include <thread>
#include <vector>

#define CK(x)                                                                                         \
    do {                                                                                              \
        hipError_t e_ = (x);                                                                          \
        if (e_ != hipSuccess) {                                                                       \
            fprintf(stderr, "ferhip: %s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            return FERHIP_E_HIP;                                                                      \
        }                                                                                             \
    } while (0)

#define FER_HDR_SLOTS 8
struct StreamState {  // slice-level state of one stream (globals `shd`, statics of RBSP_encode)
    int frame_num, poc_lsb, idr_pic_id, first_idr_done, frames_done, have_dpb;
};

struct ferhip_ctx {
    FerDev d;
    ferhip_params p;
    hipStream_t st;
    hipStream_t st_hi;   // high-priority stream for the latency-bound per-diagonal chains
    hipStream_t st_aux;  // the sort of the reference picture's positions runs here, beside k_me_pre (run_picture)
    hipEvent_t ev_a, ev_b, ev_c, ev_d;
    int overlap_sort;    // ferhip_tune: 1 / 2 = the HBM-bound sort is enqueued beside the VALU-bound stage-3 search (st_aux / st_hi)
    int device;  // HIP device the context lives on; every entry point re-selects it (callers may use any thread)
    std::vector<StreamState> ss;
    std::vector<void *> allocs;
    FerSortTmp sort;
    uint32_t *h_hdr;       // pinned [S][4]: the slot of the header ring that the next picture fills
    uint32_t *h_hdr_ring;  // pinned [FER_HDR_SLOTS][S][4]; a pinned source is read when the copy executes, not when it is
    hipEvent_t hdr_ev[8];  // enqueued, so a slot is rewritten only after the copy that last used it has run
    int hdr_slot;
    uint32_t *h_len;       // pinned [S]
    int *h_status;         // pinned [S]
    unsigned long long *h_sad;
    std::vector<int> types;
    uint8_t *planes[2];    // two picture sets, swapped after every picture
    // asynchronous ingest (ferhip_upload_frames): pinned host pictures -> stage[k] on a copy stream, double buffered
    hipStream_t st_copy;
    uint8_t *stage[2];
    hipEvent_t up_done[2], up_used[2];
    int up_next, up_ready;  // slot the next upload fills; uploads waiting to be made current
    int cur_set;
    bool refprep_valid;
    // live kernel timing with HIP events on the launch stream (bench.py roofline leg)
    bool prof;
    struct Span { int phase; hipEvent_t a, b; long launches; };
    std::vector<Span> spans;
    double prof_ms[FERHIP_NPHASE];
    long prof_launches[FERHIP_NPHASE];
};

struct ProfScope {
    ferhip_ctx *c;
    int idx;
    hipStream_t s;
    ProfScope(ferhip_ctx *c_, int phase, long launches, hipStream_t s_ = nullptr) : c(c_), idx(-1), s(s_ ? s_ : c_->st)
    {
        if (!c->prof) return;
        ferhip_ctx::Span sp{phase, nullptr, nullptr, launches};
        hipEventCreate(&sp.a);
        hipEventCreate(&sp.b);
        hipEventRecord(sp.a, s);
        c->spans.push_back(sp);
        idx = (int)c->spans.size() - 1;
    }
    ~ProfScope()
    {
        if (idx >= 0) hipEventRecord(c->spans[idx].b, s);
    }
};

static const int k_qpc[52] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17,
                              18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31, 32, 32, 33,
                              34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39};

// qPiToQPc (F/inttransform.cpp:8-14) for chroma_qp_index_offset 0: what the per-macroblock shims of fer_legacy.hip need
extern "C" int ferhip_chroma_qp(int qpy) { return k_qpc[qpy < 0 ? 0 : (qpy > 51 ? 51 : qpy)]; }

template <typename T>
static int dalloc(ferhip_ctx *c, T **p, size_t n)
{
    void *v = nullptr;
    if (hipMalloc(&v, n * sizeof(T) + 256) != hipSuccess) return FERHIP_E_HIP;
    if (hipMemset(v, 0, n * sizeof(T) + 256) != hipSuccess) return FERHIP_E_HIP;
    c->allocs.push_back(v);
    *p = (T *)v;
    return 0;
}

static void bind_planes(ferhip_ctx *c)
{
    FerDev &d = c->d;
    size_t fsz = d.ysz + 2 * d.csz;
    uint8_t *cur = c->planes[c->cur_set], *ref = c->planes[c->cur_set ^ 1];
    d.curY = cur;
    d.curCb = cur + (size_t)d.S * d.ysz;
    d.curCr = cur + (size_t)d.S * (d.ysz + d.csz);
    d.refY = ref;
    d.refCb = ref + (size_t)d.S * d.ysz;
    d.refCr = ref + (size_t)d.S * (d.ysz + d.csz);
    (void)fsz;
}

extern "C" const char *ferhip_version(void) { return "ferhip 0.2 (gfx950)"; }

// ---- device / pinned memory for hosts without a HIP binding of their own (cgo, JNI, ctypes ...): the pictures of
// ferhip_set_frames(host = 0), the buffers of ferhip_copy_rbsp and the pinned sources of ferhip_upload_frames
extern "C" void *ferhip_mem_alloc(size_t bytes, int kind)  // kind 0 = device (HBM), 1 = pinned host
{
    void *p = nullptr;
    hipError_t e = kind ? hipHostMalloc(&p, bytes) : hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}
extern "C" void ferhip_mem_free(void *p, int kind)
{
    if (!p) return;
    if (kind)
        hipHostFree(p);
    else
        hipFree(p);
}
// synchronous copy between host and device memory (direction inferred by the runtime)
extern "C" int ferhip_mem_copy(void *dst, const void *src, size_t bytes)
{
    if (!dst || !src) return FERHIP_E_ARG;
    CK(hipMemcpy(dst, src, bytes, hipMemcpyDefault));
    return 0;
}

static int ctx_create(ferhip_ctx **out, int W, int H, int S, const ferhip_params *p, bool decode_only);
extern "C" int ferhip_create(ferhip_ctx **out, int W, int H, int S, const ferhip_params *p)
{
    return ctx_create(out, W, H, S, p, false);
}
// decode_only: the encoder-side structures (interpolated planes, features, sort, search lists, RBSP) stay empty
static int ctx_create(ferhip_ctx **out, int W, int H, int S, const ferhip_params *p, bool decode_only)
{
    if (!out || !p || W <= 0 || H <= 0 || (W & 15) || (H & 15) || S <= 0 || W > 16384 || H > 16384) return FERHIP_E_ARG;
    if (p->qp < 0 || p->qp > 51 || p->window < 16 || p->intra_every <= 0) return FERHIP_E_ARG;
    // the motion kernels address a stream's planes and records with 24-bit multiplies and 32-bit byte offsets: a padded
    // plane stays below 2^24 samples (4K is 8.4 M; the reference itself stops at 10 000 macroblocks)
    if (!decode_only && (size_t)(W + FER_IP_L + FER_IP_R) * (size_t)(H + FER_IP_T + FER_IP_B) >= ((size_t)1 << 24)) return FERHIP_E_UNSUP;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        fprintf(stderr, "ferhip: no HIP device; the hot path has no CPU fallback\n");
        return FERHIP_E_HIP;
    }
    ferhip_ctx *c = new ferhip_ctx();
    memset(&c->d, 0, sizeof c->d);
    c->st = c->st_hi = c->st_aux = nullptr;
    c->ev_a = c->ev_b = c->ev_c = c->ev_d = nullptr;
    c->overlap_sort = 0;  // measured: kernels of two HIP streams do not share the GPU here (the sort stretches to k_me_pre's length)
    c->h_hdr = c->h_hdr_ring = c->h_len = nullptr;
    c->h_status = nullptr;
    c->h_sad = nullptr;
    for (int i = 0; i < FER_HDR_SLOTS; i++) c->hdr_ev[i] = nullptr;
    c->planes[0] = c->planes[1] = nullptr;
    c->st_copy = nullptr;
    c->stage[0] = c->stage[1] = nullptr;
    c->up_done[0] = c->up_done[1] = c->up_used[0] = c->up_used[1] = nullptr;
    c->up_next = c->up_ready = 0;
    c->p = *p;
    FerDev &d = c->d;
    d.W = W;
    d.H = H;
    d.Wc = W / 2;
    d.Hc = H / 2;
    d.mbw = W / 16;
    d.mbh = H / 16;
    d.nmb = d.mbw * d.mbh;
    d.S = S;
    d.qp = p->qp;
    d.qpc = k_qpc[p->qp];  // chroma_qp_index_offset == 0 (F/headers_and_parameter_sets.cpp:490)
    for (int k = 0; k < 2; k++) {  // the three LevelScale / LevelQuantize values of each QP (F/scaleTransform.cpp:32-40)
        static const int v[6][3] = {{10, 16, 13}, {11, 18, 14}, {13, 20, 16}, {14, 23, 18}, {16, 25, 20}, {18, 29, 23}};
        const int m = (k ? d.qpc : d.qp) % 6;
        for (int c = 0; c < 3; c++) {
            const int ls = 16 * v[m][c];
            d.lsq[k][c] = (int16_t)ls;
            d.lsq[k][3 + c] = (int16_t)((65536 + ls) / (2 * ls));
        }
    }
    d.window = p->window;
    d.maxdiff_set = p->maxdiff;
    d.basic = p->basic ? 1 : 0;
#ifdef FER_PROBE
    d.dbg = getenv("FER_DBG") ? atoi(getenv("FER_DBG")) : 0;
#else
    d.dbg = 0;  // no environment variable changes what the shipped library computes
#endif
    d.ysz = (size_t)W * H;
    d.csz = d.ysz / 4;
    d.resolve_wgs = 6144;
    d.resolve_group = S;  // all streams of a ticket queue in one group (measured: 4.9 ms against 6.1 with a short last group)
    {
        int lo = 0, hi = 0;  // numerically lower = higher priority
        if (hipGetDevice(&c->device) != hipSuccess || hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess ||
            hipStreamCreateWithPriority(&c->st, hipStreamNonBlocking, lo) != hipSuccess ||
            hipStreamCreateWithPriority(&c->st_hi, hipStreamNonBlocking, hi) != hipSuccess ||
            hipStreamCreateWithPriority(&c->st_aux, hipStreamNonBlocking, lo) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_c, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_d, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_a, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_b, hipEventDisableTiming) != hipSuccess) {
            ferhip_destroy(c);
            return FERHIP_E_HIP;
        }
    }
    size_t fsz = d.ysz * 3 / 2;
    int rc = 0;
    rc |= dalloc(c, &c->planes[0], fsz * S);
    rc |= dalloc(c, &c->planes[1], fsz * S);
    d.ipitch = W + FER_IP_L + FER_IP_R;
    d.iplane = (((size_t)d.ipitch * (H + FER_IP_T + FER_IP_B)) + 255) & ~(size_t)255;
    d.ioff = FER_IP_T * d.ipitch + FER_IP_L;
    rc |= dalloc(c, &d.interp, decode_only ? (size_t)1 : (size_t)(d.iplane * 16 * S));
    d.feat = nullptr;  // the 16-plane feature table exists only as a test read-back (FERHIP_BUF_FEAT)
    rc |= dalloc(c, &d.feat0, decode_only ? (size_t)1 : (size_t)(d.ysz * 6 * S));
    rc |= dalloc(c, &d.sort_pos, decode_only ? (size_t)1 : (size_t)(d.ysz * S));
    rc |= dalloc(c, &d.sort_rec, decode_only ? (size_t)1 : (size_t)(d.ysz * S * 3));
    d.ktw_shift = 3;  // column tiles of the bucket index: at most 64 per row, at least 8 columns wide
    while (((W + (1 << d.ktw_shift) - 1) >> d.ktw_shift) > 64) d.ktw_shift++;
    d.kt = (W + (1 << d.ktw_shift) - 1) >> d.ktw_shift;
    const size_t nbins = (size_t)S * 16384 * d.kt + 1;
    rc |= dalloc(c, &d.kol2, decode_only ? (size_t)1 : (size_t)(nbins));
    rc |= dalloc(c, &d.brange, decode_only ? (size_t)1 : (size_t)S * 16384 * 8);
    rc |= dalloc(c, &d.bmodal, decode_only ? (size_t)1 : (size_t)S * 16384 * 4);
    d.nlists = W * H / FER_BRANGE_MIN + 1;
    rc |= dalloc(c, &d.boutl, decode_only ? (size_t)1 : (size_t)S * d.nlists * FER_OUTL);
    rc |= dalloc(c, &d.nbig, (size_t)S);
    rc |= dalloc(c, &d.zero_cnt, (size_t)S);
    size_t nm = (size_t)d.nmb * S;
    rc |= dalloc(c, &d.mb_type, nm);
    rc |= dalloc(c, &d.mv, nm * 8);
    rc |= dalloc(c, &d.mvd, nm * 8);
    rc |= dalloc(c, &d.cbp, nm * 2);
    rc |= dalloc(c, &d.tc, nm * 24);
    rc |= dalloc(c, &d.i4mode, nm * 16);
    rc |= dalloc(c, &d.i4flag, nm * 16);
    rc |= dalloc(c, &d.chroma_mode, nm);
    rc |= dalloc(c, &d.levels, nm * FER_LEVELS);
    rc |= dalloc(c, &d.mbsize, nm * 2);
    rc |= dalloc(c, &d.suma, decode_only ? (size_t)1 : (size_t)(nm * 20));
    rc |= dalloc(c, &d.st3, decode_only ? (size_t)1 : (size_t)(nm * 4 * 33 * 3));
    rc |= dalloc(c, &d.st3n, decode_only ? (size_t)1 : (size_t)(nm * 4));
    rc |= dalloc(c, &d.st2, decode_only ? (size_t)1 : (size_t)(nm * 4 * FER_ST2_CAP * 2));
    rc |= dalloc(c, &d.st2n, decode_only ? (size_t)1 : (size_t)(nm * 4));
    rc |= dalloc(c, &d.v0, decode_only ? (size_t)1 : (size_t)(nm * 4));
    rc |= dalloc(c, &d.spec_hdr, decode_only ? (size_t)1 : (size_t)(nm * 4));
    rc |= dalloc(c, &d.spec_l1, decode_only ? (size_t)1 : (size_t)(nm * 4 * 17));
    rc |= dalloc(c, &d.spec_l2, decode_only ? (size_t)1 : (size_t)(nm * 4 * 33));
    rc |= dalloc(c, &d.spec_stat, (size_t)8);
    d.speculate = 1;
    rc |= dalloc(c, &d.chain, (size_t)64);
    rc |= dalloc(c, &d.timing, (size_t)64);
    rc |= dalloc(c, &d.chain64, decode_only ? (size_t)1 : (size_t)(nm * 4));
    rc |= dalloc(c, &d.mb_bits, ((size_t)d.nmb + 1) * S);
    d.bits_cap_words = ((size_t)d.nmb * 1024 + 4096) / 4;
    rc |= dalloc(c, &d.bits, decode_only ? (size_t)1 : (size_t)(d.bits_cap_words * S));
    rc |= dalloc(c, &d.hdr, (size_t)4 * S);
    rc |= dalloc(c, &d.out_bytes, (size_t)S);
    rc |= dalloc(c, &d.status, (size_t)S);
    rc |= dalloc(c, &d.sad, (size_t)S);
    rc |= dalloc(c, &d.stats, (size_t)5 * S);
    rc |= dalloc(c, &d.dec_qp, nm);
    rc |= dalloc(c, &d.dec_state, (size_t)4 * S);
    rc |= dalloc(c, &d.dec_cac, (size_t)128 * S);
    int n = W * H;
    c->sort.tmp_bytes = fer_sort_tmp_bytes(n, S);
    rc |= dalloc(c, &c->sort.rec_tmp, decode_only ? (size_t)1 : (size_t)((size_t)n * S * 3));
    rc |= dalloc(c, &c->sort.rec1, decode_only ? (size_t)1 : (size_t)((size_t)n * S));
    rc |= dalloc(c, &c->sort.keyT, decode_only ? (size_t)1 : (size_t)((size_t)n * S));
    rc |= dalloc(c, &c->sort.dig2, decode_only ? (size_t)1 : (size_t)((size_t)n * S));
    rc |= dalloc(c, &c->sort.skey, decode_only ? (size_t)1 : (size_t)((size_t)n * S));
    uint8_t *tmp = nullptr;
    rc |= dalloc(c, &tmp, decode_only ? (size_t)1 : (size_t)(c->sort.tmp_bytes));
    c->sort.tmp = tmp;
    if (rc) {
        ferhip_destroy(c);
        return FERHIP_E_HIP;
    }
    if (hipHostMalloc((void **)&c->h_hdr_ring, sizeof(uint32_t) * 4 * S * FER_HDR_SLOTS) != hipSuccess ||
        hipHostMalloc((void **)&c->h_len, sizeof(uint32_t) * S) != hipSuccess ||
        hipHostMalloc((void **)&c->h_status, sizeof(int) * S) != hipSuccess ||
        hipHostMalloc((void **)&c->h_sad, sizeof(unsigned long long) * S) != hipSuccess) {
        ferhip_destroy(c);
        return FERHIP_E_HIP;
    }
    for (int i = 0; i < FER_HDR_SLOTS; i++)
        if (hipEventCreateWithFlags(&c->hdr_ev[i], hipEventDisableTiming) != hipSuccess) {
            ferhip_destroy(c);
            return FERHIP_E_HIP;
        }
    c->hdr_slot = 0;
    c->h_hdr = c->h_hdr_ring;
    c->ss.assign(S, StreamState{0, 0, 0, 0, 0, 0});
    c->types.assign(S, 2);
    c->cur_set = 0;
    c->refprep_valid = false;
    c->prof = false;
    memset(c->prof_ms, 0, sizeof c->prof_ms);
    memset(c->prof_launches, 0, sizeof c->prof_launches);
    bind_planes(c);
    // the buffers were cleared on the null stream, which the context's non-blocking streams do not wait for: without
    // this the clearing of a large buffer can land after the first picture's kernels have written into it
    if (hipDeviceSynchronize() != hipSuccess) {
        ferhip_destroy(c);
        return FERHIP_E_HIP;
    }
    *out = c;
    return 0;
}

extern "C" void ferhip_destroy(ferhip_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->st) hipStreamSynchronize(c->st);
    if (c->st_hi) hipStreamSynchronize(c->st_hi);
    if (c->st_aux) hipStreamSynchronize(c->st_aux);
    for (void *p : c->allocs) hipFree(p);
    if (c->h_hdr_ring) hipHostFree(c->h_hdr_ring);
    for (int i = 0; i < FER_HDR_SLOTS; i++)
        if (c->hdr_ev[i]) hipEventDestroy(c->hdr_ev[i]);
    if (c->h_len) hipHostFree(c->h_len);
    if (c->h_status) hipHostFree(c->h_status);
    if (c->h_sad) hipHostFree(c->h_sad);
    if (c->st_copy) {
        hipStreamSynchronize(c->st_copy);
        hipStreamDestroy(c->st_copy);
    }
    for (int i = 0; i < 2; i++) {
        if (c->up_done[i]) hipEventDestroy(c->up_done[i]);
        if (c->up_used[i]) hipEventDestroy(c->up_used[i]);
    }
    if (c->st) hipStreamDestroy(c->st);
    if (c->st_hi) hipStreamDestroy(c->st_hi);
    if (c->st_aux) hipStreamDestroy(c->st_aux);
    if (c->ev_c) hipEventDestroy(c->ev_c);
    if (c->ev_d) hipEventDestroy(c->ev_d);
    if (c->ev_a) hipEventDestroy(c->ev_a);
    if (c->ev_b) hipEventDestroy(c->ev_b);
    delete c;
}

// [S][Y|U|V] interleaved per stream  <->  plane-major [Y of all streams][U ...][V ...]
// I420 pictures, stream-major [S][Y|Cb|Cr] <-> the context's plane-major picture set [plane][S][...], both in
// device memory: one launch instead of three copies per stream.  16 bytes per thread (plane sizes are multiples of 64).
__global__ __launch_bounds__(256) void k_repack(uint8_t *set, uint8_t *frames, size_t ysz, size_t csz, int S, int to_set)
{
    const size_t fsz = ysz + 2 * csz;
    const int s = blockIdx.y;
    for (size_t o = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; o < fsz; o += (size_t)gridDim.x * blockDim.x * 16) {
        size_t po;
        if (o < ysz)
            po = (size_t)s * ysz + o;
        else if (o < ysz + csz)
            po = (size_t)S * ysz + (size_t)s * csz + (o - ysz);
        else
            po = (size_t)S * (ysz + csz) + (size_t)s * csz + (o - ysz - csz);
        uint4 *a = (uint4 *)(set + po), *b = (uint4 *)(frames + (size_t)s * fsz + o);
        if (to_set)
            *a = *b;
        else
            *b = *a;
    }
}

static int copy_frames(ferhip_ctx *c, uint8_t *set, const uint8_t *src, uint8_t *dst, hipMemcpyKind kind)
{
    FerDev &d = c->d;
    size_t fsz = d.ysz * 3 / 2;
    if (kind == hipMemcpyDeviceToDevice && (((uintptr_t)(src ? src : dst)) & 15) == 0) {
        hipLaunchKernelGGL(k_repack, dim3(256, d.S), dim3(256), 0, c->st, set, (uint8_t *)(src ? src : dst), d.ysz, d.csz, d.S,
                           src ? 1 : 0);
        return 0;
    }
    for (int s = 0; s < d.S; s++) {
        uint8_t *py = set + (size_t)s * d.ysz;
        uint8_t *pu = set + (size_t)d.S * d.ysz + (size_t)s * d.csz;
        uint8_t *pv = set + (size_t)d.S * (d.ysz + d.csz) + (size_t)s * d.csz;
        if (src) {
            const uint8_t *f = src + (size_t)s * fsz;
            CK(hipMemcpyAsync(py, f, d.ysz, kind, c->st));
            CK(hipMemcpyAsync(pu, f + d.ysz, d.csz, kind, c->st));
            CK(hipMemcpyAsync(pv, f + d.ysz + d.csz, d.csz, kind, c->st));
        } else {
            uint8_t *f = dst + (size_t)s * fsz;
            CK(hipMemcpyAsync(f, py, d.ysz, kind, c->st));
            CK(hipMemcpyAsync(f + d.ysz, pu, d.csz, kind, c->st));
            CK(hipMemcpyAsync(f + d.ysz + d.csz, pv, d.csz, kind, c->st));
        }
    }
    return 0;
}

extern "C" int ferhip_set_frames(ferhip_ctx *c, const void *src, int host)
{
    if (!c || !src) return FERHIP_E_ARG;
    (void)hipSetDevice(c->device);
    int rc = copy_frames(c, c->planes[c->cur_set], (const uint8_t *)src, nullptr,
                         host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice);
    if (rc) return rc;
    if (host) CK(hipStreamSynchronize(c->st));
    return 0;
}

// ---- asynchronous ingest: ReadFromY4M's successor for many streams (row f3) ----
// ferhip_upload_frames starts the H2D copy of the NEXT pictures ([S][W*H*3/2], pinned host memory) on a copy stream
// of its own and returns; up to two uploads may be in flight.  ferhip_set_frames_uploaded makes the oldest one the
// current picture (the repack runs on the encode stream behind the copy).  So the upload of picture t + 1 overlaps
// the encode of picture t.
extern "C" int ferhip_upload_frames(ferhip_ctx *c, const void *pinned_src)
{
    if (!c || !pinned_src) return FERHIP_E_ARG;
    (void)hipSetDevice(c->device);
    FerDev &d = c->d;
    const size_t bytes = d.ysz * 3 / 2 * d.S;
    if (!c->st_copy) {
        // first call: the copy stream, two staging sets, their events.  A failure on the way leaves the context as it
        // was before the call (st_copy null), so that a later call tries again instead of meeting half a setup.
        hipStream_t stc = nullptr;
        CK(hipStreamCreateWithFlags(&stc, hipStreamNonBlocking));
        bool ok = true;
        for (int i = 0; i < 2 && ok; i++) {
            if (!c->stage[i]) ok = dalloc(c, &c->stage[i], bytes) == 0;
            if (ok && !c->up_done[i]) ok = hipEventCreateWithFlags(&c->up_done[i], hipEventDisableTiming) == hipSuccess;
            if (ok && !c->up_used[i]) ok = hipEventCreateWithFlags(&c->up_used[i], hipEventDisableTiming) == hipSuccess;
        }
        if (ok) ok = hipDeviceSynchronize() == hipSuccess;  // dalloc clears on the null stream
        if (!ok) {
            (void)hipGetLastError();
            hipStreamDestroy(stc);
            fprintf(stderr, "ferhip_upload_frames: could not set up the staging buffers\n");
            return FERHIP_E_HIP;
        }
        c->st_copy = stc;
    }
    if (c->up_ready >= 2) return FERHIP_E_STATE;
    const int k = c->up_next;
    CK(hipStreamWaitEvent(c->st_copy, c->up_used[k], 0));  // the repack that last read this slot
    CK(hipMemcpyAsync(c->stage[k], pinned_src, bytes, hipMemcpyHostToDevice, c->st_copy));
    CK(hipEventRecord(c->up_done[k], c->st_copy));
    c->up_next ^= 1;
    c->up_ready++;
    return 0;
}

extern "C" int ferhip_set_frames_uploaded(ferhip_ctx *c)
{
    if (!c) return FERHIP_E_ARG;
    if (c->up_ready <= 0) return FERHIP_E_STATE;
    (void)hipSetDevice(c->device);
    const int k = (c->up_next + 2 - c->up_ready) & 1;  // oldest upload in flight
    CK(hipStreamWaitEvent(c->st, c->up_done[k], 0));
    int rc = copy_frames(c, c->planes[c->cur_set], c->stage[k], nullptr, hipMemcpyDeviceToDevice);
    if (rc) return rc;
    CK(hipEventRecord(c->up_used[k], c->st));
    c->up_ready--;
    return 0;
}

extern "C" int ferhip_set_reference(ferhip_ctx *c, const void *src)
{
    if (!c || !src) return FERHIP_E_ARG;
    (void)hipSetDevice(c->device);
    int rc = copy_frames(c, c->planes[c->cur_set ^ 1], (const uint8_t *)src, nullptr, hipMemcpyHostToDevice);
    if (rc) return rc;
    CK(hipStreamSynchronize(c->st));
    for (auto &s : c->ss) s.have_dpb = 1;
    c->refprep_valid = false;
    return 0;
}

extern "C" int ferhip_get_recon(ferhip_ctx *c, void *dst, int host)
{
    if (!c || !dst) return FERHIP_E_ARG;
    (void)hipSetDevice(c->device);
    // after encode_picture the reconstruction is the reference set
    int rc = copy_frames(c, c->planes[c->cur_set ^ 1], nullptr, (uint8_t *)dst,
                         host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice);
    if (rc) return rc;
    CK(hipStreamSynchronize(c->st));
    return 0;
}

// ---- host bit writer for headers (MSB first, F/rbsp_IO.cpp:123)
struct HBits {
    unsigned long long v;
    int n;
    void put(int k, unsigned x)
    {
        v = (v << k) | (unsigned long long)x;
        n += k;
    }
    void ue(unsigned x)
    {
        int p = 0;
        while (((x + 1) >> (p + 1)) != 0) p++;
        put(p, 0);
        put(1, 1);
        if (p) put(p, x + 1 - (1u << p));
    }
    void se(int x) { ue(x <= 0 ? (unsigned)(-x) * 2u : (unsigned)x * 2u - 1u); }
};

struct ByteW {
    uint8_t *b;
    size_t cap, nbits;
    void put(int k, unsigned x)
    {
        for (int i = k - 1; i >= 0; i--) {
            size_t by = nbits >> 3;
            if (by < cap) {
                if ((nbits & 7) == 0) b[by] = 0;
                b[by] |= (uint8_t)(((x >> i) & 1u) << (7 - (nbits & 7)));
            }
            nbits++;
        }
    }
    void ue(unsigned x)
    {
        int p = 0;
        while (((x + 1) >> (p + 1)) != 0) p++;
        put(p, 0);
        put(1, 1);
        if (p) put(p, x + 1 - (1u << p));
    }
    void se(int x) { ue(x <= 0 ? (unsigned)(-x) * 2u : (unsigned)x * 2u - 1u); }
    size_t trailing()
    {
        put(1, 1);
        while (nbits & 7) put(1, 0);
        return nbits >> 3;
    }
};

// sps_write, F/headers_and_parameter_sets.cpp:305-391
extern "C" size_t ferhip_write_sps(ferhip_ctx *c, uint8_t *rbsp, size_t cap)
{
    ByteW w{rbsp, cap, 0};
    w.put(8, 66);
    w.put(1, 1);
    w.put(1, 1);
    w.put(1, 0);
    w.put(5, 0);
    w.put(8, 41);
    w.ue(0);
    w.ue(5);  // log2_max_frame_num 9
    w.ue(0);
    w.ue(6);  // log2_max_pic_order_cnt_lsb 10
    w.ue(1);
    w.put(1, 0);
    w.ue((unsigned)(c->d.mbw - 1));
    w.ue((unsigned)(c->d.mbh - 1));
    w.put(1, 1);
    w.put(1, 1);
    w.put(1, 0);
    w.put(1, 0);
    return w.trailing();
}

// pps_write, F/headers_and_parameter_sets.cpp:478-513 (weighted_bipred_idc field carries the value 1)
extern "C" size_t ferhip_write_pps(ferhip_ctx *c, uint8_t *rbsp, size_t cap)
{
    ByteW w{rbsp, cap, 0};
    w.ue(0);
    w.ue(0);
    w.put(1, 0);
    w.put(1, 0);
    w.ue(0);
    w.ue(0);
    w.ue(0);
    w.put(1, 0);
    w.put(2, 1);
    w.se(14 + c->p.qp - 26);
    w.se(0);
    w.se(0);
    w.put(1, 0);
    w.put(1, 0);
    w.put(1, 0);
    return w.trailing();
}

// writeNAL, F/nal.cpp:261-299
extern "C" size_t ferhip_write_nal(int nal_ref_idc, int nal_type, const uint8_t *rbsp, size_t n, uint8_t *out)
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
        zc = rbsp[i] == 0 ? zc + 1 : 0;
    }
    return pos;
}

// slice header of stream s for this picture: shd_write, F/headers_and_parameter_sets.cpp:172-239
static void build_header(ferhip_ctx *c, int s, int nal_type)
{
    StreamState &t = c->ss[s];
    int slice_type;
    if (nal_type == FERHIP_NAL_IDR) {  // F/rbsp_encoding.cpp:142-164
        slice_type = 2;
        if (!t.first_idr_done) {
            t.first_idr_done = 1;
            t.idr_pic_id = 0;
        } else if (t.frame_num == 0) {
            t.idr_pic_id++;
        } else {
            t.idr_pic_id = 0;
        }
        t.frame_num = 0;
        t.poc_lsb = 0;
    } else {
        slice_type = 0;
        t.frame_num++;
        t.poc_lsb += 2;
    }
    HBits h{0, 0};
    h.ue(0);
    h.ue((unsigned)slice_type);
    h.ue(0);
    h.put(9, (unsigned)t.frame_num & 511u);
    if (nal_type == FERHIP_NAL_IDR) h.ue((unsigned)t.idr_pic_id);
    h.put(10, (unsigned)t.poc_lsb & 1023u);
    if (slice_type == 0) {
        h.put(1, 0);  // num_ref_idx_active_override_flag
        h.put(1, 0);  // ref_pic_list_modification_flag_l0
        h.put(1, 0);  // adaptive_ref_pic_marking_mode_flag
    } else {
        h.put(1, 0);  // no_output_of_prior_pics_flag
        h.put(1, 0);  // long_term_reference_flag
    }
    h.se(-14);
    c->h_hdr[s * 4 + 0] = (uint32_t)(h.v >> 32);
    c->h_hdr[s * 4 + 1] = (uint32_t)h.v;
    c->h_hdr[s * 4 + 2] = (uint32_t)h.n;
    c->h_hdr[s * 4 + 3] = (uint32_t)slice_type;
    c->types[s] = slice_type;
}

// Start filling the next slot of the header ring (waits for the copy that used it FER_HDR_SLOTS pictures ago) ...
static int hdr_begin(ferhip_ctx *c)
{
    c->hdr_slot = (c->hdr_slot + 1) % FER_HDR_SLOTS;
    CK(hipEventSynchronize(c->hdr_ev[c->hdr_slot]));
    c->h_hdr = c->h_hdr_ring + (size_t)c->hdr_slot * 4 * c->d.S;
    return 0;
}
// ... and send it
static int hdr_upload(ferhip_ctx *c)
{
    CK(hipMemcpyAsync(c->d.hdr, c->h_hdr, sizeof(uint32_t) * 4 * c->d.S, hipMemcpyHostToDevice, c->st));
    CK(hipEventRecord(c->hdr_ev[c->hdr_slot], c->st));
    return 0;
}

// selectNALUnitType, F/ref_frames.cpp:185-234: nt[s] in = request (AUTO / IDR / SLICE), out = decision
static int decide_types(ferhip_ctx *c, const int *nal_type, std::vector<int> &nt)
{
    FerDev &d = c->d;
    const int S = d.S;
    nt.assign(S, 0);
    bool need_sad = false;
    for (int s = 0; s < S; s++) {
        int req = nal_type ? nal_type[s] : FERHIP_NAL_AUTO;
        StreamState &t = c->ss[s];
        if (req == FERHIP_NAL_IDR || req == FERHIP_NAL_SLICE) {
            nt[s] = (!t.have_dpb) ? FERHIP_NAL_IDR : req;
        } else if (!t.have_dpb || t.frames_done % c->p.intra_every == 0) {
            nt[s] = FERHIP_NAL_IDR;
        } else {
            nt[s] = -1;
            need_sad = true;
        }
    }
    if (need_sad) {
        {
            ProfScope ps(c, FERHIP_PH_FRAME_SAD, 1);
            fer_launch_frame_sad(d, c->st);
        }
        CK(hipMemcpyAsync(c->h_sad, d.sad, sizeof(unsigned long long) * S, hipMemcpyDeviceToHost, c->st));
        CK(hipStreamSynchronize(c->st));
        for (int s = 0; s < S; s++)
            if (nt[s] == -1) nt[s] = c->h_sad[s] > ((unsigned long long)d.nmb << 12) ? FERHIP_NAL_IDR : FERHIP_NAL_SLICE;
    }
    return 0;
}

extern "C" int ferhip_select_nal_type(ferhip_ctx *c, int *nal_type_out)
{
    if (!c || !nal_type_out) return FERHIP_E_ARG;
    (void)hipSetDevice(c->device);
    std::vector<int> nt;
    int rc = decide_types(c, nullptr, nt);
    if (rc) return rc;
    for (int s = 0; s < c->d.S; s++) nal_type_out[s] = nt[s];
    return 0;
}

static int run_picture(ferhip_ctx *c, int *nal_type)
{
    FerDev &d = c->d;
    const int S = d.S;
    std::vector<int> nt;
    int rc0 = decide_types(c, nal_type, nt);
    if (rc0) return rc0;
    bool anyP = false, anyI = false;
    if (hdr_begin(c)) return FERHIP_E_HIP;
    for (int s = 0; s < S; s++) {
        build_header(c, s, nt[s]);
        if (nal_type) nal_type[s] = nt[s];
        anyP |= c->types[s] == 0;
        anyI |= c->types[s] == 2;
    }
    if (hdr_upload(c)) return FERHIP_E_HIP;
    const int ndiag = d.mbw + 2 * (d.mbh - 1);
    if (anyP) {
        // The radix sort of the reference picture's positions and its bucket index are a stream of HBM traffic that only
        // the bucket walk needs; the stage-3 search (k_me_pre) is bound by VALU issue and needs only the quarter-sample
        // planes and the plane-0 features.  The two run side by side: the sort on st_aux, k_me_pre on st.
        bool sort_aside = false;
        if (!c->refprep_valid) {
            {
                ProfScope ps(c, FERHIP_PH_INTERP, 1);
                fer_launch_interp(d, c->st);
            }
            // BasicInterEncoding never walks the buckets (F/moestimation.cpp:470): the sorted order is not built
            if (!d.basic) {
                {
                    ProfScope ps(c, FERHIP_PH_SORT_KEYS, 1);
                    fer_launch_sort_keys(d, c->sort, c->st);
                }
                hipStream_t ss = c->st;
                if (c->overlap_sort) {
                    sort_aside = true;
                    ss = c->overlap_sort == 2 ? c->st_hi : c->st_aux;
                    CK(hipEventRecord(c->ev_c, c->st));
                    CK(hipStreamWaitEvent(ss, c->ev_c, 0));
                }
                {
                    ProfScope ps(c, FERHIP_PH_SORT, 1, ss);
                    fer_launch_sort_radix(d, c->sort, ss);
                }
                {
                    ProfScope ps(c, FERHIP_PH_SORT_FINISH, 1, ss);
                    fer_launch_sort_finish(d, c->sort, ss);
                }
                if (sort_aside) CK(hipEventRecord(c->ev_d, ss));
            }
        }
        {
            ProfScope ps(c, FERHIP_PH_ME_PRE, 1);
            fer_launch_me_pre(d, c->st);
        }
        if (sort_aside) CK(hipStreamWaitEvent(c->st, c->ev_d, 0));
        {
            ProfScope ps(c, FERHIP_PH_ME_WALK, 1);
            fer_launch_me_walk(d, c->st);
        }
        if (d.speculate) {
            ProfScope ps(c, FERHIP_PH_ME_SPEC, 1);
            fer_launch_me_spec(d, c->st);
        }
        {
            // the per-diagonal chain is latency bound: run it on the high-priority stream so that its small
            // launches are dispatched ahead of other contexts' throughput kernels
            CK(hipEventRecord(c->ev_a, c->st));
            CK(hipStreamWaitEvent(c->st_hi, c->ev_a, 0));
            {
                ProfScope ps(c, FERHIP_PH_ME_RESOLVE, 1, c->st_hi);
                d.serial = d.serial % 0x7ffffff0 + 1;  // validates this picture's words in chain64
                fer_launch_me_resolve(d, c->st_hi);
            }
            CK(hipEventRecord(c->ev_b, c->st_hi));
            CK(hipStreamWaitEvent(c->st, c->ev_b, 0));
        }
        {
            ProfScope ps(c, FERHIP_PH_P_RESID, 1);
            fer_launch_basic_stat(d, c->st);  // BasicInterEncoding only: brojTipova of the discarded exhaustive pass
            fer_launch_p_resid(d, c->st);
        }
    }
    if (anyI) {
        CK(hipEventRecord(c->ev_a, c->st));
        CK(hipStreamWaitEvent(c->st_hi, c->ev_a, 0));
        {
            ProfScope ps(c, FERHIP_PH_INTRA, ndiag, c->st_hi);
            fer_launch_intra(d, c->st_hi);
        }
        CK(hipEventRecord(c->ev_b, c->st_hi));
        CK(hipStreamWaitEvent(c->st, c->ev_b, 0));
    }
    {
        ProfScope ps(c, FERHIP_PH_CAVLC, 1);
        fer_launch_cavlc(d, c->st);
    }
    CK(hipGetLastError());
    // the reconstruction becomes the reference picture (frameDeepCopy, F/ref_frames.cpp:17)
    c->cur_set ^= 1;
    bind_planes(c);
    c->refprep_valid = false;
    for (int s = 0; s < S; s++) {
        c->ss[s].have_dpb = 1;
        c->ss[s].frames_done++;
    }
    return 0;
}

extern "C" int ferhip_encode_picture_dev(ferhip_ctx *c, int *nal_type, const uint8_t **d_rbsp, size_t *stride,
                                         const uint32_t **d_rbsp_len)
{
    if (!c) return FERHIP_E_ARG;
    (void)hipSetDevice(c->device);
    int rc = run_picture(c, nal_type);
    if (rc) return rc;
    if (d_rbsp) *d_rbsp = (const uint8_t *)c->d.bits;
    if (stride) *stride = c->d.bits_cap_words * 4;
    if (d_rbsp_len) *d_rbsp_len = c->d.out_bytes;
    return 0;
}

// RBSP of the last picture -> caller buffers, asynchronously on the context's stream (so it is ordered before the
// next picture reuses the device buffer): bytes_per_stream bytes of every stream's RBSP to dst + s * dst_stride and
// the S lengths to len_dst.  host = 1: dst / len_dst are (pinned) host memory.  ferhip_sync() waits.
extern "C" int ferhip_copy_rbsp(ferhip_ctx *c, void *dst, size_t dst_stride, size_t bytes_per_stream, uint32_t *len_dst, int host)
{
    if (!c || !dst || !len_dst) return FERHIP_E_ARG;
    (void)hipSetDevice(c->device);
    FerDev &d = c->d;
    const size_t cap = d.bits_cap_words * 4;
    if (bytes_per_stream > cap) bytes_per_stream = cap;
    if (bytes_per_stream > dst_stride) return FERHIP_E_ARG;
    const hipMemcpyKind kind = host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    CK(hipMemcpy2DAsync(dst, dst_stride, d.bits, cap, bytes_per_stream, (size_t)d.S, kind, c->st));
    CK(hipMemcpyAsync(len_dst, d.out_bytes, sizeof(uint32_t) * d.S, kind, c->st));
    return 0;
}

extern "C" int ferhip_sync(ferhip_ctx *c)
{
    if (!c) return FERHIP_E_ARG;
    (void)hipSetDevice(c->device);
    CK(hipStreamSynchronize(c->st));
    CK(hipStreamSynchronize(c->st_hi));
    return 0;
}

extern "C" int ferhip_encode_picture(ferhip_ctx *c, int *nal_type, uint8_t *rbsp, size_t rbsp_stride, uint32_t *rbsp_len)
{
    if (!c || !rbsp || !rbsp_len) return FERHIP_E_ARG;
    (void)hipSetDevice(c->device);
    int rc = run_picture(c, nal_type);
    if (rc) return rc;
    FerDev &d = c->d;
    CK(hipMemcpyAsync(c->h_len, d.out_bytes, sizeof(uint32_t) * d.S, hipMemcpyDeviceToHost, c->st));
    CK(hipMemcpyAsync(c->h_status, d.status, sizeof(int) * d.S, hipMemcpyDeviceToHost, c->st));
    CK(hipStreamSynchronize(c->st));
    for (int s = 0; s < d.S; s++) {
        if (c->h_status[s]) {
            fprintf(stderr, "ferhip: stream %d device status 0x%x\n", s, c->h_status[s]);
            return FERHIP_E_DEVICE;
        }
        rbsp_len[s] = c->h_len[s];
        if (c->h_len[s] > rbsp_stride) return FERHIP_E_ARG;
        CK(hipMemcpyAsync(rbsp + (size_t)s * rbsp_stride, (const uint8_t *)d.bits + (size_t)s * d.bits_cap_words * 4,
                          c->h_len[s], hipMemcpyDeviceToHost, c->st));
    }
    CK(hipStreamSynchronize(c->st));
    return 0;
}

extern "C" int ferhip_encode_streams(ferhip_ctx *c, const uint8_t *frames, int nframes, uint8_t *out, size_t out_stride,
                                     size_t *out_len, uint8_t *recon)
{
    if (!c || !frames || !out || !out_len || nframes <= 0) return FERHIP_E_ARG;
    FerDev &d = c->d;
    const int S = d.S;
    size_t fsz = d.ysz * 3 / 2;
    size_t rstride = d.bits_cap_words * 4;
    std::vector<uint8_t> rbsp(rstride * S);
    std::vector<uint32_t> len(S);
    std::vector<int> nt(S);
    uint8_t hdr[64];
    for (int s = 0; s < S; s++) {
        size_t pos = 0, n;
        n = ferhip_write_sps(c, hdr, sizeof hdr);
        pos += ferhip_write_nal(1, 7, hdr, n, out + (size_t)s * out_stride + pos);
        n = ferhip_write_pps(c, hdr, sizeof hdr);
        pos += ferhip_write_nal(1, 8, hdr, n, out + (size_t)s * out_stride + pos);
        out_len[s] = pos;
    }
    for (int f = 0; f < nframes; f++) {
        int rc = ferhip_set_frames(c, frames + (size_t)f * S * fsz, 1);
        if (rc) return rc;
        for (int s = 0; s < S; s++) nt[s] = FERHIP_NAL_AUTO;
        rc = ferhip_encode_picture(c, nt.data(), rbsp.data(), rstride, len.data());
        if (rc) return rc;
        for (int s = 0; s < S; s++) {
            if (out_len[s] + (size_t)len[s] * 3 / 2 + 16 > out_stride) return FERHIP_E_ARG;
            out_len[s] += ferhip_write_nal(1, nt[s], rbsp.data() + (size_t)s * rstride, len[s],
                                           out + (size_t)s * out_stride + out_len[s]);
        }
        if (recon) {
            rc = ferhip_get_recon(c, recon + (size_t)f * S * fsz, 1);
            if (rc) return rc;
        }
    }
    return 0;
}

// The context's streams are non-blocking: a synchronous copy on the null stream does not wait for them.
static hipError_t ctx_sync(ferhip_ctx *c)
{
    hipError_t e = hipStreamSynchronize(c->st);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(c->st_aux);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(c->st_hi);
}

extern "C" int ferhip_get_stats(ferhip_ctx *c, int *out)
{
    if (!c || !out) return FERHIP_E_ARG;
    (void)hipSetDevice(c->device);
    CK(ctx_sync(c));
    CK(hipMemcpy(out, c->d.stats, sizeof(int) * 5 * c->d.S, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int ferhip_status(ferhip_ctx *c, int *out)
{
    if (!c || !out) return FERHIP_E_ARG;
    (void)hipSetDevice(c->device);
    CK(ctx_sync(c));
    CK(hipMemcpy(out, c->d.status, sizeof(int) * c->d.S, hipMemcpyDeviceToHost));
    return 0;
}

// launch-shape knobs (results never depend on them)
extern "C" int ferhip_tune(ferhip_ctx *c, int key, int value)
{
    if (!c) return FERHIP_E_ARG;
    switch (key) {
    case FERHIP_TUNE_RESOLVE_WGS:
        if (value < 1 || value > 65535) return FERHIP_E_ARG;  // (any grid takes every row: workgroups move from queue to queue)
        c->d.resolve_wgs = value;
        return 0;
    case FERHIP_TUNE_RESOLVE_GROUP:
        if (value < 1) return FERHIP_E_ARG;
        c->d.resolve_group = value < c->d.S ? value : c->d.S;  // (more streams than the context has add nothing)
        return 0;
    case FERHIP_TUNE_OVERLAP_SORT:
        if (value < 0 || value > 2) return FERHIP_E_ARG;  // (2: on the high-priority stream -- an experiment)
        c->overlap_sort = value;
        return 0;
    case FERHIP_TUNE_SPECULATE:
        if (value != 0 && value != 1) return FERHIP_E_ARG;
        c->d.speculate = value;
        return 0;
    default: return FERHIP_E_ARG;
    }
}

extern "C" int ferhip_profile(ferhip_ctx *c, int enable)
{
    if (!c) return FERHIP_E_ARG;
    (void)hipSetDevice(c->device);
    c->prof = enable != 0;
    return 0;
}

extern "C" int ferhip_get_profile(ferhip_ctx *c, double *ms, long *launches, int reset)
{
    if (!c || !ms || !launches) return FERHIP_E_ARG;
    (void)hipSetDevice(c->device);
    CK(hipStreamSynchronize(c->st));
    for (auto &sp : c->spans) {
        float t = 0;
        CK(hipEventElapsedTime(&t, sp.a, sp.b));
        c->prof_ms[sp.phase] += t;
        c->prof_launches[sp.phase] += sp.launches;
        hipEventDestroy(sp.a);
        hipEventDestroy(sp.b);
    }
    c->spans.clear();
    for (int i = 0; i < FERHIP_NPHASE; i++) {
        ms[i] = c->prof_ms[i];
        launches[i] = c->prof_launches[i];
    }
    if (reset) {
        memset(c->prof_ms, 0, sizeof c->prof_ms);
        memset(c->prof_launches, 0, sizeof c->prof_launches);
    }
    return 0;
}

// ---- per-stage entry points
static void set_all_types(ferhip_ctx *c, int slice_type)
{
    if (hdr_begin(c)) return;
    for (int s = 0; s < c->d.S; s++) {
        c->h_hdr[s * 4 + 0] = c->h_hdr[s * 4 + 1] = 0;
        c->h_hdr[s * 4 + 2] = 1;
        c->h_hdr[s * 4 + 3] = (uint32_t)slice_type;
        c->types[s] = slice_type;
    }
    (void)hdr_upload(c);
}

extern "C" int ferhip_fill_interpolated(ferhip_ctx *c)
{
    if (!c) return FERHIP_E_ARG;
    (void)hipSetDevice(c->device);
    set_all_types(c, 0);
    fer_launch_refprep(c->d, c->sort, nullptr, c->st);
    CK(hipStreamSynchronize(c->st));
    CK(hipGetLastError());
    c->refprep_valid = true;
    return 0;
}

extern "C" int ferhip_inter_encoding(ferhip_ctx *c)
{
    if (!c) return FERHIP_E_ARG;
    (void)hipSetDevice(c->device);
    set_all_types(c, 0);
    if (!c->refprep_valid) fer_launch_refprep(c->d, c->sort, nullptr, c->st);
    c->refprep_valid = true;
    {
        ProfScope ps(c, FERHIP_PH_ME_PRE, 1);
        fer_launch_me_pre(c->d, c->st);
    }
    {
        ProfScope ps(c, FERHIP_PH_ME_WALK, 1);
        fer_launch_me_walk(c->d, c->st);
    }
    if (c->d.speculate) {
        ProfScope ps(c, FERHIP_PH_ME_SPEC, 1);
        fer_launch_me_spec(c->d, c->st);
    }
    {
        ProfScope ps(c, FERHIP_PH_ME_RESOLVE, 1);
        c->d.serial = c->d.serial % 0x7ffffff0 + 1;
        fer_launch_me_resolve(c->d, c->st);
    }
    fer_launch_basic_stat(c->d, c->st);
    fer_launch_p_resid(c->d, c->st);  // partition merge + mvd share the residual wavefront of the macroblock
    CK(hipStreamSynchronize(c->st));
    CK(hipGetLastError());
    return 0;
}

extern "C" size_t ferhip_read_buffer(ferhip_ctx *c, int which, void *dst, size_t cap)
{
    if (!c || !dst) return 0;
    (void)hipSetDevice(c->device);
    if (ctx_sync(c) != hipSuccess) return 0;
    FerDev &d = c->d;
    size_t nm = (size_t)d.nmb * d.S;
    const void *src = nullptr;
    size_t n = 0;
    switch (which) {
    case FERHIP_BUF_INTERP: {  // the planes without their margins
        n = d.ysz * 16 * d.S;
        if (n > cap) return 0;
        for (int pl = 0; pl < 16 * d.S; pl++)
            if (hipMemcpy2D((uint8_t *)dst + (size_t)pl * d.ysz, (size_t)d.W, d.interp + (size_t)pl * d.iplane + d.ioff, (size_t)d.ipitch,
                            (size_t)d.W, (size_t)d.H, hipMemcpyDeviceToHost) != hipSuccess)
                return 0;
        return n;
    }
    case FERHIP_BUF_FEAT: {
        // refFrameKar[0..4][frac] for every position: the searches derive what they need of it on chip, the full table
        // is built here on request from the interpolated planes (parity tests of row a16)
        n = d.ysz * 96 * d.S * 2;
        if (n > cap) return 0;
        if (!d.feat && dalloc(c, &d.feat, d.ysz * 96 * d.S)) return 0;
        if (hipDeviceSynchronize() != hipSuccess) return 0;
        fer_launch_features(d, c->st);
        src = d.feat;
        break;
    }
    case FERHIP_BUF_SORTPOS: src = d.sort_pos; n = d.ysz * d.S * 4; break;
    case FERHIP_BUF_KOLIKO: {  // the reference's koliko[] = first level of the bucket index, relative to the stream's segment
        n = (size_t)16385 * d.S * 4;
        if (n > cap) return 0;
        if (hipStreamSynchronize(c->st) != hipSuccess) return 0;
        for (int s = 0; s < d.S; s++) {
            int *o = (int *)dst + (size_t)s * 16385;
            if (hipMemcpy2D(o, 4, d.kol2 + (size_t)s * 16384 * d.kt, (size_t)d.kt * 4, 4, 16385, hipMemcpyDeviceToHost) != hipSuccess)
                return 0;
            for (int a = 0; a <= 16384; a++) o[a] -= (int)((size_t)s * d.ysz);
        }
        return n;
    }
    case FERHIP_BUF_MBTYPE: src = d.mb_type; n = nm * 4; break;
    case FERHIP_BUF_MV: src = d.mv; n = nm * 16; break;
    case FERHIP_BUF_MVD: src = d.mvd; n = nm * 16; break;
    case FERHIP_BUF_LEVELS: src = d.levels; n = nm * FER_LEVELS * 2; break;
    case FERHIP_BUF_CBP: src = d.cbp; n = nm * 2; break;
    case FERHIP_BUF_TC: src = d.tc; n = nm * 24; break;
    case FERHIP_BUF_I4MODE: src = d.i4mode; n = nm * 16; break;
    case FERHIP_BUF_TIMING: src = d.timing; n = 64 * 8; break;
    case FERHIP_BUF_ST2N: src = d.st2n; n = nm * 16; break;
    case FERHIP_BUF_ST2: src = d.st2; n = nm * 4 * FER_ST2_CAP * 8; break;
    case FERHIP_BUF_SPEC_STAT: src = d.spec_stat; n = 8 * 8; break;
    case FERHIP_BUF_MBSIZE: src = d.mbsize; n = nm * 8; break;
    case FERHIP_BUF_CUR:
    case FERHIP_BUF_REF: {
        n = d.ysz * 3 / 2 * d.S;
        if (n > cap) return 0;
        uint8_t *set = c->planes[which == FERHIP_BUF_CUR ? c->cur_set : c->cur_set ^ 1];
        if (copy_frames(c, set, nullptr, (uint8_t *)dst, hipMemcpyDeviceToHost)) return 0;
        if (hipStreamSynchronize(c->st) != hipSuccess) return 0;
        return n;
    }
    default: return 0;
    }
    if (n > cap) return 0;
    if (hipStreamSynchronize(c->st) != hipSuccess) return 0;
    if (hipMemcpy(dst, src, n, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return n;
}


// =====================================================================================
// Decoder driver: decode() / RBSP_decode() of F/fer_h264.cpp:26-53, F/rbsp_decoding.cpp:17-367.
// Host side: Annex-B scan + emulation-prevention removal (F/nal.cpp:68-223), parameter sets and
// slice header (F/headers_and_parameter_sets.cpp:245-298,398-537) -- a few dozen bits per NAL.
// The macroblock loop runs on the device (fer_decode.hip).
// =====================================================================================
struct HostBR {
    const uint8_t *b;
    size_t n, pos;
    unsigned bit()
    {
        size_t by = pos >> 3;
        unsigned v = by < n ? (b[by] >> (7 - (pos & 7))) & 1u : 0u;
        pos++;
        return v;
    }
    unsigned bits(int k)
    {
        unsigned v = 0;
        for (int i = 0; i < k; i++) v = (v << 1) | bit();
        return v;
    }
    unsigned ue()
    {
        int z = 0;
        while (z < 24 && bit() == 0) z++;  // the reference searches a 24-bit window (F/expgolomb.cpp:122)
        return (1u << z) - 1u + bits(z);
    }
    int se()
    {
        int v = (int)ue();
        return (v & 1) ? (v + 1) / 2 : -v / 2;
    }
};

struct DecHdr {
    int have_sps, have_pps, W, H, log2_max_frame_num, poc_type, log2_max_poc_lsb;
    int pic_init_qp, chroma_qp_offset, deblock_ctl, constrained_intra;
    // slice-header state the reference keeps in globals between slices: the active reference count is only ever set
    // by an override (never reset to the PPS default), the list-modification flag and its entry count only by P slices
    int nref_active_minus1, mod_flag, mod_copies;
};

// seq_parameter_set_rbsp, F/headers_and_parameter_sets.cpp:398-470.  Returns 0, or FERHIP_E_UNSUP for syntax the
// slice parser does not implement (the reference would mis-decode it silently).
static int dec_parse_sps(DecHdr &h, HostBR &r)
{
    const unsigned profile_idc = r.bits(8);
    r.bits(16);
    r.ue();
    if (profile_idc >= 100) return FERHIP_E_UNSUP;  // High profiles carry chroma_format_idc ... here
    h.log2_max_frame_num = (int)r.ue() + 4;
    h.poc_type = (int)r.ue();
    h.log2_max_poc_lsb = 0;
    if (h.poc_type == 0) {
        h.log2_max_poc_lsb = (int)r.ue() + 4;
    } else if (h.poc_type == 1) {
        r.bits(1);
        r.se();
        r.se();
        int n = (int)r.ue();
        for (int i = 0; i < n; i++) r.se();
    }
    r.ue();
    r.bits(1);
    int wmb = (int)r.ue() + 1, hmu = (int)r.ue() + 1, fmo = (int)r.bits(1);
    if (!fmo) return FERHIP_E_UNSUP;  // field / MBAFF coding
    h.W = wmb * 16;
    h.H = hmu * 16;
    h.have_sps = 1;
    return 0;
}

// pic_parameter_set_rbsp, F/headers_and_parameter_sets.cpp:520-537
static int dec_parse_pps(DecHdr &h, HostBR &r)
{
    r.ue();
    r.ue();
    if (r.bits(1)) return FERHIP_E_UNSUP;  // entropy_coding_mode_flag: CABAC
    r.bits(1);
    if (r.ue() > 0) return FERHIP_E_UNSUP;  // slice groups
    if (r.ue() > 0) return FERHIP_E_UNSUP;  // num_ref_idx_l0_default_active_minus1: several reference indices
    r.ue();
    r.bits(3);
    h.pic_init_qp = r.se() + 26;
    r.se();
    h.chroma_qp_offset = r.se();
    h.deblock_ctl = (int)r.bits(1);
    h.constrained_intra = (int)r.bits(1);
    h.have_pps = 1;
    return 0;
}

// slice header -> info[4] = {rbsp bytes, first bit of slice_data, slice_type % 5, SliceQPy}; returns 0 or error
static int dec_parse_slice_header(DecHdr &h, const uint8_t *rbsp, size_t n, int nal_type, int ref_idc, uint32_t *info,
                                  int &override_flag)
{
    if (!h.have_sps || !h.have_pps) return FERHIP_E_STATE;
    HostBR r{rbsp, n, 0};
    r.ue();
    int st = (int)r.ue() % 5;
    r.ue();
    r.bits(h.log2_max_frame_num);
    if (nal_type == 5) r.ue();
    r.bits(h.log2_max_poc_lsb);
    if (st == 0 || st == 1 || st == 3) {
        override_flag = (int)r.bits(1);
        if (override_flag) h.nref_active_minus1 = (int)r.ue();  // only tells the macroblock layer whether ref_idx is coded
    }
    if (st != 2 && st != 4) {  // ref_pic_list_modification (F/headers_and_parameter_sets.cpp:196-215)
        h.mod_flag = (int)r.bits(1);
        h.mod_copies = 0;
        if (h.mod_flag) {
            unsigned idc;
            int guard = 0;
            do {
                idc = r.ue();
                if (idc <= 2) {
                    r.ue();
                    h.mod_copies++;
                }
            } while (idc != 3 && ++guard < 64 && r.pos < n * 8);
        }
    }
    if (ref_idc != 0) {
        if (nal_type == 5) {
            r.bits(2);
        } else if (r.bits(1)) {
            unsigned op;
            do {
                op = r.ue();
                if (op == 1 || op == 3) r.ue();
                if (op == 2) r.ue();
                if (op == 3 || op == 6) r.ue();
                if (op == 4) r.ue();
            } while (op != 0);
        }
    }
    int qp = h.pic_init_qp + r.se();
    if (h.deblock_ctl == 1) {
        if (r.ue() != 1) {
            r.se();
            r.se();
        }
    }
    if (st != 0 && st != 2) return FERHIP_E_UNSUP;
    info[0] = (uint32_t)n;
    info[1] = (uint32_t)r.pos;
    // slice type | ref_idx coded in sub-macroblock prediction (the reference tests the override FLAG there,
    // F/rbsp_decoding.cpp:156) << 8 | active reference count - 1 (what it tests in mb_pred, :217) << 16
    info[2] = (uint32_t)st | ((uint32_t)(override_flag ? 1 : 0) << 8) | ((uint32_t)std::min(h.nref_active_minus1, 255) << 16);
    info[3] = (uint32_t)qp;
    return 0;
}

// split an Annex-B stream like findNALstart/findNALend/parseNAL (4-byte start codes only)
struct ByteView {  // bytes owned elsewhere: a stream's RBSP store (split_stream) or the caller's buffer (ferhip_dec_nal)
    const uint8_t *p = nullptr;
    size_t n = 0;
    const uint8_t *data() const { return p; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
};
struct NalRef {
    int type, ref_idc;
    ByteView rbsp;
};
// next position i in [from, n - 2) with s[i] == 0, s[i+1] == 0 and s[i+2] in `third` (two allowed values), or npos;
// zero bytes are rare in entropy-coded data, so the scan is driven by memchr
static size_t find_zz(const uint8_t *s, size_t from, size_t n, uint8_t t0, uint8_t t1)
{
    while (from + 2 < n) {
        const uint8_t *p = (const uint8_t *)memchr(s + from, 0, n - 2 - from);
        if (!p) break;
        size_t i = (size_t)(p - s);
        if (s[i + 1] == 0 && (s[i + 2] == t0 || s[i + 2] == t1)) return i;
        from = i + 1;
    }
    return (size_t)-1;
}
// `store` receives the RBSP of every NAL unit back to back (never more than the stream itself) and must outlive `out`
static void split_stream(const uint8_t *s, size_t n, std::vector<NalRef> &out, std::vector<uint8_t> &store)
{
    if (store.size() < n) store.resize(n);
    uint8_t *w = store.data();
    size_t pos = 0;
    for (;;) {
        size_t st = (size_t)-1;
        for (size_t i = pos; i + 3 < n;) {  // 00 00 00 01
            size_t z = find_zz(s, i, n - 1, 0, 0);
            if (z == (size_t)-1) break;
            if (s[z + 3] == 1) {
                st = z + 4;
                break;
            }
            i = z + 1;
        }
        if (st == (size_t)-1) break;
        size_t en = find_zz(s, st, n, 0, 1);
        if (en == (size_t)-1) en = n;
        pos = en;
        if (en <= st) continue;
        NalRef nal;
        nal.ref_idc = (s[st] & 0x7f) >> 5;
        nal.type = s[st] & 0x1f;
        uint8_t *w0 = w;
        size_t from = st + 1;
        for (;;) {  // drop the emulation prevention byte of every 00 00 03
            size_t z = find_zz(s, from, en, 3, 3);
            if (z == (size_t)-1) break;
            memcpy(w, s + from, z + 2 - from);
            w += z + 2 - from;
            from = z + 3;
        }
        if (from < en) {
            memcpy(w, s + from, en - from);
            w += en - from;
        }
        nal.rbsp.p = w0;
        nal.rbsp.n = (size_t)(w - w0);
        if (nal.rbsp.empty()) break;
        out.push_back(std::move(nal));
    }
}

// Window buffers of the decode twin.  ferhip_decode_streams keeps one arena per process between calls (releasing
// tens of GB costs more than a window's reconstruction); it is tied to the HIP device it was allocated on and
// can be dropped with ferhip_decode_release().  A streaming decoder (ferhip_dec_*) owns a small one of its own.
struct DecArena {
    void *dev = nullptr;
    size_t bytes = 0;
    int device = -1;
    uint8_t *d_rbsp = nullptr, *h_rbsp = nullptr;  // slices of a window: device buffer and pinned staging
    size_t rbsp_cap = 0;
    bool busy = false, cached = false;
};
static DecArena g_dec_arena;
static std::mutex g_dec_arena_mu;
static thread_local std::vector<std::vector<uint8_t>> tl_rbsp_store;  // ferhip_decode_streams: the streams' RBSP, per calling thread
static void dec_arena_free(DecArena *a)
{
    if (a->dev) hipFree(a->dev);
    if (a->d_rbsp) hipFree(a->d_rbsp);
    if (a->h_rbsp) hipHostFree(a->h_rbsp);
    a->dev = nullptr;
    a->d_rbsp = a->h_rbsp = nullptr;
    a->bytes = a->rbsp_cap = 0;
    a->device = -1;
}
static DecArena *dec_arena_acquire(size_t need, int device, bool use_cache)
{
    DecArena *a = nullptr;
    if (use_cache) {
        std::lock_guard<std::mutex> lk(g_dec_arena_mu);
        if (!g_dec_arena.busy) {
            a = &g_dec_arena;
            a->busy = true;
            a->cached = true;
        }
    }
    if (!a) {
        a = new DecArena();
        a->busy = true;
    }
    if (a->device != device && a->device >= 0) {  // allocated on another device: its pointers are useless here
        int cur = device;
        (void)hipSetDevice(a->device);
        dec_arena_free(a);
        (void)hipSetDevice(cur);
    }
    a->device = device;
    if (a->bytes < need) {
        if (a->dev) hipFree(a->dev);
        a->dev = nullptr;
        a->bytes = 0;
        if (hipMalloc(&a->dev, need) != hipSuccess) {
            (void)hipGetLastError();
            a->dev = nullptr;
            std::lock_guard<std::mutex> lk(g_dec_arena_mu);
            if (a->cached)
                a->busy = false;
            else
                delete a;
            return nullptr;
        }
        a->bytes = need;
    }
    return a;
}
static void dec_arena_release(DecArena *a)
{
    if (!a) return;
    if (a->cached) {
        std::lock_guard<std::mutex> lk(g_dec_arena_mu);
        a->busy = false;
    } else {
        dec_arena_free(a);
        delete a;
    }
}

extern "C" int ferhip_decode_release(void)
{
    std::vector<std::vector<uint8_t>>().swap(tl_rbsp_store);
    std::lock_guard<std::mutex> lk(g_dec_arena_mu);
    if (g_dec_arena.busy) return FERHIP_E_STATE;
    if (g_dec_arena.device >= 0) {
        int cur = 0;
        (void)hipGetDevice(&cur);
        (void)hipSetDevice(g_dec_arena.device);
        dec_arena_free(&g_dec_arena);
        (void)hipSetDevice(cur);
    }
    return 0;
}

// One decoding session: S streams of one picture size, a window of up to TWmax pictures per stream parsed by one
// launch, parameter sets and the state the reference keeps in globals (mb_qp_delta, ChromaACLevel) per stream.
struct DecSession {
    ferhip_ctx *c = nullptr;
    int S = 0;
    std::vector<DecHdr> hs;  // parameter sets of every stream (the QP of this codec lives in the PPS)
    DecArena *ar = nullptr;
    DecBatch B;
    uint32_t *d_info = nullptr;
    size_t TWmax = 0, nm = 0, fsz = 0;
    std::vector<uint32_t> info, hdr;
    std::vector<char> anyP, anyAny, keep;
    // A picture whose slice carries an empty reference-list modification is NOT stored as the reference picture
    // (modificationProcess, F/ref_frames.cpp:130-183): the stream's last decoded picture (what `frame` holds, what a
    // slice that ends early leaves in place) then differs from its reference picture and waits in `hold`.
    uint8_t *hold = nullptr;
    std::vector<char> held;
    double t_pack = 0, t_parse = 0, t_recon = 0;
};

static double dec_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void dec_session_close(DecSession &ss)
{
    if (ss.c) {
        (void)hipSetDevice(ss.c->device);
        hipStreamSynchronize(ss.c->st);
    }
    dec_arena_release(ss.ar);
    ss.ar = nullptr;
    if (ss.hold) hipFree(ss.hold);
    ss.hold = nullptr;
    if (ss.c) ferhip_destroy(ss.c);
    ss.c = nullptr;
}

// T = pictures per stream the caller expects (sizes the window); use_cache = take the process-wide arena
static int dec_session_open(DecSession &ss, int W, int H, int S, size_t T, bool use_cache)
{
    ferhip_params p = {26, 0, 16, 3, 1 << 30};
    int rc = ctx_create(&ss.c, W, H, S, &p, true);
    if (rc) return rc;
    FerDev &d = ss.c->d;
    ss.S = S;
    ss.nm = (size_t)S * d.nmb;
    ss.fsz = (size_t)W * H * 3 / 2;
    const size_t nm = ss.nm;
    const size_t per_pic = nm * (4 + 16 + 2 + 24 + 16 + 16 + 1 + FER_LEVELS * 2 + 1 + 1);
    size_t budget = (size_t)48e9, freeb = 0, totalb = 0;
    if (hipMemGetInfo(&freeb, &totalb) == hipSuccess) {
        std::lock_guard<std::mutex> lk(g_dec_arena_mu);
        if (use_cache && !g_dec_arena.busy && g_dec_arena.device == ss.c->device) freeb += g_dec_arena.bytes;
        budget = std::min(budget, freeb / 2);
    }
    size_t TWmax = std::min<size_t>(std::max<size_t>(budget / per_pic, 1), 256);
    TWmax = std::max<size_t>(std::min(TWmax, T), 1);
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    uint8_t *base = nullptr;
    size_t sizes[16];
    for (;;) {  // halve the window until its buffers fit
        const size_t sz[16] = {TWmax * nm * 4, TWmax * nm * 16, TWmax * nm * 2, TWmax * nm * 24, TWmax * nm * 16, TWmax * nm * 16,
                               TWmax * nm, TWmax * nm * FER_LEVELS * 2, TWmax * nm, TWmax * nm, TWmax * S * 16, TWmax * S * 16,
                               TWmax * S * 16, TWmax * S * 256, TWmax * S * 256, TWmax * S * 24};
        size_t need = 0;
        for (int i = 0; i < 16; i++) {
            sizes[i] = sz[i];
            need += up(sz[i]) + 256;
        }
        ss.ar = dec_arena_acquire(need, ss.c->device, use_cache);
        if (ss.ar) break;
        if (TWmax == 1) {
            dec_session_close(ss);
            return FERHIP_E_HIP;
        }
        TWmax = (TWmax + 1) / 2;
    }
    ss.TWmax = TWmax;
    base = (uint8_t *)ss.ar->dev;
    size_t off = 0;
    auto wmalloc = [&](size_t bytes) -> void * {
        void *v = base + off;
        off += up(bytes) + 256;
        return v;
    };
    DecBatch &B = ss.B;
    B.mb_type = (int *)wmalloc(sizes[0]);
    B.mv = (short *)wmalloc(sizes[1]);
    B.cbp = (uint8_t *)wmalloc(sizes[2]);
    B.tc = (uint8_t *)wmalloc(sizes[3]);
    B.i4mode = (uint8_t *)wmalloc(sizes[4]);
    B.i4flag = (uint8_t *)wmalloc(sizes[5]);
    B.chroma_mode = (uint8_t *)wmalloc(sizes[6]);
    B.levels = (int16_t *)wmalloc(sizes[7]);
    B.dec_qp = (uint8_t *)wmalloc(sizes[8]);
    B.carry = (uint8_t *)wmalloc(sizes[9]);
    B.hdr = (uint32_t *)wmalloc(sizes[10]);
    B.state = (int *)wmalloc(sizes[11]);
    B.summ = (int *)wmalloc(sizes[12]);
    B.cac_in = (int16_t *)wmalloc(sizes[13]);
    B.cac_out = (int16_t *)wmalloc(sizes[14]);
    ss.d_info = (uint32_t *)wmalloc(sizes[15]);
    ss.hs.assign(S, DecHdr{});
    ss.held.assign(S, 0);
    return 0;
}

// the three planes of stream s from one plane-major picture set to another (same stream order)
static int dec_copy_stream(ferhip_ctx *c, uint8_t *dst, const uint8_t *src, int s)
{
    const FerDev &d = c->d;
    const size_t S = (size_t)d.S;
    const size_t off[3] = {(size_t)s * d.ysz, S * d.ysz + (size_t)s * d.csz, S * (d.ysz + d.csz) + (size_t)s * d.csz};
    const size_t len[3] = {d.ysz, d.csz, d.csz};
    for (int k = 0; k < 3; k++)
        if (hipMemcpyAsync(dst + off[k], src + off[k], len[k], hipMemcpyDeviceToDevice, c->st) != hipSuccess) return FERHIP_E_HIP;
    return 0;
}

// Decode pictures [t0, t0 + TW) of every stream: slices[s][t] = the slice NAL of picture t of stream s (streams may
// be shorter).  out (host, may be NULL) receives picture t of stream s at (t * S + s) * fsz; pictures[s] counts.
static int dec_session_window(DecSession &ss, const std::vector<std::vector<const NalRef *>> &slices, size_t t0, size_t TW,
                              uint8_t *out, int *pictures)
{
    ferhip_ctx *c = ss.c;
    FerDev &d = c->d;
    const int S = ss.S;
    const size_t nm = ss.nm;
    DecBatch &B = ss.B;
    DecArena *ar = ss.ar;
    double ta = dec_now();
    ss.info.assign(TW * S * 6, 0);
    ss.hdr.assign(TW * S * 4, 0);
    ss.anyP.assign(TW, 0);
    ss.anyAny.assign(TW, 0);
    ss.keep.assign(TW * S, 1);
    // slice headers and the offsets of the slices in the window's RBSP buffer
    size_t total = 0;
    for (size_t t = 0; t < TW; t++)
        for (int s = 0; s < S; s++) {
            uint32_t *in = &ss.info[(t * S + s) * 6];
            uint32_t *hd = &ss.hdr[(t * S + s) * 4];
            hd[3] = 2;
            if (t0 + t >= slices[s].size()) continue;
            const NalRef &n = *slices[s][t0 + t];
            int ov = 0;
            int rc = dec_parse_slice_header(ss.hs[s], n.rbsp.data(), n.rbsp.size(), n.type, n.ref_idc, in, ov);
            if (rc) return rc;
            in[4] = (uint32_t)total;
            in[5] = (uint32_t)(total >> 32);
            total += (n.rbsp.size() + 15) & ~(size_t)15;
            // what the kernels need of this stream's PPS travels with the picture
            hd[0] = (uint32_t)ss.hs[s].chroma_qp_offset;
            hd[1] = (uint32_t)ss.hs[s].constrained_intra;
            hd[3] = in[2] & 255u;
            ss.anyP[t] |= (in[2] & 255u) == 0;
            ss.anyAny[t] = 1;
            ss.keep[t * S + s] = !ss.hs[s].mod_flag || ss.hs[s].mod_copies > 0;
        }
    if (total + 64 > ar->rbsp_cap) {
        if (ar->d_rbsp) hipFree(ar->d_rbsp);
        if (ar->h_rbsp) hipHostFree(ar->h_rbsp);
        ar->d_rbsp = ar->h_rbsp = nullptr;
        ar->rbsp_cap = total + total / 4 + 4096;
        if (hipMalloc((void **)&ar->d_rbsp, ar->rbsp_cap) != hipSuccess || hipHostMalloc((void **)&ar->h_rbsp, ar->rbsp_cap) != hipSuccess) {
            ar->rbsp_cap = 0;
            return FERHIP_E_HIP;
        }
    }
    {  // gather the slices into the pinned staging buffer with a few threads, then one H2D copy
        const int nth = std::max(1, std::min(std::min(S, 16), (int)std::thread::hardware_concurrency()));
        auto gather = [&](int k) {
            for (int s = k; s < S; s += nth)
                for (size_t t = 0; t < TW; t++) {
                    if (t0 + t >= slices[s].size()) continue;
                    const NalRef &n = *slices[s][t0 + t];
                    const uint32_t *in = &ss.info[(t * S + s) * 6];
                    memcpy(ar->h_rbsp + (((size_t)in[5] << 32) | in[4]), n.rbsp.data(), n.rbsp.size());
                }
        };
        if (nth == 1) {
            gather(0);
        } else {
            std::vector<std::thread> th;
            for (int k = 0; k < nth; k++) th.emplace_back(gather, k);
            for (auto &x : th) x.join();
        }
    }
    if (hipMemcpyAsync(ar->d_rbsp, ar->h_rbsp, total, hipMemcpyHostToDevice, c->st) != hipSuccess ||
        hipMemcpyAsync(ss.d_info, ss.info.data(), ss.info.size() * 4, hipMemcpyHostToDevice, c->st) != hipSuccess ||
        hipMemcpyAsync(B.hdr, ss.hdr.data(), ss.hdr.size() * 4, hipMemcpyHostToDevice, c->st) != hipSuccess)
        return FERHIP_E_HIP;
    B.TW = (int)TW;
    B.rbsp = ar->d_rbsp;
    B.info = ss.d_info;
    ss.t_pack += dec_now() - ta;
    ta = dec_now();
    fer_launch_decode_parse(d, B, c->st);
    // the host vectors above are read by the copies when they execute: the synchronisation below covers them
    if (hipMemcpyAsync(c->h_status, d.status, sizeof(int) * S, hipMemcpyDeviceToHost, c->st) != hipSuccess ||
        hipStreamSynchronize(c->st) != hipSuccess || hipGetLastError() != hipSuccess)
        return FERHIP_E_HIP;
    for (int s = 0; s < S; s++)
        if (c->h_status[s]) {
            fprintf(stderr, "ferhip: stream %d decode status 0x%x\n", s, c->h_status[s]);
            return (c->h_status[s] & FER_ERR_DEC_UNSUPPORTED) ? FERHIP_E_UNSUP : FERHIP_E_DEVICE;
        }
    ss.t_parse += dec_now() - ta;
    ta = dec_now();
    for (size_t t = 0; t < TW; t++) {
        if (!ss.anyAny[t]) break;
        // `frame` keeps the previous picture where the parser does not reach (F/rbsp_decoding.cpp:77)
        if (hipMemcpyAsync(c->planes[c->cur_set], c->planes[c->cur_set ^ 1], d.ysz * 3 / 2 * S, hipMemcpyDeviceToDevice, c->st) !=
            hipSuccess)
            return FERHIP_E_HIP;
        for (int s = 0; s < S; s++)  // ... which, for a stream whose last picture was not stored as reference, waits in `hold`
            if (ss.held[s] && dec_copy_stream(c, c->planes[c->cur_set], ss.hold, s)) return FERHIP_E_HIP;
        FerDev ds = d;  // this picture's slice of the window
        const size_t o = t * nm;
        ds.mb_type = B.mb_type + o;
        ds.mv = B.mv + o * 8;
        ds.cbp = B.cbp + o * 2;
        ds.tc = B.tc + o * 24;
        ds.i4mode = B.i4mode + o * 16;
        ds.i4flag = B.i4flag + o * 16;
        ds.chroma_mode = B.chroma_mode + o;
        ds.levels = B.levels + o * FER_LEVELS;
        ds.dec_qp = B.dec_qp + o;
        ds.hdr = B.hdr + t * S * 4;
        ds.dec_state = B.state + t * S * 4;
        fer_launch_decode_recon(ds, ss.anyP[t] != 0, true, c->st);
        c->cur_set ^= 1;  // the decoded picture becomes the reference (modificationProcess -> frameDeepCopy)
        bind_planes(c);
        if (out) {
            int rc = ferhip_get_recon(c, out + (t0 + t) * S * ss.fsz, 1);
            if (rc) return rc;
        }
        for (int s = 0; s < S; s++) {
            if (t0 + t >= slices[s].size()) continue;
            if (!ss.keep[t * S + s]) {
                // not stored: the picture moves to `hold`, the stream's reference picture (still intact in the other
                // set) moves back into the reference set
                if (!ss.hold) {
                    if (hipMalloc((void **)&ss.hold, d.ysz * 3 / 2 * S + 256) != hipSuccess) return FERHIP_E_HIP;
                }
                if (dec_copy_stream(c, ss.hold, c->planes[c->cur_set ^ 1], s) ||
                    dec_copy_stream(c, c->planes[c->cur_set ^ 1], c->planes[c->cur_set], s))
                    return FERHIP_E_HIP;
                ss.held[s] = 1;
            } else {
                ss.held[s] = 0;
            }
        }
        if (pictures)
            for (int s = 0; s < S; s++)
                if (t0 + t < slices[s].size()) pictures[s]++;
    }
    if (hipStreamSynchronize(c->st) != hipSuccess || hipGetLastError() != hipSuccess) return FERHIP_E_HIP;
    ss.t_recon += dec_now() - ta;
    return 0;
}

// decode() for S Annex-B streams side by side.  All streams must carry the same picture size.
// out: host [T][S][W*H*3/2] (T = max_pictures); pictures[s] = pictures decoded of stream s.
extern "C" int ferhip_decode_streams(const uint8_t *const *streams, const size_t *lens, int S, uint8_t *out,
                                     int max_pictures, int *pictures, int *W_out, int *H_out)
{
    if (!streams || !lens || S <= 0 || !pictures) return FERHIP_E_ARG;
    if (out && max_pictures <= 0) return FERHIP_E_ARG;  // `out` holds max_pictures pictures per stream: its size must be known
    const bool verbose = getenv("FER_DEC_TIMING") != nullptr;
    double t_start = dec_now();
    std::vector<std::vector<NalRef>> nals(S);
    // the RBSP of every stream, kept by the calling thread between calls like the device arena: 240 MB of fresh heap per
    // call (128 1080p GOPs) cost 0.25 s in page faults and unmapping, a fifth of the decode itself
    std::vector<std::vector<uint8_t>> &rbsp_store = tl_rbsp_store;
    if ((int)rbsp_store.size() < S) rbsp_store.resize(S);
    {  // NAL splitting is host work per stream: spread it over a few threads
        const int nth = std::max(1, std::min(std::min(S, 16), (int)std::thread::hardware_concurrency()));
        std::vector<std::thread> th;
        std::vector<std::vector<uint8_t>> &store = rbsp_store;
        for (int k = 0; k < nth; k++)
            th.emplace_back([&, k]() {
                for (int s = k; s < S; s += nth) split_stream(streams[s], lens[s], nals[s], store[s]);
            });
        for (auto &x : th) x.join();
    }
    // parameter sets per stream (the last SPS / PPS of a stream wins, as in the reference, which keeps one of each)
    std::vector<DecHdr> hs(S, DecHdr{});
    for (int s = 0; s < S; s++) {
        for (auto &n : nals[s]) {
            HostBR r{n.rbsp.data(), n.rbsp.size(), 0};
            int rc = 0;
            if (n.type == 7)
                rc = dec_parse_sps(hs[s], r);
            else if (n.type == 8)
                rc = dec_parse_pps(hs[s], r);
            if (rc) return rc;
        }
        if (!hs[s].have_sps || !hs[s].have_pps) return FERHIP_E_ARG;
        if (hs[s].W != hs[0].W || hs[s].H != hs[0].H) return FERHIP_E_ARG;
    }
    if (W_out) *W_out = hs[0].W;
    if (H_out) *H_out = hs[0].H;
    const double t_split = dec_now();
    // the slice NALs of every stream, in order
    std::vector<std::vector<const NalRef *>> slices(S);
    size_t T = 0;
    for (int s = 0; s < S; s++) {
        for (auto &n : nals[s])
            if (n.type == 1 || n.type == 5) slices[s].push_back(&n);
        T = std::max(T, slices[s].size());
        pictures[s] = 0;
    }
    if (max_pictures > 0) T = std::min(T, (size_t)max_pictures);
    for (int s = 0; s < S; s++)
        if (slices[s].size() > T) slices[s].resize(T);
    DecSession ss;
    int rc = dec_session_open(ss, hs[0].W, hs[0].H, S, T, true);
    if (rc) return rc;
    ss.hs = hs;
    const double t_open = dec_now();
    // Slice data is bit-serial, so the parser's parallelism is pictures: a window of TW pictures of all streams is
    // parsed by one launch (one wavefront each), then reconstructed picture by picture.
    for (size_t t0 = 0; t0 < T && rc == 0; t0 += ss.TWmax) rc = dec_session_window(ss, slices, t0, std::min(ss.TWmax, T - t0), out, pictures);
    if (verbose)
        fprintf(stderr, "ferhip_decode_streams: %d streams, %zu pictures, window %zu: split %.3f s, context + buffers %.3f s, "
                        "pack+H2D %.3f s, parse %.3f s, reconstruction %.3f s\n", S, T, ss.TWmax, t_split - t_start, t_open - t_split,
                ss.t_pack, ss.t_parse, ss.t_recon);
#ifdef FER_PROBE
    {
        long long tm[64];
        hipDeviceSynchronize();
        hipMemcpy(tm, ss.c->d.timing, sizeof tm, hipMemcpyDeviceToHost);
        double n = tm[52] > 0 ? (double)tm[52] : 1.0;
        fprintf(stderr, "k_dec_parse stream 0: %lld MBs, us per MB: header %.2f residual %.2f tail %.2f skip/loop %.2f\n", tm[52],
                tm[48] / n / 100, tm[49] / n / 100, tm[50] / n / 100, tm[51] / n / 100);
    }
#endif
    const double t_free0 = dec_now();
    dec_session_close(ss);
    if (verbose) fprintf(stderr, "ferhip_decode_streams: release %.3f s, total %.3f s\n", dec_now() - t_free0, dec_now() - t_start);
    return rc;
}

// ---- streaming decoder: RBSP_decode(NALunit) of F/rbsp_decoding.cpp:17 for one stream, NAL unit by NAL unit ----
struct ferhip_dec {
    DecSession ss;
    DecHdr h{};
    bool open = false;
    int pictures = 0;
};

extern "C" int ferhip_dec_create(ferhip_dec **out)
{
    if (!out) return FERHIP_E_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        fprintf(stderr, "ferhip: no HIP device; the hot path has no CPU fallback\n");
        return FERHIP_E_HIP;
    }
    *out = new ferhip_dec();
    return 0;
}

extern "C" void ferhip_dec_destroy(ferhip_dec *dc)
{
    if (!dc) return;
    if (dc->open) dec_session_close(dc->ss);
    delete dc;
}

extern "C" int ferhip_dec_nal(ferhip_dec *dc, int nal_unit_type, int nal_ref_idc, const uint8_t *rbsp, size_t n, uint8_t *picture,
                              int *got_picture, int *width, int *height)
{
    if (!dc || !rbsp || n == 0) return FERHIP_E_ARG;
    if (got_picture) *got_picture = 0;
    HostBR r{rbsp, n, 0};
    if (nal_unit_type == 7) {  // fill_sps + init_h264_structures + AllocateMemory
        DecHdr hn = dc->h;
        int rc = dec_parse_sps(hn, r);
        if (rc) return rc;
        if (dc->open && (hn.W != dc->h.W || hn.H != dc->h.H)) {
            dec_session_close(dc->ss);
            dc->open = false;
        }
        dc->h = hn;
        if (!dc->open) {
            dc->ss = DecSession();
            rc = dec_session_open(dc->ss, hn.W, hn.H, 1, 1, false);
            if (rc) return rc;
            dc->open = true;
        }
    } else if (nal_unit_type == 8) {
        int rc = dec_parse_pps(dc->h, r);
        if (rc) return rc;
    } else if (nal_unit_type == 5 || nal_unit_type == 1) {
        if (!dc->open) return FERHIP_E_STATE;
        (void)hipSetDevice(dc->ss.c->device);
        dc->ss.hs[0] = dc->h;
        NalRef nal;
        nal.type = nal_unit_type;
        nal.ref_idc = nal_ref_idc;
        nal.rbsp.p = rbsp;
        nal.rbsp.n = n;
        std::vector<std::vector<const NalRef *>> slices(1);
        slices[0].push_back(&nal);
        int rc = dec_session_window(dc->ss, slices, 0, 1, picture, nullptr);
        if (rc) return rc;
        dc->h = dc->ss.hs[0];  // the slice header leaves state behind (reference count override, list modification)
        dc->pictures++;
        if (got_picture) *got_picture = 1;
    }  // every other NAL unit type (SEI, AUD ...) is ignored, as in the reference
    if (width) *width = dc->h.W;
    if (height) *height = dc->h.H;
    return 0;
}

// ---- block-level KATs: the reference's per-block entry points (F/quantizationTransform.h, F/scaleTransform.h) as
// batched device calls over the SAME device functions the production kernels use (fer_dev.h)
#define KAT_FWD_RESIDUAL 0
#define KAT_INV_RESIDUAL 1
#define KAT_FWD_DC_LUMA 2
#define KAT_INV_DC_LUMA 3
#define KAT_FWD_DC_CHROMA 4
#define KAT_INV_DC_CHROMA 5
#define KAT_SCAN 6
#define KAT_INV_SCAN 7
__global__ void k_block_kat(int op, int qP, const int32_t *in, int32_t *out, int flag, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int a[16], t[16], b[16];
    for (int k = 0; k < 16; k++) {
        a[k] = in[i * 16 + k];
        b[k] = 0;
    }
    switch (op) {
    case KAT_FWD_RESIDUAL:
        fwd4x4(a, t);
        quant4x4(t, b, qP, flag != 0);
        break;
    case KAT_INV_RESIDUAL: inv4x4(a, b, qP, flag != 0); break;
    case KAT_FWD_DC_LUMA: fwd_dc_luma(a, b, qP); break;
    case KAT_INV_DC_LUMA: inv_dc_luma(a, b, qP); break;
    case KAT_FWD_DC_CHROMA: fwd_dc_chroma(a, b, qP); break;   // 2x2 raster in slots 0..3
    case KAT_INV_DC_CHROMA: inv_dc_chroma(a, b, qP); break;
    case KAT_SCAN:  // transformScan: flag = Intra16x16AC (15 entries from index 1)
        for (int k = flag ? 1 : 0; k < 16; k++) b[k - (flag ? 1 : 0)] = a[c_zz[k]];
        break;
    default:  // transformInverseScan
        for (int k = 0; k < 16; k++) b[c_zz[k]] = a[k];
        break;
    }
    for (int k = 0; k < 16; k++) out[i * 16 + k] = b[k];
}

static int block_kat(int op, int qP, const int32_t *in, int32_t *out, int flag, size_t n)
{
    if (!in || !out || qP < 0 || qP > 51) return FERHIP_E_ARG;
    if (n == 0) return 0;
    int32_t *di = nullptr, *dout = nullptr;
    CK(hipMalloc((void **)&di, n * 64));
    if (hipMalloc((void **)&dout, n * 64) != hipSuccess) {
        hipFree(di);
        return FERHIP_E_HIP;
    }
    int rc = 0;
    if (hipMemcpy(di, in, n * 64, hipMemcpyHostToDevice) != hipSuccess) rc = FERHIP_E_HIP;
    if (!rc) {
        hipLaunchKernelGGL(k_block_kat, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, op, qP, di, dout, flag, n);
        if (hipGetLastError() != hipSuccess || hipMemcpy(out, dout, n * 64, hipMemcpyDeviceToHost) != hipSuccess) rc = FERHIP_E_HIP;
    }
    hipFree(di);
    hipFree(dout);
    return rc;
}

extern "C" int ferhip_forward_residual(int qP, const int32_t *in, int32_t *out, int keep_dc, size_t n)
{
    return block_kat(KAT_FWD_RESIDUAL, qP, in, out, keep_dc, n);
}
extern "C" int ferhip_inverse_residual(int qP, const int32_t *in, int32_t *out, int keep_dc, size_t n)
{
    return block_kat(KAT_INV_RESIDUAL, qP, in, out, keep_dc, n);
}
extern "C" int ferhip_forward_dc_luma_intra(int qP, const int32_t *in, int32_t *out, size_t n) { return block_kat(KAT_FWD_DC_LUMA, qP, in, out, 0, n); }
extern "C" int ferhip_inverse_dc_luma_intra(int qP, const int32_t *in, int32_t *out, size_t n) { return block_kat(KAT_INV_DC_LUMA, qP, in, out, 0, n); }
extern "C" int ferhip_forward_dc_chroma(int qP, const int32_t *in, int32_t *out, size_t n) { return block_kat(KAT_FWD_DC_CHROMA, qP, in, out, 0, n); }
extern "C" int ferhip_inverse_dc_chroma(int qP, const int32_t *in, int32_t *out, size_t n) { return block_kat(KAT_INV_DC_CHROMA, qP, in, out, 0, n); }
extern "C" int ferhip_transform_scan(const int32_t *in, int32_t *out, int intra16x16_ac, size_t n) { return block_kat(KAT_SCAN, 0, in, out, intra16x16_ac, n); }
extern "C" int ferhip_transform_inverse_scan(const int32_t *in, int32_t *out, size_t n) { return block_kat(KAT_INV_SCAN, 0, in, out, 0, n); }
