// fer_fileio.hip -- Y4M ingest / YUV-Y4M emit around the hot path (row f3 of SURVEY.md 8f), host code only.
//
// The reference reads its input with LoadY4MHeader() / ReadFromY4M() (F/fileIO.cpp:228-346) and writes decoded
// pictures with writeToY4M() / writeToYUV() (F/fileIO.cpp:100-176), all through the global `frame` and the FILE
// handles `yuvinput` / `yuvoutput`.  Those names are exported here with the reference's behaviour:
//   * the picture size comes from the " W" and " H" tokens of the stream header, cropped to multiples of 16
//     (:242-243) around the centre (:290-293, chroma at half the offsets :317-320);
//   * a picture starts after the line that begins with "FRAME" (frame parameters are skipped with the line);
//   * a short read ends the stream (ReadFromY4M returns -1);
//   * writeToY4M writes the header "YUV4MPEG2 C420jpeg W%d H%d F24:1 Ip A1:1\n" once, then "FRAME\n" + I420.
// The context-based reader (ferhip_y4m_*) cuts the same coded pictures into a caller buffer -- pinned memory when
// they are headed for ferhip_set_frames -- so that many streams can be ingested without the globals.
#include "../../include/ferhip.h"
#include "../../include/ferhip_legacy.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

struct ferhip_y4m {
    FILE *f;
    bool own;
    int inW, inH, W, H;
    std::vector<unsigned char> buf;  // one input picture
};

// the header line: up to the first '\n'; W / H from the " W<n>" and " H<n>" tokens (F/fileIO.cpp:233-239)
static int y4m_parse_header(ferhip_y4m *y)
{
    char line[1024];
    size_t n = 0;
    int ch;
    while (n + 1 < sizeof line && (ch = fgetc(y->f)) != EOF) {
        if (ch == '\n') break;
        line[n++] = (char)ch;
    }
    line[n] = 0;
    const char *w = strstr(line, " W"), *h = strstr(line, " H");
    if (!w || !h || sscanf(w + 2, "%d", &y->inW) != 1 || sscanf(h + 2, "%d", &y->inH) != 1) return FERHIP_E_ARG;
    if (y->inW < 16 || y->inH < 16) return FERHIP_E_ARG;
    y->W = y->inW & ~15;
    y->H = y->inH & ~15;
    y->buf.resize((size_t)y->inW * y->inH + 2 * ((size_t)(y->inW >> 1) * (y->inH >> 1)));
    return 0;
}

extern "C" int ferhip_y4m_open(ferhip_y4m **out, const char *path, int *in_w, int *in_h, int *coded_w, int *coded_h)
{
    if (!out || !path) return FERHIP_E_ARG;
    FILE *f = fopen(path, "rb");
    if (!f) return FERHIP_E_ARG;
    ferhip_y4m *y = new ferhip_y4m{f, true, 0, 0, 0, 0, {}};
    int rc = y4m_parse_header(y);
    if (rc) {
        fclose(f);
        delete y;
        return rc;
    }
    if (in_w) *in_w = y->inW;
    if (in_h) *in_h = y->inH;
    if (coded_w) *coded_w = y->W;
    if (coded_h) *coded_h = y->H;
    *out = y;
    return 0;
}

extern "C" void ferhip_y4m_close(ferhip_y4m *y)
{
    if (!y) return;
    if (y->own && y->f) fclose(y->f);
    delete y;
}

// centre crop of one plane (F/fileIO.cpp:290-333)
static void crop_plane(const unsigned char *in, int inW, int top, int left, unsigned char *out, int W, int H)
{
    for (int r = 0; r < H; r++) memcpy(out + (size_t)r * W, in + (size_t)(top + r) * inW + left, (size_t)W);
}

// next picture as coded-size I420 [W*H*3/2]; returns 0, or 1 at the end of the stream
extern "C" int ferhip_y4m_read(ferhip_y4m *y, unsigned char *dst)
{
    if (!y || !dst) return FERHIP_E_ARG;
    // the line that starts the picture: "FRAME" + optional parameters + '\n'
    int ch, m = 0;
    const char tag[] = "FRAME";
    while (m < 5) {
        ch = fgetc(y->f);
        if (ch == EOF) return 1;
        m = ch == tag[m] ? m + 1 : (ch == 'F' ? 1 : 0);
    }
    while ((ch = fgetc(y->f)) != EOF && ch != '\n') {
    }
    if (ch == EOF) return 1;
    if (fread(y->buf.data(), 1, y->buf.size(), y->f) != y->buf.size()) return 1;  // "End of stream found."
    const int inW = y->inW, inH = y->inH, W = y->W, H = y->H;
    const size_t luma = (size_t)inW * inH, chroma = luma >> 2;
    int top = (inH - H) >> 1, left = (inW - W) >> 1;
    crop_plane(y->buf.data(), inW, top, left, dst, W, H);
    top >>= 1;
    left >>= 1;
    crop_plane(y->buf.data() + luma, inW >> 1, top, left, dst + (size_t)W * H, W >> 1, H >> 1);
    crop_plane(y->buf.data() + luma + chroma, inW >> 1, top, left, dst + (size_t)W * H * 5 / 4, W >> 1, H >> 1);
    return 0;
}

extern "C" int ferhip_y4m_write_header(void *file, int W, int H)
{
    FILE *f = (FILE *)file;
    if (!f) return FERHIP_E_ARG;
    return fprintf(f, "YUV4MPEG2 C420jpeg W%d H%d F24:1 Ip A1:1%c", W, H, 0x0a) > 0 ? 0 : FERHIP_E_ARG;
}

extern "C" int ferhip_y4m_write_frame(void *file, const unsigned char *i420, int W, int H, int with_frame_line)
{
    FILE *f = (FILE *)file;
    if (!f || !i420) return FERHIP_E_ARG;
    if (with_frame_line && fprintf(f, "FRAME%c", 0x0a) <= 0) return FERHIP_E_ARG;
    const size_t n = (size_t)W * H * 3 / 2;
    return fwrite(i420, 1, n, f) == n ? 0 : FERHIP_E_ARG;
}

// ---------------------------------------------------------------- legacy names (global `frame`)
extern "C" {
FILE *yuvinput = nullptr;
FILE *yuvoutput = nullptr;
int inputWidth = 0, inputHeight = 0;
}
static ferhip_y4m *g_in = nullptr;
static bool g_out_header_done = false;

// The reference allocates `frame` at SPS time (init_h264_structures[_encoder], F/h264_globals.cpp:224-297) and never
// frees it.  A host may also have pointed `frame` at buffers of its own: only planes allocated here are ever released.
static unsigned char *g_own[3] = {nullptr, nullptr, nullptr};
static void frame_alloc_if_needed()
{
    if (!frame.L) frame.L = g_own[0] = new unsigned char[(size_t)frame.Lwidth * frame.Lheight];
    if (!frame.C[0]) frame.C[0] = g_own[1] = new unsigned char[(size_t)frame.Cwidth * frame.Cheight];
    if (!frame.C[1]) frame.C[1] = g_own[2] = new unsigned char[(size_t)frame.Cwidth * frame.Cheight];
}
extern "C" void ferhip_legacy_frame_alloc(void) { frame_alloc_if_needed(); }
// a new picture size: drop the planes (the library's own are freed, a host's are merely forgotten)
extern "C" void ferhip_legacy_frame_drop(void)
{
    unsigned char **pl[3] = {&frame.L, &frame.C[0], &frame.C[1]};
    for (int i = 0; i < 3; i++) {
        if (*pl[i] && *pl[i] == g_own[i]) delete[] g_own[i];
        g_own[i] = nullptr;
        *pl[i] = nullptr;
    }
}

extern "C" void LoadY4MHeader(void)
{
    if (!yuvinput) return;
    if (g_in) ferhip_y4m_close(g_in);
    g_in = new ferhip_y4m{yuvinput, false, 0, 0, 0, 0, {}};
    if (y4m_parse_header(g_in)) {
        fprintf(stderr, "LoadY4MHeader: no W / H tokens in the stream header\n");
        delete g_in;
        g_in = nullptr;
        return;
    }
    inputWidth = g_in->inW;
    inputHeight = g_in->inH;
    // planes the library allocated for another picture size are dropped: ReadFromY4M copies W x H samples into them
    if (frame.L && (frame.Lwidth != g_in->W || frame.Lheight != g_in->H)) ferhip_legacy_frame_drop();
    frame.Lwidth = g_in->W;
    frame.Lheight = g_in->H;
    frame.Cwidth = frame.Lwidth >> 1;
    frame.Cheight = frame.Lheight >> 1;
}

extern "C" int ReadFromY4M(void)
{
    if (!g_in) return -1;
    frame_alloc_if_needed();
    std::vector<unsigned char> pic((size_t)g_in->W * g_in->H * 3 / 2);
    if (ferhip_y4m_read(g_in, pic.data()) != 0) {
        printf("End of stream found.\n");
        return -1;
    }
    const size_t ys = (size_t)g_in->W * g_in->H, cs = ys / 4;
    memcpy(frame.L, pic.data(), ys);
    memcpy(frame.C[0], pic.data() + ys, cs);
    memcpy(frame.C[1], pic.data() + ys + cs, cs);
    return 0;
}

static void write_frame_planes(FILE *f)
{
    fwrite(frame.L, 1, (size_t)frame.Lwidth * frame.Lheight, f);
    fwrite(frame.C[0], 1, (size_t)frame.Cwidth * frame.Cheight, f);
    fwrite(frame.C[1], 1, (size_t)frame.Cwidth * frame.Cheight, f);
}

extern "C" void writeToYUV(void)
{
    if (!yuvoutput || !frame.L) return;
    write_frame_planes(yuvoutput);
}

extern "C" void writeToY4M(void)
{
    if (!yuvoutput || !frame.L) return;
    if (!g_out_header_done) {
        ferhip_y4m_write_header(yuvoutput, frame.Lwidth, frame.Lheight);
        g_out_header_done = true;
    }
    fprintf(yuvoutput, "FRAME%c", 0x0a);
    write_frame_planes(yuvoutput);
}

// the reference keeps "first frame" flags in function statics (one file per process); a host that opens another
// file calls this first
extern "C" void ferhip_fileio_reset(void)
{
    if (g_in) ferhip_y4m_close(g_in);
    g_in = nullptr;
    g_out_header_done = false;
}
