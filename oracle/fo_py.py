"""ctypes binding of the CPU oracle (oracle/libfo.so).  TEST INFRASTRUCTURE ONLY: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent


def build():
    subprocess.run(["make", "-s", "-C", str(HERE), "libfo.so", "fo_cli", "ref"], check=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        p = HERE / "libfo.so"
        if not p.exists():
            build()
        L = C.CDLL(str(p))
        vp, i, sz = C.c_void_p, C.c_int, C.c_size_t
        L.fo_create.restype = vp
        L.fo_create.argtypes = [i, i]
        L.fo_destroy.argtypes = [vp]
        L.fo_set_params.argtypes = [vp, i, i, i, i, i]
        L.fo_encode_stream.restype = sz
        L.fo_encode_stream.argtypes = [vp, vp, i, vp, sz, vp]
        L.fo_encode_slice.restype = sz
        L.fo_encode_slice.argtypes = [vp, i, vp, sz]
        L.fo_fill_interpolated.argtypes = [vp]
        L.fo_gen_frame.argtypes = [i, i, i, C.c_uint64, i, vp, vp, vp]
        L.fo_decode_stream.argtypes = [vp, sz, vp, vp, vp]
        for n in ("fo_dbg_plane", "fo_dbg_interp", "fo_dbg_sorted", "fo_dbg_mv", "fo_dbg_cbp"):
            getattr(L, n).restype = vp
            getattr(L, n).argtypes = [vp, i]
        L.fo_dbg_kar.restype = vp
        L.fo_dbg_kar.argtypes = [vp, i, i]
        for n in ("fo_dbg_koliko", "fo_dbg_mb_type", "fo_dbg_tc_l", "fo_dbg_tc_c", "fo_dbg_i4mode", "fo_dbg_type_count"):
            getattr(L, n).restype = vp
            getattr(L, n).argtypes = [vp]
        L.fo_dbg_set_dpb.argtypes = [vp, vp, vp, vp]
        L.fo_dbg_set_frame.argtypes = [vp, vp, vp, vp]
        L.fo_forwardResidual.argtypes = [i, vp, vp, i]
        L.fo_inverseResidual.argtypes = [i, vp, vp, i]
        for n_ in ("fo_forwardDCLumaIntra", "fo_forwardDCChroma", "fo_inverseDCLumaIntra", "fo_inverseDCChroma"):
            getattr(L, n_).argtypes = [i, vp, vp]
        L.fo_scan.argtypes = [vp, vp, i]
        L.fo_invscan.argtypes = [vp, vp]
        L.fo_cavlc_encode_block.restype = C.c_uint
        L.fo_cavlc_encode_block.argtypes = [vp, vp, i, i, vp]
        L.fo_bw_init.argtypes = [vp, vp, sz]
        L.fo_dbg_mbsize.restype = vp
        L.fo_dbg_mbsize.argtypes = [vp]
        L.fo_dbg_levels.restype = vp
        L.fo_dbg_levels.argtypes = [vp]
        L.fo_dbg_set_mb.argtypes = [vp, i, i, i, i]
        L.fo_dbg_set_mv.argtypes = [vp, i, i, i, i, i]
        L.fo_mc_sub.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, i, i]
        L.fo_quantizationTransform.argtypes = [vp, vp, vp, vp, i]
        L.fo_transformDecoding4x4Luma.argtypes = [vp, vp, vp, i, i]
        L.fo_transformDecoding16x16Luma.argtypes = [vp, vp, vp, vp, i]
        L.fo_transformDecodingChroma.argtypes = [vp, vp, vp, vp, i, i]
        L.fo_transformDecodingPSkip.argtypes = [vp, vp, vp, vp, i]
        _lib = L
    return _lib


def _arr(ptr, n, dt):
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(n,))


class Oracle:
    def __init__(self, W, H, qp=12, window=16, maxdiff=3, intra_every=30, basic=0):
        self.L = lib()
        self.W, self.H = W, H
        self.nmb = (W // 16) * (H // 16)
        self.fsz = W * H * 3 // 2
        self.c = C.c_void_p(self.L.fo_create(W, H))
        self.L.fo_set_params(self.c, qp, basic, window, maxdiff, intra_every)

    def close(self):
        if self.c:
            self.L.fo_destroy(self.c)
            self.c = None

    def encode_stream(self, frames):
        a = np.ascontiguousarray(frames, np.uint8).reshape(-1, self.fsz)
        T = a.shape[0]
        out = np.empty(T * self.fsz * 2 + 65536, np.uint8)
        rec = np.empty((T, self.fsz), np.uint8)
        n = self.L.fo_encode_stream(self.c, a.ctypes.data, T, out.ctypes.data, out.size, rec.ctypes.data)
        return bytes(out[:n]), rec

    def _split(self, f):
        f = np.ascontiguousarray(f, np.uint8)
        ys = self.W * self.H
        return f[:ys], f[ys: ys + ys // 4], f[ys + ys // 4:]

    def set_frame(self, f):
        y, u, v = self._split(f)
        self.L.fo_dbg_set_frame(self.c, y.ctypes.data, u.ctypes.data, v.ctypes.data)

    def set_dpb(self, f):
        y, u, v = self._split(f)
        self.L.fo_dbg_set_dpb(self.c, y.ctypes.data, u.ctypes.data, v.ctypes.data)

    def encode_slice(self, nal_type):
        out = np.empty(self.fsz * 2 + 65536, np.uint8)
        n = self.L.fo_encode_slice(self.c, nal_type, out.ctypes.data, out.size)
        return bytes(out[:n])

    def fill_interpolated(self):
        self.L.fo_fill_interpolated(self.c)

    def frame(self):
        ys = self.W * self.H
        return np.concatenate([_arr(self.L.fo_dbg_plane(self.c, k), ys if k == 0 else ys // 4, np.uint8).copy()
                               for k in range(3)])

    def interp(self, f):
        return _arr(self.L.fo_dbg_interp(self.c, f), self.W * self.H, np.uint8).reshape(self.H, self.W).copy()

    def kar(self, k, f):
        a = _arr(self.L.fo_dbg_kar(self.c, k, f), (self.W + 8) * (self.H + 8), np.int32).reshape(self.H + 8, self.W + 8)
        return a[: self.H, : self.W].copy()

    def sorted(self, k):
        return _arr(self.L.fo_dbg_sorted(self.c, k), self.W * self.H, np.int32).copy()

    def koliko(self):
        return _arr(self.L.fo_dbg_koliko(self.c), 16385, np.int32).copy()

    def mb_type(self):
        return _arr(self.L.fo_dbg_mb_type(self.c), self.nmb, np.int32).copy()

    def mv(self):
        x = _arr(self.L.fo_dbg_mv(self.c, 0), self.nmb * 16, np.int32).reshape(self.nmb, 4, 4)[:, :, 0]
        y = _arr(self.L.fo_dbg_mv(self.c, 1), self.nmb * 16, np.int32).reshape(self.nmb, 4, 4)[:, :, 0]
        return np.stack([x, y], -1).copy()

    def stats(self):
        """brojTipova[5]: P_Skip, 16x16, 16x8, 8x16, 8x8 macroblocks so far"""
        return _arr(self.L.fo_dbg_type_count(self.c), 5, np.int32).copy()

    def mbsize(self):
        """coded_mb_size of the Intra16x16 / Intra4x4 alternative of every macroblock of the last I picture"""
        return _arr(self.L.fo_dbg_mbsize(self.c), self.nmb * 2, np.int32).reshape(self.nmb, 2).copy()

    # ---- one macroblock at a time (per-macroblock KATs)
    LEVELS = (("lumaLevel", 256), ("dc16", 16), ("ac16", 256), ("cdc", 8), ("cac", 128))

    def set_mb(self, cur, mb_type, slice_type, qp):
        self.L.fo_dbg_set_mb(self.c, cur, mb_type, slice_type, qp)

    def levels(self):
        a = _arr(self.L.fo_dbg_levels(self.c), 664, np.int32)
        out, o = {}, 0
        for n, k in self.LEVELS:
            out[n] = a[o:o + k].copy()
            o += k
        return out

    def set_levels(self, **kw):
        a = _arr(self.L.fo_dbg_levels(self.c), 664, np.int32)
        o = 0
        for n, k in self.LEVELS:
            if n in kw:
                a[o:o + k] = np.asarray(kw[n], np.int32).reshape(-1)
            o += k

    def quantization_transform(self, predL, predCb, predCr, reconstruct):
        p = [np.ascontiguousarray(x, np.int32) for x in (predL, predCb, predCr)]
        self.L.fo_quantizationTransform(self.c, p[0].ctypes.data, p[1].ctypes.data, p[2].ctypes.data, int(reconstruct))

    def mc_sub(self, mb, sub, part, mvx, mvy):
        """MotionCompensateSubMBPart against the context's dpb -> (4x4 luma, 2x2 Cb, 2x2 Cr)"""
        self.L.fo_dbg_set_mv(self.c, mb, sub, part, mvx, mvy)
        pl, pr, pb = np.zeros((16, 16), np.int32), np.zeros((8, 8), np.int32), np.zeros((8, 8), np.int32)
        self.L.fo_mc_sub(self.c, pl.ctypes.data, pr.ctypes.data, pb.ctypes.data, self.L.fo_dbg_plane(self.c, 3), self.L.fo_dbg_plane(self.c, 4),
                         self.L.fo_dbg_plane(self.c, 5), mb, sub, part)
        oy, ox = ((sub & 2) << 2) + ((part & 2) << 1), ((sub & 1) << 3) + ((part & 1) << 2)
        return pl[oy:oy + 4, ox:ox + 4].copy(), pb[oy // 2:oy // 2 + 2, ox // 2:ox // 2 + 2].copy(), pr[oy // 2:oy // 2 + 2, ox // 2:ox // 2 + 2].copy()

    def cbp(self):
        return np.stack([_arr(self.L.fo_dbg_cbp(self.c, k), self.nmb, np.int32) for k in range(2)], -1).copy()


class _BW(C.Structure):
    _fields_ = [("buf", C.c_void_p), ("cap", C.c_size_t), ("nbits", C.c_size_t)]


def cavlc_encode_block(coef, max_num_coeff, nC):
    """residual_block_cavlc_write of one block with nC given -> (bytes MSB first, bits, TotalCoeff)"""
    L = lib()
    buf = np.zeros(128, np.uint8)
    w = _BW(buf.ctypes.data, buf.size, 0)
    c = np.ascontiguousarray(coef, np.int32)
    tc = C.c_int(0)
    n = L.fo_cavlc_encode_block(C.byref(w), c.ctypes.data, int(max_num_coeff), int(nC), C.byref(tc))
    return buf[:(n + 7) // 8].tobytes(), int(n), tc.value


def forward_residual(qp, blocks, keep_dc=False):
    L = lib()
    a = np.ascontiguousarray(blocks, np.int32).reshape(-1, 16)
    out = np.empty_like(a)
    for k in range(a.shape[0]):
        L.fo_forwardResidual(qp, a[k].ctypes.data, out[k].ctypes.data, int(keep_dc))
    return out


def inverse_residual(qp, blocks, keep_dc=False):
    L = lib()
    a = np.ascontiguousarray(blocks, np.int32).reshape(-1, 16)
    out = np.empty_like(a)
    for k in range(a.shape[0]):
        L.fo_inverseResidual(qp, a[k].ctypes.data, out[k].ctypes.data, int(keep_dc))
    return out


def gen_frame(W, H, t, seed=1234, noise=2):
    L = lib()
    b = np.empty(W * H * 3 // 2, np.uint8)
    ys = W * H
    L.fo_gen_frame(W, H, t, seed, noise, b.ctypes.data, b[ys:].ctypes.data, b[ys + ys // 4:].ctypes.data)
    return b


def decode_stream_md5(stream):
    """Decode an Annex-B stream; returns (pictures, list of I420 pictures)."""
    import hashlib
    L = lib()
    frames = []
    CB = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)
    state = {}

    def cb(cptr, user):
        if "W" not in state:
            # first two ints of fo_ctx are W, H
            wh = _arr(cptr, 2, np.int32)
            state["W"], state["H"] = int(wh[0]), int(wh[1])
        W, H = state["W"], state["H"]
        ys = W * H
        frames.append(np.concatenate([_arr(L.fo_dbg_plane(C.c_void_p(cptr), k), ys if k == 0 else ys // 4, np.uint8).copy()
                                      for k in range(3)]))

    a = np.frombuffer(stream, np.uint8)
    n = L.fo_decode_stream(a.ctypes.data, a.size, CB(cb), None, None)
    return n, frames, state


def block_op(name, blocks, qp=0, flag=None):
    """oracle twin of h264_fer_amd.ferhip.block_op (16-int32 records; chroma DC uses slots 0..3)"""
    L = lib()
    a = np.ascontiguousarray(blocks, np.int32).reshape(-1, 16)
    out = np.zeros_like(a)
    for k in range(a.shape[0]):
        if name == "forward_dc_luma_intra":
            L.fo_forwardDCLumaIntra(qp, a[k].ctypes.data, out[k].ctypes.data)
        elif name == "inverse_dc_luma_intra":
            L.fo_inverseDCLumaIntra(qp, a[k].ctypes.data, out[k].ctypes.data)
        elif name == "forward_dc_chroma":
            i4, o4 = a[k, :4].copy(), np.zeros(4, np.int32)
            L.fo_forwardDCChroma(qp, i4.ctypes.data, o4.ctypes.data)
            out[k, :4] = o4
        elif name == "inverse_dc_chroma":
            i4, o4 = a[k, :4].copy(), np.zeros(4, np.int32)
            L.fo_inverseDCChroma(qp, i4.ctypes.data, o4.ctypes.data)
            out[k, :4] = o4
        elif name == "transform_scan":
            L.fo_scan(a[k].ctypes.data, out[k].ctypes.data, int(bool(flag)))
        else:
            L.fo_invscan(a[k].ctypes.data, out[k].ctypes.data)
    return out
