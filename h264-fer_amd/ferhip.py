"""ctypes binding of include/ferhip.h (no torch types cross the C ABI)."""
import ctypes as C
import os
from pathlib import Path

import numpy as np

NAL_SLICE, NAL_IDR, NAL_AUTO = 1, 5, 0

BUF = dict(INTERP=1, FEAT=2, SORTPOS=3, KOLIKO=4, MBTYPE=5, MV=6, MVD=7, LEVELS=8, CBP=9, TC=10, I4MODE=11,
           CUR=12, REF=13, TIMING=14, ST2N=15, ST2=16, SPEC_STAT=17, MBSIZE=18)
TUNE_RESOLVE_WGS, TUNE_RESOLVE_GROUP, TUNE_SPECULATE, TUNE_OVERLAP_SORT = 1, 2, 3, 4
_BUF_DTYPE = {1: np.uint8, 2: np.uint16, 3: np.uint32, 4: np.int32, 5: np.int32, 6: np.int16, 7: np.int16,
              8: np.int16, 9: np.uint8, 10: np.uint8, 11: np.uint8, 12: np.uint8, 13: np.uint8, 14: np.int64, 15: np.int32, 16: np.int32,
              17: np.uint64, 18: np.int32}


class FerHipError(RuntimeError):
    pass


class Params(C.Structure):
    _fields_ = [("qp", C.c_int), ("basic", C.c_int), ("window", C.c_int), ("maxdiff", C.c_int),
                ("intra_every", C.c_int)]


def lib_path():
    return Path(__file__).resolve().parent / "libferhip.so"


_lib = None


def load_library():
    """Load libferhip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not p.exists():
        raise FerHipError(f"{p} missing: build it with __graft_entry__.build() (hipcc --offload-arch=gfx950)")
    lib = C.CDLL(str(p))
    vp, i, sz = C.c_void_p, C.c_int, C.c_size_t
    lib.ferhip_version.restype = C.c_char_p
    lib.ferhip_create.argtypes = [C.POINTER(vp), i, i, i, C.POINTER(Params)]
    lib.ferhip_destroy.argtypes = [vp]
    lib.ferhip_destroy.restype = None
    lib.ferhip_set_frames.argtypes = [vp, vp, i]
    lib.ferhip_mem_alloc.argtypes = [sz, i]
    lib.ferhip_mem_alloc.restype = vp
    lib.ferhip_mem_free.argtypes = [vp, i]
    lib.ferhip_mem_free.restype = None
    lib.ferhip_mem_copy.argtypes = [vp, vp, sz]
    lib.ferhip_set_reference.argtypes = [vp, vp]
    lib.ferhip_upload_frames.argtypes = [vp, vp]
    lib.ferhip_set_frames_uploaded.argtypes = [vp]
    lib.ferhip_encode_picture.argtypes = [vp, C.POINTER(i), vp, sz, C.POINTER(C.c_uint32)]
    lib.ferhip_encode_picture_dev.argtypes = [vp, C.POINTER(i), C.POINTER(vp), C.POINTER(sz), C.POINTER(vp)]
    lib.ferhip_select_nal_type.argtypes = [vp, C.POINTER(i)]
    lib.ferhip_copy_rbsp.argtypes = [vp, vp, sz, sz, vp, i]
    lib.ferhip_sync.argtypes = [vp]
    lib.ferhip_get_recon.argtypes = [vp, vp, i]
    lib.ferhip_write_sps.argtypes = [vp, vp, sz]
    lib.ferhip_write_sps.restype = sz
    lib.ferhip_write_pps.argtypes = [vp, vp, sz]
    lib.ferhip_write_pps.restype = sz
    lib.ferhip_write_nal.argtypes = [i, i, vp, sz, vp]
    lib.ferhip_write_nal.restype = sz
    lib.ferhip_encode_streams.argtypes = [vp, vp, i, vp, sz, C.POINTER(sz), vp]
    lib.ferhip_get_stats.argtypes = [vp, C.POINTER(i)]
    lib.ferhip_status.argtypes = [vp, C.POINTER(i)]
    lib.ferhip_fill_interpolated.argtypes = [vp]
    lib.ferhip_inter_encoding.argtypes = [vp]
    lib.ferhip_read_buffer.argtypes = [vp, i, vp, sz]
    lib.ferhip_read_buffer.restype = sz
    lib.ferhip_profile.argtypes = [vp, i]
    lib.ferhip_tune.argtypes = [vp, i, i]
    lib.ferhip_get_profile.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_long), i]
    lib.ferhip_decode_streams.argtypes = [C.POINTER(C.c_char_p), C.POINTER(sz), i, vp, i, C.POINTER(i), C.POINTER(i), C.POINTER(i)]
    lib.ferhip_decode_release.argtypes = []
    lib.ferhip_dec_create.argtypes = [C.POINTER(vp)]
    lib.ferhip_dec_destroy.argtypes = [vp]
    lib.ferhip_dec_destroy.restype = None
    lib.ferhip_dec_nal.argtypes = [vp, i, i, vp, sz, vp, C.POINTER(i), C.POINTER(i), C.POINTER(i)]
    lib.ferhip_y4m_open.argtypes = [C.POINTER(vp), C.c_char_p, C.POINTER(i), C.POINTER(i), C.POINTER(i), C.POINTER(i)]
    lib.ferhip_y4m_read.argtypes = [vp, vp]
    lib.ferhip_y4m_close.argtypes = [vp]
    lib.ferhip_y4m_close.restype = None
    lib.ferhip_y4m_write_header.argtypes = [vp, i, i]
    lib.ferhip_y4m_write_frame.argtypes = [vp, vp, i, i, i]
    lib.ferhip_forward_residual.argtypes = [i, vp, vp, i, sz]
    lib.ferhip_inverse_residual.argtypes = [i, vp, vp, i, sz]
    for n_ in ("forward_dc_luma_intra", "inverse_dc_luma_intra", "forward_dc_chroma", "inverse_dc_chroma"):
        getattr(lib, "ferhip_" + n_).argtypes = [i, vp, vp, sz]
    lib.ferhip_transform_scan.argtypes = [vp, vp, i, sz]
    lib.ferhip_transform_inverse_scan.argtypes = [vp, vp, sz]
    _lib = lib
    return lib


def _chk(rc, what):
    if rc != 0:
        raise FerHipError(f"{what} failed with code {rc}")


class DeviceBuffer:
    """HBM (or pinned host) memory through the library's own runtime (ferhip_mem_*), for callers without torch."""

    def __init__(self, nbytes, pinned=False):
        self.lib = load_library()
        self.nbytes, self.kind = nbytes, int(pinned)
        self.ptr = self.lib.ferhip_mem_alloc(nbytes, self.kind)
        if not self.ptr:
            raise FerHipError(f"ferhip_mem_alloc({nbytes}) failed")

    def upload(self, arr, offset=0):
        a = np.ascontiguousarray(arr)
        _chk(self.lib.ferhip_mem_copy(C.c_void_p(self.ptr + offset), a.ctypes.data, a.nbytes), "ferhip_mem_copy")

    def download(self, nbytes=None, offset=0, dtype=np.uint8):
        out = np.empty((nbytes or self.nbytes) // np.dtype(dtype).itemsize, dtype)
        _chk(self.lib.ferhip_mem_copy(out.ctypes.data, C.c_void_p(self.ptr + offset), out.nbytes), "ferhip_mem_copy")
        return out

    def free(self):
        if self.ptr:
            self.lib.ferhip_mem_free(C.c_void_p(self.ptr), self.kind)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class FerHip:
    """One encoder context = S independent streams of W x H pictures on one GPU.

    Mirrors Starter::PostaviParametre / PokreniKoder / NastaviKoder / DohvatiStatistiku.
    """

    def __init__(self, width, height, nstreams=1, qp=12, window=16, maxdiff=3, intra_every=30, basic=0):
        self.lib = load_library()
        self.W, self.H, self.S = width, height, nstreams
        self.nmb = (width // 16) * (height // 16)
        self.fsz = width * height * 3 // 2
        self.params = Params(qp, basic, window, maxdiff, intra_every)
        self.ctx = C.c_void_p()
        _chk(self.lib.ferhip_create(C.byref(self.ctx), width, height, nstreams, C.byref(self.params)), "ferhip_create")

    def close(self):
        if self.ctx:
            self.lib.ferhip_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- pictures
    def set_frames(self, frames):
        a = np.ascontiguousarray(frames, dtype=np.uint8).reshape(self.S, self.fsz)
        _chk(self.lib.ferhip_set_frames(self.ctx, a.ctypes.data, 1), "ferhip_set_frames")

    def set_frames_device(self, dptr):
        _chk(self.lib.ferhip_set_frames(self.ctx, C.c_void_p(int(dptr)), 0), "ferhip_set_frames(dev)")

    def upload_frames(self, host_ptr):
        """start the asynchronous H2D copy of the next pictures (pinned host memory, [S][fsz])"""
        _chk(self.lib.ferhip_upload_frames(self.ctx, C.c_void_p(int(host_ptr))), "ferhip_upload_frames")

    def set_frames_uploaded(self):
        _chk(self.lib.ferhip_set_frames_uploaded(self.ctx), "ferhip_set_frames_uploaded")

    def set_reference(self, frames):
        a = np.ascontiguousarray(frames, dtype=np.uint8).reshape(self.S, self.fsz)
        _chk(self.lib.ferhip_set_reference(self.ctx, a.ctypes.data), "ferhip_set_reference")

    def encode_picture(self, nal_types=None):
        """RBSP_encode for one picture of every stream -> (list of rbsp bytes, nal types)."""
        nt = (C.c_int * self.S)(*([NAL_AUTO] * self.S if nal_types is None else nal_types))
        stride = self.nmb * 1024 + 4096
        buf = np.empty((self.S, stride), np.uint8)
        ln = (C.c_uint32 * self.S)()
        _chk(self.lib.ferhip_encode_picture(self.ctx, nt, buf.ctypes.data, stride, ln), "ferhip_encode_picture")
        return [bytes(buf[s, : ln[s]]) for s in range(self.S)], list(nt)

    def encode_picture_device(self, nal_types=None):
        nt = (C.c_int * self.S)(*([NAL_AUTO] * self.S if nal_types is None else nal_types))
        p, st, pl = C.c_void_p(), C.c_size_t(), C.c_void_p()
        _chk(self.lib.ferhip_encode_picture_dev(self.ctx, nt, C.byref(p), C.byref(st), C.byref(pl)),
             "ferhip_encode_picture_dev")
        return p.value, st.value, pl.value, list(nt)

    def copy_rbsp_device(self, dst_ptr, len_ptr, stride=None, nbytes=None):
        """RBSP + lengths of the last picture -> device buffers, on the library's stream (asynchronous)."""
        stride = stride or (self.nmb * 1024 + 4096)
        _chk(self.lib.ferhip_copy_rbsp(self.ctx, C.c_void_p(int(dst_ptr)), stride, nbytes or stride, C.c_void_p(int(len_ptr)), 0),
             "ferhip_copy_rbsp")

    def copy_rbsp_host(self, dst, lens, nbytes):
        """... -> host arrays dst [S][stride] uint8 (pinned for asynchrony), lens [S] uint32; sync() before reading"""
        _chk(self.lib.ferhip_copy_rbsp(self.ctx, C.c_void_p(int(dst.ctypes.data if hasattr(dst, "ctypes") else dst.data_ptr())),
                                       int(dst.strides[0] if hasattr(dst, "strides") else dst.stride(0)), nbytes,
                                       C.c_void_p(int(lens.ctypes.data if hasattr(lens, "ctypes") else lens.data_ptr())), 1),
             "ferhip_copy_rbsp")

    def sync(self):
        _chk(self.lib.ferhip_sync(self.ctx), "ferhip_sync")

    def get_recon(self):
        out = np.empty((self.S, self.fsz), np.uint8)
        _chk(self.lib.ferhip_get_recon(self.ctx, out.ctypes.data, 1), "ferhip_get_recon")
        return out

    def encode_streams(self, frames, want_recon=False):
        """frames: [T][S][fsz] uint8 -> (list of Annex-B byte strings, recon or None)."""
        a = np.ascontiguousarray(frames, dtype=np.uint8)
        T = a.size // (self.S * self.fsz)
        a = a.reshape(T, self.S, self.fsz)
        stride = 64 + T * (self.nmb * 1024 + 4096) * 3 // 2
        out = np.empty((self.S, stride), np.uint8)
        ln = (C.c_size_t * self.S)()
        rec = np.empty((T, self.S, self.fsz), np.uint8) if want_recon else None
        _chk(self.lib.ferhip_encode_streams(self.ctx, a.ctypes.data, T, out.ctypes.data, stride, ln,
                                            rec.ctypes.data if want_recon else None), "ferhip_encode_streams")
        return [bytes(out[s, : ln[s]]) for s in range(self.S)], rec

    def sps_pps(self):
        b = np.empty(64, np.uint8)
        o = np.empty(128, np.uint8)
        n = self.lib.ferhip_write_sps(self.ctx, b.ctypes.data, 64)
        m = self.lib.ferhip_write_nal(1, 7, b.ctypes.data, n, o.ctypes.data)
        sps = bytes(o[:m])
        n = self.lib.ferhip_write_pps(self.ctx, b.ctypes.data, 64)
        m = self.lib.ferhip_write_nal(1, 8, b.ctypes.data, n, o.ctypes.data)
        return sps, bytes(o[:m])

    def write_nal(self, nal_type, rbsp):
        r = np.frombuffer(rbsp, np.uint8)
        o = np.empty(len(rbsp) * 3 // 2 + 16, np.uint8)
        m = self.lib.ferhip_write_nal(1, nal_type, r.ctypes.data, len(rbsp), o.ctypes.data)
        return bytes(o[:m])

    # --- stage entry points / state read-back
    def fill_interpolated(self):
        _chk(self.lib.ferhip_fill_interpolated(self.ctx), "ferhip_fill_interpolated")

    def inter_encoding(self):
        _chk(self.lib.ferhip_inter_encoding(self.ctx), "ferhip_inter_encoding")

    def read(self, name):
        which = BUF[name]
        n = self.nmb * self.S
        px = self.W * self.H * self.S
        count = {1: px * 16, 2: px * 96, 3: px, 4: 16385 * self.S, 5: n, 6: n * 8, 7: n * 8, 8: n * 400, 9: n * 2,
                 10: n * 24, 11: n * 16, 12: self.fsz * self.S, 13: self.fsz * self.S, 14: 64, 15: n * 4, 16: n * 4 * 384 * 2, 17: 8, 18: n * 2}[which]
        out = np.empty(count, _BUF_DTYPE[which])
        got = self.lib.ferhip_read_buffer(self.ctx, which, out.ctypes.data, out.nbytes)
        if got != out.nbytes:
            raise FerHipError(f"ferhip_read_buffer({name}) returned {got}, expected {out.nbytes}")
        return out

    PHASES = ("interp", "me_pre", "me_resolve", "p_resid", "intra", "cavlc", "frame_sad", "me_spec", "sort", "me_walk",
              "sort_keys", "sort_finish")
    NPHASE = 12

    def tune(self, key, value):
        _chk(self.lib.ferhip_tune(self.ctx, key, value), "ferhip_tune")

    def profile(self, enable=True):
        _chk(self.lib.ferhip_profile(self.ctx, int(enable)), "ferhip_profile")

    def get_profile(self, reset=True):
        """{phase: (milliseconds, launches)} measured with HIP events on the launch stream."""
        ms = (C.c_double * self.NPHASE)()
        ln = (C.c_long * self.NPHASE)()
        _chk(self.lib.ferhip_get_profile(self.ctx, ms, ln, int(reset)), "ferhip_get_profile")
        return {n: (ms[k], ln[k]) for k, n in enumerate(self.PHASES)}

    def stats(self):
        a = (C.c_int * (5 * self.S))()
        _chk(self.lib.ferhip_get_stats(self.ctx, a), "ferhip_get_stats")
        return np.array(a).reshape(self.S, 5)

    def status(self):
        a = (C.c_int * self.S)()
        _chk(self.lib.ferhip_status(self.ctx, a), "ferhip_status")
        return list(a)


def forward_residual(qp, blocks, keep_dc=False):
    """forwardResidual of F/quantizationTransform.h on n 4x4 int32 blocks (device)."""
    lib = load_library()
    a = np.ascontiguousarray(blocks, np.int32).reshape(-1, 16)
    out = np.empty_like(a)
    _chk(lib.ferhip_forward_residual(qp, a.ctypes.data, out.ctypes.data, int(keep_dc), a.shape[0]),
         "ferhip_forward_residual")
    return out


def inverse_residual(qp, blocks, keep_dc=False):
    lib = load_library()
    a = np.ascontiguousarray(blocks, np.int32).reshape(-1, 16)
    out = np.empty_like(a)
    _chk(lib.ferhip_inverse_residual(qp, a.ctypes.data, out.ctypes.data, int(keep_dc), a.shape[0]),
         "ferhip_inverse_residual")
    return out


class MbJob(C.Structure):
    """ferhip_mb_job of include/ferhip.h"""
    _fields_ = [("op", C.c_int32), ("cls", C.c_int32), ("qp", C.c_int32), ("qpc", C.c_int32), ("reconstruct", C.c_int32), ("blk", C.c_int32),
                ("srcY", C.c_int32 * 256), ("srcCb", C.c_int32 * 64), ("srcCr", C.c_int32 * 64),
                ("predY", C.c_int32 * 256), ("predCb", C.c_int32 * 64), ("predCr", C.c_int32 * 64),
                ("lumaLevel", C.c_int32 * 256), ("dc16", C.c_int32 * 16), ("ac16", C.c_int32 * 256), ("cdc", C.c_int32 * 8), ("cac", C.c_int32 * 128)]


class MbResult(C.Structure):
    _fields_ = [("lumaLevel", C.c_int32 * 256), ("dc16", C.c_int32 * 16), ("ac16", C.c_int32 * 256), ("cdc", C.c_int32 * 8), ("cac", C.c_int32 * 128),
                ("recY", C.c_int32 * 256), ("recCb", C.c_int32 * 64), ("recCr", C.c_int32 * 64)]


MBU_QT, MBU_DEC4, MBU_DEC16, MBU_DECC, MBU_SKIP = range(5)


def mb_unit(jobs):
    """ferhip_mb_unit: jobs = list of dicts (fields of ferhip_mb_job, arrays as numpy) -> list of dicts of numpy arrays"""
    lib = load_library()
    n = len(jobs)
    J = (MbJob * n)()
    R = (MbResult * n)()
    for k, j in enumerate(jobs):
        for name, val in j.items():
            if isinstance(val, (int, np.integer)):
                setattr(J[k], name, int(val))
            else:
                a = np.ascontiguousarray(val, np.int32).reshape(-1)
                C.memmove(getattr(J[k], name), a.ctypes.data, a.nbytes)
    lib.ferhip_mb_unit.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    _chk(lib.ferhip_mb_unit(J, R, n), "ferhip_mb_unit")
    return [{f: np.frombuffer(getattr(R[k], f), np.int32).copy() for f, _ in MbResult._fields_} for k in range(n)]


def cavlc_blocks(coef, nC, max_num_coeff):
    """ferhip_cavlc_blocks -> (bits [n][64] uint8, nbits [n], total_coeff [n])"""
    lib = load_library()
    c = np.ascontiguousarray(coef, np.int32).reshape(-1, 16)
    n = c.shape[0]
    nc = np.ascontiguousarray(nC, np.int32)
    mx = np.ascontiguousarray(max_num_coeff, np.int32)
    bits = np.zeros((n, 64), np.uint8)
    nb = np.zeros(n, np.uint32)
    tc = np.zeros(n, np.int32)
    lib.ferhip_cavlc_blocks.argtypes = [C.c_void_p] * 3 + [C.c_size_t] + [C.c_void_p] * 3
    _chk(lib.ferhip_cavlc_blocks(c.ctypes.data, nc.ctypes.data, mx.ctypes.data, n, bits.ctypes.data, nb.ctypes.data, tc.ctypes.data), "ferhip_cavlc_blocks")
    return bits, nb, tc


def mc_sub_mb_parts(ref_i420, width, height, desc):
    """ferhip_mc_sub_mb_parts: desc [n][5] = mb, subMbIdx, subMbPartIdx, mvx, mvy -> (predL [n][4][4], predCb [n][2][2], predCr [n][2][2])"""
    lib = load_library()
    r = np.ascontiguousarray(ref_i420, np.uint8)
    d = np.ascontiguousarray(desc, np.int32).reshape(-1, 5)
    n = d.shape[0]
    pl, pb, pr = np.zeros((n, 4, 4), np.int32), np.zeros((n, 2, 2), np.int32), np.zeros((n, 2, 2), np.int32)
    lib.ferhip_mc_sub_mb_parts.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    _chk(lib.ferhip_mc_sub_mb_parts(r.ctypes.data, width, height, d.ctypes.data, n, pl.ctypes.data, pb.ctypes.data, pr.ctypes.data), "ferhip_mc_sub_mb_parts")
    return pl, pb, pr


def block_op(name, blocks, qp=0, flag=None):
    """The per-block entry points of F/quantizationTransform.h / F/scaleTransform.h on n 16-int32 records (device):
    forward_dc_luma_intra, inverse_dc_luma_intra, forward_dc_chroma, inverse_dc_chroma, transform_scan,
    transform_inverse_scan."""
    lib = load_library()
    a = np.ascontiguousarray(blocks, np.int32).reshape(-1, 16)
    out = np.empty_like(a)
    f = getattr(lib, "ferhip_" + name)
    if name == "transform_scan":
        rc = f(a.ctypes.data, out.ctypes.data, int(bool(flag)), a.shape[0])
    elif name == "transform_inverse_scan":
        rc = f(a.ctypes.data, out.ctypes.data, a.shape[0])
    else:
        rc = f(qp, a.ctypes.data, out.ctypes.data, a.shape[0])
    _chk(rc, "ferhip_" + name)
    return out


class Decoder:
    """Streaming decoder for one stream: RBSP_decode(NALunit), NAL unit by NAL unit (ferhip_dec_*)."""

    def __init__(self):
        self.lib = load_library()
        self.h = C.c_void_p()
        _chk(self.lib.ferhip_dec_create(C.byref(self.h)), "ferhip_dec_create")
        self.W = self.H = 0

    def nal(self, nal_unit_type, nal_ref_idc, rbsp):
        """-> decoded picture (uint8 [W*H*3/2]) for slice NAL units, else None"""
        r = np.frombuffer(rbsp, np.uint8)
        got, W, H = C.c_int(), C.c_int(), C.c_int()
        pic = np.empty(self.W * self.H * 3 // 2, np.uint8) if nal_unit_type in (1, 5) else None
        _chk(self.lib.ferhip_dec_nal(self.h, nal_unit_type, nal_ref_idc, r.ctypes.data, len(rbsp),
                                     pic.ctypes.data if pic is not None else None, C.byref(got), C.byref(W), C.byref(H)),
             "ferhip_dec_nal")
        self.W, self.H = W.value, H.value
        return pic if got.value else None

    def close(self):
        if self.h:
            self.lib.ferhip_dec_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def unescape_nal(nal):
    """Annex-B NAL unit (start code + header + payload) -> (type, ref_idc, rbsp) like getNAL (F/nal.cpp:68-223)"""
    body = nal[4:]
    out = bytearray()
    z = 0
    for b in body[1:]:
        if z >= 2 and b == 3:
            z = 0
            continue
        out.append(b)
        z = z + 1 if b == 0 else 0
    return body[0] & 31, (body[0] >> 5) & 3, bytes(out)


class Y4MReader:
    """LoadY4MHeader / ReadFromY4M (F/fileIO.cpp:228-346): centre crop to multiples of 16."""

    def __init__(self, path):
        self.lib = load_library()
        self.h = C.c_void_p()
        iw, ih, w, h = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _chk(self.lib.ferhip_y4m_open(C.byref(self.h), str(path).encode(), C.byref(iw), C.byref(ih), C.byref(w), C.byref(h)),
             "ferhip_y4m_open")
        self.in_size, self.W, self.H = (iw.value, ih.value), w.value, h.value

    def read(self):
        pic = np.empty(self.W * self.H * 3 // 2, np.uint8)
        rc = self.lib.ferhip_y4m_read(self.h, pic.ctypes.data)
        if rc == 1:
            return None
        _chk(rc, "ferhip_y4m_read")
        return pic

    def close(self):
        if self.h:
            self.lib.ferhip_y4m_close(self.h)
            self.h = C.c_void_p()


def decode_streams(streams, max_pictures, want_pictures=True):
    """decode() for a list of Annex-B byte strings of equal picture size -> (recon [T][S][fsz], pictures, W, H).
    want_pictures=False leaves the decoded pictures on the device (recon is None): what tools/bench_decode.py times."""
    lib = load_library()
    S = len(streams)
    arr = (C.c_char_p * S)(*streams)
    lens = (C.c_size_t * S)(*[len(s) for s in streams])
    pics = (C.c_int * S)()
    W, H = C.c_int(), C.c_int()
    if max_pictures <= 0:
        raise FerHipError("decode_streams: max_pictures must be positive (it sizes the output)")
    out = None
    if want_pictures:  # the output is sized from the first SPS of stream 0 (found by start code, not by splitting the stream)
        at = 0
        while True:
            at = streams[0].find(b"\x00\x00\x01", at)
            if at < 0 or at + 3 >= len(streams[0]):
                raise FerHipError("decode_streams: no sequence parameter set in stream 0")
            at += 3
            if (streams[0][at] & 31) == 7:
                break
        end = streams[0].find(b"\x00\x00\x01", at)
        w, h = _sps_size(streams[0][at + 1:end if end >= 0 else len(streams[0])].replace(b"\x00\x00\x03", b"\x00\x00"))
        out = np.empty((max_pictures, S, w * h * 3 // 2), np.uint8)
    _chk(lib.ferhip_decode_streams(arr, lens, S, out.ctypes.data if want_pictures else None, max_pictures, pics,
                                   C.byref(W), C.byref(H)), "ferhip_decode_streams")
    return out, list(pics), W.value, H.value


def _sps_size(rbsp):
    """width/height from an SPS RBSP (host-side helper for buffer sizing only)."""
    bits = "".join(f"{b:08b}" for b in rbsp)
    pos = [24]

    def ue():
        z = 0
        while bits[pos[0]] == "0":
            z += 1
            pos[0] += 1
        pos[0] += 1
        v = int(bits[pos[0]:pos[0] + z] or "0", 2)
        pos[0] += z
        return (1 << z) - 1 + v

    ue()
    ue()
    poc = ue()
    if poc == 0:
        ue()
    ue()
    pos[0] += 1
    wmb = ue() + 1
    hmu = ue() + 1
    fmo = int(bits[pos[0]])
    return wmb * 16, (2 - fmo) * hmu * 16
