"""GPU parity at BASELINE.json's configurations, through the C ABI (libferhip.so).

Small cases compare byte for byte with committed golden streams; the full-size cases compare
with the oracle run on the same seeded input and add size-independent properties: the oracle
decoder reproduces the GPU reconstruction from the GPU bitstream, streams in a batch do not
influence each other, and repeated runs are identical.
"""
import hashlib
import json
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / "golden"


@pytest.mark.parametrize("case", ["qcif_i_2f_qp12", "qcif_ippp_4f_qp12_w16", "qcif_ippp_4f_qp28_w32", "qcif_skip_5f_qp12"])
def test_gpu_matches_committed_goldens(pkg, case):
    m = json.loads((GOLD / "goldens.json").read_text())[case]
    frames = np.stack([pkg.gen_frame(m["W"], m["H"], 0 if m.get("static") else t, m["seed"], m["noise"]) for t in range(m["T"])])
    g = pkg.FerHip(m["W"], m["H"], 1, qp=m["qp"], window=m["window"], maxdiff=m["maxdiff"], intra_every=m["intra_every"])
    streams, rec = g.encode_streams(frames[:, None], want_recon=True)
    assert g.status() == [0]
    assert streams[0] == (GOLD / f"{case}.264").read_bytes()
    assert hashlib.sha256(rec.tobytes()).hexdigest() == m["recon_sha256"]
    if "skip" in case:
        assert g.stats()[0][0] > 0, "the case is meant to contain P_Skip macroblocks"


def _check_against_oracle(pkg, fo, W, H, T, S, qp, window, intra_every, noise=2, check_streams=(0,)):
    frames = np.stack([np.stack([pkg.gen_frame(W, H, t, 1234 + s, noise) for s in range(S)]) for t in range(T)])
    g = pkg.FerHip(W, H, S, qp=qp, window=window, maxdiff=3, intra_every=intra_every)
    streams, rec = g.encode_streams(frames, want_recon=True)
    assert g.status() == [0] * S
    counts = g.stats()
    for s in check_streams:
        o = fo.Oracle(W, H, qp=qp, window=window, maxdiff=3, intra_every=intra_every)
        ref, ref_rec = o.encode_stream(frames[:, s])
        ref_counts = o.stats()
        o.close()
        assert np.array_equal(rec[:, s], ref_rec), f"recon of stream {s}"
        assert streams[s] == ref, f"bitstream of stream {s}"
        assert list(counts[s]) == list(ref_counts), f"brojTipova of stream {s}"
    # property: the oracle DEcoder turns the GPU bitstream back into the GPU reconstruction
    n, dec, _ = fo.decode_stream_md5(streams[-1])
    assert n == T
    ys = W * H
    for t in range(T):
        assert np.array_equal(dec[t][:ys], rec[t, S - 1][:ys])
    g.close()
    return frames, streams, rec


_ORACLE_CACHE = {}


def _oracle_stream(fo, frames, key, W, H, **kw):
    if key not in _ORACLE_CACHE:
        o = fo.Oracle(W, H, **kw)
        ref, ref_rec = o.encode_stream(frames)
        cnt = [int(v) for v in o.stats()]
        o.close()
        _ORACLE_CACHE[key] = (ref, ref_rec, cnt)
    return _ORACLE_CACHE[key]


@pytest.mark.parametrize("S,window,wgs,group,spec", [(17, 16, 1, 4, 1), (17, 32, 3, 4, 1), (24, 16, 7, 1, 1), (24, 32, 8, 3, 1),
                                                      (24, 16, 64, 4, 0), (24, 32, 3072, 100, 1), (17, 32, 3, 2, 0)])
def test_many_streams_take_the_per_xcd_queues(pkg, fo, S, window, wgs, group, spec):
    """S >= 16 switches the motion chain to its eight ticket queues (one per XCD; what bench.py runs).  Every stream is
    compared with the oracle -- bitstream, reconstruction, brojTipova -- under launch shapes that cannot cover the eight
    XCDs (1, 3, 7 workgroups: rows are then taken by workgroups that move from queue to queue), with short last stream
    groups, and with the speculative pre-pass on and off.  Every fifth stream is a still sequence (P_Skip macroblocks)."""
    W, H, T = 176, 144, 4
    def src(s, t):
        return pkg.gen_frame(W, H, 0 if s % 5 == 4 else t, 1234 + s, 0 if s % 5 == 4 else 2)
    frames = np.stack([np.stack([src(s, t) for s in range(S)]) for t in range(T)])
    g = pkg.FerHip(W, H, S, qp=12, window=window, maxdiff=3, intra_every=30)
    g.tune(pkg.TUNE_RESOLVE_WGS, wgs)
    g.tune(pkg.TUNE_RESOLVE_GROUP, group)
    g.tune(pkg.TUNE_SPECULATE, spec)
    streams, rec = g.encode_streams(frames, want_recon=True)
    assert g.status() == [0] * S
    counts = g.stats()
    st = g.read("SPEC_STAT")
    g.close()
    for s in range(S):
        ref, ref_rec, ref_counts = _oracle_stream(fo, frames[:, s], (s, window), W=W, H=H, qp=12, window=window, maxdiff=3, intra_every=30)
        assert streams[s] == ref, f"bitstream of stream {s}"
        assert np.array_equal(rec[:, s], ref_rec), f"recon of stream {s}"
        assert [int(v) for v in counts[s]] == ref_counts, f"brojTipova of stream {s}"
    assert counts[4][0] > 0, "the still streams are meant to contain P_Skip macroblocks"
    parts, hits, skq, skh = (int(v) for v in st[:4])
    assert parts > 0 and hits <= parts and skh <= skq and skq == S * (T - 1) * (W // 16) * (H // 16)
    assert (hits > 0) == bool(spec)


def test_720p_intra_config(pkg, fo):
    """BASELINE configs[1]: 720p I-frame encode (4x4 transform + quant + CAVLC), bit-exact."""
    _check_against_oracle(pkg, fo, 1280, 720, 2, 2, qp=12, window=16, intra_every=1, check_streams=(0, 1))


def test_720p_intra_qp28(pkg, fo):
    _check_against_oracle(pkg, fo, 1280, 720, 1, 1, qp=28, window=16, intra_every=1)


def test_1080p_ippp_config(pkg, fo):
    """BASELINE configs[2]: 1080p IPPP, WindowSize 32 (+-16 integer search)."""
    frames, streams, rec = _check_against_oracle(pkg, fo, 1920, 1072, 3, 2, qp=12, window=32, intra_every=30)
    # property: batching does not couple streams -- stream 1 alone gives the same bytes
    g = pkg.FerHip(1920, 1072, 1, qp=12, window=32, maxdiff=3, intra_every=30)
    alone, _ = g.encode_streams(frames[:, 1:2])
    assert alone[0] == streams[1]
    # property: idempotent across contexts
    g2 = pkg.FerHip(1920, 1072, 1, qp=12, window=32, maxdiff=3, intra_every=30)
    again, _ = g2.encode_streams(frames[:, 1:2])
    assert again[0] == alone[0]


def test_1080p_qp28(pkg, fo):
    _check_against_oracle(pkg, fo, 1920, 1072, 2, 1, qp=28, window=32, intra_every=30)


def test_4k_ippp_config(pkg, fo):
    """BASELINE configs[3] picture size (32 400 MBs, beyond the reference's own 10 000-MB arrays)."""
    _check_against_oracle(pkg, fo, 3840, 2160, 2, 1, qp=28, window=32, intra_every=8)


def test_ragged_sizes_and_edges(pkg, fo):
    """one-MB-wide / one-MB-high pictures and a non-multiple-of-64 width exercise every edge rule"""
    for (W, H) in [(16, 16), (16, 64), (64, 16), (48, 32), (208, 112)]:
        _check_against_oracle(pkg, fo, W, H, 3, 1, qp=12, window=16, intra_every=30, noise=1)


def test_1080p_full_gop(pkg, fo):
    """A whole GOP of the benchmark workload (30 pictures, 1080p, window 32): stream 0 bit-exact against the oracle
    over the full P chain; stream 1 decoded by the GPU decoder == decoded by the oracle decoder (the reference decoder's
    quirks can make a decoded picture differ from the encoder's reconstruction, so that is not the yardstick)."""
    W, H, T = 1920, 1072, 30
    frames = np.stack([np.stack([pkg.gen_frame(W, H, t, 1234 + s, 2) for s in range(2)]) for t in range(T)])
    g = pkg.FerHip(W, H, 2, qp=12, window=32, maxdiff=3, intra_every=30)
    streams, rec = g.encode_streams(frames, want_recon=True)
    assert g.status() == [0, 0]
    g.close()
    o = fo.Oracle(W, H, qp=12, window=32, maxdiff=3, intra_every=30)
    ref, ref_rec = o.encode_stream(frames[:, 0])
    o.close()
    assert streams[0] == ref
    assert np.array_equal(rec[:, 0], ref_rec)
    out, pics, w, h = pkg.decode_streams([streams[1]], T)
    assert pics == [T] and (w, h) == (W, H)
    n, dec, _ = fo.decode_stream_md5(streams[1])
    assert n == T
    for t in range(T):
        assert np.array_equal(out[t, 0], dec[t]), f"picture {t}"
    ys = W * H
    for t in range(T):  # luma is not touched by those quirks
        assert np.array_equal(out[t, 0][:ys], rec[t, 1][:ys])


@pytest.mark.parametrize("window", [48, 64])
def test_other_window_sizes(pkg, fo, window):
    """WindowSize values other than 16 / 32 take the kernels' general (not window-specialised) code: ordered list
    insertion instead of the rank selection, the un-tiled wide search."""
    _check_against_oracle(pkg, fo, 352, 288, 3, 2, qp=20, window=window, intra_every=30, check_streams=(0, 1))


@pytest.mark.parametrize("kind,W,H", [("flat", 352, 288), ("half", 352, 288), ("bars", 352, 288), ("half", 1920, 64), ("thirds", 1920, 48),
                                      ("drift", 352, 288), ("drift", 1920, 64), ("drift", 1280, 96)])
def test_flat_areas_overflow_the_candidate_lists(pkg, fo, kind, W, H):
    """Large flat areas: thousands of positions share one feature vector, the stage-2 candidate set of a partition
    outgrows what k_me_walk keeps; k_me_resolve then looks for the winners around the predictor (resolve_crowded) with
    the summary k_me_walk leaves (last step, distance bound, candidates of distance 0).  The wide, low pictures make the
    column range of the 280-diamond matter: partitions deep inside the flat area never see the textured part's
    positions of the same sums, partitions near the border do (the bound is then small and the ring scan long).
    "drift": the flat value moves by 2 per picture and MAXDIFF is 1, so the flat macroblocks are NOT P_Skip and every flat
    partition goes through the crowded stage 2 with the flat area of the previous picture 128 buckets away."""
    T = 3
    maxdiff = 1 if kind == "drift" else 3
    frames = []
    for t in range(T):
        f = pkg.gen_frame(W, H, t, 77, 2).copy()
        y = f[: W * H].reshape(H, W)
        if kind == "flat":
            y[:] = 128
        elif kind == "half":
            y[:, : W // 2] = 100
        elif kind == "drift":
            y[:, : W // 2] = 100 + 2 * t
        elif kind == "thirds":  # two flat areas of different values around a textured one
            y[:, : W // 3] = 60
            y[:, 2 * W // 3:] = 200
        else:  # letterbox
            y[:32] = 16
            y[-32:] = 16
        frames.append(f)
    frames = np.stack(frames)[:, None]
    g = pkg.FerHip(W, H, 1, qp=20, window=32, maxdiff=maxdiff, intra_every=30)
    streams, rec = g.encode_streams(frames, want_recon=True)
    assert g.status() == [0]
    g.close()
    o = fo.Oracle(W, H, qp=20, window=32, maxdiff=maxdiff, intra_every=30)
    ref, ref_rec = o.encode_stream(frames[:, 0])
    o.close()
    assert streams[0] == ref and np.array_equal(rec[:, 0], ref_rec)


@pytest.mark.parametrize("kind", ["bars", "quarter", "moving", "black_white"])
def test_zero_sum_blocks_follow_the_reference_bucket_defect(pkg, fo, kind):
    """Black (Y = 0) areas: 8x8 sums of 0 hit the reference's counting sort defect (bucket 0 is left out of the prefix
    sum, F/moestimation.cpp:153) -- every other bucket sits early, sum-0 positions overwrite or are overwritten by
    regular ones in arrival order, the end of the array keeps the previous picture's entries.  Reproduced, not refused."""
    W, H, T = 352, 288, 5
    frames = []
    for t in range(T):
        f = pkg.gen_frame(W, H, t, 31, 2).copy()
        y = f[: W * H].reshape(H, W)
        if kind == "bars":
            y[:40] = 0
            y[-40:] = 0
        elif kind == "quarter":
            y[: H // 2, : W // 2] = 0
        elif kind == "moving":   # the black area grows and moves: the left-over end of the array changes every picture
            y[16 * t: 16 * t + 64 + 8 * t, 32 + 24 * t: 200 + 8 * t] = 0
        else:
            y[:48] = 0
            y[-48:] = 255
        frames.append(f)
    frames = np.stack(frames)[:, None]
    g = pkg.FerHip(W, H, 1, qp=16, window=32, maxdiff=3, intra_every=30)
    streams, rec = g.encode_streams(frames, want_recon=True)
    assert g.status() == [0]
    g.close()
    o = fo.Oracle(W, H, qp=16, window=32, maxdiff=3, intra_every=30)
    ref, ref_rec = o.encode_stream(frames[:, 0])
    o.close()
    assert np.array_equal(rec[:, 0], ref_rec)
    assert streams[0] == ref


def test_scene_cut_forces_idr(pkg, fo):
    """selectNALUnitType: frame SAD above 16/pixel turns a P picture into IDR (F/ref_frames.cpp:210-228)."""
    W, H = 176, 144
    a = pkg.gen_frame(W, H, 0, 1, 2)
    b = pkg.gen_frame(W, H, 1, 1, 2)
    c = (255 - pkg.gen_frame(W, H, 2, 2, 2)).astype(np.uint8)
    c[: W * H] = np.clip(c[: W * H].astype(int), 16, 235).astype(np.uint8)
    frames = np.stack([a, b, c, c])[:, None]
    g = pkg.FerHip(W, H, 1, qp=12, window=16, maxdiff=3, intra_every=30)
    streams, rec = g.encode_streams(frames, want_recon=True)
    o = fo.Oracle(W, H, qp=12, window=16, maxdiff=3, intra_every=30)
    ref, ref_rec = o.encode_stream(frames[:, 0])
    assert streams[0] == ref and np.array_equal(rec[:, 0], ref_rec)
    types = [n[4] & 31 for n in pkg.split_nals(ref)]
    assert types == [7, 8, 5, 1, 5, 1]


def test_mixed_picture_types_in_one_call(pkg, fo):
    """Streams of one context take different picture types in the same call (a scene cut in stream 1 only,
    different IntraEvery phase is not possible, so the IDR comes from selectNALUnitType): the P-only kernels --
    sort, persistent row chain, residual -- must skip the IDR stream's rows and leave its state alone."""
    W, H = 176, 144
    s0 = [pkg.gen_frame(W, H, t, 5, 2) for t in range(5)]
    s1 = [pkg.gen_frame(W, H, t, 9, 2) for t in range(2)]
    cut = (255 - pkg.gen_frame(W, H, 2, 11, 2)).astype(np.uint8)
    cut[: W * H] = np.clip(cut[: W * H].astype(int), 16, 235).astype(np.uint8)
    s1 += [cut, pkg.gen_frame(W, H, 3, 9, 2), cut]   # IDR at t = 2, 3 and 4 (every change exceeds the threshold)
    frames = np.stack([np.stack(s0), np.stack(s1)], axis=1)
    g = pkg.FerHip(W, H, 2, qp=12, window=16, maxdiff=3, intra_every=30)
    streams, rec = g.encode_streams(frames, want_recon=True)
    assert g.status() == [0, 0]
    kinds = []
    for s in range(2):
        o = fo.Oracle(W, H, qp=12, window=16, maxdiff=3, intra_every=30)
        ref, ref_rec = o.encode_stream(frames[:, s])
        o.close()
        assert streams[s] == ref and np.array_equal(rec[:, s], ref_rec)
        kinds.append([n[4] & 31 for n in pkg.split_nals(ref)][2:])
    assert kinds[0] == [5, 1, 1, 1, 1] and kinds[1][2] == 5 and 1 in kinds[1][:2]


def test_bad_arguments_are_rejected(pkg):
    with pytest.raises(pkg.FerHipError):
        pkg.FerHip(100, 144, 1)          # not a multiple of 16
    with pytest.raises(pkg.FerHipError):
        pkg.FerHip(176, 144, 0)          # no streams


@pytest.mark.parametrize("W,H,T,qp,window,noise", [(176, 144, 5, 12, 16, 2), (176, 144, 4, 12, 32, 0), (1920, 1072, 2, 12, 32, 2),
                                                    (64, 48, 2, 12, 96, 2)])  # WindowSize 96: 9409 candidates per partition, beyond a 13-bit index
def test_basic_inter_encoding_matches_oracle(pkg, fo, W, H, T, qp, window, noise):
    """BasicInterEncoding = 1 (F/moestimation.cpp:394-397,470): the exhaustive pass whose vectors the reference
    discards leaves only brojTipova behind; stages 2 and 3 of the feature search are skipped.  Bitstream,
    reconstruction and the counters (P_Skip counted twice, the discarded pass's own 16x16 / 8x8 verdicts) must be
    the oracle's."""
    S = 2 if W < 1000 else 1   # (the oracle's literal exhaustive pass takes a minute per 1080p P picture)
    # noise 0 = still content: P_Skip macroblocks
    frames = np.stack([np.stack([pkg.gen_frame(W, H, t if noise else 0, 99 + s, noise) for s in range(S)]) for t in range(T)])
    g = pkg.FerHip(W, H, S, qp=qp, window=window, maxdiff=3, intra_every=30, basic=1)
    streams, rec = g.encode_streams(frames, want_recon=True)
    assert g.status() == [0] * S
    st = g.stats()
    g.close()
    for s in range(S):
        o = fo.Oracle(W, H, qp=qp, window=window, maxdiff=3, intra_every=30, basic=1)
        ref, ref_rec = o.encode_stream(frames[:, s])
        cnt = o.stats()
        o.close()
        assert streams[s] == ref, f"bitstream of stream {s}"
        assert np.array_equal(rec[:, s], ref_rec)
        assert list(st[s]) == list(cnt), f"brojTipova of stream {s}: {st[s]} vs {cnt}"
    if noise == 0:
        assert st[0][0] > 0 and st[0][0] % 2 == 0   # P_Skip macroblocks, each counted twice


def test_legacy_global_seam_matches_oracle(pkg, fo):
    """Drive the reference's own entry points (RBSP_encode, selectNALUnitType, global `frame`)
    exactly like encode()/NastaviEncode() do (F/fer_h264.cpp:55-134)."""
    import ctypes as C
    lib = pkg.load_library()

    class NALunit(C.Structure):
        _fields_ = [("forbidden_zero_bit", C.c_ubyte), ("nal_ref_idc", C.c_uint), ("nal_unit_type", C.c_uint),
                    ("NumBytesInRBSP", C.c_uint), ("rbsp_byte", C.POINTER(C.c_ubyte))]

    class Frame(C.Structure):
        _fields_ = [("Lwidth", C.c_int), ("Lheight", C.c_int), ("Cwidth", C.c_int), ("Cheight", C.c_int),
                    ("L", C.POINTER(C.c_ubyte)), ("C", C.POINTER(C.c_ubyte) * 2)]

    W, H, T = 64, 48, 4
    frame = Frame.in_dll(lib, "frame")
    for name, v in (("_qParameter", 12), ("BasicInterEncoding", 0), ("WindowSize", 16), ("MAXDIFF_SET", 3), ("IntraEvery", 30)):
        C.c_int.in_dll(lib, name).value = v
    ys = W * H
    L = (C.c_ubyte * ys)()
    Cb = (C.c_ubyte * (ys // 4))()
    Cr = (C.c_ubyte * (ys // 4))()
    frame.Lwidth, frame.Lheight, frame.Cwidth, frame.Cheight = W, H, W // 2, H // 2
    frame.L = C.cast(L, C.POINTER(C.c_ubyte))
    frame.C[0] = C.cast(Cb, C.POINTER(C.c_ubyte))
    frame.C[1] = C.cast(Cr, C.POINTER(C.c_ubyte))
    buf = (C.c_ubyte * 500000)()
    nu = NALunit(0, 1, 7, 0, C.cast(buf, C.POINTER(C.c_ubyte)))
    lib.RBSP_encode.argtypes = [C.POINTER(NALunit)]
    lib.RBSP_encode.restype = None
    out = bytearray()
    g = pkg.FerHip(16, 16, 1)  # only for write_nal()
    for t in (7, 8):
        nu.nal_unit_type = t
        lib.RBSP_encode(C.byref(nu))
        out += g.write_nal(t, bytes(buf[: nu.NumBytesInRBSP]))
    frames = np.stack([pkg.gen_frame(W, H, t, 21, 2) for t in range(T)])
    recs = []
    for t in range(T):
        C.c_int.in_dll(lib, "currFrameCount").value = t
        C.memmove(L, frames[t][:ys].ctypes.data, ys)
        C.memmove(Cb, frames[t][ys: ys + ys // 4].ctypes.data, ys // 4)
        C.memmove(Cr, frames[t][ys + ys // 4:].ctypes.data, ys // 4)
        nu.nal_unit_type = lib.selectNALUnitType()
        lib.RBSP_encode(C.byref(nu))
        assert nu.NumBytesInRBSP > 0
        out += g.write_nal(nu.nal_unit_type, bytes(buf[: nu.NumBytesInRBSP]))
        recs.append(np.concatenate([np.frombuffer(L, np.uint8), np.frombuffer(Cb, np.uint8), np.frombuffer(Cr, np.uint8)]).copy())
    o = fo.Oracle(W, H, qp=12, window=16, maxdiff=3, intra_every=30)
    ref, ref_rec = o.encode_stream(frames)
    assert bytes(out) == ref
    assert np.array_equal(np.stack(recs), ref_rec)
    # `frame` pointed at buffers of this test: do not leave them behind for the library to write into
    frame.L = None
    frame.C[0] = None
    frame.C[1] = None


def _oracle_decode(fo, stream):
    n, frames, _ = fo.decode_stream_md5(stream)
    return np.stack(frames)


@pytest.mark.parametrize("case", ["qcif_i_2f_qp12", "qcif_ippp_4f_qp12_w16", "qcif_ippp_4f_qp28_w32", "qcif_skip_5f_qp12"])
def test_gpu_decoder_matches_oracle_decoder_on_goldens(pkg, fo, case):
    """BASELINE configs[4] at QCIF: CAVLC parse + dequant + inverse transform + prediction on the GPU,
    bit-exact YUV against the oracle decoder (which reproduces the reference's md5 on drugi.264)."""
    stream = (GOLD / f"{case}.264").read_bytes()
    ref = _oracle_decode(fo, stream)
    out, pics, W, H = pkg.decode_streams([stream, stream], ref.shape[0])
    assert pics == [ref.shape[0]] * 2 and (W, H) == (176, 144)
    for s in range(2):
        assert np.array_equal(out[:, s], ref), case


def test_gpu_decoder_1080p(pkg, fo):
    """BASELINE configs[4]: 1080p decode of the IPPP stream."""
    W, H, T = 1920, 1072, 3
    frames = np.stack([pkg.gen_frame(W, H, t, 1234, 2) for t in range(T)])
    g = pkg.FerHip(W, H, 1, qp=12, window=32, maxdiff=3, intra_every=30)
    streams, rec = g.encode_streams(frames[:, None], want_recon=True)
    g.close()
    ref = _oracle_decode(fo, streams[0])
    out, pics, w, h = pkg.decode_streams([streams[0]], T)
    assert pics == [T] and (w, h) == (W, H)
    assert np.array_equal(out[:, 0], ref)
    assert np.array_equal(out[:, 0], rec[:, 0])  # decoder output == encoder reconstruction


def test_gpu_decoder_survives_damaged_slice_data(pkg, fo):
    """Bytes of the slice data overwritten at random: the parse either reaches the end of the picture or reports a syntax /
    unsupported error through the return code -- it never hangs or leaves the library unusable (every loop of k_dec_parse is
    bounded by the picture or the block, every table index by construction) -- and the undamaged stream still decodes bit-exactly
    right after."""
    clean = (GOLD / "qcif_ippp_4f_qp12_w16.264").read_bytes()
    ref = _oracle_decode(fo, clean)
    rng = np.random.default_rng(5)
    outcomes = {"ok": 0, "error": 0}
    for trial in range(24):
        bad = bytearray(clean)
        lo = 64 + int(rng.integers(0, len(bad) - 200))  # behind the parameter sets and the first slice header
        for k in range(int(rng.integers(1, 6))):
            bad[min(lo + int(rng.integers(0, 64)), len(bad) - 1)] = int(rng.integers(1, 256))  # (no new start codes: never 0)
        try:
            out, pics, w, h = pkg.decode_streams([bytes(bad), clean], ref.shape[0])
            outcomes["ok"] += 1
            assert np.array_equal(out[:, 1], ref)  # the damaged neighbour does not disturb the other stream of the batch
        except pkg.FerHipError:
            outcomes["error"] += 1
    out, pics, w, h = pkg.decode_streams([clean], ref.shape[0])
    assert pics == [ref.shape[0]] and np.array_equal(out[:, 0], ref)
    assert outcomes["ok"] + outcomes["error"] == 24


@pytest.mark.parametrize("W,H,S", [(3840, 64, 3), (6400, 32, 2), (12800, 16, 1)])
def test_gpu_decoder_wide_pictures(pkg, fo, W, H, S):
    """The parse kernel keeps a row of neighbour context per picture in LDS and is instantiated for 16, 8, 4 or 1 pictures
    per workgroup, whichever fits: 240, 400 and 800 macroblocks per row take the 8, 4 and 1 instantiations (1080p: 16)."""
    T = 3
    frames = np.stack([np.stack([pkg.gen_frame(W, H, t, 77 + s, 2) for s in range(S)]) for t in range(T)])
    g = pkg.FerHip(W, H, S, qp=20, window=16, maxdiff=3, intra_every=30)
    streams, rec = g.encode_streams(frames, want_recon=True)
    assert g.status() == [0] * S
    g.close()
    out, pics, w, h = pkg.decode_streams(streams, T)
    assert pics == [T] * S and (w, h) == (W, H)
    assert np.array_equal(out, rec)  # decoder output == encoder reconstruction
    assert np.array_equal(out[:, 0], _oracle_decode(fo, streams[0]))


def test_gpu_decoder_reproduces_reference_md5_on_drugi(pkg):
    """The reference's own fixture F/drugi.264 (x264 baseline, 640x480, 1000 pictures, intra MBs in P
    slices, mb_qp_delta != 0) decoded on the GPU gives the md5 of the REFERENCE's output recorded in
    SURVEY.md section 4 -- a pin of the HIP decode path that does not go through the oracle."""
    stream = (GOLD / "drugi.264").read_bytes()
    out, pics, W, H = pkg.decode_streams([stream], 1000)
    assert pics == [1000] and (W, H) == (640, 480)
    h = hashlib.md5()
    h.update(b"YUV4MPEG2 C420jpeg W640 H480 F24:1 Ip A1:1\n")
    for t in range(1000):
        h.update(b"FRAME\n")
        h.update(out[t, 0].tobytes())
    assert h.hexdigest() == "346891974ac8cafcc6bb72706e34f950"


def test_streaming_decoder_nal_by_nal(pkg, fo):
    """ferhip_dec_nal: one stream fed NAL unit by NAL unit (what RBSP_decode does), against the oracle decoder."""
    for case in ["qcif_ippp_4f_qp12_w16", "qcif_skip_5f_qp12"]:
        stream = (GOLD / f"{case}.264").read_bytes()
        ref = _oracle_decode(fo, stream)
        d = pkg.Decoder()
        pics = []
        for nal in pkg.split_nals(stream):
            t, idc, rbsp = pkg.unescape_nal(nal)
            p = d.nal(t, idc, rbsp)
            if p is not None:
                pics.append(p)
        d.close()
        assert (d.W, d.H) == (176, 144)
        assert np.array_equal(np.stack(pics), ref), case


def test_legacy_rbsp_decode_seam(pkg, tmp_path):
    """The reference's decode() loop (F/fer_h264.cpp:26-53): getNAL -> RBSP_decode(nu), output appended to `yuvoutput`
    by writeToY4M.  Driven through the exported legacy names on the reference's own fixture drugi.264 (first 40
    pictures NAL by NAL) and checked against the batch decoder, whose whole-file md5 is the reference's."""
    import ctypes as C
    lib = pkg.load_library()
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]

    class NALunit(C.Structure):
        _fields_ = [("forbidden_zero_bit", C.c_ubyte), ("nal_ref_idc", C.c_uint), ("nal_unit_type", C.c_uint),
                    ("NumBytesInRBSP", C.c_uint), ("rbsp_byte", C.POINTER(C.c_ubyte))]

    class Frame(C.Structure):
        _fields_ = [("Lwidth", C.c_int), ("Lheight", C.c_int), ("Cwidth", C.c_int), ("Cheight", C.c_int),
                    ("L", C.POINTER(C.c_ubyte)), ("C", C.POINTER(C.c_ubyte) * 2)]

    frame = Frame.in_dll(lib, "frame")   # the SPS call sizes and allocates it, as in the reference
    frame.L = None
    frame.C[0] = None
    frame.C[1] = None
    lib.RBSP_decode.argtypes = [NALunit]
    lib.RBSP_decode.restype = None
    stream = (GOLD / "drugi.264").read_bytes()
    NP = 40
    nals = pkg.split_nals(stream[:4_000_000])
    want, pics, W, H = pkg.decode_streams([stream], NP)
    assert pics == [NP]
    out = tmp_path / "dec.y4m"
    lib.ferhip_fileio_reset()
    fout = libc.fopen(str(out).encode(), b"wb")
    C.c_void_p.in_dll(lib, "yuvoutput").value = fout
    buf = (C.c_ubyte * 500000)()
    n = 0
    for nal in nals:
        t, idc, rbsp = pkg.unescape_nal(nal)
        if t in (1, 5):
            if n == NP:
                break
            n += 1
        C.memmove(buf, rbsp, len(rbsp))
        lib.RBSP_decode(NALunit(0, idc, t, len(rbsp), C.cast(buf, C.POINTER(C.c_ubyte))))
    libc.fclose(fout)
    C.c_void_p.in_dll(lib, "yuvoutput").value = None
    lib.ferhip_fileio_reset()
    expect = b"YUV4MPEG2 C420jpeg W%d H%d F24:1 Ip A1:1\n" % (W, H) + b"".join(b"FRAME\n" + want[t, 0].tobytes() for t in range(NP))
    assert out.read_bytes() == expect


def test_decoder_batch_with_different_qp_per_stream(pkg, fo):
    """Streams of one decode call carry their own PPS: the QP of this codec lives there (pic_init_qp = 14 + qp,
    slice_qp_delta = -14, F/headers_and_parameter_sets.cpp:489,186)."""
    W, H, T = 176, 144, 3
    frames = np.stack([pkg.gen_frame(W, H, t, 4, 2) for t in range(T)])
    streams = []
    for qp in (12, 28, 20):
        o = fo.Oracle(W, H, qp=qp, window=16, maxdiff=3, intra_every=30)
        s, _ = o.encode_stream(frames)
        o.close()
        streams.append(s)
    out, pics, w, h = pkg.decode_streams(streams, T)
    assert pics == [T] * 3
    for k, s in enumerate(streams):
        assert np.array_equal(out[:, k], _oracle_decode(fo, s)), k
    with pytest.raises(pkg.FerHipError):
        pkg.decode_streams(streams, 0)   # the output buffer is sized by max_pictures


def test_device_path_with_explicit_types_back_to_back(pkg, fo):
    """ferhip_encode_picture_dev with explicit picture types never synchronises between pictures: the slice headers
    of consecutive pictures must not overwrite each other in the pinned staging buffer (header ring).  Same pictures
    through the device-input / device-output path of two contexts driven from two threads (what bench.py times)
    against the oracle."""
    import threading
    W, H, T, S = 176, 144, 6, 2
    types = [pkg.ferhip.NAL_IDR if t % 3 == 0 else pkg.ferhip.NAL_SLICE for t in range(T)]
    res, errs = {}, []

    def run(k):
        try:
            frames = np.stack([np.stack([pkg.gen_frame(W, H, t, 40 + 10 * k + s, 2) for s in range(S)]) for t in range(T)])
            g = pkg.FerHip(W, H, S, qp=16, window=16, maxdiff=3, intra_every=1000)
            fsz, stride = g.fsz, g.nmb * 1024 + 4096
            dev = pkg.DeviceBuffer(T * S * fsz)
            dev.upload(frames)
            keep = pkg.DeviceBuffer(T * S * stride)
            lens = pkg.DeviceBuffer(T * S * 4)
            for t in range(T):
                g.set_frames_device(dev.ptr + t * S * fsz)
                p, strd, pl, nt = g.encode_picture_device([types[t]] * S)
                assert strd == stride and nt == [types[t]] * S
                g.copy_rbsp_device(keep.ptr + t * S * stride, lens.ptr + t * S * 4)   # on the library's stream, asynchronous
            assert g.status() == [0] * S
            res[k] = (frames, keep.download().reshape(T, S, stride), lens.download(dtype=np.uint32).reshape(T, S), g)
            for b_ in (dev, keep, lens):
                b_.free()
        except BaseException as ex:  # noqa: BLE001
            errs.append(ex)

    th = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs
    for k in range(2):
        frames, keep, lens, g = res[k]
        sps, pps = g.sps_pps()
        for s in range(S):
            o = fo.Oracle(W, H, qp=16, window=16, maxdiff=3, intra_every=3)
            ref, _ = o.encode_stream(frames[:, s])
            o.close()
            got = sps + pps + b"".join(g.write_nal(types[t], keep[t, s, : lens[t, s]].tobytes()) for t in range(T))
            assert got == ref, (k, s)
        g.close()


def test_gop_shards_over_two_contexts_equal_one_continuous_run(pkg, fo):
    """The multi-GPU partitioning on one GPU: 4 closed GOPs of one sequence, sharded round-robin (gops_of_rank) over
    2 "ranks" = 2 encoder contexts driven by 2 host threads, every GOP encoded by libferhip as its own stream, NAL
    units merged on the host in GOP order -- equal to ONE continuous oracle run over the 16 pictures."""
    import threading
    W, H, G, T = 352, 288, 4, 4
    frames = np.stack([pkg.gen_frame(W, H, t, 17, 2) for t in range(G * T)])
    world = 2
    res = {}

    def rank_main(rank):
        mine = pkg.gops_of_rank(G, world, rank)
        fr = np.stack([frames[g * T:(g + 1) * T] for g in mine], axis=1)     # [T][len(mine)][fsz]
        g_ = pkg.FerHip(W, H, len(mine), qp=20, window=32, maxdiff=3, intra_every=T)
        streams, _ = g_.encode_streams(fr)
        assert g_.status() == [0] * len(mine)
        g_.close()
        for k, g in enumerate(mine):
            res[g] = streams[k]

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    merged = pkg.merge_gop_streams([res[g] for g in range(G)])
    o = fo.Oracle(W, H, qp=20, window=32, maxdiff=3, intra_every=T)
    whole, _ = o.encode_stream(frames)
    o.close()
    assert merged == whole


def test_bench_ranks_shard_the_4k_gops(pkg):
    """bench.py --gpus 2 starts two rank processes itself (gloo rehearsal: both ranks on this GPU), each rank encodes
    its closed GOPs of the 64-picture 4K sequence with libferhip, the host merges the NAL units: the SHA-256 must be
    that of ONE continuous oracle run (tests/golden/goldens.json, BASELINE configs[3])."""
    import json
    import subprocess
    import sys
    root = Path(__file__).resolve().parent.parent
    out = subprocess.run([sys.executable, str(root / "bench.py"), "--config", "4k", "--gpus", "2", "--dist-backend", "gloo",
                          "--steps", "1", "--warmup", "0", "--cpu-frames", "0"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["output_check"]["ok"] is True and line["output_check"]["bytes"] == 54988382


_F4_PLANS = {
    # sub-8x8 partitions only
    "sub_partitions": [dict(), dict(), dict()],
    # ref_idx_l0 coded: override flag (sub_mb_pred) and the count it leaves behind (mb_pred of LATER slices too)
    "ref_idx": [dict(override=True, active=1), dict(), dict(override=True, active=0), dict(), dict(override=True, active=3)],
    # list modification: an empty one keeps the picture out of the reference slot, one with entries stores it
    "list_modification": [dict(modification=[]), dict(), dict(modification=[(0, 0)]), dict(modification=[]),
                          dict(modification=[]), dict(modification=[(1, 2), (2, 0)]), dict()],
    # everything, and slices whose last macroblocks fall to the reference's more_rbsp_data() heuristic
    "mixed": [dict(override=True, active=1, modification=[]), dict(early_end=True), dict(modification=[], early_end=True),
              dict(override=True, active=0), dict(modification=[(0, 1)], mvd_range=12), dict(mvd_range=40, p_skip=0.05)],
}


@pytest.mark.parametrize("plan", sorted(_F4_PLANS))
def test_decoder_sub_partitions_ref_idx_and_list_modification(pkg, fo, plan):
    """Row f4: P slices the reference's encoder never writes but its decoder parses (tests/pslice_synth.py writes
    them bit by bit after the IDR picture of a golden stream).  Batch decode (two different streams side by side, so
    the not-stored pictures of one stream must not disturb the other) and NAL-by-NAL decode against the oracle."""
    import pslice_synth as ps
    base = (GOLD / "qcif_ippp_4f_qp12_w16.264").read_bytes()
    a = ps.make_stream(pkg.split_nals, base, 7, _F4_PLANS[plan])
    b = ps.make_stream(pkg.split_nals, base, 8, _F4_PLANS["sub_partitions"] + _F4_PLANS[plan][:3])
    ra, rb = _oracle_decode(fo, a), _oracle_decode(fo, b)
    assert ra.shape[0] == len(_F4_PLANS[plan]) + 1
    assert (ra[1] != ra[0]).any()  # the P pictures do move
    out, pics, W, H = pkg.decode_streams([a, b], max(ra.shape[0], rb.shape[0]))
    assert pics == [ra.shape[0], rb.shape[0]]
    for t in range(ra.shape[0]):
        assert np.array_equal(out[t, 0], ra[t]), f"{plan}: picture {t}"
    assert np.array_equal(out[:rb.shape[0], 1], rb)
    d = pkg.Decoder()
    got = [p for p in (d.nal(*pkg.unescape_nal(n)) for n in pkg.split_nals(a)) if p is not None]
    d.close()
    assert np.array_equal(np.stack(got), ra)
