"""CPU tests of the oracle itself: pins against the reference's own data.

  * drugi.264 -- a fixture the reference ships (F/drugi.264).  The md5 of the reference's decoded
    Y4M output is recorded in SURVEY.md section 4; the oracle decoder must reproduce it.
  * oracle/_ref/libfer_leaf.so -- the reference's own leaf sources (bit writer, Exp-Golomb, CAVLC
    tables, level tables) compiled where they lie; the oracle's tables and coders are compared
    entry by entry.  Skipped when the library is not present (it is built only where the reference
    tree exists).
  * encode -> decode round trip of the oracle, and the committed golden streams.
"""
import ctypes as C
import hashlib
import json
from pathlib import Path

import numpy as np
import pytest

HERE = Path(__file__).resolve().parent
GOLD = HERE / "golden"
REF_MD5_DRUGI = "346891974ac8cafcc6bb72706e34f950"  # SURVEY.md section 4 (reference output, 1000 frames)


def test_decoder_reproduces_reference_md5_on_drugi(fo):
    data = (GOLD / "drugi.264").read_bytes()
    n, frames, st = fo.decode_stream_md5(data)
    assert n == 1000 and (st["W"], st["H"]) == (640, 480)
    h = hashlib.md5()
    h.update(b"YUV4MPEG2 C420jpeg W640 H480 F24:1 Ip A1:1\n")  # writeToY4M, F/fileIO.cpp:134-176
    for f in frames:
        h.update(b"FRAME\n")
        h.update(f.tobytes())
    assert h.hexdigest() == REF_MD5_DRUGI


def _leaf():
    p = HERE.parent / "oracle" / "_ref" / "libfer_leaf.so"
    if not p.exists():
        pytest.skip("oracle/_ref/libfer_leaf.so not built (reference tree absent)")
    return C.CDLL(str(p))


def _arr(lib, name, ctype, shape):
    n = int(np.prod(shape))
    a = (ctype * n).in_dll(lib, name)
    return np.ctypeslib.as_array(a).reshape(shape).copy()


def test_cavlc_tables_match_reference_leaf(fo):
    ref = _leaf()
    L = fo.lib()

    def mine(name, shape):
        return _arr(L, name, C.c_uint8, shape).astype(np.int64)

    ct_len, ct_code = mine("fo_ct_len", (3, 4, 17)), mine("fo_ct_code", (3, 4, 17))
    names = ["CoeffTokenCodesCoder_nC_0_to_2", "CoeffTokenCodesCoder_nC_2_to_4", "CoeffTokenCodesCoder_nC_4_to_8"]
    for cls, nm in enumerate(names):
        rl = _arr(ref, nm + "_length", C.c_int, (17, 4))
        rc = _arr(ref, nm + "_data_int", C.c_uint, (17, 4))
        for tc in range(17):
            for t1 in range(min(tc, 3) + 1):
                assert ct_len[cls, t1, tc] == rl[tc, t1], (cls, tc, t1)
                assert ct_code[cls, t1, tc] == rc[tc, t1], (cls, tc, t1)
    rl = _arr(ref, "CoeffTokenCodesCoder_nC_8_to_max_length", C.c_int, (17, 4))
    rc = _arr(ref, "CoeffTokenCodesCoder_nC_8_to_max_data_int", C.c_uint, (17, 4))
    for tc in range(17):
        for t1 in range(min(tc, 3) + 1):
            assert rl[tc, t1] == 6 and rc[tc, t1] == (3 if tc == 0 else ((tc - 1) << 2) | t1)
    dl, dc = mine("fo_ctdc_len", (4, 5)), mine("fo_ctdc_code", (4, 5))
    rl = _arr(ref, "CoeffTokenCodeTableCoder_ChromaDC_length", C.c_int, (17, 4))
    rc = _arr(ref, "CoeffTokenCodeTableCoder_ChromaDC_data_int", C.c_uint, (17, 4))
    for tc in range(5):
        for t1 in range(min(tc, 3) + 1):
            assert dl[t1, tc] == rl[tc, t1] and dc[t1, tc] == rc[tc, t1], (tc, t1)
    tl, tcde = mine("fo_tz_len", (15, 16)), mine("fo_tz_code", (15, 16))
    rl = _arr(ref, "TotalZerosCodeTableCoder_4x4_length", C.c_int, (15, 16))
    rc = _arr(ref, "TotalZerosCodeTableCoder_4x4_data_int", C.c_uint, (15, 16))
    for k in range(15):
        for z in range(16 - k):
            assert tl[k, z] == rl[k, z] and tcde[k, z] == rc[k, z], (k, z)
    tl, tcde = mine("fo_tzdc_len", (3, 4)), mine("fo_tzdc_code", (3, 4))
    rl = _arr(ref, "TotalZerosCodeTableCoder_ChromaDC_length", C.c_int, (3, 4))
    rc = _arr(ref, "TotalZerosCodeTableCoder_ChromaDC_data_int", C.c_uint, (3, 4))
    for k in range(3):
        for z in range(4 - k):
            assert tl[k, z] == rl[k, z] and tcde[k, z] == rc[k, z]
    bl, bc = mine("fo_rb_len", (6, 7)), mine("fo_rb_code", (6, 7))
    rl = _arr(ref, "RunBeforeCodeTableCoder_length", C.c_int, (6, 7))
    rc = _arr(ref, "RunBeforeCodeTableCoder_data_int", C.c_uint, (6, 7))
    for zl in range(6):
        for r in range(zl + 2):
            assert bl[zl, r] == rl[zl, r] and bc[zl, r] == rc[zl, r]


def test_level_codes_match_reference_leaf(fo):
    """closed-form level_prefix/suffix of the oracle == the table the reference generates
    (generate_residual_level_tables, F/residual_tables.cpp:940-1010)."""
    ref = _leaf()
    getattr(ref, "_Z30generate_residual_level_tablesv")()  # C++ linkage: generate_residual_level_tables()
    tab = _arr(ref, "levelcode_to_outputstream", C.c_int, (5056, 7, 4))
    L = fo.lib()
    L.fo_cavlc_encode_block.restype = C.c_uint
    # probe through single-coefficient blocks: level v at suffixLength 0 as first non-T1 level
    for sl in range(7):
        for code in list(range(0, 300)) + [1000, 2000, 4000, 5055 if sl == 6 else 4125 if sl == 0 else 3000]:
            if tab[code, sl, 0] == 30:
                continue
            pre, ss, suf = tab[code, sl, 1], tab[code, sl, 2], tab[code, sl, 3]
            if sl == 0:
                e = (code, 0, 0) if code < 14 else ((14, 4, code - 14) if code < 30 else (15, 12, code - 30))
            elif code < (15 << sl):
                e = (code >> sl, sl, code & ((1 << sl) - 1))
            else:
                e = (15, 12, code - (15 << sl))
            assert (pre, suf) == (e[0], e[2]) and (ss == e[1] or (sl == 0 and pre < 14)), (code, sl)


def test_expgolomb_and_bitwriter_match_reference_leaf(fo):
    ref = _leaf()
    L = fo.lib()
    fn = {n: getattr(ref, m) for n, m in dict(init="_Z23init_expgolomb_UC_codesv", writer="_Z13initRawWriterPhj",
                                              ue="_Z12expGolomb_UCj", se="_Z12expGolomb_SCi", put="_Z12writeRawBitsij",
                                              ones="_Z9writeOnesi", flush="_Z16flushWriteBufferv").items()}
    fn["init"]()
    buf_r = (C.c_ubyte * 4096)()
    buf_m = (C.c_ubyte * 4096)()
    fn["writer"](buf_r, C.c_uint(4096))

    class BW(C.Structure):
        _fields_ = [("buf", C.c_void_p), ("cap", C.c_size_t), ("nbits", C.c_size_t)]

    w = BW()
    L.fo_bw_init(C.byref(w), buf_m, 4096)
    rng = np.random.default_rng(3)
    for v in list(range(0, 70)) + [int(x) for x in rng.integers(0, 9999, 200)]:
        fn["ue"](C.c_uint(v))
        L.fo_bw_ue(C.byref(w), C.c_uint(v))
        sv = int(v // 2 - 40)
        fn["se"](C.c_int(sv))
        L.fo_bw_se(C.byref(w), C.c_int(sv))
        n = int(rng.integers(1, 25))
        x = int(rng.integers(0, 1 << n))
        fn["put"](C.c_int(n), C.c_uint(x))
        L.fo_bw_put(C.byref(w), C.c_int(n), C.c_uint(x))
    fn["ones"](C.c_int(1))
    fn["flush"]()
    L.fo_bw_put(C.byref(w), 1, 1)
    nbytes = (w.nbits + 7) // 8
    assert bytes(buf_r[: nbytes - 1]) == bytes(buf_m[: nbytes - 1])
    assert C.c_uint.in_dll(ref, "RBSP_write_current_byte").value * 8 + C.c_uint.in_dll(ref, "RBSP_write_current_bit").value == w.nbits


def test_level_scale_tables_match_reference_formula(fo):
    L = fo.lib()
    # LevelQuantize printed at F/quantizationTransform.cpp:24-32, first row of each m
    expect_q = {0: (205, 158, 128), 1: (186, 146, 114), 2: (158, 128, 102), 3: (146, 114, 89), 4: (128, 102, 82),
                5: (114, 89, 71)}
    for m, (a, b, c) in expect_q.items():
        assert (L.fo_level_quantize(m, 0, 0), L.fo_level_quantize(m, 0, 1), L.fo_level_quantize(m, 1, 1)) == (a, b, c)
        assert L.fo_level_scale(m, 0, 0) == 16 * (10, 11, 13, 14, 16, 18)[m]


@pytest.mark.parametrize("case", ["qcif_i_2f_qp12", "qcif_ippp_4f_qp12_w16", "qcif_ippp_4f_qp28_w32", "qcif_skip_5f_qp12"])
def test_oracle_matches_committed_goldens_and_roundtrips(pkg, fo, case):
    meta = json.loads((GOLD / "goldens.json").read_text())[case]
    frames = np.stack([pkg.gen_frame(meta["W"], meta["H"], 0 if meta.get("static") else t, meta["seed"], meta["noise"])
                       for t in range(meta["T"])])
    assert hashlib.sha256(frames.tobytes()).hexdigest() == meta["source_sha256"]
    o = fo.Oracle(meta["W"], meta["H"], qp=meta["qp"], window=meta["window"], maxdiff=meta["maxdiff"],
                  intra_every=meta["intra_every"])
    stream, rec = o.encode_stream(frames)
    o.close()
    assert stream == (GOLD / f"{case}.264").read_bytes()
    assert hashlib.sha256(rec.tobytes()).hexdigest() == meta["recon_sha256"]
    # the encoder's in-place reconstruction equals what the oracle decoder produces from the stream
    n, dec, _ = fo.decode_stream_md5(stream)
    assert n == meta["T"]
    for t in range(n):
        if case == "qcif_skip_5f_qp12":
            # the reference decoder re-applies stale chroma AC levels to cbp==0 macroblocks
            # (clear_residual_structures, F/residual.cpp:28-49): luma must still match
            ys = meta["W"] * meta["H"]
            assert np.array_equal(dec[t][:ys], rec[t][:ys])
        else:
            assert np.array_equal(dec[t], rec[t]), t


def test_block_transforms_roundtrip_identity(fo):
    """forward core is built as the fixed-point inverse of the decoder's inverse transform
    (F/quantizationTransform.cpp:41): dequant(quant(fwd(x))) stays within the quantiser step."""
    rng = np.random.default_rng(5)
    x = rng.integers(-200, 201, size=(2000, 16), dtype=np.int32)
    for qp in (10, 12, 20, 28):
        y = fo.inverse_residual(qp, fo.forward_residual(qp, x))
        step = 0.625 * 2 ** (qp / 6)
        err = np.abs(y - x)
        assert err.max() <= 5 * step + 2 and err.mean() <= step
