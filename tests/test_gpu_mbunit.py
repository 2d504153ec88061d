"""Per-macroblock known-answer tests (SURVEY.md 8b "per-MB", 8c): the reference's macroblock-level functions, one at a
time, HIP (through the C ABI of libferhip.so) against the oracle on seeded random inputs.

  a13  residual_block_cavlc_write / _size   every nC class {0-1, 2-3, 4-7, >= 8, -1}, maxNumCoeff 16 / 15 / 4, > 10^4 blocks
  a17  MotionCompensateSubMBPart            luma at all 16 quarter-sample phases, chroma at all 64 eighth-sample phases,
                                            vectors that reach far outside the picture (edge clamping)
  a1-a9 quantizationTransform and the transformDecoding* drivers, every macroblock class and QP class
  a12  coded_mb_size                        both alternatives of every macroblock of an I picture, incl. the stale-mb_type quirk
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _random_levels(rng, n, maxn):
    """level vectors the way a quantiser leaves them: mostly zero, a run of +-1 at the tail, now and then large levels (escape codes)"""
    c = np.zeros((n, 16), np.int32)
    for i in range(n):
        kind = rng.integers(0, 6)
        if kind == 0:
            continue                                               # empty block
        dens = rng.choice([0.1, 0.3, 0.6, 1.0])
        mask = rng.random(maxn) < dens
        mag = rng.choice([1, 1, 1, 2, 3, 8, 40, 300, 2000])
        v = rng.integers(1, mag + 1, maxn) * rng.choice([-1, 1], maxn)
        if kind == 1:
            v = rng.choice([-1, 1], maxn)                          # trailing ones only
        if kind == 2:
            v[rng.integers(0, maxn)] = rng.choice([-1, 1]) * rng.integers(16, 2040)   # level_prefix 14 / 15 escapes (12-bit suffix: |level| <= 2063)
            mask[:] = mask | (np.arange(maxn) == np.argmax(np.abs(v)))
        c[i, :maxn] = np.where(mask, v, 0)
    return c


def test_cavlc_block_writer_kat(pkg, fo):
    rng = np.random.default_rng(2024)
    total = 0
    for maxn, ncs in ((16, (0, 1, 2, 3, 4, 7, 8, 16)), (15, (0, 1, 2, 3, 4, 7, 8, 16)), (4, (-1,))):
        for nC in ncs:
            n = 700
            coef = _random_levels(rng, n, maxn)
            bits, nb, tc = pkg.cavlc_blocks(coef, np.full(n, nC), np.full(n, maxn))
            for i in range(n):
                ref_bytes, ref_n, ref_tc = fo.cavlc_encode_block(coef[i], maxn, nC)
                assert nb[i] == ref_n, (maxn, nC, i, coef[i])
                assert tc[i] == ref_tc
                assert bits[i, :(ref_n + 7) // 8].tobytes() == ref_bytes, (maxn, nC, i, coef[i])
            total += n
    assert total > 10000


def test_motion_compensate_sub_mb_part_kat(pkg, fo):
    W, H = 64, 48
    rng = np.random.default_rng(7)
    ref = rng.integers(0, 256, W * H * 3 // 2, dtype=np.uint8)
    o = fo.Oracle(W, H)
    o.set_dpb(ref)
    nmb = (W // 16) * (H // 16)
    desc = []
    for fy in range(8):          # every eighth-sample chroma phase = every quarter-sample luma phase twice over
        for fx in range(8):
            for _ in range(6):
                mb, sub, part = rng.integers(0, nmb), rng.integers(0, 4), rng.integers(0, 4)
                ix, iy = rng.integers(-12, 13), rng.integers(-12, 13)
                desc.append((mb, sub, part, ix * 8 + fx, iy * 8 + fy))
    for mb in range(nmb):        # far outside the picture in every direction, every sub-block
        for k in range(16):
            desc.append((mb, k >> 2, k & 3, int(rng.choice([-300, -77, 75, 301])), int(rng.choice([-250, -61, 59, 251]))))
    desc = np.array(desc, np.int32)
    pl, pb, pr = pkg.mc_sub_mb_parts(ref, W, H, desc)
    for i, (mb, sub, part, mvx, mvy) in enumerate(desc):
        rl, rb, rr = o.mc_sub(int(mb), int(sub), int(part), int(mvx), int(mvy))
        assert np.array_equal(pl[i], rl), ("luma", desc[i])
        assert np.array_equal(pb[i], rb), ("Cb", desc[i])
        assert np.array_equal(pr[i], rr), ("Cr", desc[i])
    o.close()


def _mb_case(rng, flat=False):
    srcY = rng.integers(0, 256, (16, 16))
    predY = np.clip(srcY + rng.integers(-40, 41, (16, 16)), 0, 255) if not flat else srcY.copy()
    srcC = rng.integers(0, 256, (2, 8, 8))
    predC = np.clip(srcC + rng.integers(-30, 31, (2, 8, 8)), 0, 255)
    return srcY, predY, srcC, predC


def test_quantization_transform_and_decoding_drivers_kat(pkg, fo):
    """quantizationTransform (F/quantizationTransform.cpp:349) for Intra16x16, inter and Intra4x4 macroblocks, with and
    without reconstruction, and the decode-side drivers of F/inttransform.cpp on the levels it produced."""
    W, H = 32, 32
    rng = np.random.default_rng(99)
    o = fo.Oracle(W, H)
    cur = 3
    for qp in (0, 12, 23, 24, 30, 36, 51):
        for mb_type, slice_type, cls in ((1, 2, 1), (0, 0, 2), (0, 2, 0)):  # I16 in an I slice, P_L0_16x16, I4x4
            for rec in (0, 1):
                srcY, predY, srcC, predC = _mb_case(rng, flat=(qp == 51))
                frame = np.zeros(W * H * 3 // 2, np.uint8)
                Y = frame[:W * H].reshape(H, W)
                Cb = frame[W * H:W * H * 5 // 4].reshape(H // 2, W // 2)
                Cr = frame[W * H * 5 // 4:].reshape(H // 2, W // 2)
                yP, xP = (cur // 2) * 16, (cur % 2) * 16
                Y[yP:yP + 16, xP:xP + 16] = srcY
                Cb[yP // 2:yP // 2 + 8, xP // 2:xP // 2 + 8] = srcC[0]
                Cr[yP // 2:yP // 2 + 8, xP // 2:xP // 2 + 8] = srcC[1]
                o.set_frame(frame)
                o.set_mb(cur, mb_type, slice_type, qp)
                o.set_levels(lumaLevel=np.zeros(256), dc16=np.zeros(16), ac16=np.zeros(256), cdc=np.zeros(8), cac=np.zeros(128))
                o.quantization_transform(predY, predC[0], predC[1], rec)
                lv = o.levels()
                of = o.frame()
                job = dict(op=pkg.MBU_QT, cls=cls, qp=qp, qpc=_qpc(qp), reconstruct=rec, srcY=srcY, srcCb=srcC[0], srcCr=srcC[1],
                           predY=predY, predCb=predC[0], predCr=predC[1])
                r = pkg.mb_unit([job])[0]
                tag = (qp, mb_type, slice_type, rec)
                if cls == 2:
                    assert np.array_equal(r["lumaLevel"], lv["lumaLevel"]), tag
                if cls == 1:
                    assert np.array_equal(r["dc16"], lv["dc16"]), tag
                    assert np.array_equal(r["ac16"].reshape(16, 16)[:, :15], lv["ac16"].reshape(16, 16)[:, :15]), tag
                assert np.array_equal(r["cdc"], lv["cdc"]), tag
                assert np.array_equal(r["cac"].reshape(8, 16)[:, :15], lv["cac"].reshape(8, 16)[:, :15]), tag
                if rec:
                    oY = of[:W * H].reshape(H, W)[yP:yP + 16, xP:xP + 16]
                    oCb = of[W * H:W * H * 5 // 4].reshape(H // 2, W // 2)[yP // 2:yP // 2 + 8, xP // 2:xP // 2 + 8]
                    oCr = of[W * H * 5 // 4:].reshape(H // 2, W // 2)[yP // 2:yP // 2 + 8, xP // 2:xP // 2 + 8]
                    if cls != 0:
                        assert np.array_equal(r["recY"].reshape(16, 16), oY), tag
                    assert np.array_equal(r["recCb"].reshape(8, 8), oCb), tag
                    assert np.array_equal(r["recCr"].reshape(8, 8), oCr), tag
                    # the decode-side drivers on the same levels give the same samples
                    if cls == 1:
                        d = pkg.mb_unit([dict(op=pkg.MBU_DEC16, qp=qp, qpc=_qpc(qp), predY=predY, dc16=r["dc16"], ac16=r["ac16"])])[0]
                        assert np.array_equal(d["recY"].reshape(16, 16), oY), tag
                    if cls == 2:
                        for blk in (0, 5, 15):
                            d = pkg.mb_unit([dict(op=pkg.MBU_DEC4, qp=qp, qpc=_qpc(qp), blk=blk, predY=predY, lumaLevel=r["lumaLevel"])])[0]
                            x0 = (0, 4, 0, 4, 8, 12, 8, 12, 0, 4, 0, 4, 8, 12, 8, 12)[blk]
                            y0 = (0, 0, 4, 4, 0, 0, 4, 4, 8, 8, 12, 12, 8, 8, 12, 12)[blk]
                            assert np.array_equal(d["recY"].reshape(16, 16)[y0:y0 + 4, x0:x0 + 4], oY[y0:y0 + 4, x0:x0 + 4]), (tag, blk)
                    d = pkg.mb_unit([dict(op=pkg.MBU_DECC, qp=qp, qpc=_qpc(qp), predCb=predC[0], predCr=predC[1], cdc=r["cdc"], cac=r["cac"])])[0]
                    assert np.array_equal(d["recCb"].reshape(8, 8), oCb) and np.array_equal(d["recCr"].reshape(8, 8), oCr), tag
        # transformDecodingP_Skip: reconstruction == prediction
        _, predY, _, predC = _mb_case(rng)
        d = pkg.mb_unit([dict(op=pkg.MBU_SKIP, qp=qp, qpc=_qpc(qp), predY=predY, predCb=predC[0], predCr=predC[1])])[0]
        assert np.array_equal(d["recY"].reshape(16, 16), predY) and np.array_equal(d["recCb"].reshape(8, 8), predC[0])
    o.close()


_QPC = (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31, 32, 32, 33,
        34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39)


def _qpc(qp):
    return _QPC[qp]  # H.264 Table 8-15 (qPiToQPc, F/inttransform.cpp:8-14) for chroma_qp_index_offset 0


@pytest.mark.parametrize("case", ["i_only", "idr_after_p_skip"])
def test_coded_mb_size_of_both_alternatives(pkg, fo, case):
    """coded_mb_size (F/rbsp_encoding.cpp:330): the bit count of the Intra16x16 and of the Intra4x4 alternative of EVERY macroblock
    of an I picture, read back from the device, against the oracle's two calls per macroblock.  `idr_after_p_skip`: the last
    picture is an IDR that follows P pictures with P_Skip macroblocks, so the Intra16x16 estimate meets the stale mb_type of
    F/residual.cpp:468."""
    W, H = 176, 144
    if case == "i_only":
        frames = np.stack([pkg.gen_frame(W, H, 0, 1234, 2)])
        ie = 30
    else:
        frames = np.stack([pkg.gen_frame(W, H, 0, 1234, 0)] * 4)   # I P P I, still content: P_Skip macroblocks
        ie = 3
    g = pkg.FerHip(W, H, 1, qp=12, window=16, maxdiff=3, intra_every=ie)
    streams, _ = g.encode_streams(frames[:, None])
    skips = int(g.stats()[0][0])
    got = g.read("MBSIZE").reshape(-1, 2)
    g.close()
    o = fo.Oracle(W, H, qp=12, window=16, maxdiff=3, intra_every=ie)
    ref, _ = o.encode_stream(frames)
    want = o.mbsize()
    o.close()
    assert streams[0] == ref
    assert np.array_equal(got, want)
    assert (want[:, 0] != want[:, 1]).any()
    if case != "i_only":
        assert skips > 0


def test_macroblock_entry_points_under_the_reference_names(pkg, fo):
    """quantizationTransform, transformDecoding*, residual_block_cavlc_write / _size, MotionCompensateSubMBPart and Decode exported
    under the reference's names, working on the reference's globals (frame, CurrMbAddr, QPy, mb_type, the level arrays, mvL0x)."""
    lib = pkg.load_library()
    W, H = 32, 32
    rng = np.random.default_rng(11)

    class Frame(C.Structure):
        _fields_ = [("Lwidth", C.c_int), ("Lheight", C.c_int), ("Cwidth", C.c_int), ("Cheight", C.c_int), ("L", C.c_void_p), ("C", C.c_void_p * 2)]

    def gint(name):
        return C.c_int.in_dll(lib, name)

    frame = Frame.in_dll(lib, "frame")
    saved = (frame.Lwidth, frame.Lheight, frame.Cwidth, frame.Cheight, frame.L, frame.C[0], frame.C[1])
    pic = rng.integers(0, 256, W * H * 3 // 2, dtype=np.uint8)
    mine = pic.copy()
    frame.Lwidth, frame.Lheight, frame.Cwidth, frame.Cheight = W, H, W // 2, H // 2
    frame.L = mine.ctypes.data
    frame.C[0] = mine.ctypes.data + W * H
    frame.C[1] = mine.ctypes.data + W * H * 5 // 4
    try:
        I16x16, I8x8 = (C.c_int * 16) * 16, (C.c_int * 8) * 8
        cur, qp = 2, 20
        gint("CurrMbAddr").value, gint("QPy").value = cur, qp
        predY = rng.integers(0, 256, (16, 16)).astype(np.int32)
        predC = rng.integers(0, 256, (2, 8, 8)).astype(np.int32)
        pl, pb, pr = I16x16(), I8x8(), I8x8()
        C.memmove(pl, predY.ctypes.data, 1024)
        C.memmove(pb, predC[0].ctypes.data, 256)
        C.memmove(pr, predC[1].ctypes.data, 256)
        o = fo.Oracle(W, H)
        for mb_type, slice_type in ((3, 2), (0, 0)):   # Intra16x16 in an I slice, P_L0_16x16 in a P slice
            mine[:] = pic
            gint("mb_type").value, gint("ferhip_legacy_slice_type").value = mb_type, slice_type
            lib.quantizationTransform(pl, pb, pr, C.c_ubyte(1))
            o.set_frame(pic)
            o.set_mb(cur, mb_type, slice_type, qp)
            o.quantization_transform(predY, predC[0], predC[1], 1)
            lv = o.levels()
            assert np.array_equal(mine, o.frame()), ("frame after quantizationTransform", mb_type)
            cdc = np.frombuffer((C.c_int * 8).in_dll(lib, "ChromaDCLevel"), np.int32)
            assert np.array_equal(cdc, lv["cdc"])
            if slice_type == 0:
                ll = np.frombuffer((C.c_int * 256).in_dll(lib, "LumaLevel"), np.int32)
                assert np.array_equal(ll, lv["lumaLevel"])
                # the decode-side driver reproduces block 6 from the levels
                before = mine.copy()
                mine[:W * H].reshape(H, W)[16:32, 0:16] = 0
                lib.transformDecoding4x4LumaResidual((C.c_int * 256).in_dll(lib, "LumaLevel"), pl, 6, qp)
                assert np.array_equal(mine[:W * H].reshape(H, W)[16 + 4:16 + 8, 8:12], before[:W * H].reshape(H, W)[16 + 4:16 + 8, 8:12])
            else:
                dc = np.frombuffer((C.c_int * 16).in_dll(lib, "Intra16x16DCLevel"), np.int32)
                assert np.array_equal(dc, lv["dc16"])
        # CAVLC: size and write agree with the oracle for a stated nC
        coef = np.array([7, -2, 0, 1, 0, 0, -1, 1, 0, 0, 0, 0, 0, 0, 0, 0], np.int32)
        for nC in (0, 3, 5, 9, -1):
            maxn = 4 if nC < 0 else 16
            gint("ferhip_legacy_nC").value = nC
            C.c_uint.in_dll(lib, "ferhip_legacy_nbits").value = 0
            cc = (C.c_int * 16)(*coef.tolist())
            lib.residual_block_cavlc_size.restype = C.c_uint
            n = lib.residual_block_cavlc_size(cc, 0, maxn - 1, maxn)
            lib.residual_block_cavlc_write(cc, 0, maxn - 1, maxn)
            ref_bytes, ref_n, _ = fo.cavlc_encode_block(coef, maxn, nC)
            assert n == ref_n and C.c_uint.in_dll(lib, "ferhip_legacy_nbits").value == ref_n
            got = bytes((C.c_ubyte * 64).in_dll(lib, "ferhip_legacy_bits"))[:(ref_n + 7) // 8]
            assert got == ref_bytes
        # motion compensation: MotionCompensateSubMBPart on an explicit picture, Decode on dpb
        lib.AllocateMemory()
        mvx = C.POINTER(C.POINTER(C.POINTER(C.c_int))).in_dll(lib, "mvL0x")
        mvy = C.POINTER(C.POINTER(C.POINTER(C.c_int))).in_dll(lib, "mvL0y")
        refpic = rng.integers(0, 256, W * H * 3 // 2, dtype=np.uint8)
        rf = Frame(W, H, W // 2, H // 2, refpic.ctypes.data, (C.c_void_p * 2)(refpic.ctypes.data + W * H, refpic.ctypes.data + W * H * 5 // 4))
        o.set_dpb(refpic)
        want_l, want_b, want_r = np.zeros((16, 16), np.int32), np.zeros((8, 8), np.int32), np.zeros((8, 8), np.int32)
        for k in range(16):
            sub, part = k >> 2, k & 3
            vx, vy = int(rng.integers(-70, 71)), int(rng.integers(-70, 71))
            mvx[cur][sub][part], mvy[cur][sub][part] = vx, vy
            l4, b2, r2 = o.mc_sub(cur, sub, part, vx, vy)
            oy, ox = ((sub & 2) << 2) + ((part & 2) << 1), ((sub & 1) << 3) + ((part & 1) << 2)
            want_l[oy:oy + 4, ox:ox + 4], want_b[oy // 2:oy // 2 + 2, ox // 2:ox // 2 + 2], want_r[oy // 2:oy // 2 + 2, ox // 2:ox // 2 + 2] = l4, b2, r2
        gl, gr, gb = I16x16(), I8x8(), I8x8()
        lib.MotionCompensateSubMBPart(gl, gr, gb, C.byref(rf), cur, 2, 1)
        assert np.array_equal(np.frombuffer(gl, np.int32).reshape(16, 16)[8:12, 4:8], want_l[8:12, 4:8])
        dpb = Frame.in_dll(lib, "dpb")
        dsaved = (dpb.Lwidth, dpb.Lheight, dpb.Cwidth, dpb.Cheight, dpb.L, dpb.C[0], dpb.C[1])
        dpb.Lwidth, dpb.Lheight, dpb.Cwidth, dpb.Cheight, dpb.L = W, H, W // 2, H // 2, rf.L
        dpb.C[0], dpb.C[1] = rf.C[0], rf.C[1]
        gl, gr, gb = I16x16(), I8x8(), I8x8()
        lib.Decode(gl, gr, gb)
        assert np.array_equal(np.frombuffer(gl, np.int32).reshape(16, 16), want_l)
        assert np.array_equal(np.frombuffer(gb, np.int32).reshape(8, 8), want_b)
        assert np.array_equal(np.frombuffer(gr, np.int32).reshape(8, 8), want_r)
        (dpb.Lwidth, dpb.Lheight, dpb.Cwidth, dpb.Cheight, dpb.L, dpb.C[0], dpb.C[1]) = dsaved
        o.close()
    finally:
        (frame.Lwidth, frame.Lheight, frame.Cwidth, frame.Cheight, frame.L, frame.C[0], frame.C[1]) = saved
