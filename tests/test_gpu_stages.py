"""GPU parity, stage by stage, against the CPU oracle (bit-exact, integer/byte work).

Every check calls libferhip.so through its C ABI and compares device state read back with
the oracle's state on the same seeded synthetic input.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H = 176, 144


def _two_frames(pkg, W=W, H=H, noise=2):
    return pkg.gen_frame(W, H, 0, 1234, noise), pkg.gen_frame(W, H, 1, 1234, noise)


def test_block_transform_kat(pkg, fo):
    rng = np.random.default_rng(7)
    blocks = rng.integers(-255, 256, size=(4096, 16), dtype=np.int32)
    blocks[::7] = 0
    from h264_fer_amd.ferhip import forward_residual, inverse_residual
    for qp in (10, 12, 23, 24, 28, 30, 36, 51):
        for keep in (False, True):
            g = forward_residual(qp, blocks, keep)
            o = fo.forward_residual(qp, blocks[:512], keep)
            assert np.array_equal(g[:512], o), (qp, keep)
            lv = rng.integers(-40, 41, size=(512, 16), dtype=np.int32)
            assert np.array_equal(inverse_residual(qp, lv, keep), fo.inverse_residual(qp, lv, keep)), (qp, keep)


# the sizes beside QCIF put a tile of the first radix pass (4096 positions in column order, k_rs_scatter<0>) on 256 columns
# (H = 16), on two or three (H = 2064) and on one or two columns of a picture taller than a tile
@pytest.mark.parametrize("W,H", [(176, 144), (1024, 16), (32, 2064), (16, 4112), (208, 80)])
def test_refprep_matches_oracle(pkg, fo, W, H):
    f0, f1 = _two_frames(pkg, W, H)
    o = fo.Oracle(W, H, qp=12, window=16)
    o.set_dpb(f0)
    o.fill_interpolated()
    g = pkg.FerHip(W, H, 1, qp=12, window=16)
    g.set_reference(f0[None])
    g.set_frames(f1[None])
    g.fill_interpolated()
    gi = g.read("INTERP").reshape(16, H, W)
    for f in range(16):
        assert np.array_equal(gi[f], o.interp(f)), f"interp plane {f}"
    gf = g.read("FEAT").reshape(H, W, 16, 6)
    for f in range(16):
        for k in range(5):
            assert np.array_equal(gf[:, :, f, k].astype(np.int32), o.kar(k, f)), (f, k)
    assert np.array_equal(g.read("KOLIKO")[:16384], o.koliko()[:16384])
    sp = g.read("SORTPOS")
    assert np.array_equal((sp >> 16).astype(np.int32), o.sorted(2))
    assert np.array_equal((sp & 0xFFFF).astype(np.int32), o.sorted(1))
    assert g.status() == [0]


@pytest.mark.parametrize("window,maxdiff,noise", [(16, 3, 2), (32, 3, 2), (16, -1, 0)])
def test_inter_decision_matches_oracle(pkg, fo, window, maxdiff, noise):
    f0, f1 = _two_frames(pkg, noise=noise)
    o = fo.Oracle(W, H, qp=12, window=window, maxdiff=maxdiff)
    rb0 = o.encode_slice(5) if False else None
    o.set_frame(f0)
    o.encode_slice(5)
    rec0 = o.frame()
    o.set_frame(f1)
    o.encode_slice(1)
    g = pkg.FerHip(W, H, 1, qp=12, window=window, maxdiff=maxdiff)
    g.set_reference(rec0[None])
    g.set_frames(f1[None])
    g.inter_encoding()
    assert g.status() == [0]
    assert np.array_equal(g.read("MBTYPE"), o.mb_type())
    assert np.array_equal(g.read("MV").reshape(-1, 4, 2).astype(np.int32), o.mv())


def _streams(pkg, n, S, W=W, H=H, noise=2):
    # stream s uses seed 1234 + s; frames [T][S][fsz]
    return np.stack([np.stack([pkg.gen_frame(W, H, t, 1234 + s, noise) for s in range(S)]) for t in range(n)])


@pytest.mark.parametrize("intra_every,qp,window,noise", [(1, 12, 16, 2), (30, 12, 16, 2), (30, 28, 32, 2), (3, 20, 16, 0)])
def test_stream_bytes_match_oracle(pkg, fo, intra_every, qp, window, noise):
    T, S = 4, 2
    frames = _streams(pkg, T, S, noise=noise)
    g = pkg.FerHip(W, H, S, qp=qp, window=window, maxdiff=3, intra_every=intra_every)
    streams, rec = g.encode_streams(frames, want_recon=True)
    assert g.status() == [0] * S
    for s in range(S):
        o = fo.Oracle(W, H, qp=qp, window=window, maxdiff=3, intra_every=intra_every)
        ref_bytes, ref_rec = o.encode_stream(frames[:, s])
        o.close()
        for t in range(T):
            assert np.array_equal(rec[t, s], ref_rec[t]), f"recon stream {s} frame {t}"
        assert streams[s] == ref_bytes, f"bitstream stream {s}: {len(streams[s])} vs {len(ref_bytes)}"


def test_dc_transform_and_scan_kats(pkg, fo):
    """Rows a3 / a4 / a5 at block level through the C ABI: forwardDCLumaIntra, InverseDCLumaIntra, forwardDCChroma,
    InverseDCChroma (F/quantizationTransform.cpp:105-178,227-282, F/scaleTransform.cpp:154-189,247-261,344-421),
    transformScan (both variants) and transformInverseScan, against the oracle, every QP class."""
    from h264_fer_amd.ferhip import block_op
    rng = np.random.default_rng(11)
    n = 384
    dc = rng.integers(-4000, 4001, size=(n, 16), dtype=np.int32)     # DC terms of 16 forward-transformed blocks
    dc[::9] = 0
    lv = rng.integers(-300, 301, size=(n, 16), dtype=np.int32)
    for qp in (0, 5, 10, 12, 17, 23, 24, 29, 35, 36, 41, 51):
        for name, data in (("forward_dc_luma_intra", dc), ("inverse_dc_luma_intra", lv),
                           ("forward_dc_chroma", dc), ("inverse_dc_chroma", lv)):
            g = block_op(name, data, qp)
            o = fo.block_op(name, data, qp)
            if "chroma" in name:
                g, o = g[:, :4], o[:, :4]
            assert np.array_equal(g, o), (name, qp)
    for ac in (False, True):
        assert np.array_equal(block_op("transform_scan", lv, flag=ac), fo.block_op("transform_scan", lv, flag=ac)), ac
    assert np.array_equal(block_op("transform_inverse_scan", lv), fo.block_op("transform_inverse_scan", lv))
    # round trip: inverse scan of the scan is the identity
    assert np.array_equal(block_op("transform_inverse_scan", block_op("transform_scan", lv)), lv)


def test_block_entry_points_under_the_reference_names(pkg, fo):
    """The unit-parity surface of SURVEY 8b: forwardResidual, transformScan, forwardDCLumaIntra, forwardDCChroma
    (F/quantizationTransform.h), transformInverseScan, inverseResidual, InverseDCLumaIntra, InverseDCChroma
    (F/scaleTransform.h) exported by libferhip under their own names with the reference's argument lists, one block per
    call, against the oracle."""
    import ctypes as C
    lib = pkg.load_library()
    rng = np.random.default_rng(5)
    I4, I16, I2 = (C.c_int * 4) * 4, C.c_int * 16, (C.c_int * 2) * 2

    def arr(t, v):
        a = t()
        C.memmove(a, np.ascontiguousarray(v, np.int32).ctypes.data, C.sizeof(a))
        return a

    def out(a, n):
        return np.frombuffer(a, np.int32, n).copy()

    for qp in (0, 12, 23, 24, 35, 36, 51):
        res = rng.integers(-255, 256, 16, dtype=np.int32)
        lev = rng.integers(-300, 301, 16, dtype=np.int32)
        dc = rng.integers(-4000, 4001, 16, dtype=np.int32)
        for keep in (0, 1):
            r = I4()
            lib.forwardResidual(qp, arr(I4, res), r, C.c_ubyte(0), C.c_ubyte(keep))
            assert np.array_equal(out(r, 16), fo.forward_residual(qp, res[None], keep)[0]), ("forwardResidual", qp, keep)
            r = I4()
            lib.inverseResidual(8, qp, arr(I4, lev), r, C.c_ubyte(keep))
            assert np.array_equal(out(r, 16), fo.inverse_residual(qp, lev[None], keep)[0]), ("inverseResidual", qp, keep)
        c = I4()
        lib.forwardDCLumaIntra(qp, arr(I4, dc), c)
        assert np.array_equal(out(c, 16), fo.block_op("forward_dc_luma_intra", dc[None], qp)[0])
        c = I4()
        lib.InverseDCLumaIntra(8, qp, arr(I4, lev), c)
        assert np.array_equal(out(c, 16), fo.block_op("inverse_dc_luma_intra", lev[None], qp)[0])
        c = I2()
        lib.forwardDCChroma(qp, arr(I2, dc[:4]), c, C.c_ubyte(0))
        assert np.array_equal(out(c, 4), fo.block_op("forward_dc_chroma", dc[None], qp)[0, :4])
        c = I2()
        lib.InverseDCChroma(8, qp, arr(I2, lev[:4]), c)
        assert np.array_equal(out(c, 4), fo.block_op("inverse_dc_chroma", lev[None], qp)[0, :4])
    lev = rng.integers(-300, 301, 16, dtype=np.int32)
    for ac in (0, 1):
        l = I16()
        lib.transformScan(arr(I4, lev), l, C.c_ubyte(ac))
        assert np.array_equal(out(l, 16 - ac), fo.block_op("transform_scan", lev[None], flag=bool(ac))[0, :16 - ac])
    c = I4()
    lib.transformInverseScan(arr(I16, lev), c)
    assert np.array_equal(out(c, 16), fo.block_op("transform_inverse_scan", lev[None])[0])


def test_randomised_configuration_sweep(pkg, fo):
    """A bounded slice of tools/sweep.py inside the suite: 20 random configurations (sizes, qp, WindowSize incl. the
    general path, MAXDIFF incl. adaptive and 0, IntraEvery, noise, still content = P_Skip heavy, flat / black boxes,
    BasicInterEncoding) -- encoder bitstream + reconstruction + counters and the decoder against the oracle."""
    import random
    rng = random.Random(20261004)
    for it in range(20):
        W, H = rng.choice([(176, 144), (352, 288), (64, 48), (128, 96), (320, 240), (16, 32), (48, 16)])
        cfg = dict(qp=rng.choice([10, 12, 16, 20, 24, 28, 30]), window=rng.choice([16, 32, 32, 32, 48]),
                   maxdiff=rng.choice([3, 3, -1, 0, 6]), intra_every=rng.choice([30, 30, 3, 2]), basic=rng.choice([0, 0, 0, 1]))
        T, S = rng.choice([3, 4, 5]), rng.choice([1, 2, 3])
        noise = rng.choice([0, 1, 2, 4])
        seeds = [rng.randrange(1, 10000) for _ in range(S)]
        still = rng.random() < 0.25
        frames = np.stack([np.stack([pkg.gen_frame(W, H, 0 if still else t, seeds[s], 0 if still else noise) for s in range(S)])
                           for t in range(T)])
        shape = rng.choice(["none", "none", "flat_box", "black_box", "black_rows"])
        if shape != "none" and W >= 64 and H >= 48:
            for t in range(T):
                for s_ in range(S):
                    y = frames[t, s_][: W * H].reshape(H, W)
                    x0, y0 = rng.randrange(0, W // 2), rng.randrange(0, H // 2)
                    if shape == "flat_box":
                        y[y0: y0 + H // 3, x0: x0 + W // 2] = rng.choice([16, 128, 235])
                    elif shape == "black_box":
                        y[y0: y0 + H // 4, x0: x0 + W // 3] = 0
                    else:
                        y[: 8 * rng.randrange(2, 5)] = 0
        tag = (it, W, H, T, S, cfg, noise, still, shape)
        g = pkg.FerHip(W, H, S, **cfg)
        streams, rec = g.encode_streams(frames, want_recon=True)
        assert g.status() == [0] * S, tag
        counts = g.stats()
        g.close()
        for s in range(S):
            o = fo.Oracle(W, H, **cfg)
            ref, rr = o.encode_stream(frames[:, s])
            oc = o.stats()
            o.close()
            assert streams[s] == ref, tag
            assert np.array_equal(rec[:, s], rr), tag
            assert list(counts[s]) == list(oc), tag
        out, pics, w, h = pkg.decode_streams(streams, T)
        assert pics == [T] * S, tag
        for s in range(S):
            assert np.array_equal(out[:, s], np.stack(fo.decode_stream_md5(streams[s])[1])), tag
