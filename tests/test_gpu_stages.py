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


def test_refprep_matches_oracle(pkg, fo):
    f0, f1 = _two_frames(pkg)
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
