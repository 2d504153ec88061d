"""Dev tool: randomised parity sweep of the GPU encoder (and decoder) against the oracle at small picture sizes --
seeds, noise levels, qp, WindowSize, MAXDIFF (incl. adaptive), IntraEvery, sizes.  Prints the failing configurations."""
import sys, random
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "oracle"))
from conftest import load_pkg
import fo_py
pkg = load_pkg()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
bad = 0
for it in range(N):
    W, H = rng.choice([(176, 144), (352, 288), (64, 48), (128, 96), (320, 240), (16, 32), (48, 16),
                       (16, 2064), (32, 4112), (1024, 16), (640, 32), (16, 528)])  # tall / wide: the tile shapes of the first radix pass
    cfg = dict(qp=rng.choice([10, 12, 16, 20, 24, 28, 30]), window=rng.choice([16, 32, 32, 32, 48]),
               maxdiff=rng.choice([3, 3, -1, 0, 6]), intra_every=rng.choice([30, 30, 3, 2]))
    T, S = rng.choice([3, 4, 5]), rng.choice([1, 2, 3, 3, 17, 19])   # >= 16 streams: the eight ticket queues of the motion chain
    tune = (rng.choice([1, 2, 5, 8, 64, 6144]), rng.choice([1, 2, 4, 64]), rng.choice([0, 1, 1, 1]))
    noise = rng.choice([0, 1, 2, 4])
    seeds = [rng.randrange(1, 10000) for _ in range(S)]
    still = rng.random() < 0.25  # static content: P_Skip heavy
    frames = np.stack([np.stack([pkg.gen_frame(W, H, 0 if still else t, seeds[s], noise if not still else 0) for s in range(S)])
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
    g = pkg.FerHip(W, H, S, **cfg)
    g.tune(pkg.TUNE_RESOLVE_WGS, tune[0])
    g.tune(pkg.TUNE_RESOLVE_GROUP, tune[1])
    g.tune(pkg.TUNE_SPECULATE, tune[2])
    streams, rec = g.encode_streams(frames, want_recon=True)
    st = g.status()
    g.close()
    ok = st == [0] * S
    why = []
    for s in range(S):
        o = fo_py.Oracle(W, H, **cfg)
        ref, rr = o.encode_stream(frames[:, s])
        o.close()
        if streams[s] != ref:
            why.append("bits%d" % s)
        if not np.array_equal(rec[:, s], rr):
            bt = [t for t in range(T) if not np.array_equal(rec[t, s], rr[t])]
            why.append("recon%d@%s" % (s, bt))
    out, pics, w, h = pkg.decode_streams(streams, T)
    # the decoder is checked against the ORACLE decoder: the reference decoder's quirks (stale chroma AC levels in
    # macroblocks without residual) can make its output differ from the encoder's reconstruction
    rec = np.stack([np.stack(fo_py.decode_stream_md5(streams[s])[1]) for s in range(S)], axis=1)
    if pics != [T] * S:
        why.append("pics%s" % pics)
    elif not np.array_equal(out, rec):
        bt = [(t, s) for t in range(T) for s in range(S) if not np.array_equal(out[t, s], rec[t, s])]
        why.append("decode@%s" % bt[:6])
        t, s_ = bt[0]
        a = out[t, s_][:W * H].reshape(H, W).astype(int); b = rec[t, s_][:W * H].reshape(H, W).astype(int)
        dm = (a != b).reshape(H // 16, 16, W // 16, 16).any(axis=(1, 3))
        ys, xs = np.nonzero(dm)
        ca = out[t, s_][W * H:].astype(int); cb = rec[t, s_][W * H:].astype(int)
        why.append("lumaMBs=%d first=%s maxdiff=%d chroma_diff=%d" % (dm.sum(), list(zip(xs[:5].tolist(), ys[:5].tolist())),
                                                                       np.abs(a - b).max(), int((ca != cb).sum())))
    ok &= not why
    if not ok:
        bad += 1
    print(("ok  " if ok else "FAIL"), W, H, T, S, cfg, "noise", noise, "still", still, shape, "tune", tune, "status", st, " ".join(why), flush=True)
print("failures:", bad, "of", N)
