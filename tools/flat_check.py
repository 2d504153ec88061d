"""Dev tool: content with large flat areas (every 8x8 block has the same features) -- does the stage-2 candidate
list overflow (status bit 0)?"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "oracle"))
from conftest import load_pkg
import fo_py
pkg = load_pkg()
W, H, T = (int(sys.argv[1]), int(sys.argv[2]), 3) if len(sys.argv) > 2 else (352, 288, 3)
import time
for name in ("flat", "half", "bars", "black_bars", "black_quarter", "black_white"):
    frames = []
    for t in range(T):
        f = pkg.gen_frame(W, H, t, 77, 2).copy()
        y = f[:W * H].reshape(H, W)
        if name == "flat":
            y[:] = 128
        elif name == "half":
            y[:, : W // 2] = 100
        elif name == "bars":
            y[:32] = 16
            y[-32:] = 16
        elif name == "black_bars":      # sum-0 blocks: the reference's bucket-0 defect
            y[:40] = 0
            y[-40:] = 0
        elif name == "black_quarter":
            y[: H // 2, : W // 2] = 0
        else:
            y[:48] = 0
            y[-48:] = 255
        frames.append(f)
    frames = np.stack(frames)[:, None]
    g = pkg.FerHip(W, H, 1, qp=20, window=32, maxdiff=3, intra_every=30)
    try:
        t0 = time.time()
        streams, rec = g.encode_streams(frames, want_recon=True)
        print(name, 'encode seconds', round(time.time() - t0, 3), flush=True)
        o = fo_py.Oracle(W, H, qp=20, window=32, maxdiff=3, intra_every=30)
        ref, rr = o.encode_stream(frames[:, 0]); o.close()
        print(name, "status", g.status(), "bits equal", streams[0] == ref, "recon equal", np.array_equal(rec[:, 0], rr))
    except Exception as e:
        print(name, "FAILED:", e, "status", g.status())
    g.close()
