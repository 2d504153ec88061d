"""Dev tool: the noisiest content at the lowest GUI quantiser (largest pictures in bytes) and the smoothest at the
highest, 1080p, against the oracle."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "oracle"))
from conftest import load_pkg
import fo_py
pkg = load_pkg()
W, H, T = 1920, 1072, 2
for qp, noise in ((10, 12), (30, 0)):
    frames = np.stack([pkg.gen_frame(W, H, t, 4242, noise) for t in range(T)])[:, None]
    g = pkg.FerHip(W, H, 1, qp=qp, window=32, maxdiff=3, intra_every=30)
    streams, rec = g.encode_streams(frames, want_recon=True)
    st = g.status()
    g.close()
    o = fo_py.Oracle(W, H, qp=qp, window=32, maxdiff=3, intra_every=30)
    ref, rr = o.encode_stream(frames[:, 0]); o.close()
    print("qp", qp, "noise", noise, "bytes per picture", len(streams[0]) // T, "status", st, "bits equal", streams[0] == ref,
          "recon equal", np.array_equal(rec[:, 0], rr), flush=True)
