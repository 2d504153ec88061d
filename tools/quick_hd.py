"""Dev helper: time the GPU encoder at 1080p and (optionally) check stream 0 against the oracle."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
from conftest import load_pkg
pkg = load_pkg()
W, H = 1920, 1072
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
T = int(sys.argv[2]) if len(sys.argv) > 2 else 3
check = len(sys.argv) > 3 and sys.argv[3] == "check"
t0 = time.time()
base = np.stack([pkg.gen_frame(W, H, t, 1234, 2) for t in range(T)])
frames = np.repeat(base[:, None, :], S, axis=1).copy()
print("gen", time.time() - t0, flush=True)
g = pkg.FerHip(W, H, S, qp=12, window=32, maxdiff=3, intra_every=30)
t0 = time.time()
streams, rec = g.encode_streams(frames, want_recon=True)
dt = time.time() - t0
print("first run", dt, "MB/s", S * T * g.nmb / dt, "bytes", [len(s) for s in streams][:4], "status", g.status(), flush=True)
g.close()
g = pkg.FerHip(W, H, S, qp=12, window=32, maxdiff=3, intra_every=30)
t0 = time.time()
streams2, _ = g.encode_streams(frames, want_recon=False)
dt = time.time() - t0
print("second run", dt, "MB/s", S * T * g.nmb / dt, flush=True)
assert streams2 == streams and all(s == streams[0] for s in streams)
print("stats", g.stats()[0])
if check:
    import fo_py
    o = fo_py.Oracle(W, H, qp=12, window=32, maxdiff=3, intra_every=30)
    t0 = time.time()
    rb, rr = o.encode_stream(base)
    print("oracle", time.time() - t0, len(rb), flush=True)
    print("bitstream equal:", rb == streams[0], "recon equal:", np.array_equal(rr, rec[:, 0]))
