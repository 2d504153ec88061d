"""Dev tool: letterboxed 1080p content (128 rows of Y = 16 top and bottom): stage-2 candidate counts and a timing of the
P picture, to look at the crowded-partition paths."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
from conftest import load_pkg
pkg = load_pkg()
W, H = 1920, 1072
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
kind = sys.argv[2] if len(sys.argv) > 2 else "letterbox"
fr = []
for t in range(2):
    f = pkg.gen_frame(W, H, t, 1234, 2).copy()
    y = f[: W * H].reshape(H, W)
    if kind == "letterbox":
        y[:128] = 16
        y[-128:] = 16
    elif kind == "flat-half":
        y[:, : W // 2] = 100
    fr.append(f)
frames = np.repeat(np.stack(fr)[:, None, :], S, axis=1).copy()
g = pkg.FerHip(W, H, S, qp=12, window=32, maxdiff=3, intra_every=30)
t0 = time.time()
streams, _ = g.encode_streams(frames)
print("encode", round(time.time() - t0, 3), "s", "status", g.status()[:2], "stats", g.stats()[0])
n = g.read("ST2N").reshape(S, -1)[0]
print("st2n: partitions", n.size, "crowded", int((n > 384).sum()), "median", int(np.median(n)), "max", int(n.max()))
