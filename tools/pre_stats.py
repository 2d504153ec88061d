"""Dev helper (FER_STATS build): how often k_me_pre's lower-bound pruning applies."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
from conftest import load_pkg
pkg = load_pkg()
W, H, S = 1920, 1072, 4
fr = np.stack([np.stack([pkg.gen_frame(W, H, t, 1234 + s, 2) for s in range(S)]) for t in range(3)])
g = pkg.FerHip(W, H, S, qp=12, window=32, maxdiff=3, intra_every=30)
g.encode_streams(fr)
t = g.read("TIMING")
n = max(int(t[48]), 1)
print("partitions", n, "fallback", t[49] / n, "avg survivors", t[50] / n, "no threshold", t[51] / n, "avg T", t[52] / n)
