"""Dev helper: time k_me_pre / k_me_resolve phases with stage-skipping FER_DBG masks."""
import os, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
from conftest import load_pkg
pkg = load_pkg()
W, H, S = 1920, 1072, int(os.environ.get("FER_TP_STREAMS", "16"))
f0 = pkg.gen_frame(W, H, 0, 1234, 2); f1 = pkg.gen_frame(W, H, 1, 1234, 2)
g = pkg.FerHip(W, H, S, qp=12, window=32, maxdiff=3, intra_every=30)
g.set_reference(np.repeat(f0[None], S, 0)); g.set_frames(np.repeat(f1[None], S, 0))
g.fill_interpolated()
g.profile(True)
for _ in range(3):
    g.inter_encoding()
if int(os.environ.get("FER_DBG", "0")) & 128:  # needs a build with EXTRA=-DFER_PROBE
    t = g.read("TIMING")
    for role in (0, 1):
        r = t[role * 8:role * 8 + 8]
        n = max(int(r[7]), 1)
        print("probe row, wavefront %d, us per partition: " % role + "  ".join(
            "%s %.2f" % (nm, r[k] / n / 100) for k, nm in enumerate(["prefetch+poll", "stage", "barrier1", "merge+publish", "barrier2"])))
if int(os.environ.get("FER_DBG", "0")) & 128:
    t = g.read("TIMING")
    n = max(int(t[39]), 1)
    print("k_me_pre sample of %d partitions, us each: " % n + "  ".join(
        "%s %.2f" % (nm, t[32 + k] / n / 100) for k, nm in enumerate(["sums", "wide", "local", "select", "sads"])))
print(os.environ.get("FER_DBG", "0"), {k: round(v[0] / 3, 2) for k, v in g.get_profile().items() if v[0] > 0})
