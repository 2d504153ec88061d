"""Dev tool: what the stage-2 walk leaves behind on content with a large flat area (bench.py --content flat-half):
per 8x8 partition the candidate count, and for crowded partitions the summary (last step, distance bound, zeros found)."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
from conftest import load_pkg
pkg = load_pkg()
W, H = 1920, 1072
fr = np.stack([pkg.gen_frame(W, H, t, 1234, 2) for t in range(3)])
Y = fr[:, :W * H].reshape(3, H, W)
Y[:, :, :W // 2] = 100
g = pkg.FerHip(W, H, 1, qp=12, window=32, maxdiff=3, intra_every=30)
g.profile(True)
t0 = time.time()
streams, rec = g.encode_streams(fr[:, None].copy(), want_recon=True)
print("encode 3 pictures: %.2f s" % (time.time() - t0), {k: round(v[0], 1) for k, v in g.get_profile().items() if v[0] > 1})
n = g.read("ST2N").reshape(-1)
st2 = g.read("ST2").reshape(-1, 384, 2)
cr = n > 384
print("partitions", n.size, "crowded", int(cr.sum()))
jend, dmin, zc = st2[cr, 40, 0], st2[cr, 40, 1], st2[cr, 41, 0]
print("crowded: jend hist", np.bincount(np.minimum(jend, 5)), "dmin hist", np.bincount(np.minimum(dmin, 8)), "zeros hist", np.bincount(zc, minlength=34)[[0, 1, 2, 8, 16, 32, 33]])
r1 = rec[1, 0, :W * H].reshape(H, W)[:, :W // 2]
print("recon of the flat half (picture 1): values", np.unique(r1, return_counts=True))
b = r1[:, :].reshape(H // 8, 8, W // 16, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
print("aligned 8x8 blocks exactly flat 100:", float((b == 100).all(1).mean()), "flat any value:", float((b == b[:, :1]).all(1).mean()))
print("ST2N of crowded partitions: min", n[cr].min(), "median", int(np.median(n[cr])), "max", n[cr].max())
nm = W // 16
idx = np.nonzero(cr)[0]
mbi, part = idx // 4, idx % 4
px = (mbi % nm) * 16 + (part & 1) * 8
print("crowded partitions by x: ", np.histogram(px, bins=[0, 200, 400, 600, 800, 960, 1100, 1300, 1920])[0])
print("jend of crowded: min", jend.min(), "median", int(np.median(jend)), "max", jend.max(), " dmin median", int(np.median(dmin)))
import collections
print("per-kernel ms again:", {k: round(v[0], 1) for k, v in g.get_profile(reset=False).items()})
import os
if int(os.environ.get("FER_DBG", "0")) & 128:  # a -DFER_PROBE build: where the middle partition row of the chain spent its time
    t = g.read("TIMING")
    for role in (0, 1):
        r = t[role * 8:role * 8 + 8]
        nn = max(int(r[7]), 1)
        print("probe row, wavefront %d, us per partition: " % role + "  ".join(
            "%s %.2f" % (nm, r[k] / nn / 100) for k, nm in enumerate(["prefetch+poll", "stage", "barrier1", "merge+publish", "barrier2"])))
    for k, nm in ((40, "slowest stage-1 call"), (41, "slowest stage-2/3 call")):
        v = int(t[k]) & 0xffffffffffffffff
        print(nm, "%.1f us at partition column %d row %d" % ((v >> 24) / 100, (v >> 12) & 4095, v & 4095))
    print("stage-2/3 total %.1f ms, calls over 200 us: %d" % (int(t[42]) / 1e5, int(t[43])))
fl = st2[cr, 41, 1]
print("crowded: general-bound fallback", int(((fl >> 30) & 1).sum()), " listed candidates: median", int(np.median(fl & 0xffff)), "max", int((fl & 0xffff).max()),
      " big slices hist", np.bincount((fl >> 16) & 15, minlength=5))
fb = np.nonzero(cr)[0][((fl >> 30) & 1) == 1]
if fb.size:
    print("fallback partitions: x", np.unique(((fb // 4) % nm) * 16 + (fb % 4 & 1) * 8)[:40], " ST2N", n[fb][:10], " summary of the first:", st2[fb[0], 40:47].tolist())
