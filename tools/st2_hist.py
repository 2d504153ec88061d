"""Dev tool: distribution of the stage-2 candidate counts per 8x8 partition on the benchmark content."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
from conftest import load_pkg
pkg = load_pkg()
W, H = 1920, 1072
f0 = pkg.gen_frame(W, H, 0, 1234, 2); f1 = pkg.gen_frame(W, H, 1, 1234, 2)
g = pkg.FerHip(W, H, 1, qp=12, window=32, maxdiff=3, intra_every=30)
g.set_reference(f0[None]); g.set_frames(f1[None])
g.fill_interpolated(); g.inter_encoding()
n = g.read("ST2N")
print("partitions", n.size, "min", n.min(), "median", int(np.median(n)), "mean", round(float(n.mean()), 1), "p90", int(np.percentile(n, 90)),
      "p99", int(np.percentile(n, 99)), "max", n.max())
for cap in (192, 256, 320, 384):
    print("  > %d: %.2f %%" % (cap, 100.0 * (n > cap).mean()))
