"""Dev tool: a completely black sequence (every 8x8 sum is 0, more than half of the array mis-filed: the reference reads
past its arrays there, so there is nothing to compare with) must still encode without error or hang."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
from conftest import load_pkg
pkg = load_pkg()
W, H, T = 352, 288, 4
f = pkg.gen_frame(W, H, 0, 1, 0).copy()
f[: W * H] = 0
frames = np.stack([f] * T)[:, None]
g = pkg.FerHip(W, H, 1, qp=20, window=32, maxdiff=3, intra_every=30)
t0 = time.time()
streams, rec = g.encode_streams(frames, want_recon=True)
print("black: %.2f s, %d bytes, status %s, recon max %d" % (time.time() - t0, len(streams[0]), g.status(), int(rec[:, 0, : W * H].max())))
out, pics, w, h = pkg.decode_streams(streams, T)
print("decoded", pics, "luma equal", np.array_equal(out[:, 0, : W * H], rec[:, 0, : W * H]))
