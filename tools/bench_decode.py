"""Decode-twin throughput (row a19): S 1080p IPPP streams produced by the GPU encoder are decoded side by side by
ferhip_decode_streams; decoded pictures stay on the device (the Annex-B input is host memory, as the reference
reads it).  Prints macroblocks/s; checks one stream's luma against the encoder's reconstruction first (chroma may differ
by the reference decoder's stale-ChromaACLevel quirk; tests compare the decoder with the oracle decoder)."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
from conftest import load_pkg
pkg = load_pkg()
W, H = 1920, 1072
S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 30
base = np.stack([pkg.gen_frame(W, H, t, 1234, 2) for t in range(T)])
g = pkg.FerHip(W, H, 1, qp=12, window=32, maxdiff=3, intra_every=30)
streams, rec = g.encode_streams(base[:, None].copy(), want_recon=True)
g.close()
out, pics, w, h = pkg.decode_streams(streams, T)
assert pics == [T] and np.array_equal(out[:, 0, :W * H], rec[:, 0, :W * H]), "decoded luma differs from the encoder reconstruction"
many = [streams[0]] * S
nmb = (W // 16) * (H // 16)
for rep in range(2):
    t0 = time.time()
    _, pics, _, _ = pkg.decode_streams(many, T, want_pictures=False)
    dt = time.time() - t0
    print("decode %d streams x %d pictures: %.3f s, %.0f macroblocks/s (%.0f pictures/s), %d bytes per stream"
          % (S, T, dt, S * T * nmb / dt, S * T / dt, len(streams[0])), flush=True)
