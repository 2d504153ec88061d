"""Dev helper (GPU box): the 1080p decode leg of bench.py's secondary configurations alone."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench
pkg = bench.load_pkg() if hasattr(bench, "load_pkg") else None
if pkg is None:
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
    from conftest import load_pkg
    pkg = load_pkg()
from h264_fer_amd.synth import gen_frames_torch
dev = torch.device("cuda:0")
import os
W, H, S, T = 1920, 1072, 16, 30
TD = int(os.environ.get("DEC_T", T))  # pictures to decode of each stream (1 = the I pictures alone)
fr = gen_frames_torch(W, H, T, S, dev, seed=1234, noise=2).cpu().numpy()
e = pkg.FerHip(W, H, S, qp=12, window=32, maxdiff=3, intra_every=30)
streams, _ = e.encode_streams(fr)
nmb = e.nmb
e.close()
for reps in [int(a) for a in sys.argv[1:]] or [8]:
    batch = streams * reps
    pkg.decode_streams(batch, TD, want_pictures=False)
    for _ in range(2):
        t0 = time.perf_counter()
        _, pics, _, _ = pkg.decode_streams(batch, TD, want_pictures=False)
        dt = time.perf_counter() - t0
        print("streams", len(batch), "decode MB/s", round(len(batch) * TD * nmb / dt, 1), "s", round(dt, 3), "all", pics == [TD] * len(batch), flush=True)
    pkg.load_library().ferhip_decode_release()
