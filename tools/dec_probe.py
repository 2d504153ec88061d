"""Dev helper (GPU box, -DFER_PROBE library): phase times of k_dec_parse for the I picture alone and for a GOP."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import torch
from conftest import load_pkg
pkg = load_pkg()
from h264_fer_amd.synth import gen_frames_torch
dev = torch.device("cuda:0")
W, H, S, T = 1920, 1072, 16, 30
fr = gen_frames_torch(W, H, T, S, dev, seed=1234, noise=2).cpu().numpy()
e = pkg.FerHip(W, H, S, qp=12, window=32, maxdiff=3, intra_every=30)
streams, _ = e.encode_streams(fr)
e.close()
for reps, t in ((2, 1), (2, 2), (8, 30)):
    batch = streams * reps
    for _ in range(2):
        t0 = time.perf_counter()
        pkg.decode_streams(batch, t, want_pictures=False)
        print("streams", len(batch), "pictures", t, "s", round(time.perf_counter() - t0, 3), flush=True)
    pkg.load_library().ferhip_decode_release()
