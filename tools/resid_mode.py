"""Dev helper (GPU box): does the two-mode behaviour of k_p_resid follow the process or the context (= its allocations)?
Creates the 256-stream 1080p context several times in one process and times k_p_resid each time."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import torch
from conftest import load_pkg
pkg = load_pkg()
from h264_fer_amd.synth import gen_frames_torch
dev = torch.device("cuda:0")
W, H, S, T = 1920, 1072, 256, 4
fr = gen_frames_torch(W, H, T, S, dev, seed=1234, noise=2)
pad = []
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    e = pkg.FerHip(W, H, S, qp=12, window=32, maxdiff=3, intra_every=30)
    e.profile(True)
    for t in range(T):
        e.set_frames_device(fr[t].data_ptr())
        e.encode_picture_device(None)
    e.sync()
    p = e.get_profile(reset=True)
    print("trial", trial, "p_resid us", round(p["p_resid"][0] * 1e3 / max(p["p_resid"][1], 1)), "me_walk us", round(p["me_walk"][0] * 1e3 / max(p["me_walk"][1], 1)),
          "status_ok", not any(e.status()), flush=True)
    e.close()
    pad.append(torch.empty((trial + 1) * 37 * 1024 * 1024, dtype=torch.uint8, device=dev))  # shift what the next context gets
