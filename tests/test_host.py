"""CPU tests of the host side: C-ABI exports, synthetic source, shard merge (gloo, 2 ranks)."""
import ctypes as C
import os
import re
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.load_library()  # raises if libferhip.so is missing: no fallback
    hdr = (ROOT / "include" / "ferhip.h").read_text()
    names = sorted(set(re.findall(r"\b(ferhip_[a-z_0-9]+)\s*\(", hdr)) - {"ferhip_ctx"})
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ferhip.h but not exported"
    legacy = (ROOT / "include" / "ferhip_legacy.h")
    if legacy.exists():
        for n in re.findall(r"^\s*(?:void|int|unsigned int)\s+([A-Za-z_0-9]+)\s*\(", legacy.read_text(), re.M):
            assert hasattr(lib, n), f"legacy symbol {n} not exported"


def test_create_without_gpu_fails_loudly(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.FerHipError):
        pkg.FerHip(176, 144, 1)


def test_encoder_picture_size_limit_is_reported(pkg):
    """The motion kernels address a stream's planes with 24-bit multiplies: an encoder context refuses pictures whose padded
    plane reaches 2^24 samples with FERHIP_E_UNSUP (-4) before anything touches the device (4K is half of the limit)."""
    import ctypes as C
    lib = pkg.load_library()
    ctx = C.c_void_p()
    from h264_fer_amd.ferhip import Params
    p = Params(12, 0, 32, 3, 30)
    assert lib.ferhip_create(C.byref(ctx), 4096, 4096, 1, C.byref(p)) == -4
    assert lib.ferhip_create(C.byref(ctx), 4095, 4096, 1, C.byref(p)) == -1  # not a multiple of 16: FERHIP_E_ARG


def test_synthetic_source_three_implementations_agree(pkg, fo):
    import torch
    from h264_fer_amd.synth import gen_frames_torch
    a = gen_frames_torch(96, 64, 3, 2, "cpu", 77, 2).numpy()
    for s in range(2):
        for t in range(3):
            f = pkg.gen_frame(96, 64, t, 77 + s, 2)
            assert np.array_equal(a[t, s], f)
            assert np.array_equal(f, fo.gen_frame(96, 64, t, 77 + s, 2))
    assert a[:, :, : 96 * 64].min() >= 16  # luma > 0 keeps every 8x8 sum out of the reference's bucket-0 defect


def test_crop_matches_reference_reader(pkg):
    f = pkg.gen_frame(1920, 1080, 0, 1, 0)
    c, W, H = pkg.crop_to_mb(f, 1920, 1080)
    assert (W, H) == (1920, 1072) and c.size == 1920 * 1072 * 3 // 2
    assert np.array_equal(c[:1920], f[4 * 1920: 5 * 1920])


def test_gop_merge_equals_single_run(pkg, fo):
    """Closed GOPs encoded as separate streams and merged == one run of the reference algorithm
    (no P_Skip in the last P picture of a GOP, see DESIGN.md 'sharding caveat')."""
    W, H = 64, 48
    frames = np.stack([pkg.gen_frame(W, H, t, 9, 2) for t in range(6)])
    o = fo.Oracle(W, H, qp=12, window=16, maxdiff=3, intra_every=3)
    whole, _ = o.encode_stream(frames)
    o.close()
    parts = []
    for g in range(2):
        o = fo.Oracle(W, H, qp=12, window=16, maxdiff=3, intra_every=3)
        s, _ = o.encode_stream(frames[3 * g: 3 * g + 3])
        o.close()
        parts.append(s)
    assert pkg.merge_gop_streams(parts) == whole


_WORKER = r'''
import os, sys, pickle
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import numpy as np, torch.distributed as dist
from conftest import load_pkg
import fo_py
pkg = load_pkg()
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
W, H, G, T = 64, 48, 4, 2
frames = np.stack([pkg.gen_frame(W, H, t, 5, 2) for t in range(G * T)])
mine = {}
for g in pkg.gops_of_rank(G, world, rank):     # frames shard by closed GOP, no data-path collective
    o = fo_py.Oracle(W, H, qp=12, window=16, maxdiff=3, intra_every=T)
    mine[g], _ = o.encode_stream(frames[g * T:(g + 1) * T]); o.close()
gathered = [None] * world
dist.all_gather_object(gathered, mine)          # host-side concatenation only
import torch
t = torch.tensor([float(rank + 1)]); dist.all_reduce(t, op=dist.ReduceOp.MAX)  # bench.py's max-over-ranks timing
if rank == 0:
    allg = {}
    for d in gathered: allg.update(d)
    merged = pkg.merge_gop_streams([allg[g] for g in range(G)])
    o = fo_py.Oracle(W, H, qp=12, window=16, maxdiff=3, intra_every=T)
    whole, _ = o.encode_stream(frames); o.close()
    assert merged == whole, "sharded merge differs from single run"
    assert t.item() == world
    print("OK", len(merged))
dist.destroy_process_group()
'''


def test_two_rank_gloo_gop_sharding(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), str(ROOT / "tests"), str(ROOT / "oracle")],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "OK" in outs[0]


def test_bench_rank_plumbing_at_world_8():
    """`bench.py --gpus 8` cold: the parent spawns eight ranks before anything touches a GPU, they rendezvous on 127.0.0.1,
    reduce the step time with MAX and gather + merge the eight 4K GOP streams in GOP order (rehearsal mode: gloo, no GPU, no
    encoder -- placeholder bytes stand for the streams)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "FER_BENCH_CHILD")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--config", "4k", "--rehearse-ranks", "1", "--dist-backend", "gloo"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["rehearsal"] and line["world"] == 8 and line["max_dt"] == 8.0 and line["gops_in_order"]
    assert line["gop_owner"] == [g % 8 for g in range(8)]   # GOP g belongs to rank g % world (gops_of_rank)


def _write_y4m(path, W, H, frames, params=False):
    with open(path, "wb") as f:
        f.write(b"YUV4MPEG2 W%d H%d F25:1 Ip A1:1 C420jpeg\n" % (W, H))
        for k, fr in enumerate(frames):
            f.write(b"FRAME Ix\n" if (params and k == 1) else b"FRAME\n")
            f.write(fr.tobytes())


def test_y4m_reader_crops_like_the_reference(pkg, tmp_path):
    """ferhip_y4m_* = LoadY4MHeader / ReadFromY4M (F/fileIO.cpp:228-346): size from the W / H tokens, centre crop to
    multiples of 16 (chroma at half the offsets), end of stream on a short read."""
    W, H, T = 100, 70, 3
    frames = [pkg.gen_frame(W, H, t, 3, 2) for t in range(T)]
    p = tmp_path / "a.y4m"
    _write_y4m(p, W, H, frames, params=True)
    with open(p, "ab") as f:
        f.write(b"FRAME\n" + b"\x00" * 100)   # a truncated picture ends the stream
    r = pkg.Y4MReader(p)
    assert r.in_size == (W, H) and (r.W, r.H) == (96, 64)
    for t in range(T):
        want, w, h = pkg.crop_to_mb(frames[t], W, H)
        got = r.read()
        assert got is not None and np.array_equal(got, want), t
    assert r.read() is None
    r.close()


def test_legacy_file_io_names(pkg, tmp_path):
    """The reference's own names (LoadY4MHeader, ReadFromY4M, writeToY4M, writeToYUV over `frame`, `yuvinput`,
    `yuvoutput`, F/fileIO.h) as exported by libferhip.so: read a Y4M, write it back, compare bytes."""
    lib = pkg.load_library()
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]

    class Frame(C.Structure):
        _fields_ = [("Lwidth", C.c_int), ("Lheight", C.c_int), ("Cwidth", C.c_int), ("Cheight", C.c_int),
                    ("L", C.POINTER(C.c_ubyte)), ("C", C.POINTER(C.c_ubyte) * 2)]

    W, H, T = 64, 48, 2
    frames = [pkg.gen_frame(W, H, t, 8, 1) for t in range(T)]
    src, dst = tmp_path / "in.y4m", tmp_path / "out.y4m"
    _write_y4m(src, W, H, frames)
    lib.ferhip_fileio_reset()
    frame = Frame.in_dll(lib, "frame")
    frame.L = None
    frame.C[0] = None
    frame.C[1] = None
    fin = libc.fopen(str(src).encode(), b"rb")
    fout = libc.fopen(str(dst).encode(), b"wb")
    C.c_void_p.in_dll(lib, "yuvinput").value = fin
    C.c_void_p.in_dll(lib, "yuvoutput").value = fout
    lib.LoadY4MHeader()
    assert (frame.Lwidth, frame.Lheight, frame.Cwidth, frame.Cheight) == (W, H, W // 2, H // 2)
    assert C.c_int.in_dll(lib, "inputWidth").value == W
    n = 0
    while lib.ReadFromY4M() != -1:
        assert bytes(frame.L[: W * H]) == frames[n][: W * H].tobytes()
        lib.writeToY4M()
        n += 1
    assert n == T
    libc.fclose(fin)
    libc.fclose(fout)
    C.c_void_p.in_dll(lib, "yuvinput").value = None
    C.c_void_p.in_dll(lib, "yuvoutput").value = None
    want = b"YUV4MPEG2 C420jpeg W%d H%d F24:1 Ip A1:1\n" % (W, H) + b"".join(b"FRAME\n" + f.tobytes() for f in frames)
    assert dst.read_bytes() == want
    lib.ferhip_fileio_reset()


def test_synthetic_p_slices_stay_in_sync_with_the_oracle_parser(pkg, fo):
    """tests/pslice_synth.py (row f4's generator): with every mvd zero, whatever else the slices carry (sub-macroblock
    types, te() reference indices, list modification, all-zero residual blocks), every P picture predicts the IDR
    picture unchanged -- which it only does if generator and oracle parser agree on every bit."""
    import pslice_synth as ps
    base = (Path(__file__).parent / "golden" / "qcif_ippp_4f_qp12_w16.264").read_bytes()
    plan = [dict(mvd_range=0), dict(override=True, active=1, mvd_range=0), dict(modification=[], mvd_range=0),
            dict(mvd_range=0), dict(modification=[(0, 0)], mvd_range=0), dict(override=True, active=0, mvd_range=0)]
    n, frames, _ = fo.decode_stream_md5(ps.make_stream(pkg.split_nals, base, 2, plan))
    assert n == len(plan) + 1
    for f in frames[1:]:
        assert np.array_equal(f, frames[0])
