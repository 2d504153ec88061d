"""Deterministic synthetic 4:2:0 source pictures (integer-only; numpy twin of oracle/fo_gen.c).

  lcg(s)   = s*6364136223846793005 + 1442695040888963407 (mod 2^64)
  tex[r][q] (64x64) = ((lcg chain from seed) >> 33) % 25 - 12
  noise(x,y,t): s = (seed ^ (0x9E3779B97F4A7C15*(t+1))) + y*W + x; s = lcg(s); s ^= s>>29; s = lcg(s)
                n = (s>>33) % (2A+1) - A
  Y = clip(16,235, 40 + tri(x+2t,192) + tri(y+t,128) + tex[(y+t)&63][(x+2t)&63] + n)
  U = 104 + tri(xc+t,96)//2 ; V = 104 + tri(yc+t,96)//2 ; tri(v,p) = m if (m:=v%p) < p//2 else p-m
"""
import numpy as np

_A = np.uint64(6364136223846793005)
_C = np.uint64(1442695040888963407)
_G = 0x9E3779B97F4A7C15
_M = (1 << 64) - 1


def _lcg(s):
    return s * _A + _C


def _tri(v, p):
    m = v % p
    return np.where(m < p // 2, m, p - m)


_tex_cache = {}


def _tex(seed):
    if seed not in _tex_cache:
        s = seed & _M
        t = np.empty(4096, np.int64)
        for i in range(4096):
            s = (s * 6364136223846793005 + 1442695040888963407) & _M
            t[i] = (s >> 33) % 25 - 12
        _tex_cache[seed] = t.reshape(64, 64)
    return _tex_cache[seed]


def gen_frame(W, H, t, seed=1234, noise=2):
    """One I420 picture (W*H*3/2 bytes) of the synthetic sequence at time t."""
    tex = _tex(seed)
    y, x = np.mgrid[0:H, 0:W].astype(np.int64)
    with np.errstate(over="ignore"):
        base = np.uint64((seed ^ ((_G * (t + 1)) & _M)) & _M)
        z = base + (y * W + x).astype(np.uint64)
        z = _lcg(z)
        z ^= z >> np.uint64(29)
        z = _lcg(z)
    n = ((z >> np.uint64(33)) % np.uint64(2 * noise + 1)).astype(np.int64) - noise if noise > 0 else 0
    v = 40 + _tri(x + 2 * t, 192) + _tri(y + t, 128) + tex[(y + t) & 63, (x + 2 * t) & 63] + n
    Y = np.clip(v, 16, 235).astype(np.uint8)
    yc, xc = np.mgrid[0:H // 2, 0:W // 2].astype(np.int64)
    U = (104 + _tri(xc + t, 96) // 2).astype(np.uint8)
    V = (104 + _tri(yc + t, 96) // 2).astype(np.uint8)
    return np.concatenate([Y.ravel(), U.ravel(), V.ravel()])


def gen_frames(W, H, n, seed=1234, noise=2, t0=0):
    return np.stack([gen_frame(W, H, t0 + t, seed, noise) for t in range(n)])


def crop_to_mb(frame, W, H):
    """Centre crop to multiples of 16 like ReadFromY4M (F/fileIO.cpp:242-243,290-333)."""
    Wc, Hc = W & ~15, H & ~15
    if (Wc, Hc) == (W, H):
        return frame, W, H
    Y = frame[: W * H].reshape(H, W)
    U = frame[W * H: W * H * 5 // 4].reshape(H // 2, W // 2)
    V = frame[W * H * 5 // 4:].reshape(H // 2, W // 2)
    ct, cl = (H - Hc) >> 1, (W - Wc) >> 1
    Y = Y[ct:ct + Hc, cl:cl + Wc]
    U = U[ct >> 1:(ct >> 1) + Hc // 2, cl >> 1:(cl >> 1) + Wc // 2]
    V = V[ct >> 1:(ct >> 1) + Hc // 2, cl >> 1:(cl >> 1) + Wc // 2]
    return np.concatenate([Y.ravel(), U.ravel(), V.ravel()]), Wc, Hc


def gen_frames_torch(W, H, n, S, device, seed=1234, noise=2, seeds=None, t0s=None):
    """Same pictures as gen_frame(), built with torch integer ops on `device`: [n][S][W*H*3/2] uint8.

    Stream s uses seed + s (or seeds[s]) and starts at time t0s[s] (default 0): closed GOPs of ONE sequence are
    streams with the same seed and t0 = first picture of the GOP.  int64 arithmetic wraps like uint64; the logical
    right shifts are emulated with masks."""
    import torch

    A = 6364136223846793005
    Cc = 1442695040888963407

    def s64(v):  # python int -> value representable in int64 (two's complement)
        v &= _M
        return v - (1 << 64) if v >= (1 << 63) else v

    def lcg(z):
        return z * A + Cc

    def lsr(z, k):  # logical shift right of an int64 tensor
        return (z >> k) & ((1 << (64 - k)) - 1)

    ys = W * H
    fsz = ys * 3 // 2
    out = torch.empty((n, S, fsz), dtype=torch.uint8, device=device)
    y = torch.arange(H, device=device, dtype=torch.int64).view(H, 1)
    x = torch.arange(W, device=device, dtype=torch.int64).view(1, W)
    yc = torch.arange(H // 2, device=device, dtype=torch.int64).view(H // 2, 1).expand(H // 2, W // 2)
    xc = torch.arange(W // 2, device=device, dtype=torch.int64).view(1, W // 2).expand(H // 2, W // 2)
    idx = y * W + x

    def tri(v, p):
        m = v % p
        return torch.where(m < p // 2, m, p - m)

    for s in range(S):
        sd = seeds[s] if seeds is not None else seed + s
        tb = t0s[s] if t0s is not None else 0
        tex = torch.from_numpy(_tex(sd)).to(device)
        for tt in range(n):
            t = tb + tt
            base = s64(sd ^ ((_G * (t + 1)) & _M))
            z = idx + base
            z = lcg(z)
            z = z ^ lsr(z, 29)
            z = lcg(z)
            if noise > 0:
                nz = lsr(z, 33) % (2 * noise + 1) - noise
            else:
                nz = 0
            v = 40 + tri(x + 2 * t, 192) + tri(y + t, 128) + tex[(y + t) & 63, (x + 2 * t) & 63] + nz
            out[tt, s, :ys] = v.clamp(16, 235).to(torch.uint8).reshape(-1)
            out[tt, s, ys: ys + ys // 4] = (104 + tri(xc + t, 96) // 2).to(torch.uint8).reshape(-1)
            out[tt, s, ys + ys // 4:] = (104 + tri(yc + t, 96) // 2).to(torch.uint8).reshape(-1)
    return out
