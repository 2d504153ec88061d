"""Syntax-level P slices for the decode twin (row f4 of SURVEY.md 8f): what the reference's ENCODER never writes but
its DECODER parses -- sub-8x8 partitions (sub_mb_type 1..3), ref_idx_l0 as te(), reference picture list modification,
num_ref_idx_active_override, slices that end early.  Test infrastructure: slices are written bit by bit here (ITU-T
H.264 7.3.3 / 7.3.5 with the field widths of the SPS / PPS the encoder emits: frame_num 9 bits, pic_order_cnt_lsb 10
bits, no deblocking control) and appended to the parameter sets and IDR picture of a committed golden stream; the
GPU decoder and the oracle decoder must then agree on every picture.

Residual data is kept to what can be written without a CAVLC encoder: coded block patterns whose blocks all carry
TotalCoeff 0 (coeff_token "1" for nC < 2, "01" for chroma DC), so mb_qp_delta is exercised while every count stays 0.
"""
import numpy as np

INTER_CBP = [0, 16, 1, 2, 4, 8, 32, 3, 5, 10, 12, 15, 47, 7, 11, 13, 14, 6, 9, 31, 35, 37, 42, 44, 33, 34, 36, 40, 39, 43,
             45, 46, 17, 18, 20, 24, 19, 21, 26, 28, 23, 27, 29, 30, 22, 25, 38, 41]  # Table 9-4, inter column


class Bits:
    def __init__(self):
        self.b = []

    def put(self, n, v):
        for i in range(n - 1, -1, -1):
            self.b.append((v >> i) & 1)

    def ue(self, v):
        v += 1
        n = v.bit_length()
        self.put(n - 1, 0)
        self.put(n, v)

    def se(self, v):
        self.ue(2 * v - 1 if v > 0 else -2 * v)

    def rbsp(self, pad):
        bits = self.b + [1]
        bits += [0] * (-len(bits) % 8)
        out = bytearray(np.packbits(np.array(bits, np.uint8)).tobytes())
        out += b"\x80" * pad  # bytes after the stop bit: keeps the reference's "more data" test true to the last MB
        return bytes(out)


def nal_unit(nal_type, ref_idc, rbsp):
    out = bytearray(b"\x00\x00\x00\x01")
    out.append((ref_idc << 5) | nal_type)
    z = 0
    for b in rbsp:
        if z >= 2 and b <= 3:
            out.append(3)
            z = 0
        out.append(b)
        z = z + 1 if b == 0 else 0
    return bytes(out)


def te_bits(w, rng):
    """one of the forms the reference's te() accepts: "1" (0), or prefix 01 + suffix bit x in {0,1} + the extra bit"""
    if rng.random() < 0.5:
        w.put(1, 1)
    else:
        w.put(2, 1)
        w.put(1, int(rng.integers(0, 2)))
        w.put(1, int(rng.integers(0, 2)))


def p_slice(rng, nmb, frame_num, poc_lsb, override, active_minus1, modification, early_end=False, intra_every=0,
            mvd_range=3, p_skip=0.25, p_resid=0.3):
    """-> (rbsp bytes, ref_idx_coded_in_mb_pred).  modification: None | [] | [(idc, value), ...]"""
    w = Bits()
    w.ue(0)
    w.ue(5 if rng.random() < 0.5 else 0)  # slice_type 0 or 5: both P
    w.ue(0)
    w.put(9, frame_num & 511)
    w.put(10, poc_lsb & 1023)
    w.put(1, 1 if override else 0)
    if override:
        w.ue(active_minus1)
    if modification is None:
        w.put(1, 0)
    else:
        w.put(1, 1)
        for idc, val in modification:
            w.ue(idc)
            w.ue(val)
        w.ue(3)
    w.put(1, 0)  # adaptive_ref_pic_marking_mode_flag
    w.se(int(rng.integers(-3, 4)))  # slice_qp_delta
    return w, override, active_minus1


def mb_layer(w, rng, nmb, ref_sub, ref_mb, mvd_range, p_skip, p_resid):
    cur = 0
    while cur < nmb:
        run = 0
        while cur + run < nmb and rng.random() < p_skip:
            run += 1
        w.ue(run)
        cur += run
        if cur >= nmb:
            break
        t = int(rng.choice([0, 1, 2, 3, 3, 3, 4, 4]))
        w.ue(t)
        mvd = lambda: (w.se(int(rng.integers(-mvd_range, mvd_range + 1))), w.se(int(rng.integers(-mvd_range, mvd_range + 1))))
        if t >= 3:
            sub = [int(rng.integers(0, 4)) for _ in range(4)]
            for s in sub:
                w.ue(s)
            if ref_sub and t != 4:
                for _ in range(4):
                    te_bits(w, rng)
            for s in sub:
                for _ in range([1, 2, 2, 4][s]):
                    mvd()
        else:
            npart = 1 if t == 0 else 2
            if ref_mb:
                for _ in range(npart):
                    te_bits(w, rng)
            for _ in range(npart):
                mvd()
        code = int(rng.integers(1, 48)) if rng.random() < p_resid else 0
        w.ue(code)
        cbp = INTER_CBP[code]
        if cbp:
            w.se(int(rng.integers(-2, 3)))  # mb_qp_delta
            for i8 in range(4):
                if cbp & (1 << i8):
                    for _ in range(4):
                        w.put(1, 1)  # coeff_token: TotalCoeff 0, nC in [0, 2)
            if (cbp >> 4) & 3:
                for _ in range(2):
                    w.put(2, 1)  # chroma DC coeff_token: TotalCoeff 0
            if (cbp >> 4) & 2:
                for _ in range(8):
                    w.put(1, 1)
        cur += 1


def make_stream(split_nals, base, seed, plan, nmb=99):
    """base: a golden Annex-B stream whose first three NAL units are SPS, PPS, IDR.  plan: one dict per P picture with
    keys override, active, modification, early_end.  -> Annex-B bytes"""
    rng = np.random.default_rng(seed)
    nals = split_nals(base)
    assert [n[4] & 31 for n in nals[:3]] == [7, 8, 5]
    out = bytearray(b"".join(nals[:3]))
    active = 0  # what the decoder's global holds: only an override ever changes it
    for k, p in enumerate(plan):
        w, override, am1 = p_slice(rng, nmb, k + 1, 2 * (k + 1), p.get("override", False), p.get("active", 0), p.get("modification"))
        if override:
            active = am1
        mb_layer(w, rng, nmb, bool(override), active > 0, p.get("mvd_range", 3), p.get("p_skip", 0.25), p.get("p_resid", 0.3))
        out += nal_unit(1, 2, w.rbsp(0 if p.get("early_end") else 2))
    return bytes(out)
