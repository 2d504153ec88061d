"""Frame sharding across GPUs: closed GOPs are the independent unit.

No collective on the data path: rank r encodes GOPs g with g % world == r as separate streams;
the host concatenates the NAL units in GOP order.  The reference would emit SPS+PPS once, so the
merge keeps the parameter sets of the first shard only.
"""


def gops_of_rank(n_gops, world, rank):
    """GOP indices encoded by `rank` (round-robin)."""
    return [g for g in range(n_gops) if g % world == rank]


def split_nals(stream):
    """Split an Annex-B stream (4-byte start codes, as the reference writes them) into NAL units."""
    out, i, n = [], 0, len(stream)
    starts = []
    while True:
        j = stream.find(b"\x00\x00\x00\x01", i)
        if j < 0:
            break
        starts.append(j)
        i = j + 4
    for k, s in enumerate(starts):
        e = starts[k + 1] if k + 1 < len(starts) else n
        out.append(stream[s:e])
    return out


def merge_gop_streams(gop_streams):
    """gop_streams: list (GOP order) of Annex-B streams, each SPS+PPS+slices -> one stream."""
    out = bytearray()
    for g, s in enumerate(gop_streams):
        for nal in split_nals(s):
            t = nal[4] & 31
            if g > 0 and t in (7, 8):
                continue
            out += nal
    return bytes(out)
