"""Dev helper: static VALU / SALU instruction counts of one kernel attributed to source lines (hipcc -gline-tables-only).
usage: python tools/isa_attr.py fer_me.hip _Z9k_me_specILi32EEv6FerDev [lines|funcs]"""
import collections, re, subprocess, sys
from pathlib import Path
root = Path(__file__).resolve().parent.parent
src_name, kern = sys.argv[1], sys.argv[2]
mode = sys.argv[3] if len(sys.argv) > 3 else "funcs"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-gline-tables-only", "-std=c++17", "-fPIC", "--offload-arch=gfx950", f"-I{root}/include",
                f"-I{root}/h264-fer_amd/csrc", "-S", "--cuda-device-only", "-o", "/tmp/isa_attr.s", str(root / "h264-fer_amd/csrc" / src_name)],
               check=True, stderr=subprocess.DEVNULL)
s = open("/tmp/isa_attr.s").read().split("\n")
files = {}
for l in s:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', l)
    if m:
        files[int(m.group(1))] = m.group(3)
i = [k for k, l in enumerate(s) if l.startswith(kern + ":")][0]
j = i
while not s[j].strip().startswith(".Lfunc_end"):
    j += 1
cur = None
cnt, scnt = collections.Counter(), collections.Counter()
for l in s[i:j]:
    t = l.strip()
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
    if m:
        cur = (files.get(int(m.group(1)), "?").split("/")[-1], int(m.group(2)))
        continue
    if t.startswith("v_"):
        cnt[cur] += 1
    elif t.startswith("s_"):
        scnt[cur] += 1
print("static VALU", sum(cnt.values()), "SALU", sum(scnt.values()))
srcs = {}
def line_text(f, n):
    if f not in srcs:
        for cand in (root / "h264-fer_amd/csrc" / f,):
            srcs[f] = cand.read_text().split("\n") if cand.exists() else []
    L = srcs[f]
    return L[n - 1].strip()[:100] if 0 < n <= len(L) else ""
if mode == "lines":
    for (f, n), c in cnt.most_common(40):
        print(c, scnt[(f, n)], f, n, line_text(f, n))
else:
    agg, sagg = collections.Counter(), collections.Counter()
    def fn_of(f, n):
        line_text(f, n)
        L = srcs.get(f, [])
        k = min(n, len(L)) - 1
        while k > 0 and not re.match(r"^(__device__|__global__|static __device__)", L[k]):
            k -= 1
        return f + ": " + (L[k][:80] if L else "")
    for (key, c) in cnt.items():
        if key:
            agg[fn_of(*key)] += c
    for (key, c) in scnt.items():
        if key:
            sagg[fn_of(*key)] += c
    for f, c in agg.most_common(16):
        print(c, sagg[f], f)
