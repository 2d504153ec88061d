"""Dev helper: per-kernel summary of a rocprofv3 --pmc counter_collection.csv (SQ counters): instructions per 8x8
partition and the fractions of wave cycles spent issuing / waiting."""
import collections
import csv
import sys

path, S = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 32
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(path)):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    if (k, r['Dispatch_Id']) not in seen:
        seen.add((k, r['Dispatch_Id']))
        cnt[k] += 1
parts = S * 32160
names = sorted({c for v in agg.values() for c in v})
print("kernel".ljust(22), "n".rjust(4), " ".join(n.replace("SQ_", "")[:14].rjust(14) for n in names))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
    if not k.startswith('k_'):
        continue
    n = cnt[k]
    wc = v.get('SQ_WAVE_CYCLES', 0)
    row = []
    for c in names:
        x = v.get(c, 0) / n
        if c.startswith('SQ_INSTS') or c == 'SQ_INST_CYCLES_SALU':
            row.append("%14.0f" % (x / parts))        # per partition
        elif wc and c != 'SQ_WAVE_CYCLES' and c != 'SQ_BUSY_CYCLES':
            row.append("%13.1f%%" % (100 * v[c] / wc))
        else:
            row.append("%14.3e" % x)
    print(k[:22].ljust(22), str(n).rjust(4), " ".join(row))
