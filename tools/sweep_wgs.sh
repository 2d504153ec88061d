# Dev helper: single-context bench over the launch-shape knobs; prints MB/s and per-kernel ms per picture
for g in ${GRP:-16}; do for w in ${WGS:-1536}; do
python bench.py --steps 1 --warmup 1 --cpu-frames 0 --secondary 0 --e2e 0 --contexts ${CTX:-1} --streams ${STREAMS:-128} --resolve-wgs $w --resolve-group $g ${EXTRA_ARGS:-} > gpurun_out/s_${g}_$w.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/s_${g}_$w.json').read().strip().splitlines()[-1])
k=d['roofline']['kernels']
print("group", $g, "wgs", $w, round(d['value']/1e6,2), {a:round(k[a]['avg_launch_us']/1e3,2) for a in k if a.startswith('me_')}, d['output_check']['rbsp_sha256'][:12])
PY
done; done
