# Dev helper: whole-bench throughput over (streams, contexts, resolve workgroups per context)
for cfg in ${CFGS:-"128:2:1536"}; do
IFS=: read s c w <<< "$cfg"
python bench.py --steps 1 --warmup 1 --cpu-frames 0 --secondary 0 --e2e 0 --contexts $c --streams $s --resolve-wgs $w > gpurun_out/c_${s}_${c}_$w.json 2>gpurun_out/c_err.log
python - <<PY
import json
d=json.loads(open('gpurun_out/c_${s}_${c}_$w.json').read().strip().splitlines()[-1])
k=d['roofline']['kernels']
print("streams", $s, "ctx", $c, "wgs", $w, round(d['value']/1e6,2), {a:round(k[a]['avg_launch_us']/1e3,1) for a in k}, flush=True)
PY
done
