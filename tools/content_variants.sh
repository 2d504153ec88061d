# Collects bench lines for the content variants (profiles/r03_content_variants.md): 32 streams, one context
for c in textured letterbox flat-half still; do
python bench.py --content $c --steps 1 --warmup 0 --cpu-frames 0 --secondary 0 --e2e 0 --streams 32 --contexts 1 > gpurun_out/bc_$c.json 2>gpurun_out/bc_$c.err
python - <<PY
import json
d=json.loads(open("gpurun_out/bc_$c.json").read().strip().splitlines()[-1])
k=d["roofline"]["kernels"]
print("$c", round(d["value"]/1e6,2), "skip", d.get("skip_fraction"), d.get("speculation"), {a:round(k[a]["avg_launch_us"]/1e3,2) for a in k}, flush=True)
PY
done
