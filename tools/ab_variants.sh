# Dev helper (GPU box): time bench.py one context under several library variants and chain settings.
# usage: tools/ab_variants.sh "base w5 win8" "4 16" -> gpurun_out/ab_<variant>_g<group>.json
cp h264-fer_amd/libferhip.so /tmp/libferhip_keep.so
for v in $1; do
  cp h264-fer_amd/var/libferhip_$v.so h264-fer_amd/libferhip.so
  for g in $2; do
    python bench.py --streams 256 --contexts 1 --steps 1 --warmup 1 --cpu-frames 0 --secondary 0 --e2e 0 --resolve-group $g $3 > gpurun_out/ab_${v}_g$g.json 2> gpurun_out/ab_${v}_g$g.err || echo "FAILED $v $g"
  done
done
cp /tmp/libferhip_keep.so h264-fer_amd/libferhip.so
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/ab_*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f,"ERR"); continue
    k=d["roofline"]["kernels"]
    print(f.split("/")[-1], d["value"], d["output_check"]["ok"], "resolve", k["me_resolve"]["avg_launch_us"], "spec", k.get("me_spec",{}).get("avg_launch_us"), "resid", k["p_resid"]["avg_launch_us"])
PY
