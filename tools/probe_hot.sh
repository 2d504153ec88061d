cp h264-fer_amd/libferhip.so /tmp/keep.so
cp h264-fer_amd/var/libferhip_probe.so h264-fer_amd/libferhip.so
for dbg in 0 512; do
FER_DBG=$dbg python bench.py --streams 128 --contexts 1 --steps 1 --warmup 1 --cpu-frames 0 --secondary 0 --e2e 0 --resolve-group 16 --probe-build 1 > gpurun_out/pr_$dbg.json 2> gpurun_out/pr_$dbg.err
done
cp /tmp/keep.so h264-fer_amd/libferhip.so
python - <<'PY'
import json
for f in ("pr_0","pr_512"):
    d=json.load(open(f"gpurun_out/{f}.json")); k=d["roofline"]["kernels"]
    print(f, d["value"], "resolve", k["me_resolve"]["avg_launch_us"], d["speculation"])
PY
