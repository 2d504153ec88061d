import json,glob
for f in sorted(glob.glob("gpurun_out/ab_*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f,"ERR"); continue
    k=d["roofline"]["kernels"]
    print(f.split("/")[-1], d["value"], d["output_check"]["ok"], " ".join(f"{n}={k[n]['avg_launch_us']:.0f}" for n in ("interp","sort_keys","sort","sort_finish","me_pre","me_walk","me_spec","me_resolve","p_resid","cavlc") if n in k))
