import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+'/*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); r=d["roofline"]; t=d["config"]["time_parallel"]
        print(f.split('/')[-1], d["value"], d["ms_per_step"], r["kernel_ms_avg"], r.get("kernel_busy_ms"), t["decode_passes"], t["windowed_decode_ms_per_step"])
    except Exception as e: print(f, 'ERR', e)
