#!/bin/bash
# Headline (bench.py --no-points --no-cpu, 4 steps) under a list of environment settings, the defaults first and last.
# usage (from the repository root): gpurun --timeout 200 -- 'bash profiles/tools/r04/ab_knobs.sh <tag> "A=1 B=2" "C=3" ...'
TAG=${1:-knobs}; shift
OUT=gpurun_out/ab_$TAG
mkdir -p $OUT
one() { # name, settings
   local t0=$SECONDS
   env $2 timeout 60 python bench.py --steps 4 --warmup 1 --no-points --no-cpu > $OUT/$1.json 2> $OUT/$1.err
   local rc=$?
   python - <<PY | tee -a $OUT/times.txt
import json
try:
    d = json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
    tp = d["config"]["time_parallel"]
    print("%-44s %9.1f MS/s %8.2f ms/step  scan %.1f  decode %.1f  passes %d  rescanned %d  (rc $rc, $((SECONDS - t0)) s)" % ("$2" or "defaults", d["value"], d["ms_per_step"],
          tp["scan_kernel_ms_per_step"], tp["windowed_decode_ms_per_step"], tp["decode_passes"], tp["chunks_rescanned"]))
except Exception as e:
    print("%-44s no line (rc $rc): %s" % ("$2", e))
PY
}
one defaults ""
i=0
for s in "$@"; do i=$((i+1)); one set$i "$s"; done
one defaults2 ""
