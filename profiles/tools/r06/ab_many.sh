#!/bin/bash
# Round 6: the headline (6 timed steps) + points for a list of library builds, each twice, then a stage log of each:
# ab_many.sh <tag> <points> lib1.so lib2.so ...   (paths relative to nfc-laboratory_amd/)
set -u
cd "$(dirname "$0")/../../.."
export TMPDIR=/tmp
TAG=$1; shift
POINTS=$1; shift
OUT=gpurun_out/r06_$TAG
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { echo build failed; tail -20 $OUT/build.log; exit 1; }
for round in 1 2; do
for lib in "$@"; do
   name=$(basename $lib .so)
   NFCGPU_LIB=$PWD/nfc-laboratory_amd/$lib timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu --points $POINTS \
      > $OUT/bench_${name}_$round.json 2> $OUT/bench_${name}_$round.err
   echo "$name round $round rc=$?"
   python - $OUT/bench_${name}_$round.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    p = d["config"]["points"]
    print("  headline %.1f MS/s %.1f ms, checked streams mismatching %s, passes %s" % (d["value"], d["ms_per_step"], d.get("parity", {}).get("streams_mismatching", "?"), d["config"]["time_parallel"]["decode_passes"]))
    for k, v in p.items():
        print("  ", k, {kk: vv for kk, vv in v.items() if kk in ("value", "ms_per_step", "slowest", "median", "fastest")}, (v.get("parity") or {}).get("streams_mismatching", ""))
except Exception as e:
    print("  (no line)", e)
PY
done
done
for lib in "$@"; do
   name=$(basename $lib .so)
   NFCGPU_LIB=$PWD/nfc-laboratory_amd/$lib NFCGPU_WINDOW_DEBUG=1 timeout 600 python bench.py --steps 2 --warmup 2 --no-cpu --no-points --check-streams 8 \
      > $OUT/stages_$name.json 2> $OUT/stages_$name.txt
   echo "== stages $name"; grep "windowed pass\|windowed stage passes" $OUT/stages_$name.txt | tail -7
done
