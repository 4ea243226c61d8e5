#!/bin/bash
# Round 6: stage logs (NFCGPU_WINDOW_DEBUG=1, a synchronisation at every mark) of the headline for a list of library builds,
# optionally with extra environment: ab_stage.sh <tag> "<ENV=..>" lib1.so lib2.so ...
set -u
cd "$(dirname "$0")/../../.."
export TMPDIR=/tmp
TAG=$1; shift
EXTRA=$1; shift
OUT=gpurun_out/r06_$TAG
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { echo build failed; tail -20 $OUT/build.log; exit 1; }
for lib in "$@"; do
   name=$(basename $lib .so)
   env $EXTRA NFCGPU_LIB=$PWD/nfc-laboratory_amd/$lib NFCGPU_WINDOW_DEBUG=1 timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu --no-points --check-streams 16 \
      > $OUT/stages_$name.json 2> $OUT/stages_$name.txt
   echo "== stages $name ($EXTRA) rc=$?"
   python - $OUT/stages_$name.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("  headline %.1f MS/s %.1f ms; checked streams mismatching: %s" % (d["value"], d["ms_per_step"], d.get("parity", {}).get("streams_mismatching", d.get("parity"))))
except Exception as e:
    print("  (no line)", e)
PY
   grep "windowed pass\|windowed stage" $OUT/stages_$name.txt | tail -12
done
