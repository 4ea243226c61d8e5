#!/bin/bash
# Round 6, A/B on the device between builds of the library: ab_libs.sh <tag> <libA.so> <libB.so> [bench points]
# (paths relative to nfc-laboratory_amd/; the libraries are built in the container and travel with the snapshot).
# Headline (6 timed steps) + the points named, each library in turn, twice (A B A B) to see the run-to-run spread.
set -u
cd "$(dirname "$0")/../../.."
export TMPDIR=/tmp
TAG=$1; shift
A=$1; shift
B=$1; shift
POINTS=${1:-fixtures_single,share_dense,single_dense}
OUT=gpurun_out/r06_$TAG
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { echo build failed; tail -20 $OUT/build.log; exit 1; }
for round in 1 2; do
for lib in $A $B; do
   name=$(basename $lib .so)
   NFCGPU_LIB=$PWD/nfc-laboratory_amd/$lib timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu --points $POINTS \
      > $OUT/bench_${name}_$round.json 2> $OUT/bench_${name}_$round.err
   echo "$name round $round rc=$?"
   python - $OUT/bench_${name}_$round.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    p = d["config"]["points"]
    print("  headline %.1f MS/s %.1f ms parity %s" % (d["value"], d["ms_per_step"], d.get("parity", {}).get("streams_mismatching", "?")))
    for k, v in p.items():
        print("  ", k, {kk: vv for kk, vv in v.items() if kk in ("value", "ms_per_step", "slowest", "median", "fastest")}, (v.get("parity") or {}).get("streams_mismatching", ""))
except Exception as e:
    print("  (no line)", e)
PY
done
done
for lib in $A $B; do
   name=$(basename $lib .so)
   NFCGPU_LIB=$PWD/nfc-laboratory_amd/$lib NFCGPU_WINDOW_DEBUG=1 timeout 600 python bench.py --steps 2 --warmup 2 --no-cpu --no-points --check-streams 8 \
      > $OUT/stages_$name.json 2> $OUT/stages_$name.txt
   echo "== stages $name"; grep "windowed pass\|windowed stage" $OUT/stages_$name.txt | tail -12
done
