#!/bin/bash
# Headline (bench.py --no-points --no-cpu, 4 steps) with every variant library under nfc-laboratory_amd/build/ab/ - the same tree,
# the wave decoder compiled with other options -, the in-tree library first and last.
# usage (from the repository root): gpurun --timeout 300 -- 'bash profiles/tools/r04/ab_libs.sh <tag> [bench flags]'
TAG=${1:-libs}; shift
OUT=gpurun_out/ab_$TAG
mkdir -p $OUT
PKG=$PWD/nfc-laboratory_amd
one() { # name, library
   local t0=$SECONDS
   NFCGPU_LIB=$2 timeout 60 python bench.py --steps 4 --warmup 1 --no-points --no-cpu "${EXTRA[@]}" > $OUT/$1.json 2> $OUT/$1.err
   local rc=$?
   python - <<PY | tee -a $OUT/times.txt
import json
try:
    d = json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
    print("%-14s %9.1f MS/s %8.2f ms/step  wave busy %.1f ms  (rc $rc, $((SECONDS - t0)) s)" % ("$1", d["value"], d["ms_per_step"], d["roofline"]["kernel_busy_ms"]))
except Exception as e:
    print("%-14s no line (rc $rc): %s" % ("$1", e))
PY
}
EXTRA=("$@")
one intree $PKG/libnfcgpu.so
for lib in $PKG/build/ab/libnfcgpu_*.so; do
   name=$(basename $lib .so); name=${name#libnfcgpu_}
   one $name $lib
done
one intree2 $PKG/libnfcgpu.so
