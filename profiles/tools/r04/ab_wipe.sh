#!/bin/bash
# Round 4, last GPU minutes: the wave decoder with and without the visits the next sample's visit wipes out (nfc_wave_fast.hpp),
# headline only (bench.py --no-points), the full build first and with the parity leg (64 streams against the reference).
# Variant libraries: nfc-laboratory_amd/build/ab/ (built from the same tree: -DNFC_WAVE_NO_WIPE, -DNFC_WAVE_WIPE_NO_AV).
# usage (from the repository root): gpurun --timeout 330 -- 'bash profiles/tools/r04/ab_wipe.sh'
OUT=gpurun_out/ab_wipe
mkdir -p $OUT
PKG=nfc-laboratory_amd
run() { # name, library, timeout, extra flags
   local name=$1 lib=$2 limit=$3; shift 3
   local t0=$SECONDS
   NFCGPU_LIB=$lib timeout $limit python bench.py --steps 4 --warmup 1 --no-points "$@" > $OUT/$name.json 2> $OUT/$name.err
   echo "$name rc=$? $((SECONDS - t0)) s" | tee -a $OUT/times.txt
   python - <<PY | tee -a $OUT/times.txt
import json
try:
    d = json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1])
    print("   ", "$name", d["value"], "MS/s", d["ms_per_step"], "ms/step", "parity:", (d.get("parity") or d.get("config", {}).get("parity") or {}))
except Exception as e:
    print("   ", "$name", "no line:", e)
PY
}
run full   $PWD/$PKG/libnfcgpu.so                    200
run nowipe $PWD/$PKG/build/ab/libnfcgpu_nowipe.so     90 --no-cpu
run fb     $PWD/$PKG/build/ab/libnfcgpu_fb.so         90 --no-cpu
run full2  $PWD/$PKG/libnfcgpu.so                     90 --no-cpu
timeout 90 python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1; echo "smoke rc=$?" | tee -a $OUT/times.txt
tail -2 $OUT/smoke.txt
