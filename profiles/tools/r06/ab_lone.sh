#!/bin/bash
# Round 6, A/B on the device: the wave decoder's build for few lanes (nfc_wave_lone.hip: one wave per SIMD, no scratch) against
# the one build for everything (NFCGPU_LONE_LANES=0), on the headline, on a GPU's share of it, on one dense stream and on the
# 18 captures; then the stage logs of both. Output under gpurun_out/r06_lone/.
set -u
cd "$(dirname "$0")/../../.."
export TMPDIR=/tmp
OUT=gpurun_out/r06_lone
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { echo build failed; tail -20 $OUT/build.log; exit 1; }
for lone in 0 1024; do
   NFCGPU_LONE_LANES=$lone timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu --points fixtures_single,share_dense,single_dense \
      > $OUT/bench_lone$lone.json 2> $OUT/bench_lone$lone.err
   echo "lone=$lone rc=$?"
   python - $OUT/bench_lone$lone.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    p = d["config"]["points"]
    print("  headline %.1f MS/s %.1f ms" % (d["value"], d["ms_per_step"]))
    for k, v in p.items():
        print("  ", k, {kk: vv for kk, vv in v.items() if kk in ("value", "ms_per_step", "slowest", "median", "fastest", "MS_per_s", "unit")})
except Exception as e:
    print("  (no line)", e)
PY
done
for lone in 0 1024; do
   NFCGPU_LONE_LANES=$lone NFCGPU_WINDOW_DEBUG=1 timeout 600 python bench.py --steps 2 --warmup 2 --no-cpu --no-points --check-streams 8 \
      > $OUT/stages_lone$lone.json 2> $OUT/stages_lone$lone.txt
   grep "windowed pass\|windowed stage" $OUT/stages_lone$lone.txt | tail -24
done
# where a lone lane's cycles go (profile build: shader-clock counters per phase), both builds
for lone in 0 1024; do
   NFCGPU_LIB=$PWD/nfc-laboratory_amd/libnfcgpu_profile.so NFCGPU_LONE_LANES=$lone NFCGPU_WINDOW_DEBUG=1 timeout 300 python profiles/tools/r05/capture_stages.py \
      test_NFC-A_106kbps_002 test_NFC-B_106kbps_001 > $OUT/capture_profile_lone$lone.out 2> $OUT/capture_profile_lone$lone.txt
   cat $OUT/capture_profile_lone$lone.out
   grep "attempt 1" -A 40 $OUT/capture_profile_lone$lone.txt | grep "===\|wave cycles\|windowed pass" | head -20
done
for lone in 0 1024; do
   NFCGPU_LONE_LANES=$lone timeout 300 python profiles/tools/r05/capture_stages.py > $OUT/capture_lone$lone.out 2> /dev/null
   cat $OUT/capture_lone$lone.out
done
