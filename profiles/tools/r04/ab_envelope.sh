#!/bin/bash
# Round 4, last GPU seconds: the headline with the envelope tracker's second walks in a kernel of their own (default) and left to
# the scan kernel (NFCGPU_ENVELOPE_KERNEL=0), the first with the parity leg; three short captures both ways.
# usage (from the repository root): gpurun --timeout 100 -- 'bash profiles/tools/r04/ab_envelope.sh'
OUT=gpurun_out/ab_envelope
mkdir -p $OUT
line() { python - <<PY | tee -a $OUT/times.txt
import json
try:
    d = json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
    tp = d["config"]["time_parallel"]
    print("%-10s %9.1f MS/s %8.2f ms/step  scan+second walks %.1f ms  rescanned %d  parity: %s" % ("$1", d["value"], d["ms_per_step"], tp["scan_kernel_ms_per_step"], tp["chunks_rescanned"], d.get("parity")))
except Exception as e:
    print("%-10s no line: %s" % ("$1", e))
PY
}
timeout 70 python bench.py --steps 4 --warmup 1 --no-points > $OUT/on.json 2> $OUT/on.err; line on
NFCGPU_ENVELOPE_KERNEL=0 timeout 40 python bench.py --steps 4 --warmup 1 --no-points --no-cpu > $OUT/off.json 2> $OUT/off.err; line off
for k in 1 0; do
   echo "== NFCGPU_ENVELOPE_KERNEL=$k" | tee -a $OUT/times.txt
   NFCGPU_ENVELOPE_KERNEL=$k NFCGPU_WINDOW_DEBUG=1 timeout 20 python profiles/tools/r04/capture_stages.py test_NFC-A_106kbps_002 test_NFC-A_424kbps_001 test_NFC-B_106kbps_001 2> $OUT/captures_$k.err | grep "rep 2" | tee -a $OUT/times.txt
   grep -A45 "third decode" $OUT/captures_$k.err | grep "stage seams" | tee -a $OUT/times.txt
done
