#!/bin/bash
# instruction counts per wave step per enabled-technology mask: pmc_masks.sh   (MASKS_IDLE=1 search-only workload, default;
# MASKS_IDLE=0 the dense bench workload)
export TMPDIR=/tmp
root=$(pwd)
out=$root/gpurun_out/masks
mkdir -p $out
for mask in 0 1 2 4 8 15; do
  (cd /tmp && NFC_BENCH_IDLE=${MASKS_IDLE:-1} NFC_BENCH_TECH_MASK=$mask rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAVES -d $out/m$mask -o run --output-format csv -- python $root/bench.py --streams 131072 --steps 2 --warmup 1 --no-cpu > $out/m$mask.log 2>&1)
  python $root/profiles/tools/summarize_pmc.py $out/m$mask > $out/m$mask.json
  python - <<PY
import json
d=json.load(open("$out/m$mask.json"))
for k,v in d["counters"].items():
    if "demod" not in k or "exact" in k: continue
    w=v["SQ_WAVES"]["last"]
    print("mask $mask %-28s VALU %7.1f SALU %7.1f BRANCH %6.1f  last launch ms %s"%(k, v["SQ_INSTS_VALU"]["last"]/w/8192, v["SQ_INSTS_SALU"]["last"]/w/8192, v["SQ_INSTS_BRANCH"]["last"]/w/8192, d["kernels"].get(k,{}).get("max_ms")))
PY
done
