#!/bin/bash
# SQ counter passes over one bench point that runs the wave decoder (run on the GPU box from the repo root):
#   profiles/tools/r03/pmc_wave.sh <tag> <point> [env assignments...]
# Counters go in separate passes (one --pmc group each), kernel trace only, as the microarchitecture guide prescribes.
tag=$1; point=$2; shift; shift
export TMPDIR=/tmp
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_LDS SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" ; do
  (cd /tmp && env "$@" rocprofv3 --kernel-trace --pmc $grp -d $out/p$i -o run --output-format csv -- python $root/bench.py --streams 64 --steps 1 --warmup 1 --no-cpu --points $point > $out/p$i.log 2>&1)
  i=$((i+1))
done
python $root/profiles/tools/summarize_pmc.py $out/p* > $out/summary.json
rm -rf $out/p*/*/*.db 2>/dev/null
python - <<PY
import json
d=json.load(open("$out/summary.json"))
for k,v in d["kernels"].items():
    print("%-40s calls %5d total %10.2f ms avg %8.3f max %8.3f"%(k[:40],v["calls"],v["total_ms"],v["avg_ms"],v["max_ms"]))
for k,v in d["counters"].items():
    if "wave" not in k and "scan" not in k: continue
    print(k)
    for c,e in sorted(v.items()):
        print("  %-24s dispatches %4d  sum %16.0f"%(c,e["dispatches"],e["mean"]*e["dispatches"]))
PY
