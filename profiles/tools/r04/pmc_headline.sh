#!/bin/bash
# SQ / instruction-cache counter passes over one submission pair of the headline (run on the GPU box): r04_pmc_headline.sh <tag> [env...]
# separate --pmc passes, kernel trace only (MI355X_MICROARCH.md)
tag=$1; shift
export TMPDIR=/tmp
root=$(pwd)
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_IFETCH" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAIT_INST_LDS"; do
  (cd /tmp && env "$@" rocprofv3 --kernel-trace --pmc $grp -d $out/p$i -o run --output-format csv -- python $root/bench.py --steps 1 --warmup 1 --no-cpu --no-points > $out/p$i.log 2>&1)
  i=$((i+1))
done
python $root/profiles/tools/summarize_pmc.py $out/p* > $out/summary.json
python - <<PY
import json
d=json.load(open("$out/summary.json"))
for k,v in d["counters"].items():
    if "wave" not in k: continue
    print(k)
    for c,e in sorted(v.items()):
        print("  %-22s dispatches %3d sum %16.0f"%(c,e["dispatches"],e["mean"]*e["dispatches"]))
for k,v in d["kernels"].items():
    print(k, v)
PY
