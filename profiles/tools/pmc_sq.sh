#!/bin/bash
# SQ counter passes over the default bench workload (run on the GPU box): pmc_sq.sh <tag> [env assignments...]
# Counters go in separate passes (one --pmc group each), kernel trace only, as the microarchitecture guide prescribes.
tag=$1; shift
export TMPDIR=/tmp
root=$(pwd)
out=$root/gpurun_out/sq_$tag
mkdir -p $out
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_IFETCH" ; do
  (cd /tmp && env "$@" rocprofv3 --kernel-trace --pmc $grp -d $out/p$i -o run --output-format csv -- python $root/bench.py --streams 131072 --steps 2 --warmup 1 --no-cpu > $out/p$i.log 2>&1)
  i=$((i+1))
done
python $root/profiles/tools/summarize_pmc.py $out/p* > $out/summary.json
python - <<PY
import json
d=json.load(open("$out/summary.json"))
for k,v in d["counters"].items():
    if "demod" not in k: continue
    waves=v.get("SQ_WAVES",{}).get("last",2048.0)
    print(k)
    for c,e in sorted(v.items()):
        print("  %-22s %14.0f  per wave-step %10.1f"%(c,e["last"],e["last"]/waves/8192))
PY
