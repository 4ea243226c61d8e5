#!/bin/bash
# Round-6 evidence, run on the GPU box from the repo root: profiles/tools/r06/round_profile.sh [tag]
# (writes gpurun_out/<tag>/; the summaries are then copied into profiles/r06/ and profiles/traffic.json)
#   1. rocprofv3 --kernel-trace --stats of the headline alone (no CPU leg, no points; 3 submissions): per-kernel calls and
#      durations (csv) - what bench.py's roofline.kernel_ms_avg has to agree with
#   2. HBM traffic of the kernels the roofline objects are about - nfc_wave_kernel (all its launches of a step) and
#      nfc_scan_kernel (its launch over the whole submission) - from FETCH_SIZE and WRITE_SIZE in separate --pmc passes
#      (kernel trace only, no other trace domain), calibrated with profiles/tools/calib_traffic.hip as the microarchitecture
#      guide prescribes for gfx950
#      ... and, new in round 6, of the WHOLE STEP: every kernel of the two submissions (scan, second walks, planes, tiles, windows,
#      wave decoder, chain, finish) summed per step - bench.py's roofline.traffic_step
#   (every rocprofv3 run under `timeout 240` and with --output-format csv: round 5 lost ten GPU-minutes to one that wrote its
#    sqlite database and then did not end)
#   3. SQ counters of the same command, a group per pass (instruction mix, wave cycles, waits, LDS, instruction cache)
# The bench line itself (the driver's command: --steps 20 --warmup 5) is run on its own.
tag=${1:-r06}
export TMPDIR=/tmp
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root

head="python $root/bench.py --no-cpu --no-points --steps 1 --warmup 1"
(cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats -d $out/trace -o run --output-format csv -- python $root/bench.py --no-cpu --no-points --steps 2 --warmup 1 > $out/trace.log 2>&1)
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/rocprofv3_kernel_stats.csv 2>/dev/null
cp $(find $out/trace -name "*domain_stats.csv" | head -1) $out/rocprofv3_domain_stats.csv 2>/dev/null

hipcc --offload-arch=gfx950 -O2 profiles/tools/calib_traffic.hip -o /tmp/calib_traffic
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 240 rocprofv3 --kernel-trace --pmc $c -d $out/pmc_$c -o run --output-format csv -- $head > $out/pmc_$c.log 2>&1)
  (cd /tmp && timeout 240 rocprofv3 --kernel-trace --pmc $c -d $out/calib_$c -o run --output-format csv -- /tmp/calib_traffic > $out/calib_$c.log 2>&1)
done

# 2b. the sequential kernel's traffic on the sources as they are (its record was round 2's): the `saturating`
#     point of the bench (131072 streams x 8192-sample buffers) behind a small headline
seq="python $root/bench.py --no-cpu --steps 1 --warmup 1 --streams 64 --samples 65536 --points saturating"
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c -d $out/seq_$c -o run --output-format csv -- $seq > $out/seq_$c.log 2>&1)
done

i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_LDS SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" ; do
  (cd /tmp && timeout 240 rocprofv3 --kernel-trace --pmc $grp -d $out/sq$i -o run --output-format csv -- $head > $out/sq$i.log 2>&1)
  i=$((i+1))
done

python profiles/tools/summarize_pmc.py $out/trace > $out/kernel_trace_summary.json
python profiles/tools/summarize_pmc.py $out/sq0 $out/sq1 $out/sq2 $out/sq3 > $out/pmc_sq_summary.json
python profiles/tools/summarize_pmc.py $out/calib_FETCH_SIZE $out/calib_WRITE_SIZE > $out/calibration_raw.json
python profiles/tools/r06/make_traffic.py $out > $out/make_traffic.log 2>&1
rm -rf $out/trace/*/*.db $out/pmc_*/*/*.db $out/seq_*/*/*.db $out/sq*/*/*.db $out/calib_*/*/*.db 2>/dev/null
find $out -name "*kernel_trace.csv" -delete 2>/dev/null
find $out -name "*agent_info.csv" -delete 2>/dev/null
ls -la $out | head -40
