#!/bin/bash
# Round-end evidence, run on the GPU box: round_profile.sh <round-tag>   (writes gpurun_out/<tag>/, copy into profiles/<tag>/)
#   1. default bench line (bench_n1.json)
#   2. rocprofv3 --kernel-trace --stats of the same command (kernel durations)
#   3. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (kernel trace only, no other trace domains),
#      calibrated with profiles/tools/calib_traffic.hip as the microarchitecture guide prescribes for gfx950
tag=${1:-r01}
export TMPDIR=/tmp
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root

python bench.py > $out/bench_n1.json 2> $out/bench_n1.err

(cd /tmp && rocprofv3 --kernel-trace --stats -d $out/trace -o run --output-format csv -- python $root/bench.py --no-cpu > $out/trace.log 2>&1)
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/rocprofv3_kernel_stats.csv 2>/dev/null
cp $(find $out/trace -name "*domain_stats.csv" | head -1) $out/rocprofv3_domain_stats.csv 2>/dev/null

hipcc --offload-arch=gfx950 -O2 profiles/tools/calib_traffic.hip -o /tmp/calib_traffic
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --kernel-trace --pmc $c -d $out/pmc_$c -o run --output-format csv -- python $root/bench.py --no-cpu --steps 2 --warmup 1 > $out/pmc_$c.log 2>&1)
  (cd /tmp && rocprofv3 --kernel-trace --pmc $c -d $out/calib_$c -o run --output-format csv -- /tmp/calib_traffic > $out/calib_$c.log 2>&1)
done
python profiles/tools/summarize_pmc.py $out/trace $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE > $out/kernel_trace_summary.json
python profiles/tools/summarize_pmc.py $out/calib_FETCH_SIZE $out/calib_WRITE_SIZE > $out/calibration.json
ls -la $out | head -30
