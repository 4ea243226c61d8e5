#!/bin/bash
# Round-2 evidence, run on the GPU box from the repo root: profiles/tools/r02/round_profile.sh [tag]
# (writes gpurun_out/<tag>/; the summaries are then copied into profiles/<tag>/ and profiles/traffic.json)
#   1. the default bench line (bench_n1.json)
#   2. rocprofv3 --kernel-trace --stats of the same command without the CPU leg: per-kernel calls and durations (csv)
#   3. HBM traffic of the two kernels the roofline objects are about - nfc_demod_fixed_kernel (headline) and
#      nfc_scan_kernel (search kernel of the time-parallel path, config5_idle point) - from FETCH_SIZE and WRITE_SIZE in
#      separate --pmc passes (kernel trace only, no other trace domain), calibrated with profiles/tools/calib_traffic.hip
#      as the microarchitecture guide prescribes for gfx950
tag=${1:-r02}
export TMPDIR=/tmp
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root

python bench.py > $out/bench_n1.json 2> $out/bench_n1.err

# headline only: nfc_demod_fixed_kernel's average here is the launch time of roofline.kernel_ms_avg (first launch of every stream
# runs the exact-modulo variant, the warm-up launches are in the average too)
(cd /tmp && rocprofv3 --kernel-trace --stats -d $out/trace -o run --output-format csv -- python $root/bench.py --no-cpu --no-points > $out/trace.log 2>&1)
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/rocprofv3_kernel_stats.csv 2>/dev/null
cp $(find $out/trace -name "*domain_stats.csv" | head -1) $out/rocprofv3_domain_stats.csv 2>/dev/null
# the points that exercise the time-parallel path (idle: nfc_scan_kernel = roofline_search; sparse: the windowed decode)
(cd /tmp && rocprofv3 --kernel-trace --stats -d $out/trace_points -o run --output-format csv -- python $root/bench.py --no-cpu --steps 1 --warmup 1 --points config5_idle,config5_sparse,single_sparse > $out/trace_points.log 2>&1)
cp $(find $out/trace_points -name "*kernel_stats.csv" | head -1) $out/rocprofv3_kernel_stats_points.csv 2>/dev/null

hipcc --offload-arch=gfx950 -O2 profiles/tools/calib_traffic.hip -o /tmp/calib_traffic
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --kernel-trace --pmc $c -d $out/pmc_$c -o run --output-format csv -- python $root/bench.py --no-cpu --steps 2 --warmup 1 --points config5_idle > $out/pmc_$c.log 2>&1)
  (cd /tmp && rocprofv3 --kernel-trace --pmc $c -d $out/calib_$c -o run --output-format csv -- /tmp/calib_traffic > $out/calib_$c.log 2>&1)
done
python profiles/tools/summarize_pmc.py $out/trace > $out/kernel_trace_summary.json
python profiles/tools/summarize_pmc.py $out/trace_points > $out/kernel_trace_summary_points.json
python profiles/tools/summarize_pmc.py $out/calib_FETCH_SIZE $out/calib_WRITE_SIZE > $out/calibration_raw.json
python profiles/tools/r02/make_traffic.py $out > $out/make_traffic.log 2>&1
rm -rf $out/trace/*/*.db $out/trace_points/*/*.db $out/pmc_*/*/*.db 2>/dev/null
rm -f $out/trace/*kernel_trace.csv $out/trace_points/*kernel_trace.csv $out/pmc_*/*kernel_trace.csv 2>/dev/null
ls -la $out | head -40
