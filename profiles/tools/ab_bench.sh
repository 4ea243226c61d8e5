#!/bin/bash
# Same-box A/B of two kernel builds: working tree = variant B, HEAD = variant A. Box-to-box differences are 3-5 %, larger
# than most single changes, so both libraries are built here, travel together and are benched alternately on one box
# (dense default workload A B A B, then the idle-carrier diagnostic A B). One gpurun call, about 4 GPU-minutes.
#   profiles/tools/ab_bench.sh
cd "$(dirname "$0")/../.." || exit 1
make -C nfc-laboratory_amd -j4 2>&1 | grep -E " error|Error"
cp nfc-laboratory_amd/libnfcgpu.so nfc-laboratory_amd/libnfcgpu_B.so
git stash -q && make -C nfc-laboratory_amd -j4 2>&1 | grep -E " error|Error"
cp nfc-laboratory_amd/libnfcgpu.so nfc-laboratory_amd/libnfcgpu_A.so
git stash pop -q
touch nfc-laboratory_amd/csrc/nfc_core.hpp
make -C nfc-laboratory_amd -j4 2>&1 | grep -E " error|Error"
/usr/local/graft/bin/gpurun --timeout 1500 -- 'for v in A B A B; do NFCGPU_LIB=$PWD/nfc-laboratory_amd/libnfcgpu_$v.so python bench.py --streams 131072 --steps 3 --warmup 1 --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$v dense\", d[\"value\"], d[\"roofline\"][\"kernel_ms_avg\"])"; done; for v in A B; do NFC_BENCH_IDLE=1 NFCGPU_LIB=$PWD/nfc-laboratory_amd/libnfcgpu_$v.so python bench.py --streams 131072 --steps 3 --warmup 1 --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$v idle\", d[\"value\"], d[\"roofline\"][\"kernel_ms_avg\"])"; done' 2>&1 | tail -7
rm -f nfc-laboratory_amd/libnfcgpu_A.so nfc-laboratory_amd/libnfcgpu_B.so
