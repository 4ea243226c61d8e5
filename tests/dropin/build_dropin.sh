#!/bin/bash
# TEST INFRASTRUCTURE. Builds the drop-in demonstration: the reference's OWN golden-vector harness
# (src/nfc-test/test-sdr/src/main/cpp/main.cpp, unmodified, compiled where it lies) linked against
#   - nfc-laboratory_amd/host/NfcDecoder.cpp  (lab::NfcDecoder implemented on the nfcgpu C ABI)
#   - libnfcgpu.so                            (HIP kernels)
#   - oracle/_ref/libnfcref_support.a         (the reference's rt-lang / hw-dev / lab-data objects, i.e. everything
#                                              of the reference EXCEPT its CPU decoder: NfcDecoder/NfcTech/NfcA/B/F/V)
# Output goes to oracle/_ref/ because it contains objects compiled from reference sources.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
HOST="$ROOT/nfc-laboratory_amd/host"
REF="${NFC_REFERENCE_ROOT:-/root/reference}"
R="$REF/src/nfc-lib"
OUT="$ROOT/oracle/_ref"

[ -d "$R" ] || { echo "reference tree not present; keeping prebuilt test-sdr-gpu" >&2; exit 0; }
[ -f "$OUT/libnfcref_support.a" ] && [ -f "$OUT/libnfcref_task.a" ] || bash "$ROOT/oracle/build_ref.sh"

INC="-I$R/lib-rt/rt-lang/src/main/include -I$R/lib-hw/hw-dev/src/main/include -I$R/lib-lab/lab-data/src/main/include \
 -I$R/lib-lab/lab-radio/src/main/include -I$R/lib-ext/nlohmann/src/main/cpp -I$ROOT/include"

g++ -std=c++17 -O2 -pthread -w $INC -c "$HOST/NfcDecoder.cpp" -o "$OUT/obj/NfcDecoder_gpu.o"
g++ -std=c++17 -O2 -pthread -w $INC "$REF/src/nfc-test/test-sdr/src/main/cpp/main.cpp" "$OUT/obj/NfcDecoder_gpu.o" \
    "$OUT/libnfcref_support.a" -L"$ROOT/nfc-laboratory_amd" -lnfcgpu -Wl,-rpath,'$ORIGIN/../../nfc-laboratory_amd' \
    -o "$OUT/test-sdr-gpu"

# the same at task level: the reference's RadioDecoderTask (unmodified) on top of the GPU decoder
TINC="$INC -I$R/lib-lab/lab-tasks/src/main/include -I$R/lib-lab/lab-tasks/src/main/cpp/tasks"
g++ -std=c++17 -O2 -pthread -w $TINC "$HERE/task_harness.cpp" "$OUT/libnfcref_task.a" "$OUT/obj/NfcDecoder_gpu.o" \
    "$OUT/libnfcref_support.a" -L"$ROOT/nfc-laboratory_amd" -lnfcgpu -Wl,-rpath,'$ORIGIN/../../nfc-laboratory_amd' \
    -o "$OUT/task-gpu"
# the application's replay pipeline (SignalStorageTask -> RadioDecoderTask), GPU decoder underneath
[ -f "$OUT/obj/SignalStorageTask.o" ] && g++ -std=c++17 -O2 -pthread -w $TINC "$HERE/replay_harness.cpp" "$OUT/obj/SignalStorageTask.o" \
    "$OUT/libnfcref_task.a" "$OUT/obj/NfcDecoder_gpu.o" "$OUT/libnfcref_support.a" -L"$ROOT/nfc-laboratory_amd" -lnfcgpu \
    -Wl,-rpath,'$ORIGIN/../../nfc-laboratory_amd' -o "$OUT/replay-gpu"

# the decoder interface driven by a script, GPU decoder underneath (same source as oracle/_ref/api-ref)
g++ -std=c++17 -O2 -pthread -w $INC "$HERE/api_harness.cpp" "$OUT/obj/NfcDecoder_gpu.o" \
    "$OUT/libnfcref_support.a" -L"$ROOT/nfc-laboratory_amd" -lnfcgpu -Wl,-rpath,'$ORIGIN/../../nfc-laboratory_amd' \
    -o "$OUT/api-gpu"
echo "built $OUT/test-sdr-gpu $OUT/task-gpu $OUT/api-gpu"
