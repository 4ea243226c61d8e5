#!/bin/bash
# TEST INFRASTRUCTURE ONLY. Builds tests/hostsim/libnfcgpu_emulated.so: the product's host runtime behind the C ABI
# (nfc-laboratory_amd/csrc/nfcgpu.hip, the file as it is) compiled against a stand-in HIP (fakehip/) whose kernel launches
# call CPU twins of the kernels (emu_kernels.cpp) built on the product's device step machine. It exists so that the C-ABI
# parity suite (tests/test_gpu_parity.py) can exercise the host runtime - batching, configurations, the clock mirror that
# selects the exact-modulo kernels, staging, the frame sink, flush / reset / close - on a box without a GPU
# (tests/test_host_runtime_emulated.py). It is not a CPU path of the product: nothing under nfc-laboratory_amd/ knows
# about it, and the real library refuses to work without a GPU (tests/test_abi.py).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
PKG="$HERE/../../nfc-laboratory_amd"
make -s -C "$PKG" build/nfc_config_fixed.inc
CXX="g++ -std=c++17 -O2 -fno-strict-aliasing -ffp-contract=off -msse3 -mno-avx -fPIC -Wall -Wno-unused-function -Wno-unknown-pragmas -I$HERE/fakehip -I$PKG/build -DNFCGPU_EMULATED_TEST_BUILD $EMU_DEFS"
$CXX -x c++ -c "$PKG/csrc/nfcgpu.hip" -o "$HERE/emu_nfcgpu.o"
$CXX -x c++ -c "$PKG/csrc/nfc_trace.hip" -o "$HERE/emu_trace.o"
$CXX -c "$HERE/emu_kernels.cpp" -o "$HERE/emu_kernels.o"
# the wave decoder's own text, 64 fibres per wave (wavesim.hpp)
$CXX -c "$HERE/emu_wave.cpp" -o "$HERE/emu_wave.o"
# in-process stand-in for RCCL (ranks are threads): the gather's rank logic without GPUs (NFCGPU_FAKE_RCCL=1)
$CXX -c "$HERE/fake_rccl.cpp" -o "$HERE/fake_rccl.o"
g++ -shared -pthread -o "$HERE/libnfcgpu_emulated.so" "$HERE/emu_nfcgpu.o" "$HERE/emu_trace.o" "$HERE/emu_kernels.o" "$HERE/emu_wave.o" "$HERE/fake_rccl.o"
rm -f "$HERE/emu_nfcgpu.o" "$HERE/emu_trace.o" "$HERE/emu_kernels.o" "$HERE/emu_wave.o" "$HERE/fake_rccl.o"
echo "built $HERE/libnfcgpu_emulated.so"
