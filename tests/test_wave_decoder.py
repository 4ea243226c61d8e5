"""The wave decoder (nfc-laboratory_amd/csrc/nfc_wave.hpp: one wavefront per lane of the time-parallel path) on the CPU:
tests/hostsim/emu_wave.cpp runs the kernel's own text with 64 fibres per wave (tests/hostsim/wavesim.hpp) inside the
emulated build of the host runtime. With NFC_EMU_WAVE_VERIFY=1 every tile is decoded twice - through the bulk paths
(prefix-sum / walked running sums, gates, folded symbol windows, the search step from bulk values) and sample by sample
through the step machine - and everything the two leave behind (decoder state, correlation and history rings, protocol
state, frame bytes) is compared bit for bit: the process aborts on the first difference. Frames are compared with the
reference decoder as everywhere else."""
import json
import os
import subprocess
import sys

import pytest

import nfc_testlib as T

DRIVER = os.path.join(T.ROOT, "tests", "time_parallel_driver.py")
EMU = os.path.join(T.ROOT, "tests", "hostsim", "libnfcgpu_emulated.so")

needs_reference = pytest.mark.skipif(T.reference_lib() is None, reason="oracle/_ref not built")


@pytest.fixture(scope="module")
def emulated(built):
    if not os.path.exists(EMU):
        subprocess.check_call(["bash", os.path.join(T.ROOT, "tests", "hostsim", "build_emulated.sh")])
    return EMU


def _run(cases, extra=None):
    # (NFCGPU_SOLO_SAMPLES=0: speculative windows also on the short captures)
    env = dict(os.environ, NFCGPU_LIB=EMU, NFCGPU_NO_TORCH="1", NFCGPU_WINDOWED_MIN="4096", NFCGPU_SCAN_CHUNK="32768", NFCGPU_SOLO_SAMPLES="0")
    env.update(extra or {})
    run = subprocess.run([sys.executable, DRIVER] + cases, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=3000)
    assert run.returncode == 0, run.stderr[-3000:]
    res = json.loads(run.stdout.strip().splitlines()[-1])
    assert res
    for r in res:
        assert r["mismatching"] == [] and r["frames"] > 0, r
        assert r["stats"]["windowed"] >= 1 and r["stats"]["fallback"] == 0, r
    return res


@needs_reference
def test_bulk_paths_leave_what_stepping_leaves_on_every_technology(emulated):
    """one capture per technology and rate family (NFC-A 106k / 212k / 424k with its BPSK answers, NFC-B, NFC-F, NFC-V, mixed
    polling), every tile verified"""
    names = ["test_NFC-A_106kbps_001", "test_NFC-A_212kbps_001", "test_NFC-A_424kbps_001", "test_NFC-B_106kbps_001", "test_NFC-F_212kbps_002",
             "test_NFC-V_26kbps_002", "test_POLL_ABF_001"]
    _run(["fixture:" + n for n in names], {"NFC_EMU_WAVE_VERIFY": "1"})


@needs_reference
def test_bulk_paths_on_dense_synthetic_streams_and_carried_state(emulated):
    """dense synthetic streams through the IQ entry, in two submissions (a lane's result becomes the stream's state), every
    tile verified"""
    _run(["synthetic"], {"NFC_EMU_WAVE_VERIFY": "1"})


@needs_reference
def test_stepping_alone_decodes_the_same_frames(emulated):
    """NFC_EMU_WAVE_VERIFY=2: no bulk path at all (every sample through the step machine, rings in LDS)"""
    _run(["fixture:test_NFC-A_106kbps_002", "fixture:test_POLL_AB_001"], {"NFC_EMU_WAVE_VERIFY": "2"})


@needs_reference
def test_short_streams_are_decoded_by_their_carry_lane_alone(emulated):
    """the default (NFCGPU_SOLO_SAMPLES = 2^15 since round 6, 2^16 in rounds 4-5, 2^18 before): a stream of up to 2^15 samples gets no
    speculative windows - one lane, one pass; a longer one does (every bundled capture is longer: the short ones are their first
    samples) - the 65536-sample buffers of the reference's task among them"""
    env = dict(os.environ, NFCGPU_LIB=EMU, NFCGPU_NO_TORCH="1", NFCGPU_WINDOWED_MIN="4096")
    run = subprocess.run([sys.executable, DRIVER, "fixture:test_NFC-A_106kbps_002@30000", "fixture:test_NFC-B_106kbps_001@32768", "fixture:test_NFC-A_106kbps_002@65536"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    results = json.loads(run.stdout.strip().splitlines()[-1])
    assert len(results) == 3
    for r in results:
        assert r["mismatching"] == [] and r["frames"] > 0, r
    for r in results[:2]:
        assert r["stats"]["windowed"] == 1 and r["stats"]["passes"] == 1 and r["stats"]["windows"] == 1, r
    assert results[2]["stats"]["windowed"] == 1 and results[2]["stats"]["windows"] > 1, results[2]
