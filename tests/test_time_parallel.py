"""The time-parallel path (scan kernel -> windows -> windowed decode -> chain, nfc_scan.h) against the reference decoder:
every fixture in one submission and split over buffers, synthetic streams through the IQ entry, input off the int16 grid
(carry lanes alone, running sums walked), a long quiet capture. Without a GPU the cases run on the emulated
runtime of tests/hostsim (the product's host runtime and device code on a stand-in HIP); with `-m gpu` on the real
library and kernels."""
import json
import os
import subprocess
import sys

import pytest

import nfc_testlib as T

DRIVER = os.path.join(T.ROOT, "tests", "time_parallel_driver.py")
EMU = os.path.join(T.ROOT, "tests", "hostsim", "libnfcgpu_emulated.so")
TUNING = os.path.join(T.ROOT, "nfc-laboratory_amd", "libnfcgpu_tuning.so")

needs_reference = pytest.mark.skipif(T.reference_lib() is None, reason="oracle/_ref not built")


def _run(cases, emulated, extra=None):
    # (NFCGPU_SOLO_SAMPLES=0: speculative windows also on the short captures - by default a stream of up to 2^16 samples is
    # decoded by its carry lane alone)
    env = dict(os.environ, NFCGPU_WINDOWED_MIN="4096", NFCGPU_SCAN_CHUNK="32768", NFCGPU_SOLO_SAMPLES="0")
    if emulated:
        env["NFCGPU_LIB"] = EMU
        env["NFCGPU_NO_TORCH"] = "1"
    else:
        # (the product library does not read the tuning switches: the same kernels behind a host runtime that does - `make tuning`)
        env["NFCGPU_LIB"] = TUNING
    env.update(extra or {})
    run = subprocess.run([sys.executable, DRIVER] + cases, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=3000)
    assert run.returncode == 0, run.stderr[-3000:]
    return json.loads(run.stdout.strip().splitlines()[-1])


def _check(results, windowed=True):
    assert results
    for r in results:
        assert r["mismatching"] == [], r
        assert r["frames"] > 0, r
        if windowed:
            assert r["stats"]["windowed"] >= 1 and r["stats"]["fallback"] == 0, r


@pytest.fixture(scope="module")
def emulated(built):
    if not os.path.exists(EMU):
        subprocess.check_call(["bash", os.path.join(T.ROOT, "tests", "hostsim", "build_emulated.sh")])
    return EMU


@needs_reference
def test_fixtures_through_the_time_parallel_path_emulated(emulated):
    _check(_run(["fixtures"], True))


@needs_reference
def test_buffers_synthetic_and_quiet_captures_emulated(emulated):
    res = _run(["buffers", "synthetic", "quiet", "carried"], True)
    _check(res)
    quiet = [r for r in res if r["name"].startswith("one exchange")][0]
    assert quiet["stats"]["windows"] <= 16, quiet  # nearly everything skipped


@needs_reference
def test_small_chunks_force_repairs_emulated(emulated):
    """chunks of 8192 samples with 1024 samples of warm-up: most seams do not verify and are walked again"""
    res = _run(["buffers"], True, {"NFCGPU_SCAN_CHUNK": "8192", "NFCGPU_SCAN_WARM": "1024"})
    _check(res)
    assert sum(r["stats"]["repairs"] for r in res) > 0


@needs_reference
def test_long_busy_submissions_of_few_streams_are_decoded_in_blocks_emulated(emulated):
    """a few long busy streams: the runtime cuts the submission into blocks of NFCGPU_BLOCK_SAMPLES and settles one after the
    other (the passes a submission needs grow with its length); same frames"""
    res = _run(["carried"], True, {"NFCGPU_BLOCK_SAMPLES": "131072"})
    _check(res)
    dense = [r for r in res if r["name"].startswith("3 dense")][0]
    assert dense["stats"]["windowed"] > 9, dense  # 3 streams x 3 submissions, each in several blocks


@needs_reference
def test_a_full_staging_sink_sends_the_submission_to_the_sequential_kernels_emulated(emulated):
    """the lanes chain their frames in a staging sink sized from an estimate; when it runs full (NFCGPU_STAGING_WORDS: a cap for
    this test) frames of live lanes may be missing, so the untouched streams are decoded sequentially: same frames, none lost"""
    res = _run(["buffers"], True, {"NFCGPU_STAGING_WORDS": "96"})
    for r in res:
        assert r["mismatching"] == [] and r["frames"] > 0, r
    assert sum(r["stats"]["fallback"] for r in res) > 0, res


@needs_reference
@pytest.mark.parametrize("knobs,expect", [
    ({"NFCGPU_WINDOWED": "0"}, "sequential"),                                  # the path switched off: sequential kernels only
    ({"NFCGPU_WINDOW_PASSES": "1", "NFCGPU_WINDOW_PASSES_FEW": "1"}, "any"),   # one pass and no more: streams that need another go the sequential way
    ({"NFCGPU_LANES_WANTED": "0", "NFCGPU_CUT_MAX": "16384"}, "windowed"),     # lanes 16384 samples apart whatever the submission
    ({"NFCGPU_SIDE_STREAM": "0"}, "windowed"),                                 # carry lanes on the stream of the windows
    ({"NFCGPU_SCAN_LANES": "1"}, "windowed"),                                  # scan chunks as long as they get (32768 samples)
    ({"NFCGPU_LONG_FIRST": "0"}, "windowed"),                                  # the run list of a pass in stream order
    ({"NFCGPU_LONG_FIRST": "2048", "NFCGPU_LANES_WANTED": "0", "NFCGPU_CUT_MAX": "16384"}, "windowed"),  # ... in six classes of length
    ({"NFCGPU_ENVELOPE_KERNEL": "0"}, "windowed"),                             # the envelope tracker's second walks left to the scan kernel
    ({"NFCGPU_ENVELOPE_FOLLOW": "0"}, "windowed"),                             # ... by the envelope kernel, a round per link of a chain
    ({"NFCGPU_PLANES_PIECE": "0"}, "windowed"),                                # the planes of a small submission a lane per chunk
    ({"NFCGPU_PLANES_BESIDE": "0"}, "windowed"),                               # a large submission's planes after the rounds of second walks
    ({"NFCGPU_PLANES_BESIDE_PIECE": "0"}, "windowed"),                         # ... beside them, a lane per chunk
    ({"NFCGPU_PLANES_BESIDE_PIECE": "512"}, "windowed"),                       # ... a lane per stored point
    ({"NFCGPU_PLANES_PIECE": "2048", "NFCGPU_ENVELOPE_FOLLOW": "1000000"}, "windowed"),  # ... a lane per four points; chains followed whatever the list
])
def test_remaining_knobs_at_non_default_values_emulated(emulated, knobs, expect):
    """every knob that is left (INTEGRATION.md lists them) decodes the same frames at a value that is not its
    default - two dense synthetic streams in three submissions"""
    res = _run(["carried_dense"], True, knobs)
    for r in res:
        assert r["mismatching"] == [] and r["frames"] > 0, r
        if expect == "sequential":
            assert r["stats"]["windowed"] == 0 and r["stats"]["passes"] == 0, r   # (the path is not tried at all)
        elif expect == "windowed":
            assert r["stats"]["windowed"] >= 1 and r["stats"]["fallback"] == 0, r


@needs_reference
def test_envelope_kernel_walks_the_short_lists_of_small_submissions_emulated(emulated):
    """A capture with the default knobs is scanned in chunks of 4096 samples, and after the first round its second walks are the
    envelope tracker's alone: nfc_envelope_kernel takes them (NFCGPU_ENVELOPE_KERNEL: the longest list it is given). Same frames,
    same windows and passes as with the scan kernel's envelope-only branch (knob 0) - and fewer chunks on the lists, in fewer
    rounds: the kernel's walk goes on through a chain of chunks that inherit a wrong envelope from each other, the scan kernel
    walks a chain a chunk per round."""
    cases = ["fixture:test_NFC-A_424kbps_001", "fixture:test_NFC-B_106kbps_001", "fixture:test_NFC-F_212kbps_004"]
    outs = {}
    for knob in ("64", "0"):
        env = dict(os.environ, NFCGPU_LIB=EMU, NFCGPU_NO_TORCH="1", NFCGPU_WINDOWED_MIN="4096", NFCGPU_WINDOW_DEBUG="1", NFCGPU_ENVELOPE_KERNEL=knob)
        run = subprocess.run([sys.executable, DRIVER] + cases, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        assert run.returncode == 0, run.stderr[-3000:]
        outs[knob] = (json.loads(run.stdout.strip().splitlines()[-1]), run.stderr.count("by the envelope kernel"))
    for r in outs["64"][0] + outs["0"][0]:
        assert r["mismatching"] == [] and r["frames"] > 0, r
    assert outs["64"][1] > 0 and outs["0"][1] == 0, (outs["64"][1], outs["0"][1])
    def but_repairs(stats):
        return {k: v for k, v in stats.items() if k != "repairs"}

    assert [but_repairs(r["stats"]) for r in outs["64"][0]] == [but_repairs(r["stats"]) for r in outs["0"][0]]
    listed = [sum(r["stats"]["repairs"] for r in outs[knob][0]) for knob in ("64", "0")]
    assert 0 < listed[0] <= listed[1], listed


@needs_reference
@pytest.mark.parametrize("knobs, beside", [({}, True), ({"NFCGPU_PLANES_BESIDE_PIECE": "0"}, True), ({"NFCGPU_PLANES_BESIDE_PIECE": "512"}, True),
                                           ({"NFCGPU_PLANES_BESIDE": "0"}, False),
                                           # chunks that are no multiple of the walk's default piece of 2048 samples (the default
                                           # sizing gives any multiple of 512 for totals of 2^28 .. 2^30 samples; here 9216 = 4.5 pieces):
                                           # the pieces have to tile the chunk, or the tail of every chunk gets no planes
                                           ({"NFCGPU_SCAN_CHUNK": "9216"}, True), ({"NFCGPU_SCAN_CHUNK": "9216", "NFCGPU_PLANES_BESIDE_PIECE": "4096"}, True),
                                           # the emulated runtime ran the low-priority stream's walk at its launch, i.e. always
                                           # before the rounds; NFC_EMU_DEFER_LOW keeps that stream's launches until somebody waits for it
                                           # (tests/hostsim/fakehip): the walk then runs after every rewrite of the rounds - the other order
                                           ({"NFC_EMU_DEFER_LOW": "1"}, True), ({"NFC_EMU_DEFER_LOW": "1", "NFCGPU_SCAN_CHUNK": "9216"}, True)])
def test_planes_written_beside_the_rounds_of_second_walks_emulated(emulated, knobs, beside):
    """Round 5: a large submission's front-end planes are written by a walk that starts when the first round's second walks are
    queued; the seam check and the envelope walks note every start state they rewrite from then on, and those chunks' planes are
    written again (a lane per stored point) when the rounds are over. Five dense streams x 2^20 (more than 4 Mi samples) with the
    default piece of that walk, a lane per chunk, a lane per point, and with the walk after the rounds: the reference's frames."""
    env = dict(os.environ, NFCGPU_LIB=EMU, NFCGPU_NO_TORCH="1", NFCGPU_WINDOW_DEBUG="1")
    env.update(knobs)
    run = subprocess.run([sys.executable, DRIVER, "beside"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=3000)
    assert run.returncode == 0, run.stderr[-3000:]
    res = json.loads(run.stdout.strip().splitlines()[-1])
    for r in res:
        assert r["mismatching"] == [] and r["frames"] > 0 and r["stats"]["fallback"] == 0, r
    again = [int(l.split(";")[1].split()[0]) for l in run.stderr.splitlines() if "planes written beside the rounds" in l]
    assert (len(again) == 1 and again[0] > 0) if beside else again == [], (again, run.stderr[-2000:])


@needs_reference
def test_planes_that_do_not_fit_the_device_are_not_fatal_emulated(emulated):
    """the wave path's front-end planes are 16 bytes per sample of the submission; when the device cannot give
    them the submission used to fail with NFCGPU_ENOMEM. It is now decoded a quarter of its length at a time (a quarter of
    the planes) and, if that does not fit either, by the sequential kernels - same frames either way."""
    full = 2 * (1 << 19) * 16
    res = _run(["planes"], True, {"NFCGPU_TEST_ALLOC_LIMIT": str(full // 2)})       # the quarter fits
    _check(res)
    assert res[0]["stats"]["windowed"] == 4 * 2 and res[0]["stats"]["fallback"] == 0, res   # four blocks of two streams
    res = _run(["planes"], True, {"NFCGPU_TEST_ALLOC_LIMIT": str(full // 8)})       # nothing fits: sequential kernels
    _check(res, windowed=False)
    assert res[0]["stats"]["windowed"] == 0 and res[0]["stats"]["fallback"] == 4 * 2, res   # (counted per block)
    # the other work buffers of a submission take the same way out (here: the lanes' decoder states, 364 bytes a lane;
    # the whole submission wants ~330 lanes, a quarter of it ~230)
    res = _run(["planes"], True, {"NFCGPU_TEST_ALLOC_LIMIT_LANES": "100000"})        # the quarter fits
    _check(res)
    assert res[0]["stats"]["windowed"] == 4 * 2 and res[0]["stats"]["fallback"] == 0, res
    res = _run(["planes"], True, {"NFCGPU_TEST_ALLOC_LIMIT_LANES": "50000"})         # nothing fits: sequential kernels
    _check(res, windowed=False)
    assert res[0]["stats"]["windowed"] == 0 and res[0]["stats"]["fallback"] == 4 * 2, res


@needs_reference
def test_random_multi_submission_scenarios_emulated(emulated):
    """a short run of profiles/tools/r02/emulated_fuzz.py (random mixes of sparse and dense streams cut into submissions at
    random samples: both paths, carried state, final-state fix-ups) - the long runs are in profiles/r02/emulated_fuzz.json"""
    fuzz = os.path.join(T.ROOT, "profiles", "tools", "r02", "emulated_fuzz.py")
    env = dict(os.environ, NFCGPU_LIB=EMU, NFCGPU_NO_TORCH="1", NFCGPU_WINDOWED_MIN="32768")
    run = subprocess.run([sys.executable, fuzz, "5", "35", "small"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-2000:]
    res = json.loads(run.stdout.strip().splitlines()[-1])
    assert res["rounds"] >= 3 and res["mismatches"] == [], res


@needs_reference
@pytest.mark.gpu
def test_random_multi_submission_scenarios_on_the_gpu(built):
    """a minute of the same generator on the real library (full-size scenarios: up to 23 streams of up to 1.3 M samples, cut
    into up to five submissions at random samples, a quarter of them off the int16 grid), default knobs"""
    fuzz = os.path.join(T.ROOT, "profiles", "tools", "r02", "emulated_fuzz.py")
    env = {k: v for k, v in os.environ.items() if not k.startswith("NFCGPU_")}
    run = subprocess.run([sys.executable, fuzz, "2026", "60"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-2000:]
    res = json.loads(run.stdout.strip().splitlines()[-1])
    assert res["rounds"] >= 5 and res["mismatches"] == [], res


@needs_reference
def test_input_off_the_grid_is_decoded_by_carry_lanes_with_walked_sums_emulated(emulated):
    """float input off the int16 grid (what a radio delivers) stays on the path: no speculative windows - off the grid no lane
    that starts inside a stream can be in the decoder's state - but the stream's carry lane decodes it, one wavefront per
    stream, the raw running sums walked in the step's order; every tile decoded twice (bulk / stepped) and compared"""
    # (the CPU run: three of the five captures and one synthetic stream; the GPU test below takes all of them)
    res = _run(["offgrid"], True, {"NFC_EMU_WAVE_VERIFY": "1", "NFC_TEST_OFFGRID_CAPTURES": "3", "NFC_TEST_OFFGRID_SYNTHETIC": "1"})
    _check(res)
    for r in res:
        assert r["stats"]["fallback"] == 0, r


@needs_reference
@pytest.mark.gpu
def test_fixtures_through_the_time_parallel_path_on_the_gpu(built):
    _check(_run(["fixtures"], False))


@needs_reference
@pytest.mark.gpu
def test_buffers_synthetic_quiet_and_offgrid_on_the_gpu(built):
    res = _run(["buffers", "synthetic", "quiet", "carried"], False)
    _check(res)
    res = _run(["offgrid"], False)
    _check(res)  # (off the grid: carry lanes with walked sums, nothing decoded by the sequential kernels)
    for r in res:
        assert r["stats"]["fallback"] == 0, r
    res = _run(["buffers"], False, {"NFCGPU_SCAN_CHUNK": "8192", "NFCGPU_SCAN_WARM": "1024"})
    _check(res)
