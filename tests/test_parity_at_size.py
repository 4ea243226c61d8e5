"""Parity at BASELINE's sizes with the library's DEFAULT routing and thresholds (no NFCGPU_* knob set): config 5's 4096
streams x 2^20 samples of sparse traffic, and dense traffic (the captures tiled end to end) in two submissions so that
state is carried from one to the next - every stream compared frame by frame with the reference decoder
(tests/parity_sweep_driver.py). On the GPU only."""
import json
import os
import subprocess
import sys

import pytest

import nfc_testlib as T

DRIVER = os.path.join(T.ROOT, "tests", "parity_sweep_driver.py")


def _sweep(kind, streams, samples, submissions):
    env = {k: v for k, v in os.environ.items() if not k.startswith("NFCGPU_")}
    run = subprocess.run([sys.executable, DRIVER, kind, str(streams), str(samples), str(submissions)], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=1500)
    assert run.returncode == 0, run.stderr[-3000:]
    return json.loads(run.stdout.strip().splitlines()[-1])


def _check(res):
    assert res["streams_mismatching"] == [], res
    assert res["frames_dropped"] == 0 and res["reference_frames"] > res["streams"], res
    assert res["streams_compared"] == res["streams"]
    return res


needs_reference = pytest.mark.skipif(T.reference_lib() is None, reason="oracle/_ref not built")


@needs_reference
@pytest.mark.gpu
def test_config5_sparse_every_stream_matches_the_reference(built):
    res = _check(_sweep("sparse", 4096, 1 << 20, 1))
    assert res["time_parallel_streams"] == 4096 and res["sequential_streams"] == 0, res


@needs_reference
@pytest.mark.gpu
def test_dense_streams_in_two_submissions_match_the_reference(built):
    res = _check(_sweep("dense", 512, 1 << 20, 2))
    # dense traffic is decoded where it is; a stream whose lanes do not settle within the pass limit is decoded sequentially
    # from its untouched state (exact either way: every stream has been compared above)
    assert res["time_parallel_streams"] + res["sequential_streams"] == 1024 and res["sequential_streams"] <= 8, res


@needs_reference
@pytest.mark.gpu
def test_many_dense_streams_of_short_submissions_match_the_reference(built):
    _check(_sweep("dense", 2048, 1 << 18, 2))


@needs_reference
@pytest.mark.gpu
def test_config5_dense_every_stream_matches_the_reference(built):
    """the headline of bench.py as it is submitted: 4096 dense streams x 2^20 samples in one submission (16 384+ lanes, lanes up
    to 131072 samples apart: another lane geometry than the 512-stream case above), default knobs, every stream compared"""
    res = _check(_sweep("dense", 4096, 1 << 20, 1))
    assert res["time_parallel_streams"] + res["sequential_streams"] == 4096 and res["sequential_streams"] <= 32, res


@needs_reference
@pytest.mark.gpu
def test_off_grid_streams_in_two_submissions_match_the_reference(built):
    """what a radio delivers - float IQ off the int16 grid of the captures (set S2: the dense set on a random phase
    per stream plus noise) - at size: 512 streams x 2^20 samples in two submissions, default knobs, EVERY stream compared. Such a
    stream stays on the time-parallel path (its carry lane decodes it alone, the running sums walked in the reference's order)."""
    res = _check(_sweep("offgrid", 512, 1 << 19, 2))
    assert res["time_parallel_streams"] == 1024 and res["sequential_streams"] == 0, res
