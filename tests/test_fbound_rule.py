"""The rule by which a lane of the time-parallel path stands although the NFC-F pulse memory it had assumed was not the true
one (nfc_fbound_admits): property check on the preamble tracker itself (tests/hostsim/fbound_check.cpp)."""
import json
import os
import subprocess

import nfc_testlib as T

SRC = os.path.join(T.ROOT, "tests", "hostsim", "fbound_check.cpp")


def test_lanes_admitted_on_another_pulse_memory_decide_and_leave_the_same(tmp_path):
    exe = str(tmp_path / "fbound_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-strict-aliasing", "-I" + os.path.join(T.ROOT, "tests", "hostsim", "fakehip"), SRC, "-o", exe])
    for seed in (1, 2):
        run = subprocess.run([exe, str(seed), "150000"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        assert run.returncode == 0, run.stderr[-2000:]
        res = json.loads(run.stdout.strip().splitlines()[-1])
        assert res["violations"] == 0
        # the check has to have had something to check: memories that differ and are admitted, with and without the record starting over
        assert res["admitted_with_another_memory"] > 20000 and res["of_them_counter_off"] > 5000 and res["of_them_threshold_off"] > 5000, res
        assert res["admitted"] - res["admitted_that_started_over"] > 100, res
