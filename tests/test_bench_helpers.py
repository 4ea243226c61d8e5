"""bench.py prints roofline.traffic from the counter record kept under profiles/ only while the record belongs to the kernel
sources at hand: the record carries a digest of csrc/ (without the host runtime) and is refused,
with the reason, when that digest is another one."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench


def test_the_committed_traffic_record_carries_source_and_commit():
    with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
        rec = json.load(f)
    for kernel in ("nfc_demod_fixed_kernel", "nfc_scan_kernel"):
        assert rec[kernel]["hbm_bytes_per_launch"] > 0
        assert rec[kernel]["from"].startswith("profiles/") and rec[kernel]["git"] and rec[kernel]["sources_sha1"]


def test_a_record_of_other_sources_or_another_shape_is_refused(tmp_path, monkeypatch):
    os.makedirs(tmp_path / "profiles")
    os.symlink(os.path.join(ROOT, "nfc-laboratory_amd"), tmp_path / "nfc-laboratory_amd")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    good = {"nfc_scan_kernel": {"streams": 4096, "samples": 1 << 20, "hbm_bytes_per_launch": 1.0, "sources_sha1": bench.sources_digest(), "git": "abc", "from": "profiles/x.json"}}
    with open(tmp_path / "profiles" / "traffic.json", "w") as f:
        json.dump(good, f)
    value, source = bench.stored_traffic("nfc_scan_kernel", 4096, 1 << 20)
    assert value == 1.0 and "abc" in source
    assert bench.stored_traffic("nfc_scan_kernel", 512, 1 << 20) == (None, None)          # another shape
    assert bench.stored_traffic("nfc_demod_fixed_kernel", 131072, 8192) == (None, None)   # no record
    good["nfc_scan_kernel"]["sources_sha1"] = "0" * 16
    with open(tmp_path / "profiles" / "traffic.json", "w") as f:
        json.dump(good, f)
    value, source = bench.stored_traffic("nfc_scan_kernel", 4096, 1 << 20)
    assert value is None and "other kernel sources" in source


def test_the_headline_of_the_drivers_command_fits_the_gpu():
    """`bench.py --gpus 1 --steps 20 --warmup 5` asked for 800 GiB (a fresh slice per step). The dataset is
    SURVEY 8(d)'s L samples per stream, a fixed number of slices of it resident whatever the step count"""
    gib = float(1 << 30)
    for steps, warmup in ((2, 1), (20, 5), (200, 50)):
        slices, resident = bench.headline_layout(steps, warmup, 1 << 20, 2)
        assert slices <= 2 and resident == slices << 20
        assert bench.headline_device_bytes(4096, 1 << 20, steps, warmup) < 200 * gib
    assert bench.headline_layout(1, 0, 1 << 20, 2) == (1, 1 << 20)
    # the held sink grows with the step count: twice the 2.04 M words a step of config 5 leaves on dense traffic
    assert bench.headline_sink_words(4096, 1 << 20, 20, 5) >= 25 * 2 * 2040000
    assert bench.headline_sink_words(512, 1 << 20, 2, 1) >= 16 << 20
    # N ranks under strong scaling: every rank's share plus a gather buffer for all of them
    assert bench.headline_device_bytes(512, 1 << 20, 20, 5, world=8) < 40 * gib
