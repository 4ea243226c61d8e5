"""CPU tests: the oracle (the real reference, oracle/_ref) is pinned against the reference's golden
vectors, and the CPU build of the device state machine (tests/hostsim) is checked against both."""
import hashlib

import numpy as np
import pytest

import nfc_testlib as T

NAMES = T.fixture_names()


def test_manifest_matches_fixture_files():
    m = T.manifest()
    assert len(m) == 18
    total = 0
    for name, info in m.items():
        raw = T.load_fixture_i16(name)
        assert raw.size == info["samples"]
        assert hashlib.sha256(raw.tobytes()).hexdigest() == info["sha256"]
        assert len(T.load_golden(name)) == info["frames"]
        total += info["frames"]
    assert total == 284


@pytest.mark.parametrize("name", NAMES)
def test_reference_oracle_reproduces_goldens(built, name):
    if T.reference_lib() is None:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    frames, _ = T.reference_decode(T.load_fixture(name))
    assert frames == T.load_golden(name)


@pytest.mark.parametrize("name", NAMES)
def test_step_machine_matches_goldens(built, name):
    frames = T.hostsim_decode(T.load_fixture(name), lane=hash(name) % 64)
    assert frames == T.load_golden(name)


@pytest.mark.parametrize("name", ["test_POLL_ABF_001", "test_NFC-A_106kbps_001", "test_NFC-V_26kbps_002"])
def test_step_machine_matches_reference_with_carrier_frames(built, name):
    if T.reference_lib() is None:
        pytest.skip("oracle/_ref not built")
    x = T.load_fixture(name)
    ref, _ = T.reference_decode(x, keep_carrier=True)
    assert T.hostsim_decode(x, keep_carrier=True) == ref


def test_reference_is_chunking_invariant(built):
    if T.reference_lib() is None:
        pytest.skip("oracle/_ref not built")
    x = T.load_fixture("test_NFC-A_106kbps_002")
    a, _ = T.reference_decode(x, chunk=65536)
    b, _ = T.reference_decode(x, chunk=4099)
    assert a == b == T.load_golden("test_NFC-A_106kbps_002")


def test_iq_magnitude_is_exact_for_axis_aligned_iq(built):
    """The synthetic IQ used by bench.py puts the int16 magnitude on one axis: sqrt(m*m) == |m| in fp32."""
    m = T.load_fixture("test_NFC-B_106kbps_001")
    iq = T.magnitude_to_iq(m, seed=7)
    mag = np.sqrt((iq[0::2] * iq[0::2]).astype(np.float32) + (iq[1::2] * iq[1::2]).astype(np.float32), dtype=np.float32)
    assert np.array_equal(mag, np.abs(m))
    assert T.hostsim_decode(iq, stride=2) == T.hostsim_decode(np.abs(m))


def test_empty_and_tiny_inputs(built):
    assert T.hostsim_decode(np.zeros(0, np.float32)) == []
    assert T.hostsim_decode(np.zeros(10, np.float32)) == []
    if T.reference_lib() is not None:
        ref, _ = T.reference_decode(np.zeros(2000, np.float32), keep_carrier=True)
        assert T.hostsim_decode(np.zeros(2000, np.float32), keep_carrier=True) == ref
