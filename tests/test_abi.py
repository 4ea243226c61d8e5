"""CPU tests of the drop-in boundary: the C-ABI library loads, exports exactly what include/nfcgpu.h
declares, and refuses to work without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest

import nfc_testlib as T

HEADER = os.path.join(T.ROOT, "include", "nfcgpu.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nfcgpu_[a-z_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for required in ["nfcgpu_init", "nfcgpu_shutdown", "nfcgpu_stream_open", "nfcgpu_stream_configure",
                     "nfcgpu_submit", "nfcgpu_submit_batch", "nfcgpu_poll", "nfcgpu_flush",
                     "nfcgpu_stream_close", "nfcgpu_strerror"]:
        assert required in names


def test_library_exports_every_declared_symbol(built):
    import nfclab_amd
    lib = ctypes.CDLL(nfclab_amd.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), name


def test_frame_layout_matches_header(built):
    import nfclab_amd
    assert ctypes.sizeof(nfclab_amd.Frame) == 7 * 4 + 4 + 3 * 8 + 512
    assert ctypes.sizeof(T.Frame) == ctypes.sizeof(nfclab_amd.Frame)
    p = nfclab_amd.default_params()
    assert p.tech_mask == 0xF and abs(p.power_level_threshold - 0.01) < 1e-9
    assert list(p.corr_threshold) == pytest.approx([0.75, 0.5, 0.5, 0.5])
    assert list(p.min_modulation_depth) == pytest.approx([0.9, 0.1, 0.1, 0.9])
    assert list(p.max_modulation_depth) == pytest.approx([1.0, 0.9, 0.9, 1.0])


def test_no_cpu_fallback_without_gpu(built):
    import torch
    import nfclab_amd
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(nfclab_amd.NfcGpuError) as e:
        nfclab_amd.NfcGpu()
    assert e.value.code == -2  # NFCGPU_ENODEV


def test_product_sources_do_not_reference_the_oracle():
    """Only tests/, bench.py's cpu_baseline and __graft_entry__.smoke may touch oracle/."""
    pkg = os.path.join(T.ROOT, "nfc-laboratory_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".so", ".o", ".pyc")):
                continue
            text = open(os.path.join(dirpath, fn), errors="ignore").read()
            assert "oracle/" not in text and "libnfcref" not in text and "hostsim" not in text.replace("tests/hostsim", ""), fn


def test_decoder_shim_refuses_loudly_without_gpu(built, tmp_path):
    """The reference's own test-sdr harness linked against the lab::NfcDecoder shim: on a box without a GPU it must
    not decode anything by other means; the constructor throws (no CPU fallback behind the reference's interface)."""
    import subprocess
    import torch
    exe = os.path.join(T.ROOT, "oracle", "_ref", "test-sdr-gpu")
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    if not os.path.exists(exe):
        pytest.skip("test-sdr-gpu not built (needs the reference tree at build time)")
    T.write_wav(str(tmp_path / "test_NFC-A_106kbps_001.wav"), T.load_fixture_i16("test_NFC-A_106kbps_001"))
    run = subprocess.run([exe, str(tmp_path) + "/"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert run.returncode != 0
    assert "PASS" not in run.stdout
    assert "no usable HIP device" in run.stdout


def _env_names(path):
    data = open(path, "rb").read()
    return sorted(set(m.decode() for m in re.findall(rb"NFCGPU_[A-Z][A-Z_]+(?=\x00)", data)))


def test_product_library_reads_no_tuning_switches(built):
    """the tuning switches of the time-parallel path are experiment switches. The product library knows three
    diagnostic variables and nothing else (the shim's own - device, stream count, block mode - are in host/NfcDecoder.cpp); the
    tuning build of the same kernels (`make tuning`: libnfcgpu_tuning.so, what the tests force paths with) reads the rest."""
    pkg = os.path.join(T.ROOT, "nfc-laboratory_amd")
    product = [n for n in _env_names(os.path.join(pkg, "libnfcgpu.so")) if not n.startswith("NFCGPU_E") and n != "NFCGPU_OK"]
    assert product == ["NFCGPU_GENERIC_KERNELS", "NFCGPU_WAVE_VERIFY_REPORT", "NFCGPU_WINDOW_DEBUG"], product
    tuning = _env_names(os.path.join(pkg, "libnfcgpu_tuning.so"))
    for name in ("NFCGPU_WINDOWED", "NFCGPU_WINDOWED_MIN", "NFCGPU_SCAN_CHUNK", "NFCGPU_SCAN_LANES", "NFCGPU_SOLO_SAMPLES", "NFCGPU_CUT_MAX", "NFCGPU_PLANES_BESIDE"):
        assert name in tuning, (name, tuning)
    lib = ctypes.CDLL(os.path.join(pkg, "libnfcgpu_tuning.so"))
    for name in declared_functions():
        assert hasattr(lib, name), name
