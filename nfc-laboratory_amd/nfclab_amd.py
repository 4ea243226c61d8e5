"""ctypes binding of libnfcgpu.so (include/nfcgpu.h) for the tests and bench.py.

This is plumbing only: the product is the C-ABI library; the reference's host language is C++ and
its drop-in host side lives in nfc-laboratory_amd/host/. Nothing here decodes on the CPU: if the
library or a GPU is missing, NfcGpu() raises.
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NFCGPU_LIB", os.path.join(HERE, "libnfcgpu.so"))

TECH_A, TECH_B, TECH_F, TECH_V = 1, 2, 4, 8
LOC_HOST, LOC_DEVICE = 0, 1

FRAME_CARRIER_OFF, FRAME_CARRIER_ON, FRAME_POLL, FRAME_LISTEN = 0x100, 0x101, 0x102, 0x103


class Params(ctypes.Structure):
    _fields_ = [
        ("sample_rate", ctypes.c_uint32),
        ("tech_mask", ctypes.c_uint32),
        ("stream_time", ctypes.c_int64),
        ("power_level_threshold", ctypes.c_float),
        ("corr_threshold", ctypes.c_float * 4),
        ("min_modulation_depth", ctypes.c_float * 4),
        ("max_modulation_depth", ctypes.c_float * 4),
    ]


class Frame(ctypes.Structure):
    _fields_ = [
        ("stream_id", ctypes.c_uint32),
        ("tech_type", ctypes.c_uint32),
        ("frame_type", ctypes.c_uint32),
        ("frame_flags", ctypes.c_uint32),
        ("frame_phase", ctypes.c_uint32),
        ("frame_rate", ctypes.c_uint32),
        ("length", ctypes.c_uint32),
        ("reserved", ctypes.c_uint32),
        ("sample_start", ctypes.c_uint64),
        ("sample_end", ctypes.c_uint64),
        ("sample_rate", ctypes.c_uint64),
        ("data", ctypes.c_uint8 * 512),
    ]

    def as_tuple(self):
        return (self.tech_type, self.frame_type, self.frame_flags, self.frame_phase, self.frame_rate,
                self.sample_start, self.sample_end, self.sample_rate, bytes(self.data[:self.length]))


class Options(ctypes.Structure):
    _fields_ = [
        ("max_streams", ctypes.c_uint32),
        ("reserved", ctypes.c_uint32),
        ("frame_sink_bytes", ctypes.c_uint64),
    ]


class Batch(ctypes.Structure):
    _fields_ = [
        ("n_streams", ctypes.c_uint32),
        ("stride", ctypes.c_uint32),
        ("location", ctypes.c_uint32),
        ("sample_rate", ctypes.c_uint32),
        ("stream_ids", ctypes.POINTER(ctypes.c_uint32)),
        ("data", ctypes.POINTER(ctypes.c_void_p)),
        ("n_samples", ctypes.POINTER(ctypes.c_uint32)),
    ]


class Stats(ctypes.Structure):
    _fields_ = [
        ("launches", ctypes.c_uint64),
        ("samples", ctypes.c_uint64),
        ("frames", ctypes.c_uint64),
        ("dropped_frames", ctypes.c_uint64),
        ("kernel_ms", ctypes.c_double),
        ("scan_ms", ctypes.c_double),
        ("window_ms", ctypes.c_double),
        ("scan_samples", ctypes.c_uint64),
        ("windows", ctypes.c_uint64),
        ("window_passes", ctypes.c_uint64),
        ("windowed_streams", ctypes.c_uint64),
        ("fallback_streams", ctypes.c_uint64),
        ("scan_repairs", ctypes.c_uint64),
        ("wave_ms", ctypes.c_double),
        ("wave_launches", ctypes.c_uint64),
        ("planes_ms", ctypes.c_double),
        ("wave_busy_ms", ctypes.c_double),
    ]


class NfcGpuError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("nfcgpu error %d: %s" % (code, message))
        self.code = code


_lib = None


def load_library(path=LIB_PATH):
    """Load libnfcgpu.so and declare prototypes. Raises OSError if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own ROCm runtime (same SONAMEs as /opt/rocm). Whichever copy is loaded first serves the
    # whole process, and torch only finds the GPU through its own copy: load torch first when it is installed.
    if os.environ.get("NFCGPU_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    lib = ctypes.CDLL(path)
    vp, u32, u64, i32 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int
    P = ctypes.POINTER
    lib.nfcgpu_default_params.argtypes = [P(Params)]
    lib.nfcgpu_default_params.restype = None
    lib.nfcgpu_init.argtypes = [i32, P(Options), P(vp)]
    lib.nfcgpu_shutdown.argtypes = [vp]
    lib.nfcgpu_stream_open.argtypes = [vp, P(Params), P(u32)]
    lib.nfcgpu_stream_open_many.argtypes = [vp, P(Params), u32, P(u32)]
    lib.nfcgpu_stream_configure.argtypes = [vp, u32, P(Params)]
    lib.nfcgpu_stream_reset.argtypes = [vp, u32]
    lib.nfcgpu_stream_close.argtypes = [vp, u32]
    lib.nfcgpu_submit.argtypes = [vp, u32, vp, u32, u32, u32]
    lib.nfcgpu_submit_batch.argtypes = [vp, P(Batch)]
    lib.nfcgpu_submit_uniform.argtypes = [vp, u32, u32, vp, u64, u32, u32, u32, u32]
    lib.nfcgpu_magnitude.argtypes = [vp, vp, u64, vp, u32]
    lib.nfcgpu_resample_radio.argtypes = [vp, vp, u64, u32, u32, vp, u64, u32, vp, u32]
    lib.nfcgpu_flush.argtypes = [vp, u32]
    lib.nfcgpu_sync.argtypes = [vp]
    lib.nfcgpu_poll.argtypes = [vp, u32, P(Frame), u32, P(u32)]
    lib.nfcgpu_pending.argtypes = [vp, u32, P(u32)]
    lib.nfcgpu_sink_device_view.argtypes = [vp, P(vp), P(vp), P(u64)]
    lib.nfcgpu_sink_attach.argtypes = [vp, vp, u64, vp]
    lib.nfcgpu_sink_hold.argtypes = [vp, i32]
    lib.nfcgpu_sink_rewind.argtypes = [vp]
    lib.nfcgpu_stats_get.argtypes = [vp, P(Stats)]
    lib.nfcgpu_stats_get_sized.argtypes = [vp, P(Stats), ctypes.c_uint32]
    lib.nfcgpu_stats_reset.argtypes = [vp]
    lib.nfcgpu_profile.argtypes = [vp, i32]
    lib.nfcgpu_comm_unique_id.argtypes = [vp]
    lib.nfcgpu_comm_init.argtypes = [vp, vp, i32, i32]
    lib.nfcgpu_comm_destroy.argtypes = [vp]
    lib.nfcgpu_gather_frames.argtypes = [vp, vp, ctypes.c_uint64, P(ctypes.c_uint32), P(ctypes.c_uint64)]
    lib.nfcgpu_gather_frames_packed.argtypes = [vp, vp, ctypes.c_uint64, P(ctypes.c_uint32)]
    lib.nfcgpu_trace_write.argtypes = [vp, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_double, ctypes.c_double, P(ctypes.c_uint32)]
    lib.nfcgpu_trace_write_frames.argtypes = [ctypes.c_char_p, vp, ctypes.c_uint32, ctypes.c_int64, ctypes.c_double, ctypes.c_double, P(ctypes.c_uint32)]
    lib.nfcgpu_read_bandwidth.argtypes = [vp, vp, ctypes.c_uint64, ctypes.c_uint32, P(ctypes.c_double)]
    lib.nfcgpu_hip_stream.argtypes = [vp]
    lib.nfcgpu_hip_stream.restype = vp
    lib.nfcgpu_strerror.argtypes = [i32]
    lib.nfcgpu_strerror.restype = ctypes.c_char_p
    lib.nfcgpu_last_error.argtypes = [vp]
    lib.nfcgpu_last_error.restype = ctypes.c_char_p
    lib.nfcgpu_version.restype = ctypes.c_char_p
    _lib = lib
    return lib


def default_params(sample_rate=0, tech_mask=0xF):
    p = Params()
    load_library().nfcgpu_default_params(ctypes.byref(p))
    p.sample_rate = sample_rate
    p.tech_mask = tech_mask
    return p


class NfcGpu:
    """One nfcgpu context (one GPU, one HIP stream)."""

    def __init__(self, device=0, max_streams=1024, frame_sink_bytes=64 << 20):
        self.lib = load_library()
        self.ctx = ctypes.c_void_p()
        opts = Options(max_streams, 0, frame_sink_bytes)
        rc = self.lib.nfcgpu_init(device, ctypes.byref(opts), ctypes.byref(self.ctx))
        if rc != 0:
            self.ctx = None
            raise NfcGpuError(rc, self.lib.nfcgpu_strerror(rc).decode())

    def _check(self, rc, allow=()):
        if rc != 0 and rc not in allow:
            msg = self.lib.nfcgpu_strerror(rc).decode()
            detail = self.lib.nfcgpu_last_error(self.ctx).decode()
            raise NfcGpuError(rc, "%s (%s)" % (msg, detail))
        return rc

    def close(self):
        if self.ctx:
            self.lib.nfcgpu_shutdown(self.ctx)
            self.ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def open(self, params=None, count=1):
        first = ctypes.c_uint32()
        p = params or default_params()
        self._check(self.lib.nfcgpu_stream_open_many(self.ctx, ctypes.byref(p), count, ctypes.byref(first)))
        return first.value

    def configure(self, stream, params):
        self._check(self.lib.nfcgpu_stream_configure(self.ctx, stream, ctypes.byref(params)))

    def reset(self, stream):
        self._check(self.lib.nfcgpu_stream_reset(self.ctx, stream))

    def close_stream(self, stream):
        self._check(self.lib.nfcgpu_stream_close(self.ctx, stream))

    def submit(self, stream, samples, sample_rate, stride=1):
        """samples: contiguous float32 numpy array (host)."""
        n = samples.size // stride
        self._check(self.lib.nfcgpu_submit(self.ctx, stream, samples.ctypes.data, n, stride, sample_rate))

    def submit_batch(self, stream_ids, pointers, counts, sample_rate, stride=1, location=LOC_HOST):
        n = len(stream_ids)
        ids = (ctypes.c_uint32 * n)(*stream_ids)
        ptrs = (ctypes.c_void_p * n)(*pointers)
        cnts = (ctypes.c_uint32 * n)(*counts)
        b = Batch(n, stride, location, sample_rate, ids, ptrs, cnts)
        self._check(self.lib.nfcgpu_submit_batch(self.ctx, ctypes.byref(b)))

    def submit_uniform(self, first, count, base_ptr, pitch_bytes, n_samples, sample_rate, stride=1, location=LOC_DEVICE):
        self._check(self.lib.nfcgpu_submit_uniform(self.ctx, first, count, base_ptr, pitch_bytes, n_samples, stride,
                                                   location, sample_rate))

    def magnitude(self, iq):
        """|IQ| of an interleaved float32 numpy array (host memory), computed on the device with the reference's
        roundings."""
        iq = np.ascontiguousarray(iq, dtype=np.float32)
        n = iq.size // 2
        out = np.empty(n, dtype=np.float32)
        self._check(self.lib.nfcgpu_magnitude(self.ctx, iq.ctypes.data, n, out.ctypes.data, LOC_HOST))
        return out

    def resample_radio(self, buffers, capacity_pairs=None):
        """Adaptive (value, offset) control points of each row of a 2-D float32 array of magnitude buffers (host memory);
        returns a list of (pairs, 2) arrays."""
        buffers = np.ascontiguousarray(buffers, dtype=np.float32)
        nb, n = buffers.shape
        cap = capacity_pairs or (n + n // 255 + 2)
        out = np.zeros((nb, 2 * cap), dtype=np.float32)
        counts = np.zeros(nb, dtype=np.uint32)
        self._check(self.lib.nfcgpu_resample_radio(self.ctx, buffers.ctypes.data, n * 4, nb, n, out.ctypes.data, 2 * cap * 4, cap,
                                                   counts.ctypes.data, LOC_HOST))
        return [out[b, :2 * counts[b]].reshape(-1, 2) for b in range(nb)]

    def resample_radio_device(self, in_ptr, in_pitch_bytes, n_buffers, n_samples, out_ptr, out_pitch_bytes, capacity_pairs, counts_ptr):
        """Same with device pointers (input, output and counts resident in HBM)."""
        self._check(self.lib.nfcgpu_resample_radio(self.ctx, in_ptr, in_pitch_bytes, n_buffers, n_samples, out_ptr, out_pitch_bytes,
                                                   capacity_pairs, counts_ptr, LOC_DEVICE))

    def flush(self, stream):
        self._check(self.lib.nfcgpu_flush(self.ctx, stream))

    def sync(self):
        self._check(self.lib.nfcgpu_sync(self.ctx))

    def poll(self, stream, capacity=4096):
        out = (Frame * capacity)()
        n = ctypes.c_uint32()
        self._check(self.lib.nfcgpu_poll(self.ctx, stream, out, capacity, ctypes.byref(n)))
        return [out[i].as_tuple() for i in range(n.value)]

    def sink_view(self):
        words, cursor, cap = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_uint64()
        self._check(self.lib.nfcgpu_sink_device_view(self.ctx, ctypes.byref(words), ctypes.byref(cursor), ctypes.byref(cap)))
        return words.value, cursor.value, cap.value

    def sink_attach(self, words_ptr, capacity_words, ctl_ptr):
        self._check(self.lib.nfcgpu_sink_attach(self.ctx, words_ptr, capacity_words, ctl_ptr))

    def sink_hold(self, hold):
        self._check(self.lib.nfcgpu_sink_hold(self.ctx, int(hold)))

    def sink_rewind(self):
        self._check(self.lib.nfcgpu_sink_rewind(self.ctx))

    def profile(self, enable):
        self._check(self.lib.nfcgpu_profile(self.ctx, int(enable)))

    def stats(self):
        s = Stats()
        self._check(self.lib.nfcgpu_stats_get_sized(self.ctx, ctypes.byref(s), ctypes.sizeof(Stats)))
        return s

    def stats_reset(self):
        self._check(self.lib.nfcgpu_stats_reset(self.ctx))

    def hip_stream(self):
        return self.lib.nfcgpu_hip_stream(self.ctx)

    # ---- multi-GPU frame gather over RCCL, behind the C ABI ----
    def comm_unique_id(self):
        """128 opaque bytes made by rank 0 (ncclGetUniqueId); carry them to the other ranks"""
        buf = ctypes.create_string_buffer(128)
        self._check(self.lib.nfcgpu_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, unique_id, rank, n_ranks):
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        self._check(self.lib.nfcgpu_comm_init(self.ctx, buf, rank, n_ranks))
        self._n_ranks = n_ranks

    def comm_destroy(self):
        self._check(self.lib.nfcgpu_comm_destroy(self.ctx))

    def gather_frames(self, gathered_ptr, capacity_words, packed=True):
        """every rank's frame records into the device buffer at gathered_ptr (RCCL behind the C ABI); returns (counts, stride):
        packed (nfcgpu_gather_frames_packed): rank r's records start at sum(counts[:r]), stride 0; otherwise
        (nfcgpu_gather_frames, the layout of rounds 1-2) at r * stride, stride = max(counts)"""
        counts = (ctypes.c_uint32 * self._n_ranks)()
        if packed:
            self._check(self.lib.nfcgpu_gather_frames_packed(self.ctx, gathered_ptr, capacity_words, counts), allow=(-6,))
            return list(counts), 0
        stride = ctypes.c_uint64()
        self._check(self.lib.nfcgpu_gather_frames(self.ctx, gathered_ptr, capacity_words, counts, ctypes.byref(stride)), allow=(-6,))
        return list(counts), int(stride.value)

    def trace_write(self, stream_id, path, range_start=0.0, range_end=0.0):
        """the frames decoded for a stream (still queued) as a .trz the reference application opens; returns the frame count"""
        n = ctypes.c_uint32()
        self._check(self.lib.nfcgpu_trace_write(self.ctx, stream_id, os.fsencode(path), range_start, range_end, ctypes.byref(n)), allow=(-6,))
        return int(n.value)

    def read_bandwidth(self, device_ptr, n_bytes, repeats=5):
        """streaming-read GB/s over a device buffer (16-byte loads): the measured HBM roofline denominator"""
        g = ctypes.c_double()
        self._check(self.lib.nfcgpu_read_bandwidth(self.ctx, device_ptr, n_bytes, repeats, ctypes.byref(g)))
        return g.value
