"""Deterministic synthetic capture streams for bench.py and the full-size tests (set S1).

Stream `s` is the concatenation of the 18 reference captures (tests/golden/wav, mono int16 @ 10 MS/s) read
circularly from a seeded offset, with an integer gain of 1 or 3/4 (floor) on the int16 grid. The IQ form puts
the magnitude on one axis (+I, +Q, -I, -Q), the axis advancing every 4096 samples from a seeded start, so that
sqrtf(I*I + Q*Q) == |m| exactly in fp32 and every box sum the decoder forms is an exact multiple of 2^-15.
The same definition is implemented for numpy (CPU oracle input) and torch (GPU-resident bench input).
"""
import json
import lzma
import os

import numpy as np

PHASE_PERIOD = 4096
_MASK = 0xFFFFFFFFFFFFFFFF


def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & _MASK
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
    return z ^ (z >> 31)


def load_template(golden_dir, names=None):
    """int16 concatenation of the fixtures (all of them by default, manifest order)."""
    with open(os.path.join(golden_dir, "manifest.json")) as f:
        manifest = json.load(f)
    names = names or sorted(manifest.keys())
    parts = []
    for name in names:
        with open(os.path.join(golden_dir, "wav", name + ".i16.xz"), "rb") as f:
            parts.append(np.frombuffer(lzma.decompress(f.read()), dtype="<i2"))
    return np.concatenate(parts)


def stream_params(stream, total):
    """(offset into the template, gain flag, starting axis) of one stream."""
    a = splitmix64((0x9E3779B97F4A7C15 * (stream + 1)) & _MASK)
    b = splitmix64(a)
    c = splitmix64(b)
    return a % total, b & 1, c & 3


def magnitude_i16(template, stream, start, length):
    total = template.size
    off, gain, _ = stream_params(stream, total)
    idx = (off + start + np.arange(length, dtype=np.int64)) % total
    k = template[idx].astype(np.int32)
    if gain:
        k = (k * 3) // 4
    return k.astype(np.int16)


def magnitude_f32(template, stream, start, length):
    """What the decoder sees after the IQ->magnitude step: |k| / 32768 (fp32)."""
    k = magnitude_i16(template, stream, start, length).astype(np.float32)
    return np.abs(k / np.float32(32768.0)).astype(np.float32)


def iq_f32(template, stream, start, length):
    k = magnitude_i16(template, stream, start, length).astype(np.float32)
    m = (k / np.float32(32768.0)).astype(np.float32)
    _, _, ph = stream_params(stream, template.size)
    axis = (((start + np.arange(length, dtype=np.int64)) // PHASE_PERIOD) + ph) & 3
    iq = np.zeros((length, 2), np.float32)
    iq[:, 0] = np.where(axis == 0, m, np.where(axis == 2, -m, 0))
    iq[:, 1] = np.where(axis == 1, m, np.where(axis == 3, -m, 0))
    return iq


def fill_iq_torch(out, template_dev, first_stream, start=0, chunk_streams=1024):
    """Fill out[S, T, 2] (float32, on the GPU) with streams first_stream .. first_stream+S-1, samples start..start+T-1."""
    import torch
    S, T, _ = out.shape
    total = template_dev.numel()
    dev = out.device
    t = torch.arange(start, start + T, device=dev, dtype=torch.int64)
    axis_t = t // PHASE_PERIOD
    for s0 in range(0, S, chunk_streams):
        s1 = min(S, s0 + chunk_streams)
        params = [stream_params(first_stream + s, total) for s in range(s0, s1)]
        off = torch.tensor([p[0] for p in params], device=dev, dtype=torch.int64)
        gain = torch.tensor([p[1] for p in params], device=dev, dtype=torch.int32)
        ph = torch.tensor([p[2] for p in params], device=dev, dtype=torch.int64)
        idx = (off[:, None] + t[None, :]) % total
        k = template_dev[idx].to(torch.int32)
        k = torch.where(gain[:, None] != 0, torch.div(k * 3, 4, rounding_mode="floor"), k)
        m = k.to(torch.float32) / 32768.0
        axis = (axis_t[None, :] + ph[:, None]) & 3
        zero = torch.zeros_like(m)
        out[s0:s1, :, 0] = torch.where(axis == 0, m, torch.where(axis == 2, -m, zero))
        out[s0:s1, :, 1] = torch.where(axis == 1, m, torch.where(axis == 3, -m, zero))
        del idx, k, m, axis, zero
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Sparse traffic (set S1q): what a monitoring receiver sees most of the time. A stream repeats one exchange of the
# fixtures (a 65536-sample stretch of a capture that begins and ends in quiet carrier) every 2^19 samples (52 ms), the
# rest is unmodulated carrier at the level the stretch begins with plus +-2 counts of deterministic noise. Everything
# stays on the int16 grid and the IQ form is the axis-aligned one of S1.
# ---------------------------------------------------------------------------------------------------------------------
SPARSE_SEG = 65536
SPARSE_PERIOD = 1 << 19


def sparse_segments(template):
    """(offsets, carrier levels in counts) of the stretches of the template usable as exchanges: quiet (spread < 6 %) first
    and last 1024 samples at the same level (3 %), something modulated in between, level >= 1500 counts."""
    a = np.abs(template.astype(np.int32))
    offs, levels = [], []
    for off in range(0, a.size - SPARSE_SEG, 4096):
        h = a[off:off + 1024]
        e = a[off + SPARSE_SEG - 1024:off + SPARSE_SEG]
        mh, me = float(np.median(h)), float(np.median(e))
        if mh < 1500 or (h.max() - h.min()) > 0.06 * mh or (e.max() - e.min()) > 0.06 * me or abs(mh - me) > 0.03 * mh:
            continue
        if a[off:off + SPARSE_SEG].min() > 0.8 * mh:
            continue
        offs.append(off)
        levels.append(int(mh))
    return np.asarray(offs, np.int64), np.asarray(levels, np.int32)


def sparse_params(stream, nseg):
    """(segment index, position of the exchange inside the period, starting axis) of one stream"""
    a = splitmix64((0xD1B54A32D192ED03 * (stream + 1)) & _MASK)
    b = splitmix64(a)
    c = splitmix64(b)
    return a % nseg, (b % (SPARSE_PERIOD - SPARSE_SEG)) // 64 * 64, c & 3


def _sparse_noise_np(t, stream):
    h = (t.astype(np.uint64) * np.uint64(2654435761) + np.uint64(stream * 40503 + 12345)) & np.uint64(0xFFFFFFFF)
    h = ((h ^ (h >> np.uint64(15))) * np.uint64(2246822519)) & np.uint64(0xFFFFFFFF)
    h = ((h ^ (h >> np.uint64(13))) * np.uint64(3266489917)) & np.uint64(0xFFFFFFFF)
    h = h ^ (h >> np.uint64(16))
    return (h % np.uint64(5)).astype(np.int32) - 2


def sparse_magnitude_f32(template, segs, stream, start, length):
    offs, levels = segs
    g, base, _ = sparse_params(stream, offs.size)
    t = start + np.arange(length, dtype=np.int64)
    u = t % SPARSE_PERIOD
    inside = (u >= base) & (u < base + SPARSE_SEG)
    idx = np.where(inside, offs[g] + u - base, 0)
    k = np.where(inside, np.abs(template[idx].astype(np.int32)), levels[g] + _sparse_noise_np(t, stream))
    return (k.astype(np.float32) / np.float32(32768.0)).astype(np.float32)


def fill_sparse_iq_torch(out, template_dev, segs, first_stream, start=0, chunk_streams=256):
    """out[S, T, 2] (float32, on the GPU) <- sparse streams first_stream .. first_stream+S-1, samples start..start+T-1"""
    import torch
    offs, levels = segs
    S, T, _ = out.shape
    dev = out.device
    t = torch.arange(start, start + T, device=dev, dtype=torch.int64)
    u = t % SPARSE_PERIOD
    axis_t = t // PHASE_PERIOD
    for s0 in range(0, S, chunk_streams):
        s1 = min(S, s0 + chunk_streams)
        params = [sparse_params(first_stream + s, offs.size) for s in range(s0, s1)]
        off = torch.tensor([int(offs[p[0]]) for p in params], device=dev, dtype=torch.int64)[:, None]
        level = torch.tensor([int(levels[p[0]]) for p in params], device=dev, dtype=torch.int64)[:, None]
        base = torch.tensor([p[1] for p in params], device=dev, dtype=torch.int64)[:, None]
        ph = torch.tensor([p[2] for p in params], device=dev, dtype=torch.int64)[:, None]
        sid = torch.tensor([first_stream + s for s in range(s0, s1)], device=dev, dtype=torch.int64)[:, None]
        inside = (u[None, :] >= base) & (u[None, :] < base + SPARSE_SEG)
        idx = torch.where(inside, off + u[None, :] - base, torch.zeros_like(off))
        h = (t[None, :] * 2654435761 + sid * 40503 + 12345) & 0xFFFFFFFF
        h = ((h ^ (h >> 15)) * 2246822519) & 0xFFFFFFFF
        h = ((h ^ (h >> 13)) * 3266489917) & 0xFFFFFFFF
        h = h ^ (h >> 16)
        k = torch.where(inside, template_dev[idx].to(torch.int64).abs(), level + (h % 5) - 2)
        m = k.to(torch.float32) / 32768.0
        axis = (axis_t[None, :] + ph) & 3
        zero = torch.zeros_like(m)
        out[s0:s1, :, 0] = torch.where(axis == 0, m, torch.where(axis == 2, -m, zero))
        out[s0:s1, :, 1] = torch.where(axis == 1, m, torch.where(axis == 3, -m, zero))
        del inside, idx, h, k, m, axis, zero
    return out
