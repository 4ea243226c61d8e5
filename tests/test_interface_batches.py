"""The C ABI used the way a multi-receiver host would (nfcgpu_submit_batch): dozens of streams in neighbouring
slots, each with its own parameters, ragged batches of magnitude or interleaved IQ samples, and in between configure / reset / flush / close-and-reopen on single
streams. The yardstick is the reference class itself: every stream is one lab::NfcDecoder of oracle/_ref/api-ref, which is
fed the equivalent call sequence (tests/dropin/api_harness.cpp), and the frames of each stream must be the same. Runs on
the emulated host runtime here (tests/test_host_runtime_emulated.py) and on the GPU with -m gpu."""
import os
import subprocess
import sys

import numpy as np
import pytest

import nfc_testlib as T
from test_oracle_goldens import _fuzz_stream

REF = os.path.join(T.ROOT, "oracle", "_ref", "api-ref")
EMU = os.path.join(T.ROOT, "tests", "hostsim", "libnfcgpu_emulated.so")
FS = 10000000
TECH = "ABFV"

DRIVER = r'''
import json, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import numpy as np
import nfclab_amd
ops = json.load(open(sys.argv[3]))
x = np.fromfile(sys.argv[4], dtype=np.float32)
out = {}
def take(gpu, sid, key):
    out.setdefault(key, []).extend([list(f[:8]) + [f[8].hex()] for f in gpu.poll(sid, capacity=int(ops["poll_capacity"]))])
    while True:
        more = gpu.poll(sid, capacity=int(ops["poll_capacity"]))
        if not more:
            break
        out[key].extend([list(f[:8]) + [f[8].hex()] for f in more])
with nfclab_amd.NfcGpu(device=0, max_streams=ops["slots"]) as gpu:
    first = gpu.open(count=ops["slots"])
    alive = {k: k for k in range(ops["slots"])}        # slot -> decoder key
    params = {k: nfclab_amd.default_params() for k in range(ops["slots"])}
    for op in ops["ops"]:
        kind = op[0]
        if kind == "batch":
            ids, parts = [], []
            stride = op[3]
            for slot, pos, n in op[1]:
                ids.append(first + slot)
                m = x[pos:pos + n]
                if stride == 2:
                    # interleaved IQ whose magnitude is exactly the capture: the phase turns by 90 degrees every 7 samples
                    iq = np.zeros((n, 2), np.float32)
                    q = (np.arange(pos, pos + n) // 7) % 4
                    iq[:, 0] = np.where(q == 0, m, np.where(q == 2, -m, 0))
                    iq[:, 1] = np.where(q == 1, m, np.where(q == 3, -m, 0))
                    m = iq.reshape(-1)
                parts.append(np.ascontiguousarray(m, dtype=np.float32))
            gpu.submit_batch(ids, [p.ctypes.data for p in parts], [p.size // stride for p in parts], op[2], stride=stride)
        elif kind == "configure":
            slot, field, tech, a, b = op[1:]
            p = params[slot]
            if field == "enable":
                p.tech_mask = (p.tech_mask | (1 << tech)) if a else (p.tech_mask & ~(1 << tech))
            elif field == "power":
                p.power_level_threshold = a
            elif field == "corr":
                p.corr_threshold[tech] = a
            else:
                p.min_modulation_depth[tech] = a
                p.max_modulation_depth[tech] = b
            gpu.configure(first + slot, p)
        elif kind == "reset":
            gpu.reset(first + op[1])
        elif kind == "flush":
            gpu.flush(first + op[1])
        elif kind == "poll":
            take(gpu, first + op[1], alive[op[1]])
        elif kind == "reopen":
            slot, key = op[1], op[2]
            take(gpu, first + slot, alive[slot])
            gpu.close_stream(first + slot)
            sid = gpu.open()
            assert sid == first + slot, (sid, first + slot)
            alive[slot] = key
            params[slot] = nfclab_amd.default_params()
    for slot, key in alive.items():
        take(gpu, first + slot, key)
json.dump({str(k): v for k, v in out.items()}, open(sys.argv[5], "w"))
'''


def _scenario(seed, total, slots=40, steps=60):
    """(ops for the C ABI driver, script for the reference harness)"""
    rng = np.random.default_rng(seed)
    ops, script = [], []
    key_of = {k: k for k in range(slots)}
    pos = {k: int(rng.integers(0, total // 2)) for k in range(slots)}
    rate = {k: FS for k in range(slots)}
    fed = {k: False for k in range(slots)}
    next_key = slots
    for _ in range(steps):
        # a ragged batch of the streams that currently run at the batch's rate
        batch_rate = int(rng.choice([FS, FS, FS, 5000000]))
        members = []
        for slot in range(slots):
            if rng.random() < 0.7 and pos[slot] < total:
                if rate[slot] != batch_rate and rng.random() < 0.9:
                    continue                           # now and then a stream does change its rate with the batch
                n = int(min(total - pos[slot], rng.choice([0, 1, 4099, 16384, int(rng.integers(1, 40000))])))
                members.append((slot, pos[slot], n))
                script.append("@%d feed %d %d %d" % (key_of[slot], pos[slot], n, batch_rate))
                pos[slot] += n
                rate[slot] = batch_rate
                fed[slot] = True
        if members:
            ops.append(["batch", members, batch_rate, int(rng.choice([1, 1, 2]))])
        for _ in range(int(rng.integers(0, 4))):
            slot = int(rng.integers(slots))
            k = key_of[slot]
            r = rng.random()
            if r < 0.5:
                field = ["enable", "power", "corr", "depth"][int(rng.integers(4))]
                tech = int(rng.integers(4))
                a = float(np.float32(rng.uniform(0.003, 0.05))) if field == "power" else float(np.float32(rng.uniform(0.1, 0.9)))
                b = float(np.float32(min(1.0, a + rng.uniform(0.05, 0.6))))
                if field == "enable":
                    a = int(rng.integers(2))
                    script.append("@%d enable %s %d" % (k, TECH[tech], a))
                elif field == "power":
                    script.append("@%d power %.9g" % (k, a))
                elif field == "corr":
                    script.append("@%d corr %s %.9g" % (k, TECH[tech], a))
                else:
                    script.append("@%d depth %s %.9g %.9g" % (k, TECH[tech], a, b))
                ops.append(["configure", slot, field, tech, a, b])
            elif r < 0.65 and fed[slot]:
                ops.append(["reset", slot])
                script.append("@%d init" % k)
            elif r < 0.8:
                ops.append(["flush", slot])
                script.append("@%d invalid" % k)
            elif r < 0.9:
                ops.append(["poll", slot])
            else:
                ops.append(["reopen", slot, next_key])
                script.append("@%d drop" % k)
                key_of[slot] = next_key
                next_key += 1
                rate[slot] = FS
                fed[slot] = False
    return {"slots": slots, "ops": ops, "poll_capacity": int(rng.choice([3, 64, 4096]))}, "\n".join(script) + "\n"


def _reference_frames(raw, script_path):
    run = subprocess.run([REF, raw, script_path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-2000:]
    script = open(script_path).read().splitlines()
    frames = {}
    for line in run.stdout.splitlines():
        w = line.split()
        if w[0][0] in "FI":
            key = int(script[int(w[0][1:]) - 1].split()[0][1:])
            frames.setdefault(str(key), []).append([int(v) for v in w[1:9]] + [w[12] if len(w) > 12 else ""])
    return frames


def _check(seed, tmp_path, env):
    import json
    x = np.abs(_fuzz_stream(9000 + seed, 400000))   # a magnitude: the IQ batches carry it as |I + jQ|
    raw = str(tmp_path / "x.f32")
    x.tofile(raw)
    ops, script = _scenario(seed, x.size)
    with open(tmp_path / "ops.json", "w") as f:
        json.dump(ops, f)
    with open(tmp_path / "script.txt", "w") as f:
        f.write(script)
    with open(tmp_path / "driver.py", "w") as f:
        f.write(DRIVER)
    run = subprocess.run([sys.executable, str(tmp_path / "driver.py"), os.path.join(T.ROOT, "nfc-laboratory_amd"), os.path.join(T.ROOT, "tests"),
                          str(tmp_path / "ops.json"), raw, str(tmp_path / "got.json")], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert run.returncode == 0, run.stdout[-3000:]
    got = json.load(open(tmp_path / "got.json"))
    want = _reference_frames(raw, str(tmp_path / "script.txt"))
    total = 0
    for key in sorted(set(got) | set(want), key=int):
        assert got.get(key, []) == want.get(key, []), "decoder %s" % key
        total += len(want.get(key, []))
    assert total > 50


needs_harness = pytest.mark.skipif(not os.path.exists(REF), reason="api-ref not built (needs the reference tree at build time)")


@needs_harness
@pytest.mark.parametrize("seed", range(3))
def test_batch_scenarios_on_the_emulated_runtime(built, seed, tmp_path):
    if not os.path.exists(EMU):
        subprocess.check_call(["bash", os.path.join(T.ROOT, "tests", "hostsim", "build_emulated.sh")])
    _check(seed, tmp_path, dict(os.environ, NFCGPU_LIB=EMU, NFCGPU_NO_TORCH="1"))


@needs_harness
@pytest.mark.gpu
def test_batch_scenario_on_the_gpu(built, tmp_path):
    _check(1, tmp_path, dict(os.environ))
