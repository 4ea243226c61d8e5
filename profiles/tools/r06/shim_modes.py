#!/usr/bin/env python3
"""Round 6: the reference's RadioDecoderTask on the GPU shim (oracle/_ref/task-gpu) in its default mode - every 65536-sample
buffer submitted and collected inside nextFrames() - under a few settings of the library's knobs, on a dense and a sparse WAV;
with --stages the stage log of the first submissions. usage: shim_modes.py [--stages] [ENV=VALUE[,ENV=VALUE]] ..."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "nfc-laboratory_amd"))
import numpy as np
import nfc_testlib as TL, synth

stages = "--stages" in sys.argv
settings = [a for a in sys.argv[1:] if a != "--stages"] or [""]
template = synth.load_template(os.path.join(ROOT, "tests", "golden"))
task = os.path.join(ROOT, "oracle", "_ref", "task-ref")
task_gpu = os.path.join(ROOT, "oracle", "_ref", "task-gpu")
with tempfile.TemporaryDirectory() as tmp:
    wavs = {}
    dense_n = (1 << 22) + 12345
    dense = synth.magnitude_f32(template, 0, 0, dense_n)
    wavs["dense"] = (os.path.join(tmp, "dense.wav"), dense_n)
    TL.write_wav(wavs["dense"][0], np.clip(np.rint(dense * 32768.0), -32768, 32767).astype(np.int16))
    sparse = synth.sparse_magnitude_f32(template, synth.sparse_segments(template), 0, 0, (1 << 23) + 12345)
    wavs["sparse"] = (os.path.join(tmp, "sparse.wav"), sparse.size)
    TL.write_wav(wavs["sparse"][0], np.clip(np.rint(sparse * 32768.0), -32768, 32767).astype(np.int16))
    for label, (path, n) in wavs.items():
        o = subprocess.run([task, path], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=600).stdout
        dn = [l.split() for l in o.splitlines() if l.startswith("DONE")]
        print("%-6s reference task: %s frames, %.2f MS/s" % (label, dn[0][2], n / float(dn[0][3]) / 1e6), flush=True)
        for setting in settings:
            env = {k: v for k, v in os.environ.items() if k != "NFCGPU_SHIM_BLOCK"}
            for kv in [s for s in setting.split(",") if s]:
                k, v = kv.split("=", 1)
                env[k] = v
            if stages:
                env["NFCGPU_WINDOW_DEBUG"] = "1"
            r = subprocess.run([task_gpu, path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env)
            dn = [l.split() for l in r.stdout.splitlines() if l.startswith("DONE")]
            print("%-6s gpu task [%s]: %s frames, %.2f MS/s" % (label, setting or "default", dn[0][2] if dn else "?", n / float(dn[0][3]) / 1e6 if dn else 0.0), flush=True)
            if stages:
                lines = [l for l in r.stderr.splitlines() if l.startswith("[nfcgpu] windowed")]
                print("\n".join(lines[40:75]))
