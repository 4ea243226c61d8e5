#!/usr/bin/env python3
"""Generate the committed parity fixtures from the reference's bundled captures.

Source: /root/reference/wav/test_*.wav (mono PCM int16 @ 10 MS/s, magnitude samples) and the
reference's own golden frame lists /root/reference/wav/test_*.json (README.md:512-540;
compared by src/nfc-test/test-sdr/src/main/cpp/main.cpp:182-218 through RawFrame::operator==).

Output (committed, travels to the GPU box where /root/reference does not exist):
  tests/golden/wav/<name>.i16.xz   raw little-endian int16 samples, xz-compressed
  tests/golden/wav/<name>.json     the reference golden JSON, byte-for-byte
  tests/golden/manifest.json       {name: {samples, sample_rate, frames, sha256 of raw int16}}

The decoder input is sample/32768.0f exactly as hw::RecordDevice::readScaledSamples<short> does
(src/nfc-lib/lib-hw/hw-dev/src/main/cpp/hw/RecordDevice.cpp:280-311).
"""
import hashlib
import json
import lzma
import os
import shutil
import sys
import wave

REF = os.environ.get("NFC_REFERENCE_ROOT", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    src = os.path.join(REF, "wav")
    dst = os.path.join(HERE, "wav")
    os.makedirs(dst, exist_ok=True)
    manifest = {}
    for fn in sorted(os.listdir(src)):
        if not fn.endswith(".wav"):
            continue
        name = fn[:-4]
        with wave.open(os.path.join(src, fn)) as w:
            assert w.getnchannels() == 1 and w.getsampwidth() == 2, fn
            rate = w.getframerate()
            raw = w.readframes(w.getnframes())
        with open(os.path.join(dst, name + ".i16.xz"), "wb") as f:
            f.write(lzma.compress(raw, preset=9 | lzma.PRESET_EXTREME))
        shutil.copyfile(os.path.join(src, name + ".json"), os.path.join(dst, name + ".json"))
        os.chmod(os.path.join(dst, name + ".json"), 0o644)
        with open(os.path.join(src, name + ".json")) as f:
            frames = len(json.load(f)["frames"])
        manifest[name] = {
            "samples": len(raw) // 2,
            "sample_rate": rate,
            "frames": frames,
            "sha256": hashlib.sha256(raw).hexdigest(),
        }
        print(name, manifest[name]["samples"], frames)
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    sys.exit(main())
