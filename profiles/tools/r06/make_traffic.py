#!/usr/bin/env python3
"""HBM bytes of the roofline kernels from the rocprofv3 counter passes of profiles/tools/r06/round_profile.sh: nfc_wave_kernel
(all its launches of one step of the headline: what bench.py's roofline.kernel_ms_avg times) and nfc_scan_kernel (its launch
over the whole submission).
usage: make_traffic.py <gpurun_out/tag>   ->  <dir>/pmc_hbm_traffic.json and profiles/traffic.json

FETCH_SIZE / WRITE_SIZE are in KiB. The calibration kernels (profiles/tools/calib_traffic.hip) read and write 8 GiB with
known byte counts; the ratio known / reported is applied to the kernels' counters (on gfx950 FETCH_SIZE reports half the
bytes of dword-per-lane reads, WRITE_SIZE is exact: MI355X_MICROARCH.md, HBM section)."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def counters(d):
    """[(kernel, grid, counter, value)] in dispatch order"""
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r.get("Dispatch_Id", 0)), r["Kernel_Name"].split("(")[0], int(r.get("Grid_Size", 0) or 0), r["Counter_Name"], float(r["Counter_Value"])))
    rows.sort()
    return rows


def main():
    out = sys.argv[1]
    CAL_BYTES = float(1 << 33)
    calib = {}
    for c, kern in (("FETCH_SIZE", "read_rows"), ("WRITE_SIZE", "write_rows")):
        vals = [v for (_, k, _, n, v) in counters(os.path.join(out, "calib_" + c)) if n == c and kern in k]
        calib[c] = CAL_BYTES / (vals[-1] * 1024.0) if vals else None

    STEPS = 2  # --steps 1 --warmup 1: two submissions of the same shape
    result = {"round": 6, "command": "profiles/tools/r06/round_profile.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, WRITE_SIZE) -- "
                                      "python bench.py --no-cpu --no-points --steps 1 --warmup 1; no other trace domains",
              "calibration": {"tool": "profiles/tools/calib_traffic.hip (8 GiB read, 8 GiB written)", "fetch_size_factor": calib["FETCH_SIZE"],
                              "write_size_factor": calib["WRITE_SIZE"]}, "kernels": {}}

    shapes = {"nfc_wave_kernel": (4096, 1 << 20), "nfc_scan_kernel": (4096, 1 << 20)}
    per = collections.defaultdict(dict)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        rows = counters(os.path.join(out, "pmc_" + c))
        for kern in shapes:
            mine = [(g, v) for (_, k, g, n, v) in rows if n == c and k == kern]
            if not mine:
                continue
            biggest = max(g for g, _ in mine)
            if kern == "nfc_wave_kernel":
                # every launch of the two steps (carry lanes, the windows of each pass, final lanes): per step
                per[kern][c] = {"dispatches": len(mine), "steps": STEPS, "last_KiB": sum(v for _, v in mine) / STEPS, "is": "sum over all launches / steps"}
            else:
                full = [v for g, v in mine if g == biggest]   # launches over the whole submission (not the re-walks of single chunks)
                per[kern][c] = {"dispatches": len(full), "grid": biggest, "last_KiB": full[-1], "mean_KiB": sum(full) / len(full)}

    try:
        git = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout.strip() or None
    except Exception:
        git = None
    if not git:
        # (the GPU box has no repository: the commit the libraries were built from, left beside them by __graft_entry__.build())
        try:
            git = open(os.path.join(ROOT, "nfc-laboratory_amd", "build", "git_head.txt")).read().strip() or None
        except Exception:
            git = None
    sys.path.insert(0, ROOT)
    import bench
    digest = bench.sources_digest()

    traffic = {}
    for kern, (streams, samples) in shapes.items():
        if "FETCH_SIZE" not in per[kern] or "WRITE_SIZE" not in per[kern] or not calib["FETCH_SIZE"] or not calib["WRITE_SIZE"]:
            continue
        rd = per[kern]["FETCH_SIZE"]["last_KiB"] * 1024.0 * calib["FETCH_SIZE"]
        wr = per[kern]["WRITE_SIZE"]["last_KiB"] * 1024.0 * calib["WRITE_SIZE"]
        alg = 8.0 * streams * samples
        result["kernels"][kern] = {"streams": streams, "samples": samples, "raw": per[kern], "hbm_read_bytes_per_launch": rd,
                                   "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr, "algorithmic_bytes_per_launch": alg,
                                   "ratio_traffic_to_algorithmic": (rd + wr) / alg}
        traffic[kern] = {"streams": streams, "samples": samples, "hbm_bytes_per_launch": rd + wr, "sources_sha1": digest, "git": git,
                         "from": "profiles/r06/pmc_hbm_traffic.json"}

    # ---- the whole step: every kernel the two submissions launched, per step (round 6) ----
    by_kernel = collections.defaultdict(lambda: {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "dispatches": 0})
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for (_, k, g, n, v) in counters(os.path.join(out, "pmc_" + c)):
            # (nfc_read_kernel is bench.py's bandwidth probe, nfc_init_kernel the opening of the streams: neither is part of a step)
            if n != c or not k.startswith("nfc_") or k in ("nfc_read_kernel", "nfc_init_kernel"):
                continue
            by_kernel[k][c] += v
            if c == "FETCH_SIZE":
                by_kernel[k]["dispatches"] += 1
    if by_kernel and calib["FETCH_SIZE"] and calib["WRITE_SIZE"]:
        step = {}
        for k, v in by_kernel.items():
            step[k] = {"launches_per_step": v["dispatches"] / STEPS, "hbm_read_bytes_per_step": v["FETCH_SIZE"] * 1024.0 * calib["FETCH_SIZE"] / STEPS,
                       "hbm_write_bytes_per_step": v["WRITE_SIZE"] * 1024.0 * calib["WRITE_SIZE"] / STEPS}
        rd = sum(x["hbm_read_bytes_per_step"] for x in step.values())
        wr = sum(x["hbm_write_bytes_per_step"] for x in step.values())
        alg = 8.0 * 4096 * (1 << 20)
        result["step"] = {"streams": 4096, "samples": 1 << 20, "kernels": step, "hbm_read_bytes_per_step": rd, "hbm_write_bytes_per_step": wr,
                          "hbm_bytes_per_step": rd + wr, "algorithmic_bytes_per_step": alg, "ratio_traffic_to_algorithmic": (rd + wr) / alg}
        traffic["step"] = {"streams": 4096, "samples": 1 << 20, "hbm_bytes_per_launch": rd + wr, "sources_sha1": digest, "git": git,
                           "from": "profiles/r06/pmc_hbm_traffic.json (every kernel of a step of the headline)"}

    # ---- the sequential kernel (the `saturating` point: 131072 streams x 8192 samples per launch) ----
    seq = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = [(g, v) for (_, k, g, n, v) in counters(os.path.join(out, "seq_" + c)) if n == c and k == "nfc_demod_fixed_kernel"]
        if vals:
            biggest = max(g for g, _ in vals)
            full = [v for g, v in vals if g == biggest]
            seq[c] = {"dispatches": len(full), "grid": biggest, "last_KiB": full[-1]}
    if "FETCH_SIZE" in seq and "WRITE_SIZE" in seq and calib["FETCH_SIZE"] and calib["WRITE_SIZE"]:
        rd = seq["FETCH_SIZE"]["last_KiB"] * 1024.0 * calib["FETCH_SIZE"]
        wr = seq["WRITE_SIZE"]["last_KiB"] * 1024.0 * calib["WRITE_SIZE"]
        alg = 8.0 * 131072 * 8192
        result["kernels"]["nfc_demod_fixed_kernel"] = {"streams": 131072, "samples": 8192, "raw": seq, "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
                                                       "hbm_bytes_per_launch": rd + wr, "algorithmic_bytes_per_launch": alg, "ratio_traffic_to_algorithmic": (rd + wr) / alg}
        traffic["nfc_demod_fixed_kernel"] = {"streams": 131072, "samples": 8192, "hbm_bytes_per_launch": rd + wr, "sources_sha1": digest, "git": git,
                                             "from": "profiles/r06/pmc_hbm_traffic.json"}

    json.dump(result, open(os.path.join(out, "pmc_hbm_traffic.json"), "w"), indent=1)
    json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
    print(json.dumps(result, indent=1))


if __name__ == "__main__":
    main()
