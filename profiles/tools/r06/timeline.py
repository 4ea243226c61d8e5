#!/usr/bin/env python3
"""Round 6: the device's timeline over the last step of the headline from a rocprofv3 kernel trace (csv): per kernel its launches and
busy time, and the time during which no kernel of ours is running (the host's round trips). usage: timeline.py <kernel_trace.csv>"""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0]
    if not name.startswith("nfc_"):
        continue
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
# steps begin with the large nfc_scan_kernel launch that follows an nfc_finish_kernel (or the first)
starts = [i for i, (s, e, n) in enumerate(rows) if n == "nfc_scan_kernel" and (i == 0 or rows[i - 1][2] in ("nfc_finish_kernel", "nfc_read_kernel", "nfc_init_kernel"))]
first = starts[-1]
step = rows[first:]
# cut at the finish kernel
for i, (s, e, n) in enumerate(step):
    if n == "nfc_finish_kernel":
        step = step[:i + 1]
        break
t0, t1 = step[0][0], max(e for s, e, n in step)
busy = []
for s, e, n in sorted(step):
    if busy and s <= busy[-1][1]:
        busy[-1][1] = max(busy[-1][1], e)
    else:
        busy.append([s, e])
covered = sum(e - s for s, e in busy)
print("step: %.2f ms from the first kernel's start to the finish kernel's end, %d launches; a kernel running %.2f ms, none %.2f ms in %d gaps" %
      ((t1 - t0) / 1e6, len(step), covered / 1e6, (t1 - t0 - covered) / 1e6, len(busy) - 1))
gaps = sorted(((busy[i + 1][0] - busy[i][1]) / 1e3 for i in range(len(busy) - 1)), reverse=True)
print("largest gaps (us):", [round(g) for g in gaps[:12]], "median", round(gaps[len(gaps) // 2]) if gaps else None)
per = collections.defaultdict(lambda: [0, 0])
for s, e, n in step:
    per[n][0] += 1
    per[n][1] += e - s
for n, (c, d) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print("  %-28s %3d launches %9.2f ms" % (n, c, d / 1e6))
print("order:")
for s, e, n in step:
    print("  %9.3f .. %9.3f ms  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, n))
