#!/usr/bin/env python3
"""Static instruction census of the device code per source function.

usage: asm_lines.py demod_g.s [kernel-symbol-substring]
The input is `hipcc -S --offload-device-only -gline-tables-only` output. Every instruction is attributed to the
last `.loc file line`; lines are mapped to the enclosing function of the header they come from (functions
are recognised by `NFC_DEV`/`__global__`/`template` definitions at column 0). Prints VALU / SALU / VMEM / LDS /
v_mov counts per function — a static census (what the wave executes when every path is live)."""
import re
import sys
import collections
import os

asm = sys.argv[1]
only = sys.argv[2] if len(sys.argv) > 2 else None   # substring of the kernel symbol to restrict the census to
inside = only is None
csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "nfc-laboratory_amd", "csrc")

files = {}
func_of = {}


def load_functions(path):
    table = []
    cur = None
    try:
        lines = open(path).read().split("\n")
    except OSError:
        return []
    for i, l in enumerate(lines, 1):
        m = re.match(r"^(?:template\s*<[^>]*>\s*)?(?:NFC_DEV|__global__|static|inline)[^;(]*?\b(\w+)\s*\(", l)
        if m and not l.rstrip().endswith(";"):
            cur = m.group(1)
        table.append(cur)
    return table


counts = collections.defaultdict(lambda: collections.Counter())
cur = ("?", 0)
for l in open(asm):
    l = l.strip()
    if only is not None:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            inside = only in m.group(1)
    m = re.match(r"\.file\s+(\d+)\s+\"([^\"]*)\"\s+\"([^\"]*)\"", l)
    if m:
        files[int(m.group(1))] = m.group(3)
        continue
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
        continue
    if not l or l.startswith(".") or l.startswith(";") or l.endswith(":"):
        continue
    if not inside:
        continue
    op = l.split()[0]
    if not re.match(r"^[vsdgb]_|^global_|^buffer_|^ds_|^flat_|^scratch_", op):
        continue
    f, line = cur
    base = os.path.basename(f)
    if base not in func_of:
        func_of[base] = load_functions(os.path.join(csrc, base))
    tab = func_of[base]
    fn = tab[line - 1] if 0 < line <= len(tab) and tab[line - 1] else base
    kind = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem"
    counts[fn][kind] += 1
    if op.startswith("v_mov") or op.startswith("v_accvgpr"):
        counts[fn]["vmov"] += 1
    if op.startswith("v_cndmask"):
        counts[fn]["cndmask"] += 1

tot = collections.Counter()
rows = sorted(counts.items(), key=lambda kv: -kv[1]["valu"])
print("%-34s %6s %6s %6s %6s %6s %6s" % ("function", "valu", "vmov", "cndmsk", "salu", "vmem", "lds"))
for fn, c in rows:
    print("%-34s %6d %6d %6d %6d %6d %6d" % (fn, c["valu"], c["vmov"], c["cndmask"], c["salu"], c["vmem"], c["lds"]))
    tot.update(c)
print("%-34s %6d %6d %6d %6d %6d %6d" % ("TOTAL", tot["valu"], tot["vmov"], tot["cndmask"], tot["salu"], tot["vmem"], tot["lds"]))
