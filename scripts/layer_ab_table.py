#!/usr/bin/env python3
"""Join the per-setting tables of scripts/layer_ab.sh (gpurun_out/<tag>/layers_<i>.txt) into one side-by-side table:
    python3 scripts/layer_ab_table.py gpurun_out/<tag> [name_0 name_1 ...]
one row per (pass, layer shape), one '<us> (<TF/s>)' column per setting, the kernel instances last."""
import glob
import os
import re
import sys


def parse(path):
    rows, setting = [], ""
    for line in open(path):
        if line.startswith("=="):
            setting = line[2:].strip()
            continue
        m = re.match(r"(\w+)\s+(\S+)\s+([\d.]+) us\s+([\d.]+) TF/s\s+(.*)", line)
        if m:
            rows.append((m.group(1), m.group(2), float(m.group(3)), float(m.group(4)), m.group(5).strip()))
    return setting, rows


def main():
    d = sys.argv[1]
    files = sorted(glob.glob(os.path.join(d, "layers_*.txt")), key=lambda p: int(re.findall(r"(\d+)\.txt", p)[0]))
    tabs = [parse(f) for f in files]
    names = sys.argv[2:] or [t[0] or f"#{i}" for i, t in enumerate(tabs)]
    for i, (s, _) in enumerate(tabs):
        print(f"# column {i}: {names[i]}   [{s}]")
    n = min(len(t[1]) for t in tabs)
    print("pass  layer".ljust(26) + "".join(f"{nm[:20]:>22}" for nm in names) + "   kernel instances")
    tot = {}
    for r in range(n):
        kind, shape = tabs[0][1][r][0], tabs[0][1][r][1]
        cells = []
        for i, (_, rows) in enumerate(tabs):
            cells.append(f"{rows[r][2]:9.1f} us ({rows[r][3]:6.1f})")
            tot.setdefault((kind, i), [0.0, 0.0])
            tot[(kind, i)][0] += rows[r][2]
            tot[(kind, i)][1] += rows[r][2] * rows[r][3]
        kern = " | ".join(re.sub(r"^conv3x3_|_kernel", "", rows[r][4]) for _, rows in tabs)
        print(f"{kind:5} {shape:19}" + "".join(f"{c:>22}" for c in cells) + "   " + kern)
    for kind in sorted({k for k, _ in tot}):
        cells = [f"{tot[(kind, i)][0] / 1e3:8.3f} ms ({tot[(kind, i)][1] / tot[(kind, i)][0]:6.1f})" for i in range(len(tabs))]
        print(f"{kind:5} {'sum, one per shape':19}" + "".join(f"{c:>22}" for c in cells))


if __name__ == "__main__":
    main()
