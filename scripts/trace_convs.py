"""Per-layer convolution timings of one training step from a rocprofv3 kernel trace (cfg3 geometry by default).

    python scripts/trace_convs.py <dir with *_kernel_trace.csv> [S f N H W Ci]
Maps the conv-class launches of the last step onto the layer sequence of the plan (forward in network order,
backward in reverse with [data gradient], weight gradient per layer) and prints duration and algorithmic TFLOP/s.
"""
import csv
import glob
import sys

d = sys.argv[1]
S, f, N, H, W, Ci = (int(v) for v in sys.argv[2:8]) if len(sys.argv) >= 8 else (2, 30, 32, 256, 256, 2)
layers = []  # (name, Cin, Cout, H, W)
for s in range(S):  # the plan runs each private encoder chain to its end before the next one
    layers += [(f"enc_in{s}.c1", Ci, f, H, W), (f"enc_in{s}.c2", f, f, H, W)]
    layers += [(f"down1_{s}.c1", f, 2 * f, H // 2, W // 2), (f"down1_{s}.c2", 2 * f, 2 * f, H // 2, W // 2)]
fs = f * S
layers += [("down2.c1", 2 * fs, 4 * fs, H // 4, W // 4), ("down2.c2", 4 * fs, 4 * fs, H // 4, W // 4),
           ("down3.c1", 4 * fs, 8 * fs, H // 8, W // 8), ("down3.c2", 8 * fs, 8 * fs, H // 8, W // 8),
           ("down4.c1", 8 * fs, 8 * fs, H // 16, W // 16), ("down4.c2", 8 * fs, 8 * fs, H // 16, W // 16),
           ("up1.c1", 16 * fs, 8 * fs, H // 8, W // 8), ("up1.c2", 8 * fs, 4 * fs, H // 8, W // 8),
           ("up2.c1", 8 * fs, 4 * fs, H // 4, W // 4), ("up2.c2", 4 * fs, 2 * fs, H // 4, W // 4),
           ("up3.c1", 4 * fs, 2 * fs, H // 2, W // 2), ("up3.c2", 2 * fs, fs, H // 2, W // 2)]
for s in range(S):
    cin = f * (S + 1)
    layers += [(f"up4_{s}.c1", cin, cin // 2, H, W), (f"up4_{s}.c2", cin // 2, f, H, W)]

fn = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(fn)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
step = rows[adam[-2]:adam[-1]]


def is_conv(k):
    return ("conv3x3_" in k or "conv_image_" in k or "wgrad_split" in k or "wgrad_kernel" in k or "wgrad_mfma" in k or "wgrad_thin_kernel" in k) and "reduce" not in k \
        and "group_sum" not in k and "pack" not in k


convs = [r for r in step if is_conv(r["Kernel_Name"])]
seq = [(l, "fwd") for l in layers]
order = []
for s in reversed(range(S)):
    order += [f"up4_{s}"]
order += ["up3", "up2", "up1", "down4", "down3", "down2"] + [f"down1_{s}" for s in reversed(range(S))] + \
         [f"enc_in{s}" for s in reversed(range(S))]
byname = {l[0]: l for l in layers}
for blk in order:
    for c in ("c2", "c1"):
        l = byname[f"{blk}.{c}"]
        if not (blk.startswith("enc_in") and c == "c1"):
            seq.append((l, "dgrad"))
        seq.append((l, "wgrad"))
print(f"{len(convs)} conv-class launches in the step, {len(seq)} expected")
tot = {}
for (l, kind), r in zip(seq, convs):
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    fl = 18.0 * l[1] * l[2] * N * l[3] * l[4]
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void mimo::", "")
    print(f"{l[0]:12s} {kind:5s} {l[1]:4d}->{l[2]:4d} @{l[3]:3d}  {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s   {k}")
    t = tot.setdefault(kind, [0.0, 0.0])
    t[0] += us
    t[1] += fl
for k, (us, fl) in tot.items():
    print(f"{k}: {us / 1e3:.2f} ms, {fl / us / 1e6:.1f} TF/s")
