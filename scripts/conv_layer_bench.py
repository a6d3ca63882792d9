"""Per-layer timing of the split16 3x3 convolutions outside the plan (kernel A/B work).

    rocprofv3 --kernel-trace -d <dir> -o t --output-format csv -- python3 scripts/conv_layer_bench.py run [reps]
    python3 scripts/conv_layer_bench.py report <dir>

`run` calls mimo_op_conv3x3_forward / _dgrad once per repetition for every layer shape of cfg3 (batch 32); `report`
reads the kernel trace and prints, per shape and direction, the fastest conv-kernel duration and its algorithmic
TFLOP/s.  Environment (MIMO_CONV_WIDE, MIMO_HIP_LIB, ...) selects the kernels; the label file is written next to the
trace through MIMO_LAYER_BENCH_LABELS.
"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [  # (N, H, W, Cin, Cout): the distinct 3x3 layers of cfg3 (S = 2, fbc = 30) at batch 32
    (32, 256, 256, 30, 30), (32, 128, 128, 30, 60), (32, 128, 128, 60, 60), (32, 64, 64, 120, 240),
    (32, 64, 64, 240, 240), (32, 32, 32, 240, 480), (32, 32, 32, 480, 480), (32, 16, 16, 480, 480),
    (32, 32, 32, 960, 480), (32, 32, 32, 480, 240), (32, 64, 64, 480, 240), (32, 64, 64, 240, 120),
    (32, 128, 128, 240, 120), (32, 128, 128, 120, 60), (32, 256, 256, 90, 45), (32, 256, 256, 45, 30),
]


def pad8(c):
    return (c + 7) // 8 * 8


def run(reps):
    import torch
    from mimo_unet_amd import _lib as L
    lib = L.load()
    only = os.environ.get("MIMO_LAYER_BENCH_ONLY")
    shapes = [s for i, s in enumerate(SHAPES) if only is None or str(i) in only.split(",")]
    if os.environ.get("MIMO_LAYER_BENCH_SHAPES"):  # "N,H,W,Cin,Cout;..." instead of the cfg3 table
        shapes = [tuple(int(v) for v in t.split(",")) for t in os.environ["MIMO_LAYER_BENCH_SHAPES"].split(";")]
    labels = []
    st = L.current_stream()
    prec = int(os.environ.get("MIMO_LAYER_BENCH_PREC", "1"))  # mimo_precision: 1 split16, 3 bf16-mixed, 4 16-mixed
    for (N, H, W, Ci, Co) in shapes:
        cip, cop = pad8(Ci), pad8(Co)
        g = torch.Generator(device="cuda").manual_seed(1)
        x = torch.randn(N, H, W, cip, device="cuda", generator=g)
        x[..., Ci:] = 0
        w = torch.randn(Co, Ci, 3, 3, device="cuda", generator=g) / (3.0 * Ci ** 0.5)
        b = torch.randn(Co, device="cuda", generator=g)
        z = torch.empty(N, H, W, cop, device="cuda")
        dz = torch.randn(N, H, W, cop, device="cuda", generator=g)
        dz[..., Co:] = 0
        dx = torch.empty(N, H, W, cip, device="cuda")
        stats = torch.zeros(2, Co, dtype=torch.float64, device="cuda")
        dw, db = torch.empty(Co, Ci, 3, 3, device="cuda"), torch.empty(Co, device="cuda")
        for _ in range(reps):
            L.check(lib.mimo_op_conv3x3_forward(x.data_ptr(), w.data_ptr(), b.data_ptr(), z.data_ptr(), stats.data_ptr(),
                                                N, H, W, Ci, cip, Co, cop, prec, st), "fwd")
            labels.append(f"fwd {Ci}->{Co}@{H} {N}")
            L.check(lib.mimo_op_conv3x3_dgrad(dz.data_ptr(), w.data_ptr(), dx.data_ptr(), N, H, W, Ci, cip, Co, cop, prec, st),
                    "dgrad")
            labels.append(f"dgrad {Ci}->{Co}@{H} {N}")
            if os.environ.get("MIMO_LAYER_BENCH_WGRAD", "1") != "0":
                L.check(lib.mimo_op_conv3x3_wgrad(x.data_ptr(), dz.data_ptr(), dw.data_ptr(), db.data_ptr(), N, H, W, Ci, cip,
                                                  Co, cop, prec, st), "wgrad")
                labels.append(f"wgrad {Ci}->{Co}@{H} {N}")
        torch.cuda.synchronize()
        del x, w, b, z, dz, dx
    with open(os.environ.get("MIMO_LAYER_BENCH_LABELS", "/tmp/layer_bench_labels.txt"), "w") as fh:
        fh.write("\n".join(labels) + "\n")


def report(d):
    fn = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
    rows = list(csv.DictReader(open(fn)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    convs = [r for r in rows if ("conv3x3_" in r["Kernel_Name"] or "wgrad_split" in r["Kernel_Name"] or "wgrad_mfma" in r["Kernel_Name"]
                                 or "wgrad_thin_kernel" in r["Kernel_Name"]) and "pack" not in r["Kernel_Name"]]
    labels = open(os.path.join(d, "labels.txt")).read().split("\n")[:-1]
    assert len(convs) == len(labels), (len(convs), len(labels))
    # MIMO_LAYER_BENCH_STAT: min (default: burst speed of a cool chip) | median | tail (mean of the last quarter of
    # the repetitions: the sustained regime of a long back-to-back run)
    stat = os.environ.get("MIMO_LAYER_BENCH_STAT", "min")
    every = {}
    for lab, r in zip(labels, convs):
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void mimo::", "")
        every.setdefault(lab, []).append((us, k))
    best = {}
    for lab, v in every.items():
        t = [u for u, _ in v]
        if stat == "median":
            us = sorted(t)[len(t) // 2]
        elif stat == "tail":
            q = t[-max(1, len(t) // 4):]
            us = sum(q) / len(q)
        else:
            us = min(t)
        best[lab] = (us, v[0][1])
    tot = {}
    for lab, (us, k) in best.items():
        kind, shp, n = lab.split()
        ci, rest = shp.split("->")
        co, h = rest.split("@")
        fl = 18.0 * int(ci) * int(co) * int(n) * int(h) * int(h)
        print(f"{kind:5s} {shp:14s} {us:8.1f} us {fl / us / 1e6:7.1f} TF/s  {k}")
        t = tot.setdefault(kind, [0.0, 0.0])
        t[0] += us
        t[1] += fl
    for k, (us, fl) in tot.items():
        print(f"{k}: {us / 1e3:.3f} ms, {fl / us / 1e6:.1f} TF/s (one launch per distinct shape)")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 2)
    else:
        report(sys.argv[2])
