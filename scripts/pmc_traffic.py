"""Derive profiles/<round>/final/pmc_traffic.json (HBM bytes per launch per convolution class, per step in
total) from the FETCH_SIZE / WRITE_SIZE counter summaries written by scripts/collect_profiles.sh.

    python scripts/pmc_traffic.py profiles/r01/final [steps_executed=3]"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_hash  # noqa: E402  (ties the summary to the kernel sources it was collected on)

D = sys.argv[1].rstrip("/") + "/"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3


def load(f):
    return {r["kernel"]: (int(r["dispatches"]), {k: float(v) for k, v in r.items() if k not in ("kernel", "dispatches")})
            for r in csv.DictReader(open(f))}


def cls(k):
    if "conv3x3_ws_kernel" in k or "conv3x3_bf16x3_kernel" in k or "conv3x3_wide_kernel" in k:
        mode = int(k.split("<")[1].split(",")[2 if "bf16x3" in k else 1].strip(" >"))  # MODE template argument
        return "conv3x3_fwd" if mode in (1, 2, 4, 6) else "conv3x3_dgrad"
    if "conv3x3_thin_fwd_kernel" in k:  # the image convolution's plain-FMA kernels (conv_thin.hip)
        return "conv3x3_fwd"
    return "conv3x3_wgrad" if ("wgrad_split" in k or "wgrad_thin_kernel" in k) else None


# Kernels whose global reads are 16 B per lane (global_load_dwordx4 / LDS-DMA dwordx4 / float4 per thread): the access
# pattern the guide's gfx950 correction is calibrated on (FETCH_SIZE tallies such a read at HALF its bytes -> x2).  Every
# other kernel (statistics / reductions over partial rows, packing, torch's own kernels: 4- or 8-byte loads) is
# uncalibrated: counted x1 in `step_bytes` and x2 in `step_bytes_upper`.
WIDE_READERS = ("conv3x3_", "wgrad_split", "wgrad_thin_kernel", "bn_relu", "bnrelu_bwd_reduce", "bn_bwd_apply", "upcat_fwd",
                "up_bwd", "pool_bwd", "head_fwd", "head_bwd_kernel", "loss_fwd", "adam_kernel", "amp_step_kernel", "maxpool_fwd",
                "fold_slice", "split_pairs", "conv_ksplit_reduce", "train_epilogue", "val_epilogue", "elem_mask_mul", "wgrad_group_sum", "wgrad_reduce")


def read_factor(kernel):
    return 2.0 if any(t in kernel for t in WIDE_READERS) else 1.0


F, W = load(D + "pmc_FETCH_SIZE.csv"), load(D + "pmc_WRITE_SIZE.csv")
agg, tot_r, tot_w, narrow_r = {}, 0.0, 0.0, 0.0
for k, (n, c) in F.items():
    if "rocclr" in k:
        continue  # one-off zero fills at plan creation
    r = c["FETCH_SIZE"] * 1024 * read_factor(k)  # KB; x2 where gfx950 tallies a wide coalesced read at half its bytes
    if read_factor(k) == 1.0:
        narrow_r += r
    w = W.get(k, (0, {"WRITE_SIZE": 0.0}))[1]["WRITE_SIZE"] * 1024
    tot_r, tot_w = tot_r + r, tot_w + w
    c_ = cls(k)
    if c_:
        a = agg.setdefault(c_, {"launches": 0, "read": 0.0, "write": 0.0})
        a["launches"] += n
        a["read"] += r
        a["write"] += w
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 "
                 "--warmup 1 --no-cpu-baseline; FETCH_SIZE in KB x2 (gfx950: a wide coalesced read is tallied at half its "
                 "bytes, MI355X_MICROARCH.md HBM section), WRITE_SIZE in KB; summed over the executed steps",
       "csrc_hash": csrc_hash(), "steps": steps, "step_bytes": round((tot_r + tot_w) / steps),
       "step_bytes_upper": round((tot_r + narrow_r + tot_w) / steps), "uncalibrated_read_gb_per_step": round(narrow_r / steps / 1e9, 3), "total_gb_per_step": {"read": round(tot_r / steps / 1e9, 2), "write": round(tot_w / steps / 1e9, 2)},
       "classes": {c_: {"launches_per_step": a["launches"] / steps, "read_bytes_per_launch": round(a["read"] / a["launches"]),
                        "write_bytes_per_launch": round(a["write"] / a["launches"]),
                        "bytes_per_launch": round((a["read"] + a["write"]) / a["launches"])} for c_, a in agg.items()}}
json.dump(out, open(D + "pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
