"""gpurun_out/other_configs.raw (scripts/collect_other_configs.sh) -> profiles/<round>/final/other_configs.jsonl

    python scripts/other_configs.py gpurun_out/other_configs.raw profiles/r01/final/other_configs.jsonl"""
import json
import sys

WHAT = {
    "cfg2": "training step, BASELINE configs[1] (3->1 ch, 256x256, S=2, fbc=21, batch 64)",
    "cfg4": "training step, BASELINE configs[3] geometry (2->1 ch, 256x256, S=4, fbc=30, batch 16 per GPU)",
    "cfg3": "cfg3 training step",
}
src, dst = sys.argv[1], sys.argv[2]
lines = [l.rstrip("\n") for l in open(src)]
out = []
for cmd, res in zip(lines[0::2], lines[1::2]):
    cmd = cmd[4:]
    d = json.loads(res)
    if "metric" in d:  # a bench.py line: keep the headline fields
        cfg = "cfg2" if "cfg2" in cmd else "cfg4" if "cfg4" in cmd else "cfg3"
        arith = ("bf16 storage + operands (bf16-mixed; reduced precision)" if "bf16-mixed" in cmd else
                 "fp16 storage + operands under GradScaler (16-mixed, the reference's production precision; reduced precision)"
                 if "16-mixed" in cmd else
                 "bf16 MFMA operands / fp32 accumulate and storage (reduced precision)" if "bf16" in cmd else
                 "fp32 MFMA" if "fp32" in cmd else "default split16 arithmetic")
        extra = ", batch 4 per GPU (the per-GPU batch of an 8-way strong-scaling run)" if "--batch 4" in cmd else ""
        e = {"what": f"{WHAT[cfg]}{extra}, {arith}", "images_per_s": d["value"], "ms_per_step": d["ms_per_step"],
             "hbm_frac_step": d.get("hbm_frac_step")}
        if "roofline" in d:
            k = d["roofline"]["kernels"]
            e["conv_ms_fwd_dgrad_wgrad"] = [k[c]["ms_per_step"] for c in ("conv3x3_fwd", "conv3x3_dgrad", "conv3x3_wgrad")]
            e["conv_tflops_fwd_dgrad_wgrad"] = [k[c]["tflops"] for c in ("conv3x3_fwd", "conv3x3_dgrad", "conv3x3_wgrad")]
            e["bandwidth_kernels_ms"] = d["roofline"]["bandwidth_kernels"]["ms_per_step"]
        d = e
    out.append({"command": cmd, **d})
with open(dst, "w") as fh:
    for d in out:
        fh.write(json.dumps(d) + "\n")
print(f"{len(out)} lines -> {dst}")
