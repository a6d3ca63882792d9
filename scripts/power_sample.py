#!/usr/bin/env python3
"""Run a command and sample the GPU's power / clocks beside it (sustained-regime evidence: is the step power-bound?).

    python3 scripts/power_sample.py <out.txt> -- <command ...>

Samples every 50 ms from the amdgpu hwmon / sysfs files when they are readable (power1_average | power1_input in uW,
freq1_input = shader clock in Hz, pp_dpm_sclk's starred line), else every ~0.5 s through `rocm-smi --json`.  Writes one
line per sample and a summary (median / p10 / p90 of the samples taken while the command was in its busy phase: power
above half of the maximum seen).  The parent process never touches the GPU."""
import glob
import json
import os
import statistics
import subprocess
import sys
import threading
import time


def our_pci_address():
    """PCI address of the GPU a child process sees as device 0 (the box may hold other tenants' GPUs)."""
    code = ("import torch; p = torch.cuda.get_device_properties(0); "
            "print('%04x:%02x:%02x' % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id))")
    try:
        return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=180).stdout.strip().splitlines()[-1]
    except Exception:
        return None


def find_sysfs(pci=None):
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        if pci and pci.lower() not in os.path.realpath(dev).lower():
            continue
        hw = glob.glob(os.path.join(dev, "hwmon", "hwmon*"))
        if not hw:
            continue
        h = hw[0]
        p = next((os.path.join(h, n) for n in ("power1_average", "power1_input") if os.path.exists(os.path.join(h, n))), None)
        if p is None:
            continue
        f = os.path.join(h, "freq1_input")
        return {"power": p, "freq": f if os.path.exists(f) else None, "cap": os.path.join(h, "power1_cap"),
                "sclk": os.path.join(dev, "pp_dpm_sclk")}
    return None


def read_num(path):
    try:
        return float(open(path).read().split()[0])
    except Exception:
        return None


def smi_sample():
    try:
        out = subprocess.run(["rocm-smi", "-P", "-c", "--json"], capture_output=True, text=True, timeout=5).stdout
        d = json.loads(out)
        card = next(iter(d.values()))
        pw = next((float(v) for k, v in card.items() if "ower" in k and "W" in k), None)
        ck = next((v for k, v in card.items() if k.startswith("sclk")), None)
        mhz = float(str(ck).strip("()").lower().replace("mhz", "")) if ck else None
        return pw, mhz
    except Exception:
        return None, None


def main():
    out_path = sys.argv[1]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    pci = our_pci_address()
    fs = find_sysfs(pci) or find_sysfs()
    samples = []
    stop = threading.Event()

    def loop():
        t0 = time.time()
        while not stop.is_set():
            if fs:
                pw = read_num(fs["power"])
                fr = read_num(fs["freq"]) if fs["freq"] else None
                samples.append((time.time() - t0, pw / 1e6 if pw is not None else None, fr / 1e6 if fr is not None else None))
                time.sleep(0.05)
            else:
                pw, mhz = smi_sample()
                samples.append((time.time() - t0, pw, mhz))
                time.sleep(0.3)

    th = threading.Thread(target=loop, daemon=True)
    th.start()
    rc = subprocess.call(cmd)
    stop.set()
    th.join(timeout=2)
    with open(out_path, "w") as f:
        f.write(f"# command: {' '.join(cmd)}\n# device 0 at PCI {pci}; source: {'sysfs ' + os.path.realpath(fs['power']) if fs else 'rocm-smi'}\n")
        if fs:
            cap = read_num(fs["cap"])
            if cap:
                f.write(f"# power cap: {cap / 1e6:.0f} W\n")
        pws = [s[1] for s in samples if s[1] is not None]
        if pws:
            hi = max(pws)
            busy = [s for s in samples if s[1] is not None and s[1] > 0.5 * hi]
            bp = sorted(s[1] for s in busy)
            bf = sorted(s[2] for s in busy if s[2] is not None)
            q = lambda v, a: v[min(len(v) - 1, int(a * len(v)))]
            f.write(f"# busy samples: {len(busy)} of {len(samples)}; power W median {statistics.median(bp):.0f} p10 {q(bp, 0.1):.0f} "
                    f"p90 {q(bp, 0.9):.0f} max {hi:.0f}")
            if bf:
                f.write(f"; shader clock MHz median {statistics.median(bf):.0f} p10 {q(bf, 0.1):.0f} p90 {q(bf, 0.9):.0f}")
            f.write("\n")
        else:
            f.write("# no power samples readable on this box\n")
        for t, pw, fr in samples:
            f.write(f"{t:8.2f} {'' if pw is None else f'{pw:7.1f}'} {'' if fr is None else f'{fr:7.0f}'}\n")
    sys.exit(rc)


if __name__ == "__main__":
    main()
