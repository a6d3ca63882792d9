"""Build-time guard (run by the Makefile): the kernels that synchronise hand-issued LDS-DMA with a hand-counted
`s_waitcnt vmcnt(N)` — conv3x3_ws_kernel, conv3x3_wide_kernel, wgrad_split_ws_kernel — must not touch scratch.
A spill or reload is a vector-memory instruction the count does not know about: the barrier could then pass before the
DMA'd weights have landed in LDS (silent wrong results).  Reads hipcc's -Rpass-analysis=kernel-resource-usage
remarks (csrc/*.res, written by the compile rule) and fails the build if any such instantiation has a non-zero
ScratchSize or spills vector registers.

    python3 scripts/check_resources.py mimo_unet_amd/csrc/*.res
"""
import re
import sys

GUARDED = ("conv3x3_ws_kernel", "conv3x3_wide_kernel", "wgrad_split_ws_kernel")


def parse(path):
    out, cur = {}, None
    for ln in open(path, errors="replace"):
        m = re.search(r"remark: Function Name: (\S+)", ln)
        if m:
            cur = m.group(1)
            out[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", ln)
        if m and cur:
            out[cur][m.group(1).strip()] = int(m.group(2))
    return out


def main(paths):
    bad, seen = [], 0
    for p in paths:
        for fn, r in parse(p).items():
            if not any(g in fn for g in GUARDED):
                continue
            seen += 1
            if r.get("ScratchSize", 0) != 0 or r.get("VGPRs Spill", 0) != 0:
                bad.append((fn, r.get("ScratchSize"), r.get("VGPRs Spill")))
    for fn, sc, sp in bad:
        print(f"check_resources: {fn}: ScratchSize {sc} bytes/lane, {sp} VGPRs spilled — its counted vmcnt waits are "
              "no longer exact", file=sys.stderr)
    print(f"check_resources: {seen} LDS-DMA kernel instantiations, {len(bad)} with scratch")
    return 1 if bad or seen == 0 else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
