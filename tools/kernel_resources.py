#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output (VGPR / SGPR / scratch / occupancy per kernel)."""
import re
import subprocess
import sys


def main(path, pattern=""):
    txt = open(path).read()
    blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
    for b in blocks:
        name = b.split("\n")[0].split(" ")[0]

        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1

        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dn = dn.replace("pcs::", "").replace("void ", "")
        dn = re.sub(r"\(.*", "", dn)
        if pattern and not re.search(pattern, dn):
            continue
        scr = g(r"ScratchSize \[bytes/lane\]")
        occ = g(r"Occupancy \[waves/SIMD\]")
        print(f"{dn:60s} vgpr={g('VGPRs'):4d} agpr={g('AGPRs'):3d} sgpr={g('SGPRs'):3d} scratch={scr:4d} occ={occ}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
