#!/bin/bash
# Prints VGPRs / scratch bytes / waves per SIMD of every kernel in one csrc file (default fuse_eval.hip).
# usage: scripts/kernel_resources.sh [file.hip]
src=${1:-/root/repo/d3fields_amd/csrc/fuse_eval.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
    -I /root/repo/include -I /root/repo/d3fields_amd/csrc -c "$src" -o /dev/null \
    -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys
name, d = None, {}
for l in sys.stdin:
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        name, d = m.group(1), {}
    for k, tag in (("VGPRs", "vgpr"), (r"ScratchSize \[bytes/lane\]", "scratch"), (r"Occupancy \[waves/SIMD\]", "occ")):
        m = re.search(r" " + k + r": (\d+)", l)
        if m:
            d[tag] = int(m.group(1))
            if tag == "occ":
                print("%-62s vgpr %3d scratch %3d occ %d" % (name[7:69], d.get("vgpr", -1), d.get("scratch", -1), d["occ"]))
'
