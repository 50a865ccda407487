#!/bin/bash
# Prints VGPRs / SGPRs / scratch bytes / waves per SIMD of every kernel in csrc files (default: the four fused field-query files).
# usage: scripts/kernel_resources.sh [file.hip ...]        (paths are relative to this script: works in any checkout)
here=$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)
root=$(dirname "$here")
[ $# -eq 0 ] && set -- fuse_direct.hip fuse_runs.hip fuse_sliced.hip fuse_window.hip
for src in "$@"; do
case "$src" in /*) ;; *) [ -f "$src" ] || src=$root/d3fields_amd/csrc/$src ;; esac
${HIPCC:-/opt/rocm/bin/hipcc} --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math ${D3F_EXTRA_FLAGS:-} \
    -I "$root/include" -I "$root/d3fields_amd/csrc" -c "$src" -o /dev/null \
    -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys
name, d = None, {}
for l in sys.stdin:
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        name, d = m.group(1), {}
    for k, tag in (("VGPRs", "vgpr"), ("TotalSGPRs", "sgpr"), (r"ScratchSize \[bytes/lane\]", "scratch"), (r"Occupancy \[waves/SIMD\]", "occ")):
        m = re.search(r" " + k + r": (\d+)", l)
        if m:
            d[tag] = int(m.group(1))
            if tag == "occ":
                print("%-62s vgpr %3d sgpr %3d scratch %3d occ %d" % (name[7:69], d.get("vgpr", -1), d.get("sgpr", -1), d.get("scratch", -1), d["occ"]))
' &
done
wait
