"""Register / scratch budget of the default kernel variants (hipcc's kernel-resource-usage remarks, no GPU needed).

A harmless-looking source change can push a kernel over an occupancy step (round 2: one extra flag test took the fp16
kernel from 165 to 212 VGPRs = 3 -> 2 waves per SIMD and 25 % of its speed); this test pins the allocations the
measured numbers in DESIGN.md were taken with."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# kernel (substring of the mangled name) -> (max VGPRs, max scratch bytes per lane)
BUDGET = {
    "17fused_eval_kernelILi0E": (128, 0),                       # dense maps, 4 waves per SIMD
    "17fused_eval_kernelILi1E": (128, 0),
    "22fused_eval_dist_kernelILi0ELi4ELi6ELb1ELb0E": (64, 0),       # the distance-only pass, four views, tiled depth lookups: 7 waves (102 SGPRs hold KRt)
    "22fused_eval_dist_kernelILi0ELi4ELi6ELb0ELb0E": (64, 0),
    "22fused_eval_dist_kernelILi0ELi4ELi6ELb0ELb1E": (72, 0),       # ... from the axis arrays of a d3f_grid
    "22fused_eval_dist_kernelILi0ELi0ELi6ELb1E": (72, 0),       # ... five to eight views
    "22fused_eval_dist_kernelILi0ELi2ELi8ELb1E": (64, 0),       # ... one or two views: 8 waves
    "22fused_eval_dist_kernelILi1ELi4ELi6ELb0E": (72, 0),       # ... eval_dist
    "22fused_eval_wide_kernelILi0E": (168, 0),                  # C = 1024 dense, 3 waves
    "21fused_eval_f16_kernelILi0E": (168, 0),                   # fp16-stored maps, 3 waves
    "22fused_eval_runs_kernelILi0ELi1ELi4ELi7E": (72, 0),       # cell runs (1,4), 7 waves
    "22fused_eval_runs_kernelILi0ELi2ELi8ELi3E": (168, 0),      # cell runs (2,8), 3 waves
    "24fused_eval_sliced_kernelILi5ELi2ELi7ELb0E": (72, 0),         # channel-sliced, 7 waves
    "24fused_eval_window_kernelILi1ELi1ELi4ELi256ELi16ELi4ELb0ELb0E": (128, 0),   # LDS texel windows of lattice bricks, pipelined point loop (V = 4): 4 waves per SIMD
    "24fused_eval_window_kernelILi1ELi1ELi4ELi256ELi16ELi8ELb0ELb0E": (128, 0),   # ... V = 8
    "24fused_eval_window_kernelILi1ELi1ELi4ELi256ELi16ELi0ELb0ELb0E": (128, 0),   # ... any other view count: plain view loop
    "24fused_eval_window_kernelILi1ELi1ELi4ELi256ELi16ELi4ELb1ELb0E": (128, 0),   # touched-texel pool (clouds behind the device-side gate), V = 4
    "24fused_eval_window_kernelILi1ELi1ELi4ELi256ELi16ELi8ELb1ELb0E": (128, 0),
    "24fused_eval_window_kernelILi1ELi1ELi4ELi256ELi16ELi0ELb1ELb0E": (128, 0),
    "24window_gate_probe_kernel": (128, 0),
    "24fused_eval_window_kernelILi1ELi1ELi4ELi256ELi16ELi4ELb0ELb1E": (128, 0),   # fp16-stored maps: 256-byte slices, v_fma_mix_f32
    "24fused_eval_window_kernelILi1ELi1ELi4ELi256ELi16ELi8ELb0ELb1E": (128, 0),
    "24fused_eval_window_kernelILi1ELi1ELi4ELi256ELi16ELi0ELb0ELb1E": (128, 0),
    "24fused_eval_sliced_kernelILi4ELi2ELi7ELb1E": (72, 0),                       # ... in the channel-sliced launch
}


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_default_kernels_keep_their_register_budget():
    out = subprocess.run(["bash", os.path.join(ROOT, "scripts", "kernel_resources.sh")], capture_output=True, text=True, timeout=600).stdout
    seen = {}
    for line in out.splitlines():
        m = re.match(r"(\S+)\s+vgpr\s+(\d+)\s+sgpr\s+(\d+)\s+scratch\s+(\d+)\s+occ\s+(\d+)", line)
        if m:
            seen[m.group(1)] = (int(m.group(2)), int(m.group(4)))
    assert seen, out[-500:]
    for key, (max_vgpr, max_scratch) in BUDGET.items():
        hits = [v for k, v in seen.items() if k.startswith(key)]
        assert hits, "kernel %s not found among %s" % (key, sorted(seen))
        vgpr, scratch = hits[0]
        assert vgpr <= max_vgpr and scratch <= max_scratch, "%s: %d VGPRs / %d B scratch, budget %d / %d" % (key, vgpr, scratch, max_vgpr, max_scratch)
    # the product build compiles only the variants the planner picks by itself, and none of them spills
    spilling = {k: v for k, v in seen.items() if v[1] > 0}
    assert not spilling, spilling
    assert len(seen) <= 44, "experiment variants leaked into the product build: %s" % sorted(seen)


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_pairwise_mfma_kernel_keeps_four_waves_per_simd():
    """The guarded matrix-core pairwise kernel: its 16 accumulator registers come out of the same 512-entry file as the VGPRs, so
    126 VGPRs + 16 AGPRs cost a wave per SIMD (round 6: 0.305 ms at three waves, 0.285 at four, __launch_bounds__(256, 4)); no spills."""
    out = subprocess.run(["bash", os.path.join(ROOT, "scripts", "kernel_resources.sh"), "corr_kernels.hip"], capture_output=True, text=True, timeout=600).stdout
    rows = [re.match(r"(\S+)\s+vgpr\s+(\d+)\s+sgpr\s+(\d+)\s+scratch\s+(\d+)\s+occ\s+(\d+)", line) for line in out.splitlines()]
    mfma = [(m.group(1), int(m.group(2)), int(m.group(4)), int(m.group(5))) for m in rows if m and "pairwise_mfma_kernel" in m.group(1)]
    assert len(mfma) == 4, out[-800:]
    for name, vgpr, scratch, occ in mfma:
        assert vgpr <= 128 and scratch == 0 and occ >= 4, (name, vgpr, scratch, occ)


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_register_rows_kernel_leaves_the_rows_alone(tmp_path):
    """fuse_rows.hip keeps 32 fused rows in v[128:255] -- registers only its asm blocks name; the compiler gets 128 VGPRs
    (amdgpu_num_vgpr) for everything else.  The ISA must show: 256 VGPRs reserved (two waves per SIMD), no AGPRs (they would come
    on top of the 256 and cost the second wave), no scratch, and not one compiler-written instruction that touches v128 and up --
    outside the asm blocks every high register would be a clobbered row."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = os.path.join(ROOT, "d3fields_amd", "csrc", "fuse_rows.hip")
    asm = str(tmp_path / "fuse_rows.s")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-I", os.path.join(ROOT, "include"),
                    "-I", os.path.dirname(src), "--cuda-device-only", "-S", src, "-o", asm], check=True, timeout=600)
    text = open(asm).read()
    meta = {k: int(v) for k, v in re.findall(r"\.(vgpr_count|agpr_count|vgpr_spill_count|private_segment_fixed_size):\s+(\d+)", text)}
    assert meta == {"vgpr_count": 256, "agpr_count": 0, "vgpr_spill_count": 0, "private_segment_fixed_size": 0}, meta
    inside, high = False, []
    for line in text.splitlines():
        if "#ASMSTART" in line:
            inside = True
        elif "#ASMEND" in line:
            inside = False
        elif not inside and not line.lstrip().startswith((";", ".")) and re.search(r"\bv(?:\[)?(1[3-9]\d|12[89]|2\d\d)\b", line):
            high.append(line.strip())
    assert not high, high[:5]
    assert text.count("s_set_gpr_idx_on") >= 4 and "v_pk_fma_f32 v[128:129]" in text          # the rows are addressed through the index mode
