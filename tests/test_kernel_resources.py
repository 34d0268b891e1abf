"""Register budget of the dense-factor kernels (no GPU needed: hipcc cross-compiles gfx950).

The two-workgroups-per-CU variant of the fused step (k_ldlt_step2, D >= 3072: config 5) only reaches two waves per SIMD while
VGPRs + AGPRs <= 256.  A change in ba_panel_body or ba_update_macro that pushes it over halves the occupancy silently and costs
config 5 a third of its factorisation speed (it happened in round 1), so the budget is pinned here -- with the library's own
compiler flags (csrc/Makefile: -mllvm -amdgpu-mfma-vgpr-form; without it the macro tile's accumulators go to 128 AGPRs on top
of the panel's VGPRs)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
def test_two_per_cu_variant_keeps_its_occupancy(tmp_path):
    src = os.path.join(ROOT, "scripts", "bench_dense.hip")
    inc = os.path.join(ROOT, "bundleadjustment_benchmarks_amd", "csrc")
    out = subprocess.run(
        ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "-I", inc, "--cuda-device-only", "-c", src, "-o",
         str(tmp_path / "bd.o"), "-Rpass-analysis=kernel-resource-usage", "-mllvm", "-amdgpu-mfma-vgpr-form"],
        capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    text = out.stderr
    usage = {}
    for m in re.finditer(r"Function Name: (\S+)", text):
        name = m.group(1)
        block = text[m.end():m.end() + 2500]
        v = re.search(r"VGPRs: (\d+)", block)
        a = re.search(r"AGPRs: (\d+)", block)
        o = re.search(r"Occupancy \[waves/SIMD\]: (\d+)", block)
        s = re.search(r"VGPRs Spill: (\d+)", block)
        if v and a and o and s:
            usage[name] = (int(v.group(1)), int(a.group(1)), int(o.group(1)), int(s.group(1)))
    two_per_cu = [k for k in usage if "k_ldlt_step2" in k]
    one_per_cu = [k for k in usage if "k_ldlt_stepI" in k]
    assert two_per_cu and one_per_cu, sorted(usage)
    for k in two_per_cu:
        vg, ag, occ, spill = usage[k]
        assert vg + ag <= 256 and occ >= 2 and spill == 0, (k, usage[k])
    for k in one_per_cu:  # one workgroup per CU by its LDS request; it must not spill
        assert usage[k][3] == 0, (k, usage[k])
