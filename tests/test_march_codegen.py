"""Code-generation guard of csrc/conv_march.hip (CPU container: hipcc cross-compiles gfx950 without a GPU).

The marching kernels keep their 24 accumulators in the registers v[160:255] BY NAME through inline asm, invisible to hipcc's
register allocator, which is capped at v0..v159 by `amdgpu_num_vgpr(160)`.  That is only sound while every register hipcc itself
allocates stays below v160, while it never parks values in accumulation registers (none are budgeted), and while nothing is
spilled to scratch (a scratch access is a vector-memory operation and would break the kernels' counted `s_waitcnt vmcnt`).
This test compiles the file with the product's flags and checks all three in the generated code."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_march_kernels_keep_hipcc_off_the_named_accumulators(tmp_path):
    from mri_epilepsy_diagnosis_amd import build
    if not os.path.exists(build.HIPCC):
        pytest.skip("no hipcc in this environment")
    src = os.path.join(build.CSRC, "conv_march.hip")
    out = tmp_path / "conv_march.s"
    cmd = [build.HIPCC] + build.FLAGS + build.EXTRA_FLAGS["conv_march.hip"] + ["-Rpass-analysis=kernel-resource-usage", "-S", "--cuda-device-only", src, "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = out.read_text()
    kernels = re.findall(r"^(_ZN5mri3d\d+conv_march\d*_kernel[^:\s]*):[^\n]*\n(.*?)s_endpgm", asm, flags=re.S | re.M)
    assert len(kernels) == 8, len(kernels)   # {fp32, bf16} x {statistics} x {bias}
    for name, body in kernels:
        assert "scratch_" not in body, "%s uses scratch memory" % name
        assert "v_accvgpr" not in body, "%s: hipcc uses accumulation registers" % name
        in_asm, worst = False, 0
        for ln in body.split("\n"):
            if "ASMSTART" in ln:
                in_asm = True
            elif "ASMEND" in ln:
                in_asm = False
            elif not in_asm and not ln.strip().startswith(";"):
                for m in re.finditer(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]", ln):
                    worst = max(worst, int(m.group(1)) if m.group(1) else int(m.group(3)))
        assert worst < 160, "%s: hipcc allocated v%d, inside the named accumulator block v160..v255" % (name, worst)
        # the MFMAs themselves only ever write the named block
        for m in re.finditer(r"v_mfma_\S+ v\[(\d+):(\d+)\]", body):
            assert 160 <= int(m.group(1)) and int(m.group(2)) <= 255, (name, m.group(0))
    blocks = re.findall(r"Function Name: (\S*conv_march\S*).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+)",
                        r.stderr, flags=re.S)
    assert len(blocks) == 8
    for name, vg, ag, scratch, occ in blocks:
        assert (int(vg), int(ag), int(scratch), int(occ)) == (256, 0, 0, 2), (name, vg, ag, scratch, occ)
