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


def test_bf16_weight_gradient_stages_by_dma_and_reads_transposed(tmp_path):
    """conv_mfma_wgrad_bf16t_kernel (csrc/conv_mfma.hip): what makes it fast is visible in the generated code and easy to lose —
    the data's way in is LDS-DMA only (`buffer_load_dwordx4 ... lds`: no vector register staging, so no v_perm transposes and no
    ds_write of tensor data in the step loop), the K = voxel operands come from `ds_read_b64_tr_b16`, nothing is spilled (a scratch
    access is a vector-memory operation and would break the counted `s_waitcnt vmcnt`), and two workgroups fit a CU (<= 256
    registers per lane; 77 824 bytes of LDS are requested at launch)."""
    from mri_epilepsy_diagnosis_amd import build
    if not os.path.exists(build.HIPCC):
        pytest.skip("no hipcc in this environment")
    src = os.path.join(build.CSRC, "conv_mfma.hip")
    out = tmp_path / "conv_mfma.s"
    cmd = [build.HIPCC] + build.FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-S", "--cuda-device-only", src, "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = out.read_text()
    kernels = re.findall(r"^(_ZN5mri3d\d+conv_mfma_wgrad_bf16t_kernel[^:\s]*):[^\n]*\n(.*?)s_endpgm", asm, flags=re.S | re.M)
    assert len(kernels) == 2, len(kernels)   # with / without the bias accumulator
    for name, body in kernels:
        assert "scratch_" not in body, "%s uses scratch memory" % name
        assert body.count("ds_read_b64_tr_b16") >= 20, (name, body.count("ds_read_b64_tr_b16"))   # 12 dY + 8 X fragment reads per plane
        assert len(re.findall(r"buffer_load_dwordx4 \S+, \S+, \S+ offen lds", body)) >= 5, name      # the wave's pieces of a step
        assert "v_perm_b32" not in body, "%s transposes in registers" % name
        assert len(re.findall(r"v_mfma_f32_16x16x32_bf16", body)) >= 54, name
    blocks = re.findall(r"Function Name: (\S*conv_mfma_wgrad_bf16t\S*).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+)",
                        r.stderr, flags=re.S)
    assert len(blocks) == 2
    for name, vg, ag, scratch, occ in blocks:
        assert int(vg) + int(ag) <= 256 and int(scratch) == 0 and int(occ) >= 2, (name, vg, ag, scratch, occ)
