"""What hipcc made of the hot kernels, read from the code object inside libhq_solver.so (no GPU): the brick kernels
must not spill -- a spilled VGPR is scratch traffic behind every load in flight -- and must keep the
occupancy their launch bounds promise (<= 128 VGPRs: two 512-thread workgroups per CU)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
SO = os.path.join(ROOT, "hercules_amd", "csrc", "libhq_solver.so")


def _kernel_notes(tmp_path):
    tools = [os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]
    if not all(os.path.exists(t) for t in tools) or not os.path.exists(SO):
        pytest.skip("llvm tools or the built library are not here")
    fat, co = str(tmp_path / "fat.bin"), str(tmp_path / "dev.co")
    subprocess.check_call([tools[0], "--dump-section", ".hip_fatbin=" + fat, SO])
    subprocess.check_call([tools[1], "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           "--input=" + fat, "--output=" + co])
    txt = subprocess.check_output([tools[2], "--notes", co], universal_newlines=True)
    kernels = {}
    for block in txt.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        kernels[name] = {k: int(v) for k, v in re.findall(r"\.(vgpr_count|vgpr_spill_count|sgpr_spill_count|"
                                                          r"private_segment_fixed_size|group_segment_fixed_size):\s+(\d+)", block)}
    return kernels


def test_brick_kernels_do_not_spill(tmp_path):
    k = _kernel_notes(tmp_path)
    found = {}
    tags = ("hq_k_brickILb0ELb0E", "hq_k_brickILb0ELb1E", "hq_k_brickILb1ELb0E", "hq_k_brickILb1ELb1E", "hq_k_brick_hetILb0ELb0E",
            "hq_k_brick_hetILb1ELb0E", "hq_k_brick_hetILb0ELb1E", "hq_k_brick_hetILb1ELb1E")
    for name, v in k.items():
        for tag in tags:
            if tag in name:
                found[tag] = v
    assert set(found) == set(tags), sorted(k)
    for tag, v in found.items():
        if tag == "hq_k_brick_hetILb1ELb1E":
            # hq_k_brick_het<PACKED, RAGGED> (round 6): the id pipeline of the ragged form on top of the packed form's 128
            # registers -- two spilled registers (12 bytes of scratch), accepted; every other form is spill-free
            assert v["vgpr_spill_count"] <= 2 and v["private_segment_fixed_size"] <= 16, (tag, v)
            assert v["vgpr_count"] <= 128, (tag, v)
            continue
        assert v["vgpr_spill_count"] == 0 and v["sgpr_spill_count"] == 0 and v["private_segment_fixed_size"] == 0, (tag, v)
        # two workgroups of 512 threads per CU = 4 waves per SIMD: <= 128 VGPRs (MI355X_MICROARCH.md, register files)
        assert v["vgpr_count"] <= 128, (tag, v)
    # hq_k_brick<PERNODE, BYCOMP>.  The uniform kernel leaves room beside its four waves per SIMD: 4 x 120 of the 512
    # registers per lane on one GPU, and 4 x 104 in the form partitions launch (plane sums component by component), so that
    # one / three waves of the exchange chain's kernels (<= 32 VGPRs, test below) per SIMD become resident on a CU two
    # brick workgroups occupy
    assert found["hq_k_brickILb0ELb0E"]["vgpr_count"] <= 120
    assert found["hq_k_brickILb0ELb1E"]["vgpr_count"] <= 104
    assert found["hq_k_brickILb0ELb0E"]["group_segment_fixed_size"] <= 80 * 1024


def test_exchange_chain_kernels_are_small(tmp_path):
    """pack / unpack / interface update / the IPC wait run beside the interior launches: few registers, no scratch."""
    k = _kernel_notes(tmp_path)
    seen = 0
    for name, v in k.items():
        if any(t in name for t in ("hq_k_pack", "hq_k_unpack", "hq_k_interface_update", "hq_k_ipc_wait")):
            seen += 1
            assert v["vgpr_count"] <= 32 and v["vgpr_spill_count"] == 0 and v["private_segment_fixed_size"] == 0, (name, v)
    assert seen >= 5


def test_single_precision_build_keeps_the_register_budget(tmp_path, monkeypatch):
    """libhq_solver_f32.so (hq_real = float): the same kernels on a float state.  Two workgroups per CU as in the fp64
    build: <= 128 VGPRs, no scratch -- except hq_k_brick_het<false> (24-byte coefficients, no hq_desc.edata), whose
    "one load, address chosen per lane" of n_t row / ring node needs ONE type and branches in the float build: 2 spilled
    registers, accepted for a form the packed one replaces wherever the caller hands over edata."""
    so = os.path.join(ROOT, "hercules_amd", "csrc", "libhq_solver_f32.so")
    if not os.path.exists(so):
        pytest.skip("libhq_solver_f32.so is not built")
    monkeypatch.setattr(sys.modules[__name__], "SO", so)
    k = _kernel_notes(tmp_path)
    seen = 0
    for name, v in k.items():
        if "hq_k_brick" not in name and "hq_k_patch" not in name:
            continue
        seen += 1
        assert v["vgpr_count"] <= 128, (name, v)
        if "hq_k_brick_hetILb0ELb1E" in name:             # ... and its RAGGED form (round 6): 6
            assert v["vgpr_spill_count"] <= 6, (name, v)
        elif "hq_k_brick_hetILb0E" in name:
            assert v["vgpr_spill_count"] <= 2, (name, v)
        else:
            assert v["vgpr_spill_count"] == 0 and v["private_segment_fixed_size"] == 0, (name, v)
    assert seen >= 10
