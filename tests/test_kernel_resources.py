"""What the compiler made of the path kernels (no GPU: hipcc cross-compiles gfx950 here): the shipped radiance kernel
must fit the register budget of 6 waves per SIMD with nothing spilled -- neither VGPRs to scratch nor SGPRs into the
lanes of a VGPR -- and six of its blocks must fit a CU's LDS.  The budget is met exactly (80 of 80), so this is the
test that notices when a change to the kernel sources tips it over."""
import importlib.util
import os

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_shipped_kernels_fit_their_budgets_without_spills():
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(REPO, "tools", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    k = kr.kernels(kr.compile_to_asm())
    rad = k["pathKernel<radiance, plain, wide>"]
    assert rad["vgpr_count"] <= 80, rad                                   # 512 / 6 = 85, allocated in eights
    assert rad["vgpr_spill_count"] == 0 and rad["sgpr_spill_count"] == 0 and rad["private_segment_fixed_size"] == 0, rad
    assert rad["group_segment_fixed_size"] * 6 <= 160 * 1024, rad         # six blocks of four waves per CU
    counts = k["pathKernel<counts, plain, binary>"]
    assert counts["vgpr_count"] <= 80 and counts["vgpr_spill_count"] == 0 and counts["private_segment_fixed_size"] == 0, counts
    # the fallback and the instrumented variants keep the 5-wave budget and spill no VGPR either
    for name, f in k.items():
        if "pathKernel" in name:
            assert f["vgpr_count"] <= 96 and f["vgpr_spill_count"] == 0, (name, f)
