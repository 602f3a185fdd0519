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
    # the pool kernels (radiance, shipped default; pacedPoolKernel for static partitions): same register budget, no VGPR spilled; the SGPRs it spills (into the lanes
    # of one VGPR, itself kept in 20 bytes of scratch) are spilled and reloaded in the service phase only -- the traversal
    # step, where the wave spends nine tenths of its time, must stay free of spill traffic.  LDS: 21 granules of 1,280 bytes
    # (pool_kernel.hpp): one byte more and the sixth block of a CU no longer fits.
    marked = kr.compile_to_asm(["-DPTX_ISA_MARKERS"])
    for name, mangled in (("poolKernel", "_ZN3ptx10poolKernelILb0E"), ("pacedPoolKernel", "_ZN3ptx15pacedPoolKernelILb0E")):
        pool = k[name + "<plain>"]
        assert pool["vgpr_count"] <= 80 and pool["vgpr_spill_count"] == 0 and pool["private_segment_fixed_size"] <= 32, (name, pool)
        assert pool["group_segment_fixed_size"] <= 21 * 1280, (name, pool)
        assert k[name + "<stats>"]["vgpr_spill_count"] == 0, (name, k[name + "<stats>"])
        by_region = kr.pool_spill_ops_by_region(marked, mangled)
        assert by_region.get("SERVICE", 0) > 0, by_region   # the markers are there
        for region in ("TRAV_BEGIN", "ENTER", "TRI", "NODE", "MERGE", "TRAV_END"):
            assert by_region.get(region, 0) == 0, (name, by_region)
    counts = k["pathKernel<counts, plain, binary>"]
    assert counts["vgpr_count"] <= 80 and counts["vgpr_spill_count"] == 0 and counts["private_segment_fixed_size"] == 0, counts
    # the fallback and the instrumented variants keep the 5-wave budget and spill no VGPR either
    for name, f in k.items():
        if "pathKernel" in name:
            assert f["vgpr_count"] <= 96 and f["vgpr_spill_count"] == 0, (name, f)
