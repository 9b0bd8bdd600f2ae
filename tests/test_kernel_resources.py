"""What the shipped binary holds, read from the gfx950 code object inside libp264amd.so (no GPU needed): the kernels of the hot
path must not spill vector registers or use scratch memory at all, and the figures DESIGN.md quotes (registers, LDS) must be
the binary's.  A claim like "no scratch" cannot drift from the build again (round 4's review found 17 spilled registers in
k_deblock and 4 in k_intra_sparse that the documents did not know about)."""
import pytest

from p264decoder_amd.tools import kernel_resources as kr

# kernels of the metric's launches and of config 2 (I pictures): no spills, no scratch
HOT = ["k_mc_sort", "k_mc", "k_intra", "k_intra_sparse", "k_deblock", "k_deblock_bs<false>"]
# B-picture launches: the second MC pass and the two-list edge info
HOT_B = ["k_mc_second", "k_deblock_bs<true>", "k_mc_sort_b"]


@pytest.fixture(scope="module")
def resources(lib):
    from p264decoder_amd import _native
    try:
        return kr.kernel_resources(_native.LIB_PATH)
    except RuntimeError as e:                         # no ROCm LLVM tools on this machine
        pytest.skip(str(e))


def test_every_kernel_is_in_the_code_object(resources):
    for k in HOT + HOT_B + ["k_tile_convert"]:
        assert k in resources, "kernel %s missing from the gfx950 code object (have: %s)" % (k, sorted(resources))


@pytest.mark.parametrize("kernel", HOT + HOT_B)
def test_hot_kernels_do_not_spill(resources, kernel):
    r = resources[kernel]
    assert r["vgpr_spill_count"] == 0, "%s spills %d vector registers" % (kernel, r["vgpr_spill_count"])
    assert r["private_segment_fixed_size"] == 0, "%s uses %d bytes of scratch per lane" % (kernel, r["private_segment_fixed_size"])


def test_occupancy_figures_of_design_md(resources):
    """DESIGN.md section 3 prices the kernels by wavefronts per SIMD: 512 registers per lane and SIMD, 160 KB of LDS per CU."""
    # k_mc: 4 wavefronts per SIMD (<= 128 registers), 4 workgroups of 256 threads per CU by LDS as well
    assert resources["k_mc"]["vgpr_count"] <= 128
    assert 4 * resources["k_mc"]["group_segment_fixed_size"] <= 160 * 1024
    # k_deblock: one workgroup of 16 wavefronts per CU = 4 per SIMD
    assert resources["k_deblock"]["vgpr_count"] <= 128
    assert resources["k_deblock"]["group_segment_fixed_size"] <= 160 * 1024
    # both intra builds: 8 wavefronts per SIMD
    for k in ("k_intra", "k_intra_sparse"):
        assert resources[k]["vgpr_count"] <= 64
