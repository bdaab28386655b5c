"""No kernel of the shipped library spills vector registers or uses scratch memory (round 3: the headline LDS-staged kernel wrote
3.4 GiB per 1 GiB frame because five spilled registers went to scratch; tools/kernel_resources.py reads the figures the compiler
recorded in the code objects embedded in the .so)."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"), reason="needs the ROCm LLVM binutils")
def test_no_kernel_spills_vector_registers_or_uses_scratch():
    import kernel_resources
    library = os.path.join(ROOT, "ogl_beamforming_amd", "libogl_beamformer_lib.so")
    assert os.path.exists(library) and shutil.which("python3")
    kernels = kernel_resources.kernels_of(library)
    assert len(kernels) > 200, "the code objects were not found: the extraction is stale"
    names = {k["demangled"].split("(")[0].split("<")[0].replace("void ", "") for k in kernels}
    for expected in ("das_kernel", "das_factored_kernel", "das_rca_separable_kernel", "das_rca_staged_kernel", "das_hercules_kernel", "das_tile_kernel",
                     "filter_kernel"):
        assert any(expected in n for n in names), expected
    bad = [(k["demangled"], k["vgpr_spill_count"], k["private_segment_fixed_size"]) for k in kernels
           if k["vgpr_spill_count"] or k["private_segment_fixed_size"]]
    assert not bad, bad
    # 64 VGPRs = 8 waves per SIMD for the 1024-thread blocks of the staged kernels (two blocks per CU)
    for k in kernels:
        if "das_rca_staged_kernel" in k["demangled"] or "das_rca_separable_kernel" in k["demangled"]:
            assert k["vgpr_count"] <= 64, k["demangled"]
        # das_tile.hip addresses its LDS from byte 0 of the dynamic segment: no static LDS may sit in front of it (hipcc once "promoted" a
        # run-time indexed register array to 16 KB of it), and one 1024-thread block per CU leaves 128 registers a thread
        if "das_tile_kernel" in k["demangled"]:
            assert k["group_segment_fixed_size"] == 0 and k["vgpr_count"] <= 128, (k["demangled"], k["group_segment_fixed_size"], k["vgpr_count"])
