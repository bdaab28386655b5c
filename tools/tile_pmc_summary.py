#!/usr/bin/env python3
"""tile_pmc_summary.py <tools/pmc_das.py summary of config 2> <out.json> -- the per-term figures of das_tile.hip's launch that
bench.py prices the kernel against (profiles/r03_pmc_tile_cfg2.json): VALU / SALU / LDS instructions and VALU- / LDS-busy cycles per
executed wave64 (voxel, channel, transmit) term."""
import json
import sys

d = json.load(open(sys.argv[1]))
c, n = d["counters"], d["dispatches_summed"]
assert any("das_tile_kernel" in k for k in d["kernels"]), d["kernels"]
terms = 1024 * 1024 * 128 * 31 / 64
out = {
    "source": "tools/pmc_das.py --config 2 (rocprofv3 --pmc, one counter group per pass), " + ", ".join(d["kernels"]) + ", BASELINE config 2 at full size",
    "kernel_source_sha16": d["kernel_source_sha16"], "dispatches_summed": n,
    "counters": {k: c[k] for k in sorted(c) if k.startswith("SQ_")},
    "executed_wave_terms_per_launch": terms,
    "valu_instructions_per_wave_term": c["SQ_INSTS_VALU"] / n / terms,
    "salu_instructions_per_wave_term": c["SQ_INSTS_SALU"] / n / terms,
    "lds_instructions_per_wave_term": c["SQ_INSTS_LDS"] / n / terms,
    "valu_busy_cycles_per_wave_term_per_simd": c["SQ_ACTIVE_INST_VALU"] / n * 4 / terms,
    "lds_busy_cycles_per_wave_term_per_cu": c["SQ_ACTIVE_INST_LDS"] / n * 4 / terms,
    "notes": "SQ_ACTIVE_INST_* are quad-cycles summed over the SIMDs (VALU) / CUs (LDS); the per-term figures divide by the 1024^2 x 128 x 31 / 64 wave64 "
             "terms one launch executes (inside the aperture or not).  A launch of T ms at f GHz lasts T f 1e6 x 256 / terms clk per wave-term per CU: "
             "compare the VALU figure (per SIMD; a CU's four SIMDs work side by side) and the LDS figure (per CU) with it.",
}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print({k: round(v, 2) for k, v in out.items() if k.endswith("_term") or "_per_wave_term_" in k})
