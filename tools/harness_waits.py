"""What the kernels of the reference harness's frames wait for (round-3 verdict item 4): one table out of
  * tools/pmc_das.py --config harness:<kind> --groups 0,10,11 (SQ wait / issue / active wave cycles, TCP miss latency),
  * tools/timeline_probe.sh (per-wave s_memrealtime stamps: resident waves over time),
  * the A/B runs of the walk / split variants (tools/build_variant.sh + tools/ab_harness.sh; DAS ms per variant, one box).
python3 tools/harness_waits.py gpurun_out/r04 profiles/r04_harness_waits.json"""
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
out = {"what": "reference harness frames (512 x 1024 view plane, 256 ch x 128 tx, cubic, F# 0.5): what the waves of the DAS kernel do with their cycles",
       "frames": {}}
for kind in ("tpw", "hercules", "forces"):
    p = os.path.join(src, f"pmc_wait_{kind}", "summary.json")
    if not os.path.exists(p):
        continue
    s = json.load(open(p)); c, n = s["counters"], s["dispatches_summed"]
    cycles = c["GRBM_GUI_ACTIVE"] / 8.0                       # summed over the 8 XCDs
    w = c["SQ_WAVE_CYCLES"]
    e = {"kernel": s["kernels"], "dispatches_summed": n, "kernel_cycles_per_launch": cycles / n,
         "wave_cycles_frac": {"parked_on_s_waitcnt": c["SQ_WAIT_ANY"] / w, "issue_stalled": c["SQ_WAIT_INST_ANY"] / w, "an_instruction_in_flight": c["SQ_ACTIVE_INST_ANY"] / w},
         "mean_resident_waves_per_cu": w * 4.0 / (cycles * 256),
         "valu_instructions_per_gather_instruction": c["SQ_INSTS_VALU"] / c["SQ_INSTS_VMEM_RD"],
         "gather_instructions_per_cu_per_launch": c["SQ_INSTS_VMEM_RD"] / 256 / n,
         "elapsed_clk_per_gather_instruction_per_cu": cycles * 256 / c["SQ_INSTS_VMEM_RD"],
         "gather_ceiling_clk_per_instruction": 16.3,
         "l1_accesses_per_gather_instruction": c["TCP_TOTAL_CACHE_ACCESSES_sum"] / c["SQ_INSTS_VMEM_RD"],
         "l1_accesses_per_clk_per_cu": c["TCP_TOTAL_CACHE_ACCESSES_sum"] / (cycles * 256),
         "l1_hit_rate": 1.0 - c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"],
         "l1_miss_latency_clk": c["TCP_TCC_READ_REQ_LATENCY_sum"] / c["TCP_TCC_READ_REQ_sum"],
         "tcp_pending_stall_frac": c["TCP_PENDING_STALL_CYCLES_sum"] / (cycles * 256)}
    out["frames"][f"harness:{kind}"] = e
p = os.path.join(src, "timeline", "r04_timeline.json")
if os.path.exists(p):
    t = json.load(open(p))
    out["timeline"] = {k: {kk: v[kk] for kk in ("kernel_span_ms", "mean_resident_waves_per_cu", "wave_duration_ms", "xcd_runs_dry_at_ms", "time_with_fewer_than_half_of_peak_resident_frac")}
                       | {"resident_waves_at_41_points": v["resident_waves_at_200_points"][::5]} for k, v in t["frames"].items()}
ab = {}
for d in ("ab_tail", "ab_w64", "ab_tile", "ab_tile_b", "ab_lpt2", "ab_lpt2b"):
    dd = os.path.join(src, d)
    if not os.path.isdir(dd):
        continue
    for f in sorted(os.listdir(dd)):
        if f.endswith(".json"):
            name, kind = f[:-5].rsplit("_", 1)
            try:
                ab.setdefault(name if d in ("ab_tail", "ab_w64") else f"{name} ({d[3:]} run)", {})[kind] = json.loads(open(os.path.join(dd, f)).read().strip().splitlines()[-1])["config"]["stage_ms"]["DAS"]
            except Exception:
                pass
out["variants_DAS_ms_one_box_each_row_its_own_run"] = ab
out["l1_access_rate_is_not_the_limit"] = {
    "why_asked": "the factored kernels show 0.97-0.98 TCP_TOTAL_CACHE_ACCESSES per clock per CU: a one-per-clock unit at its limit?",
    "probe": "rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE -- tools/bin/microbench, gather_probe<16 bytes, pattern>, window resident in L1 (gpurun_out/r04/clock/mb_pmc.txt)",
    "contiguous": {"accesses_per_instruction": 22.0, "clk_per_instruction_per_cu": 16.4, "accesses_per_clk_per_cu": 1.35},
    "das_like (0..8 B between neighbouring lanes)": {"accesses_per_instruction": 18.9, "clk_per_instruction_per_cu": 16.4, "accesses_per_clk_per_cu": 1.15},
    "random lines": {"accesses_per_instruction": 64.8, "clk_per_instruction_per_cu": 33.0, "accesses_per_clk_per_cu": 1.95},
    "reading": "the L1 takes up to ~2 accesses per clock; an instruction costs max(16.3 clk, accesses / 2).  At 18-19 accesses per instruction the harness frames pay the 16.3 clk of the address path, not the L1's rate."}
out["reading"] = ("TPW / FORCES (das_factored_kernel): two pipes share the time -- the vector-memory address path is busy 0.87 of it counted in EXECUTED wave64 gather instructions "
                  "(16.3 clk each; 0.77 counted in bytes of terms that pass the f-number test: 12 % of the lanes of executing waves are masked by the aperture test) and the VALU 0.77.  A wave is "
                  "parked on s_waitcnt 41-42 % of its life, issue-stalled 27 %, has an instruction in flight 30-32 %; L1 hit 0.89-0.93, an L1 miss comes back in 216-255 clk (L2 hits), TCP pending stalls 6 %.  "
                  "The timeline shows where the rest goes: a wave lives 7.5 ms of a 17 ms launch (two generations per slot), so the chip drains for the last 30 % of the launch "
                  "(resident waves fall 4096 -> 0 roughly linearly) -- but the variants that shorten the tail (deepest bands first; the channel loop split over 2 or 4 waves; one-wave blocks) "
                  "are no faster (TPW 16.0 -> 16.3-17.5 ms): with fewer waves resident each one runs faster, the pipes are what is shared.  HERCULES: VALU-bound (issue-stalled 45 %, "
                  "15.6 VALU instructions per gather).  Floor of this formulation on this grid: max(address path, VALU) with both near 0.8-0.87 -- a few per cent from better overlap, "
                  "not a factor; a different formulation (tile-wide staging) was measured slower on 0.23 mm pixels in round 3 (docs/NOTEBOOK.md 3.1c).")
# what the counters say after the round's two changes to these kernels (tap-weight cubic; waves of a block in step): the committed PMC summary
try:
    bound = json.load(open(os.path.join(os.path.dirname(dst), "r04_das_bound.json")))
    traffic = json.load(open(os.path.join(os.path.dirname(dst), "das_traffic.json")))
    after = {}
    for key in ("harness:tpw", "harness:hercules", "harness:forces"):
        for kernel, e in bound.get(key, {}).items():
            if isinstance(e, dict) and "valu_busy_frac" in e:
                after[key] = {"kernel": kernel, "valu_busy_frac": e["valu_busy_frac"], "ta_busy_frac": e.get("ta_busy_frac"), "l1_hit_rate": e.get("l1_hit_rate"),
                              "l2_hit_rate": e.get("l2_hit_rate"), "wave_cycles": e.get("wave_cycles"), "l1_miss_latency_cycles": e.get("l1_miss_latency_cycles"),
                              "hbm_GB_per_launch": (traffic.get(key, {}).get(kernel, {}).get("hbm_bytes_per_launch") or 0) / 1e9,
                              "kernel_source_sha16": e.get("kernel_source_sha16")}
    out["after_tap_weights_and_waves_in_step"] = {"frames": after, "before": "the `frames` table above (L1 hit 0.93 / 0.93 / 0.89, L2 hit 0.71 / 0.45 / 0.61, HBM-side 22.7 / 40.9 / 48.0 GB per launch for TPW / HERCULES / FORCES)",
        "reading": "a barrier per outer element (HERCULES, 8 rows per block) / per channel chunk (factored kernel) keeps the waves of a block on the same RF rows at the same time: "
                   "the lines one wave pulls serve its neighbours (L1 hit up), the block asks L2 once (L2 hit up), HBM-side bytes fall 2-8 x"}
except (OSError, ValueError, KeyError):
    pass
out["variants"] = {"base": "as shipped", "lpt": "bands of the plane walk deepest first", "lpt_split2 / lpt_split4 / split4": "channel loop split over 2 / 4 waves of a block (4 x the waves, a quarter of the life each)",
                   "lpt_w64": "one-wave blocks (tile 64 x 1) instead of four-wave blocks (256 x 1)",
                   "shallow_first (lpt2 / lpt2b runs)": "after das_hercules.hip took the deepest-first order: the HERCULES kernel with the old order (base = what ships)",
                   "tile64x4 / tile128x2": "blocks of 64 x 4 / 128 x 2 voxels (lateral x depth) instead of 256 x 1 (its own box: compare with the `base (tile run)` rows)"}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
