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
for d in ("ab_tail", "ab_w64"):
    dd = os.path.join(src, d)
    if not os.path.isdir(dd):
        continue
    for f in sorted(os.listdir(dd)):
        if f.endswith(".json"):
            name, kind = f[:-5].rsplit("_", 1)
            try:
                ab.setdefault(name, {})[kind] = json.loads(open(os.path.join(dd, f)).read().strip().splitlines()[-1])["config"]["stage_ms"]["DAS"]
            except Exception:
                pass
out["variants_DAS_ms_one_box_each_row_its_own_run"] = ab
out["variants"] = {"base": "as shipped", "lpt": "bands of the plane walk deepest first", "lpt_split2 / lpt_split4 / split4": "channel loop split over 2 / 4 waves of a block (4 x the waves, a quarter of the life each)",
                   "lpt_w64": "one-wave blocks (tile 64 x 1) instead of four-wave blocks (256 x 1)"}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
