#!/usr/bin/env python3
"""summarize_profiles.py -- turn tools/pmc_das.py summaries into the tracked evidence files:

  profiles/das_traffic.json     HBM-side bytes per DAS launch of the whole metric frames (FETCH_SIZE x 2 on
                                gfx950 + WRITE_SIZE), with the hash of the kernel sources they were taken from
                                (bench.py prints `traffic: null` when the sources have changed since)
  profiles/r02_das_bound.json   per shipping DAS kernel and configuration: VALU busy fraction, TA busy fraction,
                                L2 hit rate, VALU instructions / transcendentals / VALU-busy cycles per gather
                                instruction -- the numbers DESIGN.md's "what bounds it" paragraphs quote

usage: summarize_profiles.py --round r02 <pmc summary.json> [...]     (run from the repository root)
A summary whose command has no --planes is a whole frame and feeds das_traffic.json as well."""
import argparse
import hashlib
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# as bench.py: the kernel's own file plus the headers every DAS kernel includes, comments and whitespace removed
KERNEL_FILES = {"das_kernel": "das.hip", "das_rca_separable_kernel": "das_separable.hip", "das_rca_staged_kernel": "das_staged.hip",
                "das_rca_staged_real_kernel": "das_staged_real.hip", "das_rca_staged_cubic_kernel": "das_staged_cubic.hip", "das_factored_kernel": "das_factored.hip", "das_tile_kernel": "das_tile.hip", "das_hercules_kernel": "das_hercules.hip"}
COMMON_SOURCES = ["das_common.h", "das_exact.h", "bf_kernels.h", "das_select.cpp"]


def kernel_source_hash(kernel):
    h = hashlib.sha256()
    for name in [KERNEL_FILES.get(kernel, "das.hip")] + COMMON_SOURCES:
        p = os.path.join(ROOT, "ogl_beamforming_amd", "csrc", name)
        if os.path.exists(p):
            text = open(p, encoding="utf-8", errors="replace").read()
            text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
            text = re.sub(r"//[^\n]*", " ", text)
            h.update(name.encode())
            h.update(re.sub(r"\s+", " ", text).strip().encode())
    return h.hexdigest()[:16]
SIMDS = 256 * 4
# compulsory HBM traffic of the DAS launch: its input RF once + the volume once (SURVEY 8d shapes)
COMPULSORY = {2: 128 * 31 * 2048 * 8 + 1024 * 1024 * 8, 3: 32 * 32 * 2048 * 4 + 256 ** 3 * 4,
              4: 256 * 75 * 2048 * 8 + 512 ** 3 * 8, 5: 256 * 128 * 1024 * 8 + 512 ** 3 * 8}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="r02")
    ap.add_argument("--merge", action="store_true", help="start from the existing files: only the (configuration, kernel) entries of the given summaries are replaced")
    ap.add_argument("summaries", nargs="+")
    args = ap.parse_args()
    traffic_path = os.path.join(ROOT, "profiles", "das_traffic.json")
    bound_path = os.path.join(ROOT, "profiles", f"{args.round}_das_bound.json")
    traffic = {"_comment": "HBM-side bytes per DAS launch (whole metric frame), rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes "
                           "(tools/pmc_das.py), KiB; FETCH_SIZE doubled per /opt/skills/guides/MI355X_MICROARCH.md (gfx950 reports half the bytes of "
                           "wide reads; Infinity-Cache hits are counted, so this is an upper bound on DRAM traffic), WRITE_SIZE exact."}
    bound = {"_comment": "PMC evidence per shipping DAS kernel (tools/pmc_das.py, one counter group per rocprofv3 pass).  SQ_ACTIVE_INST_VALU is in "
                         "quad-cycles summed over SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs; TA_BUSY_avr is the mean over TA instances."}
    replaced = set()
    if args.merge:
        for target, path in ((traffic, traffic_path), (bound, bound_path)):
            try:
                for k, v in json.load(open(path)).items():
                    if not k.startswith("_"):
                        target[k] = v
            except (OSError, ValueError):
                pass
    for path in args.summaries:
        s = json.load(open(path))
        c = s["counters"]
        m = re.search(r"--config (\S+)", s["command"])
        config = m.group(1) if m else "4"
        config = int(config) if config.isdigit() else config
        planes = re.search(r"--planes (\d+)", s["command"])
        kernel = s["kernels"][0].split("<")[0] if s["kernels"] else "?"
        n = s["dispatches_summed"] or 1
        key = f"config{config}" if isinstance(config, int) else config
        if "GRBM_GUI_ACTIVE" in c:
            cycles = c["GRBM_GUI_ACTIVE"] / 8.0                       # shader-clock cycles the launches were resident, all dispatches
            e = {"command": s["command"], "kernel_source_sha16": kernel_source_hash(kernel), "tree_sha16_at_run": s["kernel_source_sha16"], "dispatches_summed": n,
                 "planes": int(planes.group(1)) if planes else "whole frame", "kernel_cycles_per_launch": cycles / n, "das_plan": s.get("das_plan")}
            if "SQ_ACTIVE_INST_VALU" in c:
                e["valu_busy_frac"] = c["SQ_ACTIVE_INST_VALU"] * 4.0 / (cycles * SIMDS)
            if "TA_BUSY_avr" in c:
                e["ta_busy_frac"] = c["TA_BUSY_avr"] / cycles
            if "SQ_LDS_IDX_ACTIVE" in c:
                e["lds_idx_active_frac"] = c["SQ_LDS_IDX_ACTIVE"] / (cycles * 256)          # cycles the LDS index unit was busy, per CU
                e["lds_bank_conflict_frac"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(1.0, c["SQ_LDS_IDX_ACTIVE"])
            if c.get("SQ_INSTS_VMEM_RD"):
                g = c["SQ_INSTS_VMEM_RD"]
                e["per_gather_instruction"] = {
                    "valu_instructions": c.get("SQ_INSTS_VALU", 0) / g, "transcendentals": c.get("SQ_INSTS_VALU_TRANS_F32", 0) / g,
                    "salu_instructions": c.get("SQ_INSTS_SALU", 0) / g, "lds_instructions": c.get("SQ_INSTS_LDS", 0) / g,
                    "valu_busy_cycles_per_simd": c.get("SQ_ACTIVE_INST_VALU", 0) * 4.0 / g,
                    "elapsed_cycles_per_simd": cycles * SIMDS / g, "elapsed_cycles_per_cu": cycles * 256 / g}
            if "SQ_WAVE_CYCLES" in c:
                e["mean_resident_waves_per_cu"] = c["SQ_WAVE_CYCLES"] * 4.0 / (cycles * 256)
            if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
                # what a resident wave does with its cycles (disjoint, MI355X_MICROARCH.md "rocprofv3 PMC slots"): parked on s_waitcnt or a
                # barrier, issue-stalled (a dependency or a busy pipe), or issuing
                w = c["SQ_WAVE_CYCLES"]
                e["wave_cycles"] = {"parked_on_waitcnt_or_barrier": c["SQ_WAIT_ANY"] / w, "issue_stalled": c.get("SQ_WAIT_INST_ANY", 0.0) / w,
                                    "issuing": c.get("SQ_ACTIVE_INST_ANY", 0.0) / w}
            if c.get("TCP_TCC_READ_REQ_LATENCY_sum") and c.get("TCP_TCC_READ_REQ_sum"):
                e["l1_miss_latency_cycles"] = c["TCP_TCC_READ_REQ_LATENCY_sum"] / c["TCP_TCC_READ_REQ_sum"]
            if c.get("TCP_PENDING_STALL_CYCLES_sum") is not None and "GRBM_GUI_ACTIVE" in c:
                e["tcp_pending_stall_frac"] = c["TCP_PENDING_STALL_CYCLES_sum"] / (cycles * 256)
            if "TCC_HIT_sum" in c:
                e["l2_hit_rate"] = c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
            if "TCP_TOTAL_CACHE_ACCESSES_sum" in c and "TCP_TCC_READ_REQ_sum" in c:
                e["l1_hit_rate"] = 1.0 - c["TCP_TCC_READ_REQ_sum"] / max(1.0, c["TCP_TOTAL_CACHE_ACCESSES_sum"])
            e["counters"] = c
            if args.merge and (key, kernel) not in replaced:
                bound.setdefault(key, {}).pop(kernel, None)             # first summary of this kernel in this call replaces the old entry
                replaced.add((key, kernel))
            have = bound.setdefault(key, {}).get(kernel)
            if have is None:
                bound[key][kernel] = e
            else:
                # a second run of the same kernel (e.g. the TA/TCP groups taken on a slab): derived fractions the first
                # run lacks are added, raw counters are kept apart under the run's own label
                for k in ("valu_busy_frac", "ta_busy_frac", "l1_hit_rate", "l2_hit_rate", "per_gather_instruction", "mean_resident_waves_per_cu",
                          "lds_idx_active_frac", "lds_bank_conflict_frac", "wave_cycles", "l1_miss_latency_cycles", "tcp_pending_stall_frac"):
                    if k in e and k not in have:
                        have[k] = e[k]
                        have.setdefault("also_from", {})[k] = s["command"]
                have.setdefault("other_runs", []).append({"command": s["command"], "counters": c})
        if not planes and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            hbm = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 / n
            traffic.setdefault(key, {})[kernel] = {
                "fetch_size_kib": c["FETCH_SIZE"] / n, "write_size_kib": c["WRITE_SIZE"] / n, "hbm_bytes_per_launch": hbm,
                "compulsory_bytes": COMPULSORY.get(config), "ratio_to_compulsory": hbm / COMPULSORY[config] if config in COMPULSORY else None,
                "l2_hit_rate": (c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"])) if "TCC_HIT_sum" in c else None,
                "launches_averaged": n, "round": args.round, "command": s["command"], "kernel_source_sha16": kernel_source_hash(kernel), "tree_sha16_at_run": s["kernel_source_sha16"],
                "das_plan": s.get("das_plan")}
    json.dump(traffic, open(traffic_path, "w"), indent=1)
    json.dump(bound, open(bound_path, "w"), indent=1)
    print("wrote", traffic_path, bound_path)


if __name__ == "__main__":
    main()
