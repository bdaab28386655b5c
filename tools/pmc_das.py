#!/usr/bin/env python3
"""pmc_das.py -- rocprofv3 PMC passes over the DAS kernel of one bench.py configuration.

Runs `rocprofv3 --pmc <one counter group> -- python3 bench.py ...` once per group (the guide's
rule: counters in their own runs, the program directly after `--`), sums every counter over the
dispatches of the DAS kernel and writes one JSON summary.  This script itself never touches the
GPU.  Typical use on the GPU box (from the repository root):

    python3 tools/pmc_das.py --config 4 --planes 64 --out gpurun_out/r02/pmc_cfg4

`--planes N` beamforms only N centre z-planes (counters scale with the planes; the ratios are what
matter), `--planes 0` the whole frame.
"""
import argparse
import csv
import glob
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GROUPS = [
    ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_INSTS_VALU_TRANS_F32",
     "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"],
    ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY",
     "SQ_WAIT_ANY", "SQ_INST_CYCLES_VMEM_RD", "SQ_THREAD_CYCLES_VALU"],
    ["SQ_INSTS_SMEM", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_CVT",
     "SQ_INSTS_VALU_INT32", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"],
    ["TA_BUSY_avr", "TA_TA_BUSY_sum", "GRBM_GUI_ACTIVE"],
    ["TA_FLAT_READ_WAVEFRONTS_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_DATA_STALLED_BY_TC_CYCLES_sum"],
    ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCP_TOTAL_ACCESSES_sum", "TCP_TA_TCP_STATE_READ_sum", "GRBM_GUI_ACTIVE"],
    ["TCP_PENDING_STALL_CYCLES_sum", "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum", "TCP_TCP_TA_DATA_STALL_CYCLES_sum",
     "TCP_GATE_EN1_sum"],
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
    ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum"],
    # what the waves wait for (round-3 verdict item 4): parked on s_waitcnt / barriers, issue-stalled, active; level = waves resident
    ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LEVEL_WAVES", "SQ_INSTS_VMEM_RD"],
    ["TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCC_READ_REQ_sum", "TCP_PENDING_STALL_CYCLES_sum", "TCP_TOTAL_CACHE_ACCESSES_sum"],
    ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_REQ_sum", "TCC_MISS_sum"],
]

# the geometry-only pair-count instantiation das_kernel<FAMILY, 0, false, false, true> is not the DAS launch
COUNT_KERNEL = re.compile(r"das_kernel<[^>]*,\s*true>")

KERNEL_SOURCES = ["das.hip", "das_common.h", "das_exact.h", "das_factored.hip", "das_tile.hip", "das_separable.hip", "das_staged.hip", "das_staged_real.hip", "das_staged_cubic.hip", "das_hercules.hip", "bf_kernels.h", "das_select.cpp"]


def kernel_source_hash():
    """sha256 over the DAS kernel sources with comments and whitespace removed: a committed PMC figure stays valid
    through edits that cannot change the code object, and is refused after any that can."""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        p = os.path.join(ROOT, "ogl_beamforming_amd", "csrc", name)
        if os.path.exists(p):
            text = open(p, encoding="utf-8", errors="replace").read()
            text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
            text = re.sub(r"//[^\n]*", " ", text)
            h.update(name.encode())
            h.update(re.sub(r"\s+", " ", text).strip().encode())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=str, default="4")
    ap.add_argument("--planes", type=int, default=0)
    ap.add_argument("--das-path", type=int, default=0)
    ap.add_argument("--out", required=True)
    ap.add_argument("--groups", default="", help="comma separated group indices (default: all)")
    ap.add_argument("--kernel", default="das_", help="substring selecting the kernel's dispatches")
    ap.add_argument("--timeout", type=int, default=400)
    args = ap.parse_args()
    out = os.path.abspath(args.out)
    os.makedirs(out, exist_ok=True)
    bench = ["python3", os.path.join(ROOT, "bench.py"), "--config", str(args.config), "--steps", "1", "--warmup", "0",
             "--no-cpu-baseline", "--das-path", str(args.das_path)]
    if args.planes:
        bench += ["--planes", str(args.planes)]
    pick = [int(v) for v in args.groups.split(",") if v] or range(len(GROUPS))
    env = dict(os.environ, TMPDIR="/tmp")
    counters, dispatches, kernel_names, failed, das_plan, clocks = {}, None, set(), [], None, []
    for gi in pick:
        group = GROUPS[gi]
        d = os.path.join(out, f"g{gi}")
        cmd = ["rocprofv3", "--pmc", *group, "--output-format", "csv", "-d", d, "-o", "r", "--"] + bench
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=args.timeout)
        except subprocess.TimeoutExpired:
            failed.append({"group": group, "why": "timeout"})
            break                                      # a timed-out GPU step: run nothing further
        if das_plan is None:
            for line in r.stdout.decode(errors="replace").splitlines():
                if line.startswith("{") and '"das_plan"' in line:
                    try:
                        das_plan = json.loads(line)["config"]["das_plan"]
                    except (ValueError, KeyError):
                        pass
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            failed.append({"group": group, "why": r.stderr.decode(errors="replace")[-400:]})
            continue
        seen = {}
        already = set(counters)          # a counter that rides along in several groups (GRBM_GUI_ACTIVE) is taken from the first
        for f in files:
            for row in csv.DictReader(open(f)):
                name = row["Kernel_Name"]
                if args.kernel not in name or COUNT_KERNEL.search(name):
                    continue
                m = re.search(r"(das_\w+)(<[^>]*>)?", name)          # "void (anonymous namespace)::das_factored_kernel<0, 2, true, false>(BfDasArgs)"
                kernel_names.add((m.group(1) + (m.group(2) or "")) if m else name[:120])
                key = row["Counter_Name"]
                if key in already:
                    continue
                counters[key] = counters.get(key, 0.0) + float(row["Counter_Value"])
                seen.setdefault(key, set()).add(row["Dispatch_Id"])
                if key == "GRBM_GUI_ACTIVE" and row.get("End_Timestamp"):
                    # shader clock the kernel ran at: busy cycles (summed over the 8 XCDs) over the dispatch's own duration (ns)
                    ns = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                    if ns > 0:
                        clocks.append(float(row["Counter_Value"]) / 8.0 / ns)
        if seen:
            dispatches = max(len(v) for v in seen.values())
    summary = {
        "command": " ".join(bench).replace(ROOT + "/", ""), "kernel_filter": args.kernel, "kernels": sorted(kernel_names),
        "dispatches_summed": dispatches, "counters": counters, "failed_groups": failed, "das_plan": das_plan,
        "kernel_source_sha16": kernel_source_hash(),
        "shader_clock_ghz_per_dispatch": [round(v, 4) for v in clocks],        # (under the profiler: counter collection serialises dispatches)
        "notes": "one rocprofv3 --pmc pass per counter group; values summed over the kernel's dispatches (bench.py runs the "
                 "geometry-only count frame on a separate kernel, excluded, plus one timed frame). SQ_*CYCLES and SQ_ACTIVE/WAIT "
                 "counters are in quad-cycles summed over waves or CUs; FETCH_SIZE/WRITE_SIZE in KiB (FETCH_SIZE under-reports "
                 "wide reads 2x on gfx950, MI355X_MICROARCH.md HBM section).",
    }
    with open(os.path.join(out, "summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print(json.dumps(summary)[:3000])
    return 0


if __name__ == "__main__":
    sys.exit(main())
