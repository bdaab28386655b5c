#!/usr/bin/env python3
"""clock_probe.py -- what the chip does while a bench.py configuration runs (round-3 verdict item 6: config 5's HERCULES kernel holds ~2.05 GHz
against 2.34 GHz in its own instruction-stream probe).  Starts the command as a CHILD process, samples `rocm-smi --json` (clocks, power,
temperature, throttle / performance-level fields: whatever this image's rocm-smi reports) every PERIOD seconds while it runs, and writes the
samples' summary.  This script itself never touches the GPU.

    python3 tools/clock_probe.py --out gpurun_out/r04/clock_cfg5.json -- python3 bench.py --config 5 --steps 6 --warmup 1 --no-cpu-baseline
    python3 tools/clock_probe.py --out gpurun_out/r04/clock_probe.json -- tools/bin/microbench
"""
import argparse
import json
import re
import subprocess
import sys
import time


def sample():
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--showperflevel", "--showuse", "--json"],
                           capture_output=True, text=True, timeout=10)
        d = json.loads(r.stdout)
    except Exception as e:           # rocm-smi missing, refused, or printing something else: say so once
        return {"error": str(e)[:200]}
    card = next(iter(d.values())) if d else {}
    out = {}
    for k, v in card.items():
        m = re.search(r"-?\d+(\.\d+)?", str(v))
        key = k.lower()
        if any(w in key for w in ("sclk", "mclk", "fclk", "socclk", "power", "temperature", "use", "performance", "throttle")):
            if m and "performance" not in key:
                out[k] = float(m.group(0))
            else:
                out[k + " (text)"] = str(v)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--period", type=float, default=0.2)
    ap.add_argument("command", nargs=argparse.REMAINDER)
    args = ap.parse_args()
    cmd = [c for c in args.command if c != "--"]
    idle = sample()
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    rows, t0 = [], time.time()
    while child.poll() is None:
        s = sample()
        s["t"] = round(time.time() - t0, 2)
        rows.append(s)
        time.sleep(args.period)
    stdout = child.stdout.read()
    json.dump({"rows": rows}, open(args.out + ".raw", "w"))          # the samples first: whatever the summary below makes of them
    line = next((l for l in reversed(stdout.splitlines()) if l.startswith("{")), None)
    keys = sorted({k for r in rows for k in r if k != "t" and isinstance(r.get(k), float)})
    # the busy part of the run: samples whose GPU use is at its plateau (or all of them when rocm-smi reports no use figure)
    use_key = next((k for k in keys if "use" in k.lower()), None)
    busy = [r for r in rows if use_key is None or (isinstance(r.get(use_key), float) and r[use_key] >= 90.0)] or rows
    summary = {"command": " ".join(cmd), "samples": len(rows), "busy_samples": len(busy), "idle_before": idle,
               "busy_mean": {k: sum(r[k] for r in busy if isinstance(r.get(k), float)) / max(1, sum(1 for r in busy if isinstance(r.get(k), float))) for k in keys},
               "busy_min": {k: min((r[k] for r in busy if isinstance(r.get(k), float)), default=None) for k in keys},
               "busy_max": {k: max((r[k] for r in busy if isinstance(r.get(k), float)), default=None) for k in keys},
               "text_fields_seen": sorted({f"{k}={v}" for r in rows for k, v in r.items() if isinstance(v, str)})[:40],
               "exit_code": child.returncode}
    if line:
        try:
            d = json.loads(line)
            summary["bench"] = {"ms_per_step": d.get("ms_per_step"), "kernel": d.get("roofline", {}).get("kernel"), "kernel_ms": d.get("roofline", {}).get("kernel_ms"),
                                "binding": {k: d.get("roofline", {}).get("binding", {}).get(k) for k in ("frac", "probe_clock_ghz", "sustained_clock_ghz_est", "frac_at_sustained_clock")}}
        except ValueError:
            pass
    json.dump({"summary": summary, "samples_every": args.period, "rows": rows[:: max(1, len(rows) // 200)]}, open(args.out, "w"), indent=1)
    print(json.dumps(summary)[:2000])
    return child.returncode


if __name__ == "__main__":
    sys.exit(main())
