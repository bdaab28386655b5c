#!/bin/bash
OUT=$PWD/gpurun_out/r04/mb_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/pmc -o mb -- /root/repo/tools/bin/microbench > $OUT/microbench.json 2> $OUT/err
cd /root/repo
python3 - <<'PY'
import csv, glob, re
rows = {}
for f in glob.glob("gpurun_out/r04/mb_pmc/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gather_probe" not in r["Kernel_Name"]: continue
        rows.setdefault((int(r["Dispatch_Id"]), re.search(r"gather_probe<[^>]*>", r["Kernel_Name"]).group(0), r["Grid_Size"], r["Workgroup_Size"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
for k in sorted(rows):
    c = rows[k]
    if "SQ_INSTS_VMEM_RD" in c and c["SQ_INSTS_VMEM_RD"]:
        print(k, "accesses/instr %.2f" % (c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / c["SQ_INSTS_VMEM_RD"]), "clk/instr/cu %.2f" % (c["GRBM_GUI_ACTIVE"] / 8 * 256 / c["SQ_INSTS_VMEM_RD"]),
              "accesses/clk/cu %.3f" % (c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / (c["GRBM_GUI_ACTIVE"] / 8 * 256)), "L2 req/instr %.2f" % (c.get("TCP_TCC_READ_REQ_sum", 0) / c["SQ_INSTS_VMEM_RD"]))
PY
