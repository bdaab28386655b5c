#!/usr/bin/env python3
"""kernel_resources.py -- register / spill / scratch / LDS figures of every gfx950 kernel in the shipped library, read from the
code objects embedded in the .so (llvm-objcopy --dump-section .hip_fatbin, the clang offload bundles split by hand,
llvm-readelf --notes): what `llvm-readelf --notes` of the build says, without rebuilding anything.

    python3 tools/kernel_resources.py [--json profiles/r03_kernel_resources.json] [library.so]
"""
import argparse
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
          "group_segment_fixed_size", "max_flat_workgroup_size")


def demangle(names):
    try:
        out = subprocess.run([os.path.join(LLVM, "llvm-cxxfilt")], input="\n".join(names), capture_output=True, text=True, check=True).stdout
        return out.strip().split("\n")
    except (OSError, subprocess.CalledProcessError):
        return names


def kernels_of(library):
    """[{name, vgpr_count, ...}] for every kernel of every gfx950 code object in `library`"""
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", library, os.path.join(tmp, "copy.so")], check=True)
        data = open(fat, "rb").read()
        for bi, m in enumerate(re.finditer(re.escape(MAGIC), data)):
            p = m.start()
            off = p + len(MAGIC)
            (count,) = struct.unpack_from("<Q", data, off)
            off += 8
            for _ in range(count):
                o, size, tlen = struct.unpack_from("<QQQ", data, off)
                off += 24
                triple = data[off:off + tlen].decode()
                off += tlen
                if "gfx950" not in triple or size == 0:
                    continue
                co = os.path.join(tmp, f"co_{bi}.o")
                open(co, "wb").write(data[p + o:p + o + size])
                notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
                for block in re.split(r"\n\s+- \.agpr_count:", "\n" + notes)[1:]:
                    block = ".agpr_count:" + block
                    name = re.search(r"\.name:\s+(\S+)", block)
                    if not name:
                        continue
                    entry = {"name": name.group(1)}
                    for f in FIELDS:
                        v = re.search(r"\." + f + r":\s+(\d+)", block)
                        entry[f] = int(v.group(1)) if v else None
                    out.append(entry)
    for e, d in zip(out, demangle([e["name"] for e in out])):
        e["demangled"] = re.sub(r"\(anonymous namespace\)::", "", d)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("library", nargs="?", default=os.path.join(ROOT, "ogl_beamforming_amd", "libogl_beamformer_lib.so"))
    ap.add_argument("--json")
    args = ap.parse_args()
    ks = kernels_of(args.library)
    spilled = [k for k in ks if k["vgpr_spill_count"] or k["private_segment_fixed_size"]]
    summary = {"library": os.path.relpath(args.library, ROOT), "kernels": len(ks),
               "with_vgpr_spills_or_scratch": [k["demangled"] for k in spilled],
               "max_sgpr_spill_count": max(k["sgpr_spill_count"] or 0 for k in ks),
               "table": sorted(ks, key=lambda k: k["demangled"])}
    if args.json:
        with open(args.json, "w") as f:
            json.dump(summary, f, indent=1)
    print(f"{len(ks)} kernels, {len(spilled)} with vector spills or scratch")
    for k in spilled:
        print("  ", k["demangled"][:110], {f: k[f] for f in ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size")})
    return 0


if __name__ == "__main__":
    sys.exit(main())
