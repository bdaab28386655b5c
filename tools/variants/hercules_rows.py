# HERCULES kernel: R rows (waves) per block instead of 4, walking the outer elements in step.  usage: ROWS=8 python3 hercules_rows.py <das_hercules.hip | das_select.cpp>
import os, sys
p=sys.argv[1]; s=open(p).read(); R=int(os.environ.get("ROWS","8"))
if p.endswith("das_hercules.hip"):
    old="\n	for (int m = 0; m < n_outer; m++) {\n"
    assert s.count(old)==1
    s=s.replace(old, old+"		__syncthreads();\n")
    old="ty_ * 4u + (threadIdx.x >> 6)"; assert s.count(old)==1; s=s.replace(old,f"ty_ * {R}u + (threadIdx.x >> 6)")
    old="hipLaunchKernelGGL((das_hercules_kernel<INTERP, CPLX, CW, PL, PD, true>), dim3(grid), dim3(256)"; assert s.count(old)==1; s=s.replace(old,old.replace("dim3(256)",f"dim3({64*R})"))
    old="hipLaunchKernelGGL((das_hercules_kernel<INTERP, CPLX, CW, PL, PD, false>), dim3(grid), dim3(256)"; assert s.count(old)==1; s=s.replace(old,old.replace("dim3(256)",f"dim3({64*R})"))
    import re
    n=len(re.findall(r"__launch_bounds__\(256\) void das_hercules_kernel", s)); assert n==1, n
    s=s.replace("__launch_bounds__(256) void das_hercules_kernel", f"__launch_bounds__({64*R}) void das_hercules_kernel")
else:
    old="q.tiles[1] = (a.size[1] + 3u) / 4u;"; assert s.count(old)==1; s=s.replace(old,f"q.tiles[1] = (a.size[1] + {R-1}u) / {R}u;")
open(p,'w').write(s)
