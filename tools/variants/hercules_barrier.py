import sys
p=sys.argv[1]; s=open(p).read()
old="\n	for (int m = 0; m < n_outer; m++) {\n"
assert s.count(old)==1
open(p,'w').write(s.replace(old, old+"		__syncthreads();          /* the four waves of a block (four adjacent rows) walk the outer elements in step */\n"))
