import sys
p=sys.argv[1]; s=open(p).read()
old="		for (int c0 = ch_begin; c0 < ch_end; c0 += CH) {\n"
assert s.count(old)==1
open(p,'w').write(s.replace(old, old+"			__syncthreads();       /* experiment: the waves of a block walk the chunks in step (whole blocks inside the grid only) */\n"))
