import sys
p=sys.argv[1]; s=open(p).read()
old="bf_plane_walk<true>("
assert s.count(old)==1; open(p,'w').write(s.replace(old,"bf_plane_walk<false>("))
