import sys
p=sys.argv[1]; s=open(p).read()
old="return tiles_u % 4u == 0 ? 4u : (tiles_u % 2u == 0 ? 2u : 1u);"
assert s.count(old)==1; open(p,'w').write(s.replace(old,"(void)tiles_u; return 1u;"))
