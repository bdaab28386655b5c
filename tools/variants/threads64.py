import sys
p=sys.argv[1]; s=open(p).read()
old="uint32_t threads = a->split_shift ? 64u << a->split_shift : 256u;"
assert s.count(old)==1; open(p,'w').write(s.replace(old,"uint32_t threads = 64u << a->split_shift;"))
