import sys
p=sys.argv[1]; s=open(p).read()
old="const uint64_t split_target = 4096;"
assert s.count(old)==1; open(p,'w').write(s.replace(old,"const uint64_t split_target = 32768;"))
