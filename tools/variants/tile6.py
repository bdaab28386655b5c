import sys
p=sys.argv[1]; s=open(p).read()
old="out.depth_axis = choose_tile(to_xdc, a.size, zcount, a.tile_shift, a.split_shift ? 6 : 8);"
assert s.count(old)==1; open(p,'w').write(s.replace(old,"out.depth_axis = choose_tile(to_xdc, a.size, zcount, a.tile_shift, 6);"))
