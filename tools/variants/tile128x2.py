import sys
p=sys.argv[1]; s=open(p).read()
old="	out.depth_axis = choose_tile(to_xdc, a.size, zcount, a.tile_shift, a.split_shift ? 6 : 8);\n"
assert s.count(old)==1
open(p,'w').write(s.replace(old, old+"	if (zcount == 1 && a.size[1] >= 4 && a.size[0] >= 128 && !a.split_shift) { a.tile_shift[0] = 7; a.tile_shift[1] = 1; a.tile_shift[2] = 0; }\n"))
