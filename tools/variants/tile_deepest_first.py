import sys
p=sys.argv[1]; s=open(p).read()
old="if (!bf_plane_walk(bid, p.blocks[0], p.blocks[1], p.band_rows, bx, by)) return;"
assert s.count(old)==1; open(p,'w').write(s.replace(old,"if (!bf_plane_walk<true>(bid, p.blocks[0], p.blocks[1], p.band_rows, bx, by)) return;"))
