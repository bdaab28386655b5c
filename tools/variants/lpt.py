import sys
p=sys.argv[1]; s=open(p).read()
old="	const uint32_t slot = j / per_band, r = j - slot * per_band;\n	const uint32_t band = 16u * (slot >> 1) + ((slot & 1u) ? 15u - xcd : xcd);"
new="	const uint32_t slot_ = j / per_band, r = j - slot_ * per_band;\n	const uint32_t bands_ = (blocks_y + band_rows - 1) / band_rows, slots_ = 2u * ((bands_ + 15u) / 16u), slot = slots_ - 1u - slot_;   /* deepest band first */\n	const uint32_t band = 16u * (slot >> 1) + ((slot & 1u) ? 15u - xcd : xcd);"
assert s.count(old)==1; open(p,'w').write(s.replace(old,new))
