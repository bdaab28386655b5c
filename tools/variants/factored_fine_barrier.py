# experiment: the waves of a block meet every 32 transmits, not only at the start of a channel chunk
import sys
p=sys.argv[1]; s=open(p).read()
old="			if (!wave_active || c0 >= ch_end) continue;\n"
assert s.count(old)==1
s=s.replace(old,"			auto idle_turns = [&]() { for (int a_ = first_transmit; a_ < A; a_++) if (((a_ - first_transmit) & 31) == 0 && a_ != first_transmit) __syncthreads(); };\n			if (!wave_active || c0 >= ch_end) { idle_turns(); continue; }\n")
old="			if (!__builtin_amdgcn_ballot_w64(any)) continue;          /* wave-uniform */\n"
assert s.count(old)==1
s=s.replace(old,"			if (!__builtin_amdgcn_ballot_w64(any)) { idle_turns(); continue; }          /* wave-uniform */\n")
old="			for (int a = first_transmit; a < A; a++) {\n				float t_index = transmit_index(a);\n"
assert s.count(old)==1
s=s.replace(old,"			for (int a = first_transmit; a < A; a++) {\n				if (((a - first_transmit) & 31) == 0 && a != first_transmit) __syncthreads();\n				float t_index = transmit_index(a);\n")
open(p,'w').write(s)
