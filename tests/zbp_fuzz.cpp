/* zbp_fuzz.cpp -- AddressSanitizer fuzz driver for the ZBP loader (csrc/zbp.cpp), CPU only.
 * Reads seed files given on the command line, then parses N random mutations of each (byte
 * flips, 32-bit field overwrites with boundary values, truncations).  The loader must reject
 * or accept every one of them without touching memory outside the buffer; ASan aborts the
 * process otherwise.  Built and run by tests/test_zbp.py::test_loader_survives_fuzzing. */
#include "../ogl_beamforming_amd/csrc/zbp.cpp"

#include <cstdint>
#include <cstdio>
#include <vector>

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rng() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

int main(int argc, char **argv)
{
	long rounds = argc > 1 ? std::atol(argv[1]) : 1000;
	static const uint32_t interesting[] = {0, 1, 2, 3, 4, 0x7F, 0xFF, 0x100, 0x101, 0xFFFF, 0x10000, 0x7FFFFFFF, 0x80000000u, 0xFFFFFFFFu, 0xFFFFFFFEu};
	unsigned long accepted = 0, rejected = 0;
	for (int f = 2; f < argc; f++) {
		std::vector<uint8_t> seed;
		if (!read_file(argv[f], seed)) { std::fprintf(stderr, "cannot read %s\n", argv[f]); return 2; }
		for (long r = 0; r < rounds; r++) {
			/* exact-size heap copy so that ASan sees any read past the end */
			size_t size = seed.size();
			if (rng() % 4 == 0) size = rng() % (seed.size() + 1);
			uint8_t *buf = (uint8_t *)std::malloc(size ? size : 1);
			std::memcpy(buf, seed.data(), size);
			int edits = 1 + (int)(rng() % 4);
			for (int e = 0; e < edits && size >= 4; e++) {
				size_t limit = size - 3 < 200 || rng() % 3 == 0 ? size - 3 : 200;      /* mostly the header fields */
				size_t at = (rng() % limit) & ~(size_t)3;
				uint32_t v = rng() % 2 ? interesting[rng() % (sizeof(interesting) / sizeof(*interesting))]
				                       : (uint32_t)(rng() % (2 * size + 16));
				if (at < 12 && rng() % 8) continue;                                    /* keep magic/version mostly intact */
				std::memcpy(buf + at, &v, 4);
			}
			BeamformerSimpleParameters bp;
			BeamformerHipZbpPayload payload;
			if (beamformer_hip_zbp_parameters(buf, size, &bp, &payload)) {
				accepted++;
				/* what a caller would do next with an accepted payload must stay inside the file */
				if (payload.size && (payload.offset > size || payload.size > size - payload.offset)) {
					std::fprintf(stderr, "accepted payload outside the file\n");
					return 1;
				}
			} else rejected++;
			std::free(buf);
		}
	}
	std::printf("accepted %lu rejected %lu\n", accepted, rejected);
	return 0;
}
