/* plan_fuzz.cpp -- AddressSanitizer + UBSan fuzz driver for the host planner (csrc/planner.cpp +
 * csrc/host_math.cpp), CPU only.  Parameter blocks arrive from clients -- through the C ABI in process, through
 * shared memory for ogl_beamformer_server -- so build_plan and the filter generators must accept or refuse ANY
 * block without reading or writing out of bounds, overflowing a signed integer or allocating without bound;
 * the sanitizers abort the process otherwise.  Accepted plans are checked for the invariants the kernels rely
 * on.  Built and run by tests/test_host_logic.py::test_planner_survives_fuzzing. */
#include "../ogl_beamforming_amd/csrc/planner.cpp"
#include "../ogl_beamforming_amd/csrc/host_math.cpp"

#include <cstdint>
#include <cstdio>
#include <cstdlib>

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rng() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
static uint32_t pick(std::initializer_list<uint32_t> v) { return *(v.begin() + rng() % v.size()); }
/* the C ABI's enum-typed fields are written by C clients with any 32-bit value: store the bits as they do */
template <typename T> static void poke(T &field, uint32_t v) { static_assert(sizeof(T) == 4, "32-bit field"); std::memcpy(&field, &v, 4); }
static float fpick()
{
	static const float v[] = {0.f, -0.f, 1.f, -1.f, 0.5f, 1e-30f, 1e30f, 25e6f, 6.25e6f, 1540.f, __builtin_inff(), -__builtin_inff(), __builtin_nanf("")};
	return rng() % 3 ? v[rng() % (sizeof v / sizeof *v)] : (float)((double)(rng() % 2000000) / 1000.0 - 1000.0);
}

int main(int argc, char **argv)
{
	long rounds = argc > 1 ? std::atol(argv[1]) : 20000;
	if (argc > 2) rng_state ^= std::strtoull(argv[2], nullptr, 0) * 0x2545F4914F6CDD1Dull;
	unsigned long accepted = 0, refused = 0;
	for (long r = 0; r < rounds; r++) {
		bf::ParameterBlock pb;
		BeamformerParameters &bp = pb.parameters;
		/* plausible base, then a few wild fields */
		bp.sample_count      = pick({1, 2, 3, 4, 5, 64, 256, 1000, 4096});
		bp.channel_count     = pick({1, 2, 15, 16, 17, 64, 255, 256});
		bp.acquisition_count = pick({1, 2, 3, 12, 16, 20, 24, 31, 128, 256});
		bp.raw_data_dimensions[0] = bp.sample_count * bp.acquisition_count;
		bp.raw_data_dimensions[1] = bp.channel_count;
		bp.decimation_rate   = pick({0, 1, 2, 3, 4, 64});
		poke(bp.interpolation_mode, pick({0, 1, 2}));
		poke(bp.acquisition_kind, pick({0, 1, 2, 3, 4, 5, 6, 7, 8}));
		poke(bp.decode_mode, pick({0, 1}));
		poke(bp.sampling_mode, pick({0, 1}));
		bp.coherency_weighting = pick({0, 1});
		bp.readi_group_count = pick({0, 1, 2, 4, 16});
		bp.readi_group       = pick({0, 1, 3});
		bp.sampling_frequency = 25e6f; bp.demodulation_frequency = 6.25e6f; bp.speed_of_sound = 1540.f; bp.f_number = 1.f;
		for (int i = 0; i < 3; i++) bp.output_points[i] = (int32_t)pick({0, 1, 2, 7, 64, 512});
		bp.output_points[3] = 1;
		for (int i = 0; i < 16; i++) { bp.das_voxel_transform[i] = (i % 5 == 0) ? 1.f : 0.f; bp.xdc_transform[i] = (i % 5 == 0) ? 1.f : 0.f; }
		pb.data_kind = (int32_t)pick({0, 1, 2, 3, 4, 5});
		pb.shader_count = pick({0, 1, 2, 3, 4, 8, 16});
		for (uint32_t i = 0; i < pb.shader_count; i++) {
			pb.shaders[i] = (int32_t)pick({0, 1, 2, 3, 4, 5, 6, 7, 8, 9});
			pb.filter_slots[i] = (uint8_t)pick({0, 1, 3, 15, 255});
		}
		for (auto &f : pb.filters) {
			poke(f.kind, pick({0, 1, 2, 7}));
			f.sampling_frequency = fpick(); f.complex = (int16_t)pick({0, 1});
			if (rng() % 2) { f.kaiser.cutoff_frequency = fpick(); f.kaiser.beta = fpick(); f.kaiser.length = pick({0, 1, 2, 36, 255, 4096, 0x7FFFFFFF}); }
			else           { f.matched_chirp.duration = fpick(); f.matched_chirp.min_frequency = fpick(); f.matched_chirp.max_frequency = fpick(); }
		}
		int wild = (int)(rng() % 4);
		for (int w = 0; w < wild; w++) {
			uint32_t v = pick({0, 1, 0x7FFFFFFF, 0x80000000u, 0xFFFFFFFFu, 0x10000, 65537, 257});
			switch (rng() % 10) {
			case 0: bp.sample_count = v; break;
			case 1: bp.channel_count = v; break;
			case 2: bp.acquisition_count = v; break;
			case 3: bp.decimation_rate = v; break;
			case 4: bp.output_points[rng() % 4] = (int32_t)v; break;
			case 5: poke(bp.interpolation_mode, v); break;
			case 6: bp.readi_group_count = v; break;
			case 7: pb.data_kind = (int32_t)(v % 8); break;          /* the C ABI refuses kinds >= 6 before planning */
			case 8: bp.sampling_frequency = fpick(); bp.time_offset = fpick(); break;
			case 9: bp.demodulation_frequency = fpick(); bp.speed_of_sound = fpick(); break;
			}
		}
		if (pb.data_kind < 0 || pb.data_kind > 5) pb.data_kind = 0;   /* lib_api.cpp validates the kind (lib .c:279-284) */
		bf::Plan plan;
		std::string error;
		if (!bf::build_plan(pb, plan, error, rng() % 2 == 0)) { refused++; if (error.empty()) { std::fprintf(stderr, "refusal without a reason\n"); return 1; } continue; }
		accepted++;
		/* invariants the executor and the kernels rely on */
		bool ok = plan.channels >= 1 && plan.channels <= 256 && plan.acquisitions >= 1 && plan.acquisitions <= 256 && plan.das_samples >= 1;
		ok = ok && plan.stages.size() <= 2 * BeamformerMaxComputeShaderStages + 2;
		ok = ok && (uint64_t)plan.channels * plan.acquisitions * plan.das_samples * (plan.iq_pipeline ? 8 : 4) < (1ull << 32);   /* 32-bit byte offsets in the DAS kernels */
		for (const bf::Stage &st : plan.stages) {
			ok = ok && st.in_kind >= 0 && st.in_kind <= 5 && st.out_kind >= 0 && st.out_kind <= 5;
			if (st.kind == BeamformerShaderKind_DAS) {
				uint32_t support = bp.interpolation_mode == 2 ? 4u : bp.interpolation_mode == 1 ? 2u : 1u;
				ok = ok && plan.das_samples >= support;
			}
			if (st.kind == BeamformerShaderKind_Filter || st.kind == BeamformerShaderKind_Demodulate)
				ok = ok && st.filter.length >= 1 && st.filter.length <= 4096 && (int)st.filter.taps.size() == st.filter.length * (st.filter.complex_taps ? 2 : 1);
		}
		if (!ok) {
			std::fprintf(stderr, "accepted plan violates an invariant (round %ld): S %u C %u A %u D %u kind %d interp %u das_samples %u stages %zu\n", r,
			             bp.sample_count, bp.channel_count, bp.acquisition_count, bp.decimation_rate, pb.data_kind, (unsigned)bp.interpolation_mode,
			             plan.das_samples, plan.stages.size());
			for (const bf::Stage &st : plan.stages) std::fprintf(stderr, "  stage %d kinds %d->%d filter length %d taps %zu complex %d\n", st.kind, st.in_kind, st.out_kind, st.filter.length, st.filter.taps.size(), (int)st.filter.complex_taps);
			return 1;
		}
	}
	std::printf("accepted %lu refused %lu\n", accepted, refused);
	return 0;
}
