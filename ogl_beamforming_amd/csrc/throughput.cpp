/* throughput.cpp -- ogl_beamformer_throughput: beamform a ZBP acquisition file on the MI355X
 * backend and report frames per second.
 *
 * Same job and same defaults as the reference's tests/throughput.c (execute_study :410-560):
 * 512 x 1 x 1024 points over lateral [-60, 60] mm x axial [10, 165] mm, f-number 0.5, cubic
 * interpolation, stages {Demodulate (unless the data is already complex), Decode, DAS}, a
 * Kaiser low-pass (L 36, beta 5.65, cutoff = half the emission frequency) for sine emissions
 * or a matched chirp filter for chirp emissions, 1000 ms timeout.  It calls the library
 * through the C ABI only.
 *
 *   ogl_beamformer_throughput [--loop] [--frame n] [--frames N] [--points x y z]
 *                             [--lateral lo hi] [--axial lo hi] [--f-number f] [--devices a,b,...] file.bp
 *
 * --frames N (default 100) bounds the run; --loop keeps going until SIGINT as the
 * reference's does; --devices spreads every frame over several GPUs (beamformer_hip_set_devices:
 * z-slabs of the grid, stitched on pull) with no other change to the calling sequence.
 */
#include "../../include/ogl_beamformer_hip.h"

#include <chrono>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>

static volatile sig_atomic_t g_should_exit;
static void on_signal(int) { g_should_exit = 1; }

static double now_seconds()
{
	return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static int usage(const char *argv0)
{
	std::fprintf(stderr,
	             "usage: %s [--loop] [--frame n] [--frames N] [--points x y z] [--lateral lo hi] [--axial lo hi]\n"
	             "          [--f-number f] [--devices a,b,...] parameters_file.bp\n"
	             "    --loop:     re-upload the data until interrupted\n"
	             "    --frame n:  use frame n of the acquisition (side files <name>_NN.zst)\n"
	             "    --frames N: number of frames to beamform (default 100)\n"
	             "    --devices a,b,...: HIP ordinals to spread each frame over (default: one device)\n", argv0);
	return 2;
}

int main(int argc, char **argv)
{
	int32_t points[3]  = {512, 1, 1024};                 /* tests/throughput.c:20-23 */
	float   axial[2]   = {10e-3f, 165e-3f};
	float   lateral[2] = {-60e-3f, 60e-3f};
	float   f_number   = 0.5f;
	bool     loop = false;
	uint32_t frame_number = 0;
	long     frames = 100;
	const char *path = nullptr;

	for (int i = 1; i < argc; i++) {
		const char *a = argv[i];
		if (!std::strcmp(a, "--loop")) loop = true;
		else if (!std::strcmp(a, "--frame") && i + 1 < argc) frame_number = (uint32_t)std::atoi(argv[++i]);
		else if (!std::strcmp(a, "--frames") && i + 1 < argc) frames = std::atol(argv[++i]);
		else if (!std::strcmp(a, "--points") && i + 3 < argc) { for (int k = 0; k < 3; k++) points[k] = std::atoi(argv[++i]); }
		else if (!std::strcmp(a, "--lateral") && i + 2 < argc) { lateral[0] = (float)std::atof(argv[++i]); lateral[1] = (float)std::atof(argv[++i]); }
		else if (!std::strcmp(a, "--axial") && i + 2 < argc) { axial[0] = (float)std::atof(argv[++i]); axial[1] = (float)std::atof(argv[++i]); }
		else if (!std::strcmp(a, "--f-number") && i + 1 < argc) f_number = (float)std::atof(argv[++i]);
		else if (!std::strcmp(a, "--devices") && i + 1 < argc) {
			int32_t  ids[8]; uint32_t n = 0;
			for (const char *c = argv[++i]; *c && n < 8; ) {
				ids[n++] = (int32_t)std::strtol(c, const_cast<char **>(&c), 10);
				if (*c == ',') c++;
			}
			if (!n || !beamformer_hip_set_devices(ids, n)) { std::fprintf(stderr, "bad --devices list\n"); return 2; }
		}
		else if (a[0] == '-') return usage(argv[0]);
		else path = a;
	}
	if (!path) return usage(argv[0]);

	static BeamformerSimpleParameters bp;
	void *rf = nullptr; uint64_t rf_size = 0;
	if (!beamformer_hip_zbp_load(path, frame_number, &bp, &rf, &rf_size)) {
		std::fprintf(stderr, "failed to load %s: %s\n", path, beamformer_hip_zbp_last_error());
		return 1;
	}

	/* tests/throughput.c:421-436 */
	const float mn[3] = {lateral[0], axial[0], 0}, mx[3] = {lateral[1], axial[1], 0};
	beamformer_hip_host_das_transform(mn, mx, points, bp.das_voxel_transform);
	bp.output_points[0] = points[0]; bp.output_points[1] = points[1]; bp.output_points[2] = points[2];
	bp.output_points[3] = 1;
	bp.f_number           = f_number;
	bp.interpolation_mode = BeamformerInterpolationMode_Cubic;
	bp.decimation_rate    = 1;

	/* :438-444 */
	if (bp.data_kind != BeamformerDataKind_Float32Complex && bp.data_kind != BeamformerDataKind_Int16Complex)
		bp.compute_stages[bp.compute_stages_count++] = BeamformerShaderKind_Demodulate;
	bp.compute_stages[bp.compute_stages_count++] = BeamformerShaderKind_Decode;
	bp.compute_stages[bp.compute_stages_count++] = BeamformerShaderKind_DAS;

	/* :446-476 */
	BeamformerFilterParameters filter{};
	filter.sampling_frequency = bp.sampling_frequency / 2;
	if (bp.emission_parameters.kind == BeamformerEmissionKind_Chirp) {
		filter.kind                        = BeamformerFilterKind_MatchedChirp;
		filter.matched_chirp.duration      = bp.emission_parameters.chirp.duration;
		filter.matched_chirp.min_frequency = bp.emission_parameters.chirp.min_frequency - bp.demodulation_frequency;
		filter.matched_chirp.max_frequency = bp.emission_parameters.chirp.max_frequency - bp.demodulation_frequency;
		filter.complex                     = 1;
	} else {
		filter.kind                    = BeamformerFilterKind_Kaiser;
		filter.kaiser.beta             = 5.65f;
		filter.kaiser.cutoff_frequency = 0.5f * bp.emission_parameters.sine.frequency;
		filter.kaiser.length           = 36;
	}
	bool ok = beamformer_create_filter(&filter, 0, 0);
	bp.compute_stage_parameters[0] = 0;
	ok = ok && beamformer_push_simple_parameters(&bp);
	beamformer_set_global_timeout(1000);
	if (!ok) {
		std::fprintf(stderr, "lib error: %s\n", beamformer_get_last_error_string());
		return 1;
	}

	/* send_frame (:398-408): the pushed size is rows x row length of the raw layout */
	static const uint32_t kind_bytes[6] = {2, 4, 4, 8, 2, 4};
	uint64_t data_size = (uint64_t)bp.raw_data_dimensions[0] * bp.raw_data_dimensions[1] * kind_bytes[bp.data_kind];
	if (data_size > rf_size || data_size > 0xFFFFFFFFull) {
		std::fprintf(stderr, "%s: the RF payload (%llu bytes) is smaller than one raw frame (%llu bytes)\n", path,
		             (unsigned long long)rf_size, (unsigned long long)data_size);
		return 1;
	}

	std::signal(SIGINT, on_signal);
	std::printf("%s: %u channels x %u transmits x %u samples, %s, %d x %d x %d points\n", path, bp.channel_count,
	            bp.acquisition_count, bp.sample_count, beamformer_get_api_version() ? "protocol v33" : "", points[0], points[1], points[2]);

	long   sent = 0, window = 0;
	double start = now_seconds(), window_start = start;
	while (!g_should_exit && (loop || sent < frames)) {
		if (!beamformer_push_data_with_compute(rf, (uint32_t)data_size, BeamformerViewPlaneTag_XZ, 0)) {
			std::fprintf(stderr, "lib error: %s\n", beamformer_get_last_error_string());
			break;
		}
		sent++; window++;
		double t = now_seconds();
		if (t - window_start >= 1.0) {                     /* the reference prints a rolling rate too (:530-550) */
			std::printf("%8.2f frames/s  %8.2f MB/s RF\n", window / (t - window_start),
			            window * (double)data_size / (t - window_start) / 1e6);
			std::fflush(stdout);
			window = 0; window_start = t;
		}
	}
	beamformer_hip_synchronize();
	double elapsed = now_seconds() - start;
	if (sent) {
		float mm[2] = {0, 0};
		beamformer_hip_frame_min_max(mm);
		std::printf("total: %ld frames in %.3f s = %.2f frames/s, %.2f MB/s RF; last frame |v| in [%g, %g]\n", sent, elapsed,
		            sent / elapsed, sent * (double)data_size / elapsed / 1e6, mm[0], mm[1]);
	}
	beamformer_hip_zbp_free(rf);
	beamformer_hip_shutdown();
	return sent > 0 ? 0 : 1;
}
