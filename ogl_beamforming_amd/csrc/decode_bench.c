/* decode_bench.c -- ogl_beamformer_decode_bench: the reference's Hadamard decode benchmark
 * (tests/decode.c) for the MI355X backend, in plain C11 against include/ogl_beamformer_lib.h.
 *
 * Same study as the reference's: for each transmit count in {2 ... 256} push a Decode-only
 * pipeline (Hadamard, Int16 RF, 4096 samples per transmit, 256 raw channels with the probe's
 * scrambled channel mapping; tests/decode.c:186-240), warm up, send 32 frames (the size of the
 * stats table) and print the average time per frame; --dump writes the compute stats table of
 * each count as the reference does.
 *
 *   ogl_beamformer_decode_bench [--loop] [--once] [--full-aperture] [--warmup n] [--dump dir]
 */
#include "../../include/ogl_beamformer_hip.h"

#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>

#define RF_TIME_SAMPLES 4096u                       /* tests/decode.c:15 */
#define AVERAGE_SAMPLES 32u                         /* rows of BeamformerComputeStatsTable.times */

static const uint32_t transmit_counts[] = {2, 4, 8, 12, 16, 20, 24, 32, 40, 48, 64, 80, 96, 128, 160, 192, 256};
#define COUNT_OF(a) (sizeof(a) / sizeof(*(a)))

static volatile sig_atomic_t should_exit;
static void on_sigint(int signo) { (void)signo; should_exit = 1; }

static double now_seconds(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* a fixed permutation of 0..255 standing in for the probe's channel wiring: "so that we still
 * get ~random~ access pattern" (tests/decode.c:202) */
static void make_channel_mapping(int16_t map[256])
{
	for (int i = 0; i < 256; i++) map[i] = (int16_t)((i * 167 + 13) & 255);     /* 167 is odd: a bijection mod 256 */
}

typedef struct { int loop, once, dump, full_aperture; unsigned warmup; const char *outdir; } Options;

static uint32_t channels_for(uint32_t transmits, int full_aperture) { return full_aperture ? 256u : transmits; }

static uint32_t data_size_for(uint32_t transmits)
{
	return RF_TIME_SAMPLES * transmits * 256u * (uint32_t)sizeof(int16_t);       /* raw_data_dim, tests/decode.c:158-164 */
}

static int send_parameters(const Options *o, uint32_t transmits)
{
	BeamformerParameters bp;
	memset(&bp, 0, sizeof(bp));
	bp.decode_mode       = BeamformerDecodeMode_Hadamard;
	bp.sample_count      = RF_TIME_SAMPLES;
	bp.channel_count     = channels_for(transmits, o->full_aperture);
	bp.acquisition_count = transmits;
	bp.raw_data_dimensions[0] = RF_TIME_SAMPLES * transmits;
	bp.raw_data_dimensions[1] = 256;
	int16_t mapping[256];
	make_channel_mapping(mapping);
	int32_t stage = BeamformerShaderKind_Decode;
	int ok = beamformer_push_parameters(&bp)
	      && beamformer_push_channel_mapping(mapping, 256)
	      && beamformer_push_pipeline(&stage, 1, BeamformerDataKind_Int16);
	beamformer_set_global_timeout(1000);
	if (!ok) fprintf(stderr, "lib error: %s\n", beamformer_get_last_error_string());
	return ok;
}

static int send_frame(const int16_t *data, uint32_t size)
{
	int ok = beamformer_push_data_with_compute((void *)data, size, BeamformerViewPlaneTag_XZ, 0);
	if (!ok && !should_exit) printf("lib error: %s\n", beamformer_get_last_error_string());
	return ok;
}

static double execute_study(const Options *o, uint32_t transmits, const int16_t *data)
{
	if (!send_parameters(o, transmits)) return -1;
	uint32_t size = data_size_for(transmits);
	for (unsigned i = 0; !should_exit && i < o->warmup; i++) if (!send_frame(data, size)) return -1;
	beamformer_hip_synchronize();
	double start = now_seconds();
	for (unsigned i = 0; !should_exit && i < AVERAGE_SAMPLES; i++) if (!send_frame(data, size)) return -1;
	beamformer_hip_synchronize();                     /* the reference's pushes block on the server; ours are asynchronous */
	return (now_seconds() - start) / AVERAGE_SAMPLES;
}

int main(int argc, char **argv)
{
	Options o;
	memset(&o, 0, sizeof(o));
	for (int i = 1; i < argc; i++) {
		if      (!strcmp(argv[i], "--loop"))          o.loop = 1;
		else if (!strcmp(argv[i], "--once"))          o.once = 1;
		else if (!strcmp(argv[i], "--full-aperture")) o.full_aperture = 1;
		else if (!strcmp(argv[i], "--warmup") && i + 1 < argc) o.warmup = (unsigned)atoi(argv[++i]);
		else if (!strcmp(argv[i], "--dump") && i + 1 < argc) { o.dump = 1; o.outdir = argv[++i]; }
		else {
			fprintf(stderr, "%s [--loop] [--once] [--full-aperture] [--warmup n] [--dump dir]\n", argv[0]);
			return 2;
		}
	}
	if (o.dump) mkdir(o.outdir, 0770);
	signal(SIGINT, on_sigint);

	BeamformerLiveImagingParameters lip;              /* tests/decode.c:300-305 */
	memset(&lip, 0, sizeof(lip));
	lip.active = 1; lip.save_enabled = 1;
	memcpy(lip.save_name_tag, "Decode Bench", 12);
	lip.save_name_tag_length = 12;
	beamformer_set_live_parameters(&lip);

	uint32_t largest = transmit_counts[COUNT_OF(transmit_counts) - 1];
	int16_t *data = (int16_t *)malloc(data_size_for(largest));
	if (!data) { fprintf(stderr, "malloc\n"); return 1; }
	unsigned seed = 12345;                            /* the reference sends uninitialised memory; any values do */
	for (size_t i = 0; i < data_size_for(largest) / sizeof(int16_t); i++) {
		seed = seed * 1664525u + 1013904223u;
		data[i] = (int16_t)((seed >> 16) % 2001u) - 1000;
	}

	int failures = 0;
	if (o.loop) {
		while (!should_exit) {
			double t = execute_study(&o, transmit_counts[0], data);
			if (t < 0) { failures++; break; }
			if (!should_exit) printf("decode %3u | %uF Average: %8.3f [ms]\n", transmit_counts[0], AVERAGE_SAMPLES, t * 1e3);
		}
	} else if (o.once) {
		failures += !(send_parameters(&o, transmit_counts[0]) && send_frame(data, data_size_for(transmit_counts[0])));
		beamformer_hip_synchronize();
	} else {
		for (size_t i = 0; !should_exit && i < COUNT_OF(transmit_counts); i++) {
			uint32_t transmits = transmit_counts[i];
			double t = execute_study(&o, transmits, data);
			if (t < 0) { failures++; continue; }
			BeamformerComputeStatsTable stats;
			memset(&stats, 0, sizeof(stats));
			beamformer_compute_timings(&stats, 1000);
			double kernel = 0;                        /* mean Decode kernel time over the table */
			for (uint32_t s = 0; s < stats.shader_count; s++)
				if (stats.shader_ids[s] == BeamformerShaderKind_Decode)
					for (int f = 0; f < 32; f++) kernel += stats.times[f][s] / 32.0;
			if (o.dump) {
				char path[1024];
				snprintf(path, sizeof(path), "%s/decode_%u.bin", o.outdir, transmits);
				FILE *f = fopen(path, "wb");
				if (f) { fwrite(&stats, sizeof(stats), 1, f); fclose(f); }
			}
			printf("decode %3u | %uF Average: %8.3f [ms]   (Decode kernel %8.3f ms, %u channels)\n", transmits, AVERAGE_SAMPLES,
			       t * 1e3, kernel * 1e3, channels_for(transmits, o.full_aperture));
			fflush(stdout);
		}
	}
	lip.active = 0;
	beamformer_set_live_parameters(&lip);
	free(data);
	beamformer_hip_shutdown();
	return failures ? 1 : 0;
}
