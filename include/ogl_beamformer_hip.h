/* ogl_beamformer_hip.h -- MI355X-specific additions to the C ABI of ogl_beamformer_lib.h.
 *
 * Nothing here exists in the reference library.  These entry points expose what an
 * in-process HIP backend can offer and the reference's shared-memory client cannot:
 * device selection, running on the caller's stream, RF that is already on the device
 * (e.g. landed by an RCCL broadcast over xGMI), sharding the output voxel grid across the
 * GPUs of a node (one process per GPU), device-side access to frames,
 * and per-stage timings taken with HIP events on the compute stream.
 *
 * Same conventions as ogl_beamformer_lib.h: returns 1 on success, 0 on failure with the
 * reason in beamformer_get_last_error().
 *
 * Threading: the library keeps one process-wide state (as the reference client keeps one
 * process-wide connection, lib/ogl_beamformer_lib.c:29-34) and takes no locks; call it from one
 * thread at a time.  Work it enqueues runs asynchronously on HIP streams; the data/compute
 * calls return before the frame is finished and beamformer_get_last_frames /
 * beamformer_hip_synchronize are the synchronisation points.
 */
#ifndef OGL_BEAMFORMER_HIP_H
#define OGL_BEAMFORMER_HIP_H

#include "ogl_beamformer_lib.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Select the HIP device the library owns.  Must precede the first call that touches the
 * device; afterwards it only succeeds for the device already in use.  Default: the value
 * of BEAMFORMER_HIP_DEVICE, else LOCAL_RANK, else 0. */
BEAMFORMER_LIB_EXPORT uint32_t beamformer_hip_set_device(int32_t device_index);
BEAMFORMER_LIB_EXPORT int32_t  beamformer_hip_get_device(void);

/* Several devices behind the same push-RF / pull-image calls (the voxel grid shards,
 * the RF does not).  ONE process, `count` devices (1..8; HIP ordinals; the same ordinal may be listed more
 * than once, which is how a one-GPU box tests the path).  Must precede the first call that touches a
 * device; afterwards it only succeeds for the set already in use (beamformer_hip_shutdown releases it).
 *   - device_indices[0] is the ingest device: beamformer_push_data_with_compute /
 *     beamformer_hip_push_device_data_with_compute land the RF there exactly as with one device;
 *   - the channel-mapped RF is then copied to every other device (hipMemcpyPeerAsync over xGMI, one copy
 *     stream per destination, three RF slots deep so the copies of frame n+1 run beside the kernels of
 *     frame n) and each device runs the whole stage list on its own contiguous z-slab of the block's
 *     grid (of its output shard, if one is set): device i of n takes planes [i P / n, (i + 1) P / n) of
 *     the P planes.  No reduction collective -- voxels are independent (das.glsl:368-407);
 *   - beamformer_get_last_frames returns whole frames: the slabs stitched in z order, each frame rounded
 *     to 64 bytes exactly as one device exports it (lib/ogl_beamformer_lib.c:656-702 semantics);
 *     a slab is bit-identical to the same planes of a one-device frame;
 *   - beamformer_hip_frame_min_max combines the per-slab extremes on the host; the Sum and display
 *     reductions run per slab and are stitched the same way; beamformer_compute_timings reports, per
 *     stage, the slowest device; beamformer_hip_get_last_frame_info describes the ingest device's slab;
 *     beamformer_hip_set_stream is refused (a stream belongs to one device). */
BEAMFORMER_LIB_EXPORT uint32_t beamformer_hip_set_devices(const int32_t *device_indices, uint32_t count);
BEAMFORMER_LIB_EXPORT uint32_t beamformer_hip_get_device_count(void);

/* Run all work on the caller's stream (a hipStream_t; 0 restores the library's own
 * stream).  Lets a host framework order its own device work (an RCCL broadcast of the RF
 * frame, a consumer of the image) against the beamformer without host synchronisation. */
BEAMFORMER_LIB_EXPORT uint32_t beamformer_hip_set_stream(void *hip_stream);

/* Restrict the frames computed from a parameter block to the z-planes
 * [z_first, z_first + z_count) of its output grid; z_count == 0 restores the whole grid.
 * Voxel coordinates are still normalised by the WHOLE grid (das.glsl:374-376), so the planes
 * of a shard are bit-identical to the same planes of an unsharded frame.  Frames of a
 * sharded block hold X*Y*z_count voxels. */
BEAMFORMER_LIB_EXPORT uint32_t beamformer_hip_set_output_shard(uint32_t parameter_slot,
                                                               uint32_t z_first, uint32_t z_count);

/* beamformer_push_data_with_compute() for RF that already resides on the library's device
 * (same layout and size rules).  The frame reads `device_data` on the library's current stream
 * -- in place by its first stage when no channel map / contrast reduction / row padding has to
 * be applied, through a copy into the RF ring otherwise; the caller may overwrite the buffer
 * from work it enqueues later on that stream, or after beamformer_hip_synchronize(). */
BEAMFORMER_LIB_EXPORT uint32_t beamformer_hip_push_device_data_with_compute(const void *device_data, uint32_t size,
                                                                            uint32_t image_plane_tag,
                                                                            uint32_t parameter_slot);

/* Block until every queued frame is complete. */
BEAMFORMER_LIB_EXPORT uint32_t beamformer_hip_synchronize(void);

typedef struct {
	void    *device_pointer;    /* valid until the frame ring wraps over it */
	uint64_t size_bytes;        /* rounded up to 64, as exported by beamformer_get_last_frames */
	uint32_t points[3];         /* x, y, z (z = shard planes) */
	uint32_t data_kind;         /* BeamformerDataKind_Float32 or _Float32Complex */
	uint32_t frame_id;
	uint32_t parameter_block;
} BeamformerHipFrameInfo;
/* The newest frame, in place on the device (no copy, no synchronisation). */
BEAMFORMER_LIB_EXPORT uint32_t beamformer_hip_get_last_frame_info(BeamformerHipFrameInfo *out);

#define BEAMFORMER_HIP_MAX_TIMED_STAGES 24
typedef struct {
	uint32_t stage_count;
	uint32_t stage_kind[BEAMFORMER_HIP_MAX_TIMED_STAGES];  /* BeamformerShaderKind; ingest = 0xFFFF */
	float    stage_ms[BEAMFORMER_HIP_MAX_TIMED_STAGES];    /* hipEvent pairs on the compute stream */
	float    frame_ms;                                     /* first event to last event */
	uint64_t das_pairs;        /* (voxel, channel, transmit) triples passing the apodization
	                              test; counted only when pair counting is enabled */
	uint64_t das_voxels;
	uint32_t das_taps;         /* 1 nearest, 2 linear, 4 cubic */
	uint32_t das_sample_bytes; /* 4 real / 8 complex float32 */
	uint32_t das_path;         /* 0 general kernel, 1 separable-delay gather kernel, 2 LDS-staged kernel, 3 per-voxel factored kernel,
	                              4 HERCULES aligned-grid kernel, 5 factored kernel with block-wide LDS staging (das_tile.hip) */
	uint32_t staged_window_violations;   /* LDS-staged kernels with the STAGED_CHECKED hook: (wave, channel) pairs in which a term's position
	                                        fell outside the staged window -- the host's window bound was wrong.  Must be 0. */
	uint32_t tile_staged_chunks;         /* das path 5 (das_tile.hip): (block, chunk of four channels) pairs whose terms were read from the windows the
	                                        block staged in LDS ... */
	uint32_t tile_gather_chunks;         /* ... and those whose spread did not fit the window: the block ran das_factored.hip's gather loop for them */
	uint32_t das_row_end_planes;         /* z-planes of the frame the ROW-END rule handed to the kernel behind the staged one (LDS-staged -> gather /
	                                        factored; block-staged factored -> factored): planes on which a term can come within reach of an end of
	                                        its RF row, where sample_rf's range test is decided by the shader's own index, evaluated exactly
	                                        (csrc/das_exact.h).  0 on acquisitions whose rows do not end inside the image; das_path then names the
	                                        kernel that took the most planes */
} BeamformerHipFrameTimings;
/* Timings of the newest frame; waits for it to finish. */
BEAMFORMER_LIB_EXPORT uint32_t beamformer_hip_get_last_frame_timings(BeamformerHipFrameTimings *out);

/* The newest frame as ONE of the devices of beamformer_hip_set_devices saw it: its slab's voxels and
 * pairs, its own event times.  (beamformer_hip_get_last_frame_timings reports the ingest device's stage
 * times with the voxel and pair counts of the whole frame and the slowest device's frame time.) */
BEAMFORMER_LIB_EXPORT uint32_t beamformer_hip_get_device_frame_timings(uint32_t device_index, BeamformerHipFrameTimings *out);

/* One device of the set and its share of the newest frame: what a scaling run needs to explain itself.
 * peer_access: 2 = the ingest device (RF arrives from the host / the caller), 1 = direct peer access to the ingest device
 * is enabled (hipDeviceCanAccessPeer said yes and hipDeviceEnablePeerAccess succeeded or was already on: the RF copy crosses
 * one xGMI link), 0 = no peer access (the runtime stages hipMemcpyPeerAsync through host memory: slower, still correct).
 * rf_checksum: sum over the 8-byte words w[i] of the RF this device's newest frame read of w[i] * (i + 1) mod 2^64, computed
 * on the device -- equal on every device of a healthy set. */
typedef struct {
	int32_t  ordinal;             /* HIP device ordinal */
	int32_t  peer_access;
	uint32_t slab_first, slab_count;   /* z planes of the newest frame beamformed here */
	float    peer_copy_ms;        /* the RF copy into this device (0 on the ingest device) */
	float    das_ms, frame_ms;    /* its DAS stage and its whole stage list (HIP events on its compute stream) */
	uint64_t rf_checksum;
	uint64_t rf_bytes;
} BeamformerHipDeviceInfo;
BEAMFORMER_LIB_EXPORT uint32_t beamformer_hip_get_device_info(uint32_t device_index, BeamformerHipDeviceInfo *out);

/* Frame graphs (BASELINE.json configs[4] names a "hipGraph-captured frame"; the reference records one command
 * list per frame, beamformer_core.c:1570-1620).  When enabled, the stage launches of a frame are captured into a
 * hipGraph, the parameter block's instantiated graph is updated in place from the capture (the frame-ring slot and
 * the RF slot move every frame, so kernel arguments do) and launched as one unit.  Frames are bit-identical either
 * way.  Off by default: measured, a replayed frame is never faster than the <= 6 direct launches it replaces and
 * 4 us slower on 15-us frames (profiles/r02_graph_probe.json), and a graph frame times as ONE segment (reported
 * under DAS) because events cannot be recorded inside it.  One device only; pair counting falls back to direct
 * launches.  beamformer_hip_frame_graph_counts reports how many frames were replayed from a graph and how many
 * graphs had to be instantiated (one per plan unless the stage topology changes). */
BEAMFORMER_LIB_EXPORT uint32_t beamformer_hip_enable_frame_graphs(uint32_t enable);
BEAMFORMER_LIB_EXPORT uint32_t beamformer_hip_frame_graph_counts(uint64_t *frames_replayed, uint64_t *graphs_instantiated);

/* When enabled, every frame also runs a geometry-only kernel that counts the triples that
 * pass the apodization test (G in BASELINE.md section 4).  Off by default. */
BEAMFORMER_LIB_EXPORT uint32_t beamformer_hip_enable_pair_counting(uint32_t enable);

/* min and max over the newest frame of |v| (complex) or v (real).  Build-defined: the
 * reference's shaders/min_max.glsl is dead code (beamformer_core.c:632-637). */
BEAMFORMER_LIB_EXPORT uint32_t beamformer_hip_frame_min_max(float out_min_max[2]);

/* Average of the `count` newest frames (all of one size and kind), written to host memory:
 * the reference's Sum stage -- shaders/sum.glsl applied once per frame, oldest first, with
 * prescale 1/count onto a cleared image (beamformer_core.c:1417-1448).  The reference's
 * planner skips Sum in every pipeline (beamformer_core.c:632-637) and so does this
 * library; the stage is offered here so frame averaging (output_points[3]) has a device
 * implementation.  out_size >= the 64-byte-rounded frame size. */
BEAMFORMER_LIB_EXPORT uint32_t beamformer_hip_sum_last_frames(uint32_t count, void *out, uint64_t out_size);

/* Display reduction of the newest frame, the step on the far side of the path: the per-voxel
 * intensity the reference's render shader computes (sample_value, shaders/render_3d.frag.glsl:
 * 50-73; defaults threshold 55 dB, gamma 1, dynamic range 50 dB, ui.c:880-883): |v| clamped to
 * 10^(threshold_db/20), normalised, raised to gamma, and -- when db_cutoff > 0 (log scale) --
 * mapped through a db_cutoff-wide dB window.  Writes one float in [0,1] per voxel (x fastest). */
BEAMFORMER_LIB_EXPORT uint32_t beamformer_hip_display_last_frame(float threshold_db, float gamma, float db_cutoff,
                                                                float *out, uint64_t out_floats);

/* The Hilbert stage.  In the reference BeamformerShaderKind_Hilbert is served by an out-of-tree
 * CUDA routine (cuda_hilbert, beamformer_internal.h:233-261) that the snapshot cannot load:
 * capabilities.hilbert is 0 (beamformer.c:262-263) and a pipeline naming the stage is refused with
 * InvalidComputeStage -- which is also what this library does by default.  Enabling this switch
 * makes the stage available with the library's OWN definition (there is nothing to be identical
 * to: parity unpinned): the analytic signal x + j H{x} along samples by a 63-tap type-III FIR
 * Hilbert transformer (Hamming window), real part = the input delayed by 31 samples, the delay added
 * to the DAS time offset; real input only; as in the reference's planner the stage is dropped when
 * the pipeline also demodulates (beamformer_core.c:567) and makes the pipeline IQ (:589). */
BEAMFORMER_LIB_EXPORT uint32_t beamformer_hip_enable_hilbert(uint32_t enable);

/* Select the DAS implementation: 0 = automatic.  Row-column frames whose receive aperture and transmit
 * steering lie along different array axes (separable delays) run the LDS-staged kernels (das_staged.hip /
 * das_staged_real.hip for linear interpolation of IQ / real samples, das_staged_cubic.hip for cubic interpolation of
 * IQ samples: a tile's delay spread inside its LDS window, at least 6 transmits per channel) or else, for linear
 * interpolation, the gather kernel (das_separable.hip); otherwise the per-voxel factored
 * kernel takes RCA-family and FORCES frames with three or more transmits, the gather kernel the remaining
 * separable ones, the general kernel everything else.  1 = always the general kernel, 2 = as automatic but
 * never the LDS-staged kernel, 3 = the LDS-staged kernel wherever its window bound holds (also below 6
 * transmits), 4 = the factored kernel wherever it applies (also ahead of the gather kernel).
 * HERCULES-family frames whose grid is aligned with the array (one lateral transducer coordinate a
 * function of the output row alone) and at least 32 voxels wide run the aligned-grid kernel
 * (das_hercules.hip) in every mode but 1; 6 = that kernel also on narrow grids (idle lanes).
 * Adding 0x10 keeps the general kernel at one thread per voxel for frames it would otherwise
 * split over channels (frames under ~4096 waves of voxels: K waves share 64 voxels, each
 * sums C/K channels, partial sums meet in LDS); adding 0x20 keeps Decode on the O(T^2) kernel
 * where it would run as a fast Walsh-Hadamard transform.  For parity testing of every path. */
typedef enum {
	BeamformerHipDasPath_Automatic        = 0,
	BeamformerHipDasPath_General          = 1,    /* das.hip for every frame */
	BeamformerHipDasPath_NoLdsStaging     = 2,    /* automatic, but das_separable.hip where das_staged.hip would run */
	BeamformerHipDasPath_PreferLdsStaged  = 3,    /* das_staged.hip wherever its window bound holds */
	BeamformerHipDasPath_PreferFactored   = 4,    /* das_factored.hip wherever the index factorises */
	BeamformerHipDasPath_HerculesAnyWidth = 6,    /* das_hercules.hip also on grids narrower than 32 voxels */
	BeamformerHipDasPath_NoChannelSplit   = 0x10, /* flag: general kernel at one thread per voxel for small frames too */
	BeamformerHipDasPath_DenseDecode      = 0x20, /* flag: Decode on the O(T^2) kernel, not the Walsh-Hadamard form */
	BeamformerHipDasPath_TileStaging      = 0x100,/* flag: das_tile.hip (factored kernel, block-wide LDS staging of cubic polynomials) wherever it is supported --
	                                                 automatic on fine grids only */
	BeamformerHipDasPath_NoTileStaging    = 0x200,/* flag: never */
} BeamformerHipDasPath;
BEAMFORMER_LIB_EXPORT uint32_t beamformer_hip_set_das_path(uint32_t mode);
/* Environment variables the library reads (none is needed in production):
 *   BEAMFORMER_HIP_DEVICE            HIP ordinal of the one-device mode (else LOCAL_RANK, else 0)
 *   BEAMFORMER_HIP_FRAME_RING_BYTES  size of the beamformed-frame ring (default 4 GiB)
 * Diagnostic switches (beamformer_hip_set_hook only -- no environment variable; they select among code paths that ship anyway):
 *   STAGED_SHAPE=u,v,w      the LDS-staged kernels only with 2^u x 2^v voxel tiles and 2^w-sample windows
 *   STAGED_CHECKED          the LDS-staged kernels' range-checked loop for every wave; it also counts window violations
 *                           (BeamformerHipFrameTimings::staged_window_violations)
 *   STAGED_NOUNIFORM        transmit tables in LDS also where the wave-uniform (global table) form applies
 *   STAGED_TABLE_CAP=bytes  largest global transmit table taken (default 2 GiB; 0: always the LDS-table fallback)
 *   DEBUG                   one line per staged plan on stderr */

/* ---- ZBP acquisition files (external/zemp_bp.h; loader tests/throughput.c:135-374) ----
 * Host only, no device needed.  The reference keeps this loader in its throughput harness;
 * it lives in the library here so every binding gets it. */
typedef struct {
	uint32_t major;                   /* header version: 1 or 2 */
	uint32_t data_kind;               /* BeamformerDataKind of the RF payload */
	uint32_t compression_kind;        /* 0 = none, 1 = zstd (ZBP_DataCompressionKind) */
	uint32_t reserved;
	uint64_t offset, size;            /* payload inside the file; size 0: the payload is the side file <name>_NN.zst */
} BeamformerHipZbpPayload;

/* Fill `out` from the bytes of a .bp file exactly as beamformer_simple_parameters_from_zbp_file
 * (tests/throughput.c:150-374) does: geometry, channel map, per-transmit focal data, emission;
 * the caller still chooses the output grid, f-number, interpolation and compute stages.
 * Every offset in the file is bounds checked.  Returns 0 on failure (see ..._zbp_last_error). */
BEAMFORMER_LIB_EXPORT uint32_t beamformer_hip_zbp_parameters(const void *file_bytes, uint64_t file_size,
                                                            BeamformerSimpleParameters *out,
                                                            BeamformerHipZbpPayload *payload);
/* Read <path> (.bp), fill `out`, and return frame `frame_number`'s RF, decompressed, in a
 * malloc'ed buffer (*rf, *rf_size) to be released with beamformer_hip_zbp_free.  zstd payloads
 * need libzstd.so.1 at run time. */
BEAMFORMER_LIB_EXPORT uint32_t beamformer_hip_zbp_load(const char *path, uint32_t frame_number,
                                                      BeamformerSimpleParameters *out, void **rf, uint64_t *rf_size);
BEAMFORMER_LIB_EXPORT void        beamformer_hip_zbp_free(void *rf);
BEAMFORMER_LIB_EXPORT const char *beamformer_hip_zbp_last_error(void);

/* The voxel-grid transform the reference's callers build with das_transform (math.c:831-920):
 * 1-D line, 2-D x-z plane, or 3-D box depending on how many of points[0..2] exceed 1; points
 * are clamped to >= 1 in place.  out16 is column major. */
BEAMFORMER_LIB_EXPORT void beamformer_hip_host_das_transform(const float min_coordinate[3], const float max_coordinate[3],
                                                            int32_t points[3], float out16[16]);

/* ---- host-side introspection (no device needed; used by tests/ to pin the host math
 * against the compiled reference and to check the planner) ---- */

/* Hadamard matrix this library uploads for Decode: order*order floats, row major,
 * Ht[order*j + i] (math.c:35-134).  Returns 0 when no construction exists. */
BEAMFORMER_LIB_EXPORT uint32_t beamformer_hip_host_hadamard(uint32_t order, float *out);

/* Filter taps this library generates for a beamformer_create_filter() request
 * (beamformer_core.c:366-398).  Returns the tap count (pairs when *complex_taps), or -1. */
BEAMFORMER_LIB_EXPORT int32_t beamformer_hip_host_filter(const BeamformerFilterParameters *filter, float *taps,
                                                         uint32_t capacity_floats, float *time_delay,
                                                         uint32_t *complex_taps);

typedef struct {
	int32_t kind;                     /* BeamformerShaderKind */
	int32_t in_kind, out_kind;        /* BeamformerDataKind */
	int64_t in_stride[3], out_stride[3];   /* sample, channel, transmit (elements) */
} BeamformerHipPlanStage;
typedef struct {
	uint32_t stage_count;
	BeamformerHipPlanStage stages[BeamformerMaxComputeShaderStages];
	uint32_t das_samples, iq_pipeline;
	float    das_sampling_frequency, das_time_offset;
	float    das_voxel_transform[16];
} BeamformerHipPlan;
/* The stage list the library would run for a parameter block (plan_compute_pipeline,
 * beamformer_core.c:553-1013, with the whole channel count as the chunk). */
BEAMFORMER_LIB_EXPORT uint32_t beamformer_hip_describe_plan(uint32_t parameter_slot, BeamformerHipPlan *out);

/* Which DAS kernel the frames of a parameter block run under the current das path mode, and why each kernel ahead of it in the
 * selection order was not taken: the library's ONE table of rules (csrc/das_select.cpp), asked instead of restated.  Needs no
 * device.  path: BeamformerHipFrameTimings::das_path numbering; -1: the pipeline has no DAS stage; -2: an acquisition kind or
 * interpolation mode the shader leaves at zero (the frame is cleared, no kernel). */
typedef struct {
	int32_t  path;
	char     kernel[48];            /* "das_rca_staged_kernel", ... */
	char     name[64];              /* "separable-delay LDS-staged kernel", ... */
	char     declined[8][160];      /* by path number: why that kernel does not run ("" for the one that does) */
	uint32_t tile_shift[3], blocks[3], split_shift;      /* per-voxel kernels (general, factored): block shape and count */
	uint32_t tile_walk;             /* 0 x,y,z; 1 z fastest; 2 y fastest; 3 view plane in XCD-balanced bands; staged kernels: + their flag bits */
	uint32_t row_end_planes;        /* planes of the shard the row-end rule hands to the kernel behind the staged one (BeamformerHipFrameTimings::das_row_end_planes) */
	uint32_t tile_window_samples;   /* block-staged factored kernel (path 5): staged window length */
	uint32_t u_axis, u_shift, v_shift, window_samples, uniform_tables, lds_bytes, threads, channel_chunk;   /* separable-delay kernels: the tile is 2^u_shift voxels
	                                   along the receive axis (voxel axis u_axis) by 2^v_shift along the transmit axis, one plane thick */
	uint32_t hercules_prepared_copy;/* HERCULES kernel: reads the {sample, difference} / polynomial copy of the DAS input */
	float    tile_spread_estimate;  /* factored-kernel frames: the host's upper bound of a 1024-voxel tile's delay spread in samples (what decides
	                                   for or against path 5; the kernel measures the real spread per block and chunk); 0 where not computed */
	uint32_t tile_estimate_shift[3];/* ... and the tile it was computed for */
	uint32_t row_ends;              /* 1: some in-aperture term of the launch may come within reach of an end of its RF row (a host bound, per plane, in
	                                   double precision): the kernel's instantiation WITH the exact row-end evaluation runs (csrc/das_exact.h); 0: the one without */
} BeamformerHipDasDescription;
BEAMFORMER_LIB_EXPORT uint32_t beamformer_hip_describe_das(uint32_t parameter_slot, BeamformerHipDasDescription *out);

/* Diagnostic switches (none is needed in production; listed above, csrc/das_select.h says what each does): set and cleared
 * (value NULL or "") through this call only -- the library reads no environment variable for them.  Returns 0 for an unknown name. */
BEAMFORMER_LIB_EXPORT uint32_t beamformer_hip_set_hook(const char *name, const char *value);

/* Free every device resource; the next call re-initialises. */
BEAMFORMER_LIB_EXPORT void beamformer_hip_shutdown(void);

#ifdef __cplusplus
}
#endif
#endif /* OGL_BEAMFORMER_HIP_H */
