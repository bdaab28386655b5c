/* bf_kernels.h -- launch interface between the host executor (C++) and the gfx950 kernels.
 * Plain structs passed by value as kernel arguments: they replace the reference's Vulkan
 * specialization constants ("Bake" structs, generated/beamformer.c:176-249) and push
 * constants (generated/beamformer.c:251-279). */
#ifndef BF_KERNELS_H
#define BF_KERNELS_H

#include <stdint.h>
#include <hip/hip_runtime_api.h>

/* generated/beamformer.c:489-521 */
static const int bf_kind_byte_size[6]     = {2, 4, 4, 8, 2, 4};
static const int bf_kind_element_size[6]  = {2, 2, 4, 4, 2, 2};
static const int bf_kind_element_count[6] = {1, 2, 1, 2, 1, 2};
static const int bf_kind_complex[6]       = {0, 1, 0, 1, 0, 1};
enum { BF_BASE_I16 = 0, BF_BASE_F32 = 1, BF_BASE_F16 = 2 };
static const int bf_kind_base[6] = {BF_BASE_I16, BF_BASE_I16, BF_BASE_F32, BF_BASE_F32, BF_BASE_F16, BF_BASE_F16};

/* per-transmit constants prepared on the host from focal_vectors / orientations
 * (das.glsl:172-202): 32 bytes each, read through wave-uniform (scalar) loads */
typedef struct {
	float    sin_a, cos_a;       /* of the steering angle */
	float    focus_x, focus_z;   /* focal_depth * (sin, cos); unused for plane waves */
	uint32_t flags;              /* BF_TX_* */
	float    pad[3];
} BfTransmit;
enum {
	BF_TX_ROWS    = 1u << 0,   /* transmit orientation == Rows (project on y,z) */
	BF_RX_ROWS    = 1u << 1,   /* receive  orientation == Rows */
	BF_TX_NONE    = 1u << 2,   /* transmit orientation == None: distance 0 */
	BF_TX_PLANE   = 1u << 3,   /* focal depth is +-inf */
	BF_RX_COLUMNS = 1u << 4,   /* receive orientation == Columns (HERCULES test, das.glsl:238) */
};

enum { BF_DAS_RCA = 0, BF_DAS_HERCULES = 1, BF_DAS_FORCES = 2, BF_DAS_READI = 3 };

typedef struct {
	float xdc_transform[16];
	float voxel_transform[16];
	float pitch[2];
	const void       *rf;             /* [channel][transmit][sample], float or float2 */
	void             *out;            /* float or float2 per voxel of the shard */
	const BfTransmit *transmits;      /* [acquisition_count] */
	const int16_t    *sparse_elements;
	const uint16_t   *readi_hadamard; /* binary16 bits, G*G */
	unsigned long long *pair_counter; /* COUNT kernels only */
	uint32_t *tile_counters;          /* das_tile.hip: [0] += (block, channel chunk) pairs that ran out of staged windows, [1] += those that took the gather loop; or null */
	int32_t  family, interpolation, complex_data, coherency_weighting;
	int32_t  acquisition_count, channel_count, sample_count, sparse;
	float    sampling_frequency, inv_sampling_frequency, demodulation_frequency;
	float    inv_speed_of_sound, time_offset, f_number;
	float    speed_of_sound;          /* with inv_speed_of_sound: div_speed_of_sound() of das_common.h */
	float    turns_per_sample;        /* demodulation_frequency / sampling_frequency: IQ phase per sample, in turns */
	float    first_transmit_weight;   /* HERCULES: 1/sqrt(acquisition_count) (das.glsl:272-273) */
	uint32_t size[3];                 /* whole output grid */
	uint32_t z_first, z_count;        /* shard of the grid computed by this launch */
	uint32_t readi_group_count, readi_group;
	uint32_t tile_shift[3];           /* log2 of the block's voxel tile extent per axis (256 voxels, or 64 with a channel split) */
	uint32_t split_shift;             /* log2 K: K waves of a block share 64 voxels, each summing C/K channels */
	uint32_t blocks[3];               /* blocks per axis */
	uint32_t depth_major;             /* tile walk: 1 = z fastest (consecutive tiles share a lateral column), 2 = y fastest (view planes:
	                                     depth lies along voxel y), 3 = view planes in XCD-balanced bands (bf_plane_walk), 0 = x, y, z */
	uint32_t band_rows;               /* depth_major == 3: tile rows per band */
	uint32_t zero_offset;             /* factored kernel: byte offset (from rf) of >= 32 zero bytes the host keeps
	                                     behind the DAS input, the gather target of out-of-range lanes */
	uint32_t tile_window_shift;       /* das_tile.hip: log2 of the staged window length (5 or 6) */
	float    edge_margin;             /* samples: a term whose index comes this close to an end of sample_rf's valid range is decided by the
	                                     shader's own expression, evaluated exactly (das_exact.h); 2^-19 of the largest index magnitude */
	uint32_t row_ends;                /* 1: some in-aperture term of this launch may come within reach of an end of its RF row (a host bound over the
	                                     launch's planes, das_select.cpp; 1 also where no bound is implemented).  The factored and HERCULES kernels
	                                     have an instantiation without any row-end code for launches where it is 0 */
} BfDasArgs;

/* tile geometry of the separable-delay fast path (das_separable.hip) */
typedef struct {
	uint32_t u_axis;          /* output axis (0 = x, 1 = y) the receive aperture runs along */
	uint32_t u_shift, v_shift;/* log2 of the tile extent along the receive / transmit axis */
	uint32_t threads;         /* (1 << u_shift) * (1 << v_shift): 256, 512 or 1024 */
	uint32_t channel_chunk;   /* channels per receive-table rebuild */
	uint32_t lds_bytes;       /* 16 * (channel_chunk << u_shift) + 16 * (transmits << v_shift) */
	uint32_t tiles[3];        /* tiles along u, along v, z planes of the shard */
	uint32_t depth_major;     /* tile walk: 1 = z fastest (consecutive tiles share a lateral column), 0 = x, y, z */
	uint32_t walk_columns;    /* depth_major: columns adjacent along u that are walked together (bf_column_walk): 4, 2 or 1, a divisor of tiles[0] */
	uint32_t window_shift;    /* staged kernel: log2 of the RF window (samples) copied to LDS per transmit */
	uint32_t zero_offset;     /* byte offset (from BfDasArgs.rf) of >= 32 zero bytes the host keeps behind
	                             the DAS input: where out-of-range lanes gather from */
	/* staged kernel (complex, linear), 64 x 16 tiles with x along the receive axis: the transmit delays and phasors of a wave are
	 * uniform and come from a global table (bf_launch_das_staged_tables writes it per frame) through scalar loads */
	uint32_t uniform;         /* 1: use `tables` */
	uint32_t window_samples;  /* 32 or 64 (= 1 << window_shift) */
	uint32_t table_stride;    /* bytes per (lateral tile row, plane) slice: 4 A4 + 16 + 16 (A4 / 4) 48, A4 = transmits rounded up to 4 */
	void    *tables;          /* tiles[1] * tiles[2] slices */
	uint32_t *violations;     /* staged kernels, range-checked loop: incremented once per wave and channel in which some term's window
	                             position fell outside the staged window -- a violated host bound (plan_staged) made loud; may be null */
} BfSeparableArgs;

/* geometry of the HERCULES fast path (das_hercules.hip): lanes of a wave lie along the output's x
 * axis; the array axis that moves with x is the OUTER loop, the other one (a function of the
 * output row y alone) the INNER loop, served from a wave-uniform table */
typedef struct {
	float   *table;           /* [size[1]][table_pitch]: squared lateral distance row y <-> inner element n */
	float   *extremes;        /* [size[1]][2]: min and max of each table row */
	uint32_t table_pitch;     /* floats per row, >= inner_count + 8 (the kernel prefetches a batch ahead) */
	uint32_t inner_count, outer_count;
	uint32_t inner_coord;     /* transducer coordinate of the inner axis: 0 = x, 1 = y */
	uint32_t inner_is_transmit;   /* inner loop walks decoded transmit elements (else receive channels) */
	uint32_t tiles[3];        /* 64-voxel x segments, groups of `rows` output rows, z planes of the shard */
	uint32_t rows;            /* output rows (= waves) of a block: 4, 8 or 16; they walk the outer elements in step (a barrier per element) */
	uint32_t depth_major;     /* tile walk: 1 = z fastest, 2 = y fastest (view planes), 3 = balanced bands (bf_plane_walk), 0 = x, y, z */
	uint32_t band_rows;       /* depth_major == 3: tile rows per band */
	uint32_t zero_offset;     /* byte offset (from BfDasArgs.rf) of >= 32 zero bytes behind the DAS input */
	/* The kernel measures squared distances in units of unit_scale2 (a float within 1e-3 of 1) and converts a
	 * distance to samples with samples_per_unit: the host picks the pair so that samples_per_unit x
	 * sqrt(unit_scale2) equals fs / c to ~1e-11 (plan_hercules), because the rounding of a single fs/c constant
	 * would be a relative bias common to every tap -- 2e-8, i.e. up to 1e-4 rad of demodulation phase on a
	 * coherent peak -- and a per-pair division costs 9 % of the kernel. */
	float    unit_scale2, samples_per_unit;
	void    *pairs;           /* IQ + linear interpolation: scratch for the {sample, difference to the next} copy of the DAS input (16 bytes per
	                             sample + 32 zero bytes), or null: the kernel then reads BfDasArgs.rf as every other kernel does */
	uint32_t phase_local;     /* IQ: the indices of one inner loop stay within ~400 turns of demodulation phase of each other, so the
	                             kernel subtracts one integer per lane and outer element instead of taking v_fract per pair */
} BfHerculesArgs;

typedef struct {
	const void *raw; void *out;
	const int16_t *channel_mapping;   /* device, [channels] */
	uint64_t in_row_bytes, out_row_bytes;
	uint32_t channels;
	int32_t  a1s2, base;              /* A1S2 contrast reduction on BF_BASE_* scalars */
	uint32_t a1s2_scalars;            /* sample_count * element_count */
} BfIngestArgs;

typedef struct {
	const void *left, *right; void *out;
	uint32_t size[3];
	int64_t  in_stride[3], out_stride[3];
	int32_t  in_kind, out_kind, interleave;
} BfReshapeArgs;

typedef struct {
	const void *in; void *out;
	const float *hadamard_t;          /* T*T floats, transposed on the host: HtT[T*i + j] = Ht[T*j + i] */
	const float *hadamard_base;       /* base*base floats B[i*base + j] when HtT = Sylvester(T/base) (x) B, else null */
	uint32_t     hadamard_base_order; /* 1, 12 or 20; 0: no such structure found, dense kernel only */
	uint32_t transmit_count, channel_count, sample_count;
	int64_t  out_stride[3];           /* sample, channel, transmit */
	int32_t  in_kind, out_kind;
} BfDecodeArgs;

typedef struct {
	const void *in; void *out;
	const float *coefficients;
	const float *phasors;             /* demodulate: {cos, -sin} of the window-local phase for index 0..D*64+L-2, or null */
	uint32_t filter_length, decimation, sample_count, batch_sample_count;
	int32_t  complex_filter, demodulate;
	float    sampling_frequency, demodulation_frequency;
	int64_t  in_stride[3], out_stride[3];
	int64_t  in_elements;
	uint32_t channels, transmits;
	int32_t  in_kind, out_kind;
} BfFilterArgs;

/* Depth-major tile walk of VOLUMES (separable gather and LDS-staged kernels).  An XCD has ~64 consecutive tiles of its run in flight (two
 * 1024-thread blocks on each of 32 CUs) and its 4 MiB L2 serves what they read.  With one column walked z fastest those are 64 consecutive
 * depths: the RF window of a (channel, transmit) drifts ~1.5 samples per plane at config 4, 96 samples over the set -- three 32-sample
 * windows apart, so most lines are pulled for one tile.  With g columns ADJACENT ALONG u walked together they are 64 / g depths (g = 4: a
 * 24-sample drift, inside one window) of columns whose transmit windows are the same and whose receive windows neighbour.  Measured on
 * config 4's staged kernel, HBM-side bytes per launch (profiles/r04_traffic.json; kernel time equal within 0.3 %): one column 288 GB,
 * g = 2 123 GB, g = 4 52 GB, g = 8 94 GB, a whole row of 16 columns 181 GB, 4 columns along v 56 GB, 2 x 2 65 GB, plane-major 96 GB.
 * The gather kernel (smaller blocks, more of them in flight, no staging loads) keeps g = 1: 280 GB against 520 with g = 4. */
static inline uint32_t bf_walk_columns(uint32_t tiles_u) { return tiles_u % 4u == 0 ? 4u : (tiles_u % 2u == 0 ? 2u : 1u); }
#ifdef __HIPCC__
static __device__ __forceinline__ void bf_column_walk(uint32_t tile, uint32_t tiles_u, uint32_t tiles_z, uint32_t g, uint32_t &tu, uint32_t &tv, uint32_t &zl)
{
	const uint32_t r = tile / g, col = r / tiles_z, per_row = tiles_u / g;
	zl = r % tiles_z;
	tu = (col % per_row) * g + tile % g;
	tv = col / per_row;
}
#endif

/* Tile walk of VIEW PLANES (depth_major == 3; depth lies along voxel y, one voxel along z -- math.c:844-885).  The work of a
 * tile grows with its depth (the f-number test culls shallow voxels), so neither a run of depth rows per XCD (idle XCDs: the
 * shallow bands finish early) nor a lateral column per XCD (an XCD's tiles in flight span half the RF rows: L2 misses x 3,
 * measured) will do.  The tile rows are cut into bands of `band_rows`; XCD k (= block id mod 8: consecutive ids land on
 * consecutive XCDs) takes bands k, 15 - k, 16 + k, 31 - k, ... -- a shallow band with a deep one -- and walks each band x
 * fastest.  Returns the number of blocks to launch. */
static inline uint32_t bf_plane_walk_blocks(uint32_t blocks_x, uint32_t blocks_y, uint32_t band_rows)
{
	uint32_t bands = (blocks_y + band_rows - 1) / band_rows;
	uint32_t slots = 2u * ((bands + 15u) / 16u);              /* bands per XCD */
	return 8u * slots * band_rows * blocks_x;
}
#ifdef __HIPCC__
template <bool DEEPEST_FIRST = false>
static __device__ __forceinline__ bool bf_plane_walk(uint32_t block_id, uint32_t blocks_x, uint32_t blocks_y, uint32_t band_rows,
                                                     uint32_t &bx, uint32_t &by)
{
	const uint32_t xcd = block_id & 7u, j = block_id >> 3;
	const uint32_t per_band = band_rows * blocks_x;
	uint32_t slot = j / per_band;
	const uint32_t r = j - slot * per_band;
	if constexpr (DEEPEST_FIRST) {
		/* an XCD's bands deepest first: the waves that live longest start first and the launch drains over the short ones.  Measured on
		 * the reference harness's plane (profiles/r04_harness_waits.json, one box): HERCULES 20.7 -> 19.7 ms; the factored kernel's frames
		 * are within +-3 % either way (TPW slower), so only das_hercules.hip asks for it */
		const uint32_t bands = (blocks_y + band_rows - 1) / band_rows;
		slot = 2u * ((bands + 15u) / 16u) - 1u - slot;
	}
	const uint32_t band = 16u * (slot >> 1) + ((slot & 1u) ? 15u - xcd : xcd);
	by = band * band_rows + r / blocks_x;
	bx = r % blocks_x;
	return by < blocks_y;
}
#endif

#ifdef __cplusplus
extern "C" {
#endif
/* every launcher returns the hipError_t of the launch; nothing synchronises */
hipError_t bf_launch_ingest(const BfIngestArgs *a, hipStream_t s);
hipError_t bf_launch_reshape(const BfReshapeArgs *a, hipStream_t s);
hipError_t bf_launch_decode(const BfDecodeArgs *a, hipStream_t s);
hipError_t bf_launch_filter(const BfFilterArgs *a, hipStream_t s);
hipError_t bf_launch_hilbert(const BfFilterArgs *a, hipStream_t s);
hipError_t bf_launch_das(const BfDasArgs *a, hipStream_t s);
hipError_t bf_launch_das_count(const BfDasArgs *a, hipStream_t s);
hipError_t bf_launch_das_separable(const BfDasArgs *a, const BfSeparableArgs *q, hipStream_t s);
hipError_t bf_launch_das_staged(const BfDasArgs *a, const BfSeparableArgs *q, hipStream_t s);
hipError_t bf_launch_das_staged_tables(const BfDasArgs *a, const BfSeparableArgs *q, hipStream_t s);
hipError_t bf_launch_das_staged_real(const BfDasArgs *a, const BfSeparableArgs *q, hipStream_t s);
hipError_t bf_launch_das_staged_cubic(const BfDasArgs *a, const BfSeparableArgs *q, hipStream_t s);
/* |v| (complex) or v (real) min/max over a frame -> out2 (device float[2]); scratch holds
 * 2*1024 floats */
hipError_t bf_launch_das_factored(const BfDasArgs *a, hipStream_t s);
hipError_t bf_launch_das_tile(const BfDasArgs *a, hipStream_t s);          /* das_tile.hip: the factored kernel with block-wide LDS staging (cubic IQ, fine grids) */
hipError_t bf_launch_das_hercules(const BfDasArgs *a, const BfHerculesArgs *q, hipStream_t s);
hipError_t bf_launch_sum(void *out, const void *in, float prescale, uint64_t bytes, hipStream_t s);
hipError_t bf_launch_display(const void *frame, uint64_t voxels, int complex_data, float threshold_db,
                             float gamma, float db_cutoff, float *out, hipStream_t s);
hipError_t bf_launch_min_max(const void *frame, uint64_t voxels, int complex_data,
                             float *scratch, float *out2, hipStream_t s);
/* out (device) = sum over the 8-byte words w[i] of the buffer of w[i] * (i + 1) mod 2^64 */
hipError_t bf_launch_rf_checksum(const void *data, uint64_t bytes, unsigned long long *out, hipStream_t s);
#ifdef __cplusplus
}
#endif
#endif
