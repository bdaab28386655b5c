/* executor.cpp -- device side of the in-process beamformer: RF ingest, per-frame stage
 * launches, frame ring, export, timings.
 *
 * Replaces the reference's two worker loops and their Vulkan plumbing:
 *   beamformer_rf_upload        (beamformer_core.c:1756-1805)  -> push_rf_and_compute: pinned
 *                                                                slot + H2D on a copy stream
 *                                                                + the ingest kernel
 *   complete_queue / Compute    (beamformer_core.c:1519-1677)  -> run_frame: stage launches in
 *                                                                stream order, one pass over
 *                                                                all channels
 *   complete_queue / Export     (beamformer_core.c:1468-1509)  -> export_last_frames
 *   beamformer_frame_next       (beamformer_core.c:440-466)    -> next_frame
 *   gpu_command_timestamp + coalesce_timing_table
 *                               (beamformer_core.c:1611-1655, :1683-1747) -> HIP event pairs
 * There is no CPU fallback: without a HIP device every entry point fails with
 * BeamformerLibErrorKind_SharedMemory.
 */
#include "context.h"
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <thread>

namespace bf {

static Context g_context;
Context &ctx() { return g_context; }

bool set_error(BeamformerLibErrorKind kind)
{
	g_context.last_error = kind;
	return false;
}

#define HIP_OK(expr) ((expr) == hipSuccess)

bool DeviceBuffer::ensure(size_t bytes)
{
	if (bytes <= size && ptr) return true;
	if (ptr) { (void)hipFree(ptr); ptr = nullptr; size = 0; }
	if (bytes == 0) bytes = 64;
	if (!HIP_OK(hipMalloc(&ptr, bytes))) { ptr = nullptr; return false; }
	size = bytes;
	return true;
}

void DeviceBuffer::release()
{
	if (ptr) (void)hipFree(ptr);
	ptr = nullptr; size = 0;
}

static uint64_t round_up(uint64_t v, uint64_t m) { return (v + m - 1) / m * m; }

/* beamformer.c:196-228 picks the first of {4, 2, 1.5, 1} GiB that fits half the device heap;
 * on a 288 GB MI355X that is always 4 GiB, so the default needs no device query.
 * BEAMFORMER_HIP_FRAME_RING_BYTES overrides it (before first use). */
uint64_t default_frame_ring_bytes()
{
	if (const char *e = std::getenv("BEAMFORMER_HIP_FRAME_RING_BYTES")) {
		unsigned long long v = std::strtoull(e, nullptr, 0);
		if (v >= (1ull << 20)) return round_up(v, 64);
	}
	return 4ull << 30;
}

static bool init_one_device(Context &c, Device &d, int ordinal, uint32_t index)
{
	if (!HIP_OK(hipSetDevice(ordinal))) return false;
	if (!HIP_OK(hipStreamCreateWithFlags(&d.own_stream, hipStreamNonBlocking))) return false;
	if (!d.stream) d.stream = d.own_stream;
	if (!HIP_OK(hipStreamCreateWithFlags(&d.copy_stream, hipStreamNonBlocking))) return false;
	if (!HIP_OK(hipStreamCreateWithFlags(&d.peer_stream, hipStreamNonBlocking))) return false;
	for (uint32_t k = 0; k < BeamformerMaxRawDataFramesInFlight; k++) {
		if (!HIP_OK(hipEventCreateWithFlags(&d.rf_landed[k], hipEventDisableTiming))) return false;
		if (!HIP_OK(hipEventCreateWithFlags(&d.rf_consumed[k], hipEventDisableTiming))) return false;
		d.consumed_pending[k] = false;
	}
	if (!d.ring.ensure(c.frame_ring_bytes)) return false;
	d.frames.assign(BeamformerMaxBacklogFrames, FrameRecord{});
	d.device = ordinal; d.index = index;
	return true;
}

/* Makes `dev` the device the executor functions act on. */
static bool select_device(uint32_t dev)
{
	Context &c = g_context;
	c.cur = &c.devices[dev];
	return HIP_OK(hipSetDevice(c.cur->device));
}

bool ensure_device()
{
	Context &c = g_context;
	if (c.device_ready) {
		c.cur = &c.devices[0];
		if (!HIP_OK(hipSetDevice(c.cur->device))) return set_error(BeamformerLibErrorKind_SharedMemory);
		return true;
	}
	int count = 0;
	if (!HIP_OK(hipGetDeviceCount(&count)) || count <= 0) return set_error(BeamformerLibErrorKind_SharedMemory);
	int      ordinals[kMaxDevices];
	uint32_t n = c.requested_count;
	if (n == 0) {
		const char *e = std::getenv("BEAMFORMER_HIP_DEVICE");
		if (!e) e = std::getenv("LOCAL_RANK");
		ordinals[0] = e ? std::atoi(e) : 0;
		n = 1;
	} else {
		for (uint32_t i = 0; i < n; i++) ordinals[i] = c.requested_devices[i];
	}
	for (uint32_t i = 0; i < n; i++)
		if (ordinals[i] < 0 || ordinals[i] >= count) return set_error(BeamformerLibErrorKind_SharedMemory);
	if (!c.frame_ring_bytes) c.frame_ring_bytes = default_frame_ring_bytes();
	bool ok = true;
	for (uint32_t i = 0; i < n && ok; i++) ok = init_one_device(c, c.devices[i], ordinals[i], i);
	/* peers copy RF slabs from the ingest device: let every device reach it directly over xGMI
	 * (errors here only mean "already enabled" or "same device": the copies work either way) */
	for (uint32_t i = 1; i < n && ok; i++) {
		if (c.devices[i].device == c.devices[0].device) continue;
		(void)hipSetDevice(c.devices[i].device);
		(void)hipDeviceEnablePeerAccess(c.devices[0].device, 0);
		(void)hipSetDevice(c.devices[0].device);
		(void)hipDeviceEnablePeerAccess(c.devices[i].device, 0);
		(void)hipGetLastError();
	}
	c.device_count = n;
	c.cur = &c.devices[0];
	c.device_ready = true;                      /* so that shutdown_device releases a partial set-up */
	if (!ok || !HIP_OK(hipSetDevice(c.devices[0].device))) { shutdown_device(); return set_error(BeamformerLibErrorKind_SharedMemory); }
	return true;
}

static void release_one_device(Device &d)
{
	if (d.device < 0) return;
	(void)hipSetDevice(d.device);
	(void)hipDeviceSynchronize();
	for (auto &p : d.plans) {
		p.hadamard_t.release(); p.hadamard_base.release(); p.readi_hadamard.release(); p.transmits.release();
		p.sparse.release(); p.mapping.release();
		for (auto &t : p.taps) t.release();
		p.taps.clear(); p.valid = false;
	}
	for (auto &b : d.raw_staging) b.release();
	for (auto &u : d.upload) {
		if (u.pinned) (void)hipHostFree(u.pinned);
		if (u.copied) (void)hipEventDestroy(u.copied);
		if (u.consumed) (void)hipEventDestroy(u.consumed);
		u = UploadSlot{};
	}
	if (d.copy_stream) (void)hipStreamDestroy(d.copy_stream);
	if (d.peer_stream) (void)hipStreamDestroy(d.peer_stream);
	d.copy_stream = d.peer_stream = nullptr;
	for (uint32_t k = 0; k < BeamformerMaxRawDataFramesInFlight; k++) {
		if (d.rf_landed[k])   (void)hipEventDestroy(d.rf_landed[k]);
		if (d.rf_consumed[k]) (void)hipEventDestroy(d.rf_consumed[k]);
		d.rf_landed[k] = d.rf_consumed[k] = nullptr; d.consumed_pending[k] = false;
	}
	for (auto &b : d.rf) b.release();
	for (auto &b : d.scratch) b.release();
	d.ring.release(); d.pair_counter.release(); d.minmax_scratch.release(); d.sum_scratch.release();
	d.hercules_table.release(); d.hercules_pairs.release(); d.staged_tables.release();
	for (auto &g : d.frame_exec) { if (g) (void)hipGraphExecDestroy(g); g = nullptr; }
	for (auto &g : d.graph_generation) g = 0;
	for (auto &t : d.timing) {
		if (t.created) for (auto &e : t.events) if (e) (void)hipEventDestroy(e);
		t = TimingSlot{};
	}
	if (d.own_stream) (void)hipStreamDestroy(d.own_stream);
	if (d.stream == d.own_stream || d.index != 0) d.stream = nullptr;   /* a caller's stream on the first device stays selected */
	d.own_stream = nullptr;
	d.frames.clear();
	d.ring_next_offset = 0; d.frame_counter = 0; d.rf_index = 0;
	d.have_sample = false; d.last_sampled_frame = 0; d.last_sampled_block = 0; d.replan_frame = 0;
	d.device = -1;
}

void shutdown_device()
{
	Context &c = g_context;
	if (!c.device_ready) return;
	for (uint32_t i = 0; i < kMaxDevices; i++) release_one_device(c.devices[i]);
	c.device_ready = false; c.device_count = 1; c.cur = &c.devices[0];
	for (auto &b : c.blocks) b.dirty |= Dirty_Parameters;   /* plans are rebuilt on next use */
}

static bool upload(DeviceBuffer &dst, const void *src, size_t bytes, hipStream_t s)
{
	if (!dst.ensure(bytes ? bytes : 64)) return false;
	if (!bytes) return true;
	return HIP_OK(hipMemcpyAsync(dst.ptr, src, bytes, hipMemcpyHostToDevice, s));
}

static uint16_t half_bits_pm1(float v) { return v < 0 ? 0xBC00 : 0x3C00; }   /* +-1 as binary16 */

/* beamformer_commit_parameter_block (beamformer_core.c:1191-1287): replan when the block
 * changed and refresh the device-side tables. */
static PlanState *commit_block(uint32_t block)
{
	Context &c = g_context;
	Device  &d = *c.cur;
	ParameterBlock &pb = c.blocks[block];
	PlanState &ps = d.plans[block];
	if (ps.valid && !pb.dirty) return &ps;
	/* with several devices every one of them replans: the change reaches them through push_multi,
	 * which commits the ingest device LAST -- only that commit clears the dirty bits */
	const bool clears_dirty = c.device_count == 1 || d.index == 0;

	std::string error;
	Plan plan;
	if (!build_plan(pb, plan, error, c.hilbert_enabled)) { ps.valid = false; ps.error = error; return nullptr; }
	ps.plan = std::move(plan);
	const BeamformerParameters &bp = pb.parameters;
	hipStream_t s = d.stream;

	/* a table a kernel of an earlier frame may still be reading must not be overwritten
	 * under it: replanning is rare, so simply drain the stream first */
	(void)hipStreamSynchronize(s);

	bool ok = true;
	ok &= upload(ps.mapping, pb.channel_mapping, sizeof(pb.channel_mapping), s);
	ok &= upload(ps.sparse,  pb.sparse_elements, sizeof(pb.sparse_elements), s);

	/* per-transmit constants (das.glsl:172-202) */
	uint32_t A = bp.acquisition_count;
	ps.transmit_table.assign(A, BfTransmit{});
	for (uint32_t a = 0; a < A; a++) {
		uint32_t txrx  = bp.single_orientation ? (bp.transmit_receive_orientation & 0xFFu) : pb.transmit_receive_orientations[a];
		float    angle = bp.single_focus ? bp.focal_vector[0] : pb.focal_vectors[a][0];
		float    depth = bp.single_focus ? bp.focal_vector[1] : pb.focal_vectors[a][1];
		uint32_t tx = (txrx >> 4) & 0xF, rx = txrx & 0xF;
		BfTransmit &t = ps.transmit_table[a];
		float rad = angle * 0.017453292519943295f;               /* GLSL radians() */
		t.sin_a = sinf(rad); t.cos_a = cosf(rad);
		t.flags = 0;
		if (tx == BeamformerRCAOrientation_None)    t.flags |= BF_TX_NONE;
		if (tx == BeamformerRCAOrientation_Rows)    t.flags |= BF_TX_ROWS;
		if (rx == BeamformerRCAOrientation_Rows)    t.flags |= BF_RX_ROWS;
		if (rx == BeamformerRCAOrientation_Columns) t.flags |= BF_RX_COLUMNS;
		if (std::isinf(depth)) { t.flags |= BF_TX_PLANE; t.focus_x = t.focus_z = 0; }
		else                   { t.focus_x = depth * t.sin_a; t.focus_z = depth * t.cos_a; }
	}
	ok &= upload(ps.transmits, ps.transmit_table.data(), sizeof(BfTransmit) * A, s);

	if (!ps.plan.hadamard_t.empty())
		ok &= upload(ps.hadamard_t, ps.plan.hadamard_t.data(), sizeof(float) * ps.plan.hadamard_t.size(), s);
	if (!ps.plan.hadamard_base.empty())
		ok &= upload(ps.hadamard_base, ps.plan.hadamard_base.data(), sizeof(float) * ps.plan.hadamard_base.size(), s);
	ps.readi_bits.clear();
	for (float v : ps.plan.readi_hadamard) ps.readi_bits.push_back(half_bits_pm1(v));
	if (!ps.readi_bits.empty())
		ok &= upload(ps.readi_hadamard, ps.readi_bits.data(), sizeof(uint16_t) * ps.readi_bits.size(), s);

	for (auto &t : ps.taps) t.release();
	ps.tap_tables.clear();
	ps.taps.assign(ps.plan.stages.size(), DeviceBuffer{});
	for (size_t i = 0; i < ps.plan.stages.size(); i++) {
		const Stage &st = ps.plan.stages[i];
		if (st.kind == BeamformerShaderKind_Filter || st.kind == BeamformerShaderKind_Demodulate ||
		    st.kind == BeamformerShaderKind_Hilbert) {
			/* taps, then -- for Demodulate -- the window-local phasors {cos, -sin}(2 pi fd index / (fs/2))
			 * of filter.glsl:99-107, in the kernel's own f32 expression */
			ps.tap_tables.emplace_back(st.filter.taps);
			std::vector<float> &table = ps.tap_tables.back();
			if (st.kind == BeamformerShaderKind_Demodulate) {
				const uint32_t window = ps.plan.decimation * 64 + (uint32_t)st.filter.length - 1;
				const float fd = bp.demodulation_frequency, fs = bp.sampling_frequency / 2;
				for (uint32_t index = 0; index < window; index++) {
					float arg = 6.28318530717958647692f * fd * (float)index / fs;
					table.push_back(cosf(arg));
					table.push_back(-sinf(arg));
				}
			}
			ok &= upload(ps.taps[i], table.data(), sizeof(float) * table.size(), s);
		}
	}
	if (ps.plan.intermediate_bytes) {
		ok &= d.scratch[0].ensure(ps.plan.intermediate_bytes + 64);
		ok &= d.scratch[1].ensure(ps.plan.intermediate_bytes + 64);
	}
	/* host vectors above must outlive the async copies out of pageable memory */
	ok &= HIP_OK(hipStreamSynchronize(s));
	if (!ok) { ps.valid = false; ps.error = "device allocation or upload failed"; return nullptr; }
	if (clears_dirty) pb.dirty = 0;
	ps.valid = true;
	ps.generation++;
	return &ps;
}

/* beamformer_frame_next (beamformer_core.c:440-466) */
static FrameRecord *next_frame(const uint32_t points[3], bool complex_frame, uint32_t block)
{
	Context &c = g_context;
	Device  &d = *c.cur;
	int kind = complex_frame ? BeamformerDataKind_Float32Complex : BeamformerDataKind_Float32;
	uint64_t bytes = round_up((uint64_t)points[0] * points[1] * points[2] * (uint64_t)bf_kind_byte_size[kind], 64);
	if (bytes > d.ring.size) return nullptr;
	if (d.ring_next_offset > d.ring.size - bytes) d.ring_next_offset = 0;
	uint64_t id = d.frame_counter++;
	FrameRecord *f = &d.frames[id % d.frames.size()];
	/* records whose storage this frame reuses stop being exportable */
	for (FrameRecord &old : d.frames)
		if (old.bytes && old.offset < d.ring_next_offset + bytes && d.ring_next_offset < old.offset + old.bytes) old.bytes = 0;
	f->offset = d.ring_next_offset; f->bytes = bytes;
	f->points[0] = points[0]; f->points[1] = points[1]; f->points[2] = points[2];
	f->data_kind = kind; f->id = (uint32_t)id; f->block = block;
	d.ring_next_offset += bytes;
	return f;
}

static uint32_t ceil_log2(uint32_t v) { uint32_t s = 0; while ((1u << s) < v) s++; return s; }

/* Shape of the 2^tile_log2-voxel block of the DAS launch.  The axis along which the transducer-space
 * depth changes fastest gets extent 1: sample indices move ~2 samples per voxel along depth
 * but only a fraction of a sample per voxel laterally, so a depth-flat tile keeps the 64 lanes
 * of a wave within a few cache lines of every (channel, transmit) row. */
static int choose_tile(const float *voxel_to_xdc, const uint32_t size[3], uint32_t zcount, uint32_t shift[3], uint32_t tile_log2 = 8)
{
	uint32_t extent[3] = {size[0], size[1], zcount};
	uint32_t full[3]   = {size[0], size[1], size[2]};
	int depth = -1; float best = -1;
	for (int i = 0; i < 3; i++) {
		if (extent[i] <= 1) continue;
		float step = std::fabs(voxel_to_xdc[4 * i + 2]) / (float)(full[i] > 1 ? full[i] - 1 : 1);
		if (step > best) { best = step; depth = i; }
	}
	uint32_t cap[3], left = tile_log2;
	for (int i = 0; i < 3; i++) { cap[i] = ceil_log2(extent[i]); shift[i] = 0; }
	int lateral[2], nl = 0;
	for (int i = 0; i < 3; i++) if (i != depth && extent[i] > 1) lateral[nl++] = i;
	uint32_t first = nl == 2 ? tile_log2 / 2 : tile_log2;
	for (int k = 0; k < nl; k++) {
		uint32_t give = cap[lateral[k]] < first ? cap[lateral[k]] : first;
		if (give > left) give = left;
		shift[lateral[k]] = give; left -= give;
	}
	for (int k = 0; k < nl && left; k++) {
		uint32_t room = cap[lateral[k]] - shift[lateral[k]];
		uint32_t give = room < left ? room : left;
		shift[lateral[k]] += give; left -= give;
	}
	if (depth >= 0 && left) {
		uint32_t give = cap[depth] < left ? cap[depth] : left;
		shift[depth] = give; left -= give;
	}
	shift[0] += left;   /* fewer voxels in total than the tile: idle lanes */
	return depth;
}

/* tile walk of the kernels that deal tiles to the XCDs in contiguous runs: the depth axis runs fastest, so that a run is a
 * lateral column at every depth (neighbouring RF windows AND the same work on every XCD: the f-number test culls shallow
 * voxels).  Volumes: depth = voxel z (1); the reference's view planes (math.c:844-885) put it on voxel y (2). */
static uint32_t tile_walk(int depth_axis, uint32_t zcount, uint32_t tile_rows, uint32_t &band_rows)
{
	band_rows = 1;
	const char *walk = std::getenv("BEAMFORMER_HIP_TILE_WALK");     /* "plane": x -> y -> z, "column": y fastest on view planes (measurement aids) */
	if (walk && walk[0] == 'p') return 0u;
	if (depth_axis != 1) return 1u;
	if (zcount != 1 || (walk && walk[0] == 'c')) return 2u;
	/* view plane: ~32 bands, four per XCD (bf_plane_walk) */
	band_rows = tile_rows / 32u ? tile_rows / 32u : 1u;
	return 3u;
}

/* Samples of delay one voxel step along x (the lane axis of the per-voxel kernels) can move a sample index: the physical
 * length of the step times fs / c.  >= 1: a COARSE grid -- neighbouring lanes read different samples of an RF row. */
static float lane_step_samples(const float *voxel_to_xdc, const BfDasArgs &a)
{
	const float n = (float)(a.size[0] > 1 ? a.size[0] - 1 : 1);
	const float dx = voxel_to_xdc[0] / n, dy = voxel_to_xdc[1] / n, dz = voxel_to_xdc[2] / n;
	return std::sqrt(dx * dx + dy * dy + dz * dz) * a.sampling_frequency * a.inv_speed_of_sound;
}

/* Can this RCA frame use the separable-delay fast path (das_separable.hip)?  Needs one
 * receive and one transmit orientation for all transmits, on different transducer axes, a
 * volume whose z axis alone carries depth, and voxel x / y axes that each move only one of
 * the two lateral coordinates -- every coefficient that must vanish has to be an exact
 * zero product, so that the tables reproduce the general kernel's per-voxel arithmetic. */
static bool plan_separable(const BfDasArgs &a, const std::vector<BfTransmit> &tx, const float *xdc, const float *vox,
                           uint32_t zcount, BfSeparableArgs &q)
{
	if (a.family != BF_DAS_RCA || tx.empty()) return false;
	const uint32_t orient = BF_TX_ROWS | BF_RX_ROWS | BF_TX_NONE;
	for (const BfTransmit &t : tx) if ((t.flags & orient) != (tx[0].flags & orient)) return false;
	const bool tx_none = (tx[0].flags & BF_TX_NONE) != 0;
	const int  r = (tx[0].flags & BF_RX_ROWS) ? 1 : 0;      /* transducer coordinate the receive aperture uses */
	const int  w = (tx[0].flags & BF_TX_ROWS) ? 1 : 0;      /* world coordinate the transmit uses */
	auto W = [&](int row, int col) { return vox[4 * col + row]; };
	auto X = [&](int row, int col) { return xdc[4 * col + row]; };
	/* transducer coordinate `row` must not move with voxel axis `col` */
	auto xdc_fixed = [&](int row, int col) {
		for (int k = 0; k < 3; k++) if (X(row, k) != 0.f && W(k, col) != 0.f) return false;
		return true;
	};
	if (a.size[0] < 2 || a.size[1] < 2) return false;
	for (int col = 0; col < 2; col++) {
		if (!xdc_fixed(2, col)) return false;                 /* transducer depth: voxel z only */
		if (!tx_none && W(2, col) != 0.f) return false;       /* world depth: voxel z only */
	}
	int u_axis = -1;
	for (int u = 0; u < 2 && u_axis < 0; u++) {
		int v = 1 - u;
		if (!xdc_fixed(r, v)) continue;                       /* receive lateral: not along v */
		if (!tx_none && W(w, u) != 0.f) continue;             /* transmit lateral: not along u */
		u_axis = u;
	}
	if (u_axis < 0) return false;

	/* Tile (U along the receive axis, V along the transmit axis), block size and the number of
	 * channels per receive-table chunk: maximise resident waves per CU (LDS: 160 KB per CU, 32
	 * waves per CU), then prefer big chunks (fewer rebuilds) and square-ish tiles. */
	const uint32_t C = (uint32_t)a.channel_count, A = (uint32_t)a.acquisition_count;
	const uint32_t lds_cu = 160u * 1024u;
	uint32_t best_waves = 0, best_score = 0;
	for (uint32_t threads_shift = 10; threads_shift >= 8; threads_shift--) {
		for (uint32_t us = 2; us + 2 <= threads_shift; us++) {
			uint32_t vs = threads_shift - us;
			if ((u_axis == 0 ? us : vs) < 4) continue;        /* >= 16 lanes of a wave along x */
			for (uint32_t chunk = 16; chunk <= 256; chunk *= 2) {
				uint32_t cc = chunk < C ? chunk : C;
				uint64_t lds = 16ull * (((uint64_t)cc << us) + ((uint64_t)A << vs));
				if (lds > lds_cu) continue;
				uint32_t blocks = (uint32_t)(lds_cu / lds);
				uint32_t by_waves = 2048u >> threads_shift;
				if (blocks > by_waves) blocks = by_waves;
				uint32_t waves = blocks << (threads_shift - 6);
				uint32_t balance = us > vs ? us - vs : vs - us;
				uint32_t score = (cc << 4) + (16 - balance);
				if (waves > best_waves || (waves == best_waves && score > best_score)) {
					best_waves = waves; best_score = score;
					q.u_shift = us; q.v_shift = vs; q.threads = 1u << threads_shift;
					q.channel_chunk = cc; q.lds_bytes = (uint32_t)lds;
				}
				if (cc == C) break;
			}
		}
	}
	if (!best_waves) return false;
	q.u_axis = (uint32_t)u_axis;
	{
		/* BEAMFORMER_HIP_TILE_WALK=plane restores the x -> y -> z walk (measurement aid) */
		const char *walk = std::getenv("BEAMFORMER_HIP_TILE_WALK");
		q.depth_major = !(walk && walk[0] == 'p');
	}
	const uint32_t best_u = q.u_shift, best_v = q.v_shift;
	uint32_t nu = a.size[u_axis], nv = a.size[1 - u_axis];
	q.tiles[0] = (nu + (1u << best_u) - 1) >> best_u;
	q.tiles[1] = (nv + (1u << best_v) - 1) >> best_v;
	q.tiles[2] = zcount;
	return true;
}

/* Upgrade a separable plan to the LDS-staged kernel (das_staged.hip) when the delay spread of a
 * tile provably fits the staging window.  The receive delay is a distance, so it changes by at
 * most one lateral voxel step (in samples) per voxel along u; the transmit delay likewise along
 * v, scaled by max|sin(angle)| when every transmit is a plane wave. */
static bool plan_staged(const BfDasArgs &a, const std::vector<BfTransmit> &tx, const float *xdc, const float *vox,
                        uint32_t zcount, BfSeparableArgs &q, bool allow_uniform = true)
{
	const bool cplx = a.complex_data != 0;                       /* das_staged.hip / das_staged_real.hip */
	const bool cubic = a.interpolation == 2;                     /* das_staged_cubic.hip: complex samples only */
	if (a.interpolation != 1 && !(cubic && cplx)) return false;
	const uint32_t C = (uint32_t)a.channel_count, A = (uint32_t)a.acquisition_count;
	/* the kernels stage through 32-bit buffer offsets and park their padding loads at 2^31 */
	if ((uint64_t)C * A * (uint64_t)a.sample_count * (cplx ? 8u : 4u) >= (1ull << 31)) return false;
	const uint32_t A4 = (A + 3u) & ~3u;                          /* the kernel pads the transmit table to whole batches of 4 */
	const int u_axis = (int)q.u_axis, v_axis = 1 - u_axis;
	const int r = (tx[0].flags & BF_RX_ROWS) ? 1 : 0, w = (tx[0].flags & BF_TX_ROWS) ? 1 : 0;
	float m[16];
	m4_mul(xdc, vox, m);
	const float samples_per_metre = a.sampling_frequency * a.inv_speed_of_sound;
	float step_u = std::fabs(m[4 * u_axis + r]) / (float)(a.size[u_axis] > 1 ? a.size[u_axis] - 1 : 1) * samples_per_metre;
	float step_v = std::fabs(vox[4 * v_axis + w]) / (float)(a.size[v_axis] > 1 ? a.size[v_axis] - 1 : 1) * samples_per_metre;
	bool all_plane = true; float max_sin = 0.f;
	for (const BfTransmit &t : tx) {
		all_plane &= (t.flags & BF_TX_PLANE) != 0;
		max_sin = std::fmax(max_sin, std::fabs(t.sin_a));
	}
	if (tx[0].flags & BF_TX_NONE) step_v = 0.f;
	else if (all_plane)           step_v *= max_sin;

	const uint32_t lds_cu = 160u * 1024u;
	uint32_t best_waves = 0, best_score = 0;
	BfSeparableArgs best = q;
	for (uint32_t threads_shift = 10; threads_shift >= 9; threads_shift--) {
		for (uint32_t vs = 4; vs <= 6; vs++) {
			if (vs + 4 > threads_shift) continue;
			uint32_t us = threads_shift - vs;
			if (us > 6) continue;
			if ((u_axis == 0 ? us : vs) < 4) continue;
			float spread = step_u * (float)((1u << us) - 1) + step_v * (float)((1u << vs) - 1);
			if (!(spread >= 0.f && spread <= 60.f)) continue;                /* also a NaN / infinite spread (wild parameters) */
			uint32_t need = (uint32_t)std::ceil(spread * 1.001f) + (cubic ? 6 : 4);   /* + taps (k - 1 .. k + 2 for cubic), floors, rounding slack */
			uint32_t ws = need <= 32 ? 5 : need <= 64 ? 6 : 0;
			if (!ws) continue;
			/* BEAMFORMER_HIP_STAGED_SHAPE="us,vs,ws": only this tile / window shape (testing every template instance;
			 * a window larger than needed is legal, a smaller one is not taken; ws = 48: the uniform variant's 48-sample window) */
			bool force_w48 = false;
			if (const char *force = std::getenv("BEAMFORMER_HIP_STAGED_SHAPE")) {
				unsigned fu = 0, fv = 0, fw = 0;
				if (std::sscanf(force, "%u,%u,%u", &fu, &fv, &fw) == 3) {
					if (fu != us || fv != vs) continue;
					if (fw == 48) { if (need > 48) continue; force_w48 = true; }
					else { if (fw < ws || fw > 6) continue; ws = fw; }
				}
			}
			/* complex samples, linear interpolation, x along the receive axis and a 64 x 16 tile: a wave's lanes share one row of the
			 * transmit axis, the transmit tables leave the LDS for a global table read through scalar loads (das_staged.hip, UNI).
			 * Measured faster than every other shape, so it is preferred wherever its window fits -- with a
			 * 48-sample window where 32 samples are too few (64-sample windows of 64 and more transmits leave no room for two
			 * blocks per CU): 63 elements per wave and pass, at most 4 passes of the 16 waves */
			const bool uniform = allow_uniform && cplx && !cubic && u_axis == 0 && threads_shift == 10 && us == 6 && vs == 4 &&
			                     !std::getenv("BEAMFORMER_HIP_STAGED_NOUNIFORM");
			uint32_t window = 1u << ws;
			/* (the 48-sample window is opt-in -- BEAMFORMER_HIP_STAGED_W48, or the shape hook: at config 4 it measured 799.0 ms against
			 * 804.5 ms for the 32 x 32 tiles with the tables in LDS, but 726 GB of HBM-side traffic per launch against 176 GB) */
			const bool want_w48 = force_w48 || (std::getenv("BEAMFORMER_HIP_STAGED_W48") && need > 32 && need <= 48);
			if (uniform && want_w48 && (A4 * 48u + 62u) / 63u <= 64u) window = 48;
			else if (force_w48) continue;
			/* window elements a thread stages per channel: 4 (complex: registers), 8 (real).  The linear kernels also rest their
			 * tap address on it -- one 16-bit shift of the element index: 4096 x 16 B and 8192 x 8 B both end at 64 KB */
			if (window != 48 && ((uint64_t)A4 << ws) > ((uint64_t)(cplx ? 4 : 8) << threads_shift)) continue;
			const uint64_t stage_elements = (uint64_t)A4 * window;
			for (uint32_t chunk = 8; chunk <= 64; chunk *= 2) {
				uint32_t cc = chunk < C ? chunk : C;
				/* transmit tables 12 B per (transmit, v), receive table, {sample, difference} windows + a zero element, floors */
				uint64_t lds = cubic ? 12ull * ((uint64_t)A4 << vs) + 16ull * ((uint64_t)cc << us) + 32ull * (stage_elements + 3) + 4ull * (A4 + cc + 1) + 128
				             : cplx ? (uniform ? 0ull : 12ull * ((uint64_t)A4 << vs)) + 16ull * ((uint64_t)cc << us) + 16ull * (stage_elements + 3) + 4ull * (A4 + cc + 1) + 128
				                    :  4ull * ((uint64_t)A4 << vs) +  8ull * ((uint64_t)cc << us) +  8ull * (stage_elements + 4) + 4ull * (A4 + cc + 1) + 128;
				lds = (lds + 15) & ~15ull;
				if (lds > lds_cu) continue;
				uint32_t blocks = (uint32_t)(lds_cu / lds), by_waves = (cubic ? 1024u : 2048u) >> threads_shift;   /* (cubic: 128 VGPRs per lane) */
				if (blocks > by_waves) blocks = by_waves;
				uint32_t waves = blocks << (threads_shift - 6);
				uint32_t balance = us > vs ? us - vs : vs - us;
				uint32_t score = (blocks >= 2 ? 1000u : 0u) + (cc << 2) + (8 - balance) + (window == 32 ? 500u : 0u) +
				                 (uniform && window == 32 ? 2000u : uniform && window == 48 ? 1500u : 0u);
				if (waves > best_waves || (waves == best_waves && score > best_score)) {
					best_waves = waves; best_score = score;
					best.u_shift = us; best.v_shift = vs; best.threads = 1u << threads_shift;
					best.channel_chunk = cc; best.lds_bytes = (uint32_t)lds; best.window_shift = ws; best.window_samples = window;
					best.uniform = uniform ? 1u : 0u;
					best.table_stride = uniform ? 4u * A4 + 16u + 16u * (A4 / 4u) * 48u : 0u;
				}
				if (cc == C) break;
			}
		}
	}
	if (std::getenv("BEAMFORMER_HIP_STAGED_CHECKED")) best.depth_major |= 2u;
	/* uniform variant: the two blocks of a CU are neighbours along u in one plane (shared rows of the global transmit table);
	 * BEAMFORMER_HIP_STAGED_WALK=column keeps the plain depth-major walk (measurement aid) */
	{
		const char *walk = std::getenv("BEAMFORMER_HIP_STAGED_WALK");
		if (best.uniform && (best.depth_major & 1u) && !(walk && walk[0] == 'c')) best.depth_major |= 4u;
	}      /* test hook: the range-checked loop for every wave */
	if (std::getenv("BEAMFORMER_HIP_DEBUG"))
		std::fprintf(stderr, "[beamformer] staged plan: step_u %.3f step_v %.3f waves %u u %u v %u w %u chunk %u lds %u uniform %u\n",
		             step_u, step_v, best_waves, best.u_shift, best.v_shift, best.window_samples, best.channel_chunk, best.lds_bytes, best.uniform);
	if (!best_waves) return false;
	q = best;
	uint32_t nu = a.size[u_axis], nv = a.size[v_axis];
	q.tiles[0] = (nu + (1u << q.u_shift) - 1) >> q.u_shift;
	q.tiles[1] = (nv + (1u << q.v_shift) - 1) >> q.v_shift;
	q.tiles[2] = zcount;
	return true;
}

/* Can this HERCULES-family frame use the aligned fast path (das_hercules.hip)?  The kernel lays
 * the 64 lanes of a wave along the output's x axis and reads the squared lateral distance along
 * the OTHER array axis from a per-output-row table, so one transducer lateral coordinate has to
 * be a function of the output row y alone: every product that would let voxel x or voxel z move
 * it must be an exact zero (then the table entry is bit-identical to the per-voxel value).
 * Everything else -- depth, the transmit distance, the coordinate along x -- stays per voxel. */
static bool plan_hercules(const BfDasArgs &a, const std::vector<BfTransmit> &tx, const float *xdc, const float *vox,
                          uint32_t zcount, bool forced, BfHerculesArgs &q)
{
	if (a.family != BF_DAS_HERCULES || tx.empty()) return false;
	if (!forced) {
		if (a.size[0] < 32 || a.split_shift) return false;       /* thin or tiny frames: the general kernel's channel split */
		if (((a.size[0] + 63u) & ~63u) > a.size[0] + a.size[0] / 3u) return false;   /* > 25 % idle lanes */
	}
	auto W = [&](int row, int col) { return vox[4 * col + row]; };
	auto X = [&](int row, int col) { return xdc[4 * col + row]; };
	/* does transducer coordinate `row` move with voxel axis `col`?  An axis of one voxel moves nothing, whatever its
	 * column of the transform holds: the view planes of math.c:844-885 keep their NORMAL there (das_transform_2d_xz: voxel z
	 * = (0, 1, 0), size 1), and the reference's own harness beamforms exactly such a plane (tests/throughput.c:20, :443-446) */
	auto moves = [&](int row, int col) {
		if (a.size[col] <= 1) return false;
		for (int k = 0; k < 3; k++) if (X(row, k) != 0.f && W(k, col) != 0.f) return true;
		return false;
	};
	int inner = -1;
	for (int coord = 0; coord < 2 && inner < 0; coord++)
		if (!moves(coord, 0) && !moves(coord, 2)) inner = coord;
	/* prefer the coordinate that does move with y when both qualify (a degenerate grid) */
	if (inner == 0 && !moves(1, 0) && !moves(1, 2) && !moves(0, 1) && moves(1, 1)) inner = 1;
	if (inner < 0) return false;
	const bool rx_cols = (tx[0].flags & BF_RX_COLUMNS) != 0;
	const int  tx_coord = rx_cols ? 1 : 0;                        /* das.glsl:238-247: transmit elements run along the other axis */
	const uint32_t A = (uint32_t)a.acquisition_count, C = (uint32_t)a.channel_count;
	const uint32_t transmits = A - (a.sparse ? 1u : 0u);
	if (!transmits || !C) return false;
	q.inner_coord       = (uint32_t)inner;
	q.inner_is_transmit = inner == tx_coord;
	q.inner_count       = q.inner_is_transmit ? transmits : C;
	q.outer_count       = q.inner_is_transmit ? C : transmits;
	q.table_pitch       = (q.inner_count + 8u + 3u) & ~3u;      /* the kernel prefetches one batch of 4 past the end */
	q.tiles[0] = (a.size[0] + 63u) / 64u;
	q.tiles[1] = (a.size[1] + 3u) / 4u;
	q.tiles[2] = zcount;
	{
		/* the axis along which the transducer-space depth changes fastest (as choose_tile finds it) */
		float m[16];
		m4_mul(xdc, vox, m);
		const uint32_t ext[3] = {a.size[0], a.size[1], zcount};
		int depth = 2; float best = -1.f;
		for (int i = 0; i < 3; i++) {
			if (ext[i] <= 1) continue;
			float step = std::fabs(m[4 * i + 2]) / (float)(a.size[i] > 1 ? a.size[i] - 1 : 1);
			if (step > best) { best = step; depth = i; }
		}
		q.depth_major = tile_walk(depth, zcount, q.tiles[1], q.band_rows);
	}
	/* unit of length: among the 8193 floats nearest 1, the s2 whose k' = float(k / sqrt(s2)) reproduces
	 * k = fs / c best as k' sqrt(s2) (errors are spread over +-3e-8, the best of 8193 lands near 1e-11).
	 * Remembered per (fs, c): frames of one plan ask again every launch. */
	{
		static float cached_fs = 0.f, cached_c = 0.f, cached_s2 = 1.f, cached_k = 0.f;
		if (cached_fs != a.sampling_frequency || cached_c != a.speed_of_sound) {
			const double k_exact = (double)a.sampling_frequency / (double)a.speed_of_sound;
			double best = 1e9;
			for (int i = -4096; i <= 4096; i++) {
				uint32_t bits = 0x3F800000u + (uint32_t)i;              /* floats around 1.0f in ulp steps */
				float s2; std::memcpy(&s2, &bits, sizeof s2);
				double s  = std::sqrt((double)s2);
				float  kk = (float)(k_exact / s);
				double err = std::fabs((double)kk * s / k_exact - 1.0);
				if (err < best) { best = err; cached_s2 = s2; cached_k = kk; }
			}
			cached_fs = a.sampling_frequency; cached_c = a.speed_of_sound;
		}
		q.unit_scale2 = cached_s2; q.samples_per_unit = cached_k;
	}
	{
		/* distances to two elements of the inner axis differ by at most their separation: at most 255 pitches (dense or
		 * sparse element indices alike), i.e. this many turns of demodulation phase inside one inner loop */
		const float span_turns = std::fabs(a.turns_per_sample) * 255.0f * std::fabs(a.pitch[inner]) * a.sampling_frequency * a.inv_speed_of_sound;
		q.phase_local = a.complex_data && span_turns < 400.0f &&       /* (false for a NaN) */
		                !std::getenv("BEAMFORMER_HIP_HERCULES_FRACT");  /* measurement aid: v_fract per pair */
	}
	return true;
}

/* Can the per-voxel factored kernel (das_factored.hip) take this frame?  It needs the sample
 * index to be a receive term plus a transmit term: RCA-family frames whose transmits all share
 * one receive orientation, and FORCES/UFORCES.  With fewer than three transmits per channel
 * chunk the receive factors are not amortised and the general kernel is as fast. */
static bool factored_applies(const BfDasArgs &a, const std::vector<BfTransmit> &tx, uint32_t mode)
{
	if (mode == 1) return false;
	int transmits = a.acquisition_count - (a.family == BF_DAS_FORCES && a.sparse ? 1 : 0);
	if (transmits < 3 && mode != 4) return false;
	if (a.family == BF_DAS_FORCES) return true;
	if (a.family != BF_DAS_RCA || tx.empty()) return false;
	for (const BfTransmit &t : tx)
		if ((t.flags & BF_RX_ROWS) != (tx[0].flags & BF_RX_ROWS)) return false;
	return true;
}

/* A timed HIP event costs ~4 us of stream time on this runtime (measured: a 0.26 MB / 256 x 256
 * frame takes 36.5 us with its five records and 15.7 us without), nothing next to a 3-D volume
 * and more than the kernels of a real-time 2-D frame.  Small frames therefore record their
 * per-stage events on one frame in kTimingSamplePeriod; the frames in between run with no event
 * at all and report the newest sampled timings in the stats table. */
constexpr uint64_t kTimingSamplePeriod = 8;
constexpr uint32_t kStagedMinTransmits = 6;      /* das_staged.hip by default from this many transmits per channel (tools/staged_threshold.py: 1.17 of the gather kernel's time at 4 transmits, 0.93 at 6-8, 0.85 at 10-12, 0.74 at 16, 0.68-0.71 at 32-75) */
constexpr uint64_t kSmallFrameBytes    = 8ull << 20;

static bool record(TimingSlot &t, uint32_t index, hipStream_t s)
{
	if (!t.sampled) return true;
	return HIP_OK(hipEventRecord(t.events[index], s));
}

static bool run_frame_stages(uint32_t block, const void *rf, int64_t rf_bytes, bool ingest_timed);

/* One frame.  With frame graphs on (beamformer_hip_enable_frame_graphs; BASELINE.json configs[4]: "hipGraph-
 * captured frame"; the reference's analogue is the one command list it records per frame,
 * beamformer_core.c:1570-1620) the stage launches are captured into a hipGraph instead of being enqueued:
 * every frame is captured (the frame-ring slot and the RF slot move from frame to frame, so kernel arguments
 * change), the block's instantiated graph is updated in place from the capture (hipGraphExecUpdate: same
 * topology, new arguments; re-instantiated when the topology changed) and launched.  The first frame of a
 * plan runs uncaptured so that every allocation a stage needs exists before anything is captured.  Per-stage
 * events cannot be recorded inside a graph: a graph frame times as one segment, reported under DAS. */
static bool run_frame(uint32_t block, const void *rf, int64_t rf_bytes, bool ingest_timed)
{
	Context &c = g_context;
	Device  &d = *c.cur;
	if (!c.frame_graphs || c.device_count != 1 || c.count_pairs) return run_frame_stages(block, rf, rf_bytes, ingest_timed);
	PlanState *ps = commit_block(block);                   /* a replan drains the stream: never inside a capture */
	if (!ps) return set_error(BeamformerLibErrorKind_InvalidComputeStage);
	if (d.graph_generation[block] != ps->generation) {
		d.graph_generation[block] = ps->generation;
		if (d.frame_exec[block]) { (void)hipGraphExecDestroy(d.frame_exec[block]); d.frame_exec[block] = nullptr; }
		return run_frame_stages(block, rf, rf_bytes, ingest_timed);
	}
	hipStream_t s = d.stream;
	TimingSlot &t = d.timing[d.frame_counter % kTimingSlots];
	if (!t.created) {
		for (auto &e : t.events) if (!HIP_OK(hipEventCreate(&e))) return set_error(BeamformerLibErrorKind_SharedMemory);
		t.created = true;
	}
	const bool sampled = t.sampled;
	const uint32_t first = ingest_timed ? 1u : 0u;          /* events[0] -> events[1] is the caller's ingest segment */
	if (sampled && !HIP_OK(hipEventRecord(t.events[first], s))) return set_error(BeamformerLibErrorKind_InvalidAccess);
	t.sampled = false;                                      /* no event records inside the capture */
	bool ok = HIP_OK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
	if (!ok) { t.sampled = sampled; return set_error(BeamformerLibErrorKind_InvalidAccess); }
	ok = run_frame_stages(block, rf, rf_bytes, ingest_timed);
	hipGraph_t graph = nullptr;
	bool ended = HIP_OK(hipStreamEndCapture(s, &graph));
	t.sampled = sampled;
	if (!ok || !ended || !graph) { if (graph) (void)hipGraphDestroy(graph); return ok ? set_error(BeamformerLibErrorKind_InvalidAccess) : false; }
	if (d.frame_exec[block]) {
		hipGraphNode_t bad = nullptr; hipGraphExecUpdateResult why;
		if (!HIP_OK(hipGraphExecUpdate(d.frame_exec[block], graph, &bad, &why))) {
			(void)hipGetLastError();
			(void)hipGraphExecDestroy(d.frame_exec[block]); d.frame_exec[block] = nullptr;
		}
	}
	if (!d.frame_exec[block]) {
		ok = HIP_OK(hipGraphInstantiate(&d.frame_exec[block], graph, nullptr, nullptr, 0));
		c.graph_instantiations++;
	}
	(void)hipGraphDestroy(graph);
	ok = ok && HIP_OK(hipGraphLaunch(d.frame_exec[block], s));
	c.graph_frames += ok;
	/* the frame as one timed segment */
	t.count = 0;
	if (ingest_timed) t.kinds[t.count++] = kStageIngest;
	t.kinds[t.count++] = (uint32_t)BeamformerShaderKind_DAS;
	if (sampled) ok = ok && HIP_OK(hipEventRecord(t.events[t.count], s));
	return ok || set_error(BeamformerLibErrorKind_InvalidAccess);
}

static bool run_frame_stages(uint32_t block, const void *rf, int64_t rf_bytes, bool ingest_timed)
{
	Context &c = g_context;
	Device  &d = *c.cur;
	PlanState *ps = commit_block(block);
	if (!ps) return set_error(BeamformerLibErrorKind_InvalidComputeStage);
	const Plan &plan = ps->plan;
	const ParameterBlock &pb = c.blocks[block];
	const BeamformerParameters &bp = pb.parameters;
	hipStream_t s = d.stream;

	TimingSlot &t = d.timing[d.frame_counter % kTimingSlots];
	if (!t.created) {
		for (auto &e : t.events) if (!HIP_OK(hipEventCreate(&e))) return set_error(BeamformerLibErrorKind_SharedMemory);
		t.created = true;
	}
	/* segment k of the frame is bracketed by events[k] and events[k+1]; events[0] was recorded
	 * in front of the ingest by the caller when ingest_timed */
	t.count = 0; t.counted = false;
	auto segment = [&](uint32_t kind) {
		if (t.count < BEAMFORMER_HIP_MAX_TIMED_STAGES) {
			t.kinds[t.count++] = kind;
			record(t, t.count, s);
		}
	};
	if (ingest_timed) segment(kStageIngest);
	else              record(t, 0, s);

	const uint32_t C = plan.channels, A = plan.acquisitions, Sd = plan.das_samples;
	const void *cur = rf;
	int64_t cur_elements_bytes = rf_bytes;
	int toggle = 0;
	bool ok = true, das_segment_done = false;
	uint32_t das_path = 0;

	for (size_t i = 0; i < plan.stages.size() && ok; i++) {
		const Stage &st = plan.stages[i];
		switch (st.kind) {
		case BeamformerShaderKind_Reshape:{
			BfReshapeArgs a{};
			a.size[0] = Sd; a.size[1] = C; a.size[2] = A;                          /* beamformer_core.c:975-977 */
			for (int k = 0; k < 3; k++) { a.in_stride[k] = st.in_stride[k]; a.out_stride[k] = st.out_stride[k]; }
			a.in_kind = st.in_kind; a.out_kind = st.out_kind;
			a.interleave = !bf_kind_complex[st.in_kind] && bf_kind_complex[st.out_kind];
			a.left  = cur;
			a.right = (const char *)cur + (size_t)Sd * C * A * (size_t)bf_kind_byte_size[st.in_kind];   /* :1384-1385 */
			a.out = d.scratch[toggle].ptr;
			ok &= HIP_OK(bf_launch_reshape(&a, s));
			cur = a.out; cur_elements_bytes = (int64_t)d.scratch[toggle].size; toggle ^= 1;
		}break;
		case BeamformerShaderKind_Decode:{
			BfDecodeArgs a{};
			a.in = cur; a.out = d.scratch[toggle].ptr;
			a.hadamard_t = (const float *)ps->hadamard_t.ptr;
			a.hadamard_base_order = (c.das_path_mode & 0x20) ? 0 : plan.hadamard_base_order;
			a.hadamard_base = a.hadamard_base_order ? (const float *)ps->hadamard_base.ptr : nullptr;
			a.transmit_count = A; a.channel_count = C; a.sample_count = Sd;
			for (int k = 0; k < 3; k++) a.out_stride[k] = st.out_stride[k];
			a.in_kind = st.in_kind; a.out_kind = st.out_kind;
			ok &= HIP_OK(bf_launch_decode(&a, s));
			cur = a.out; cur_elements_bytes = (int64_t)d.scratch[toggle].size; toggle ^= 1;
		}break;
		case BeamformerShaderKind_Hilbert:{
			BfFilterArgs a{};
			a.in = cur; a.out = d.scratch[toggle].ptr;
			a.coefficients  = (const float *)ps->taps[i].ptr;
			a.filter_length = (uint32_t)st.filter.length;
			a.sample_count  = Sd;
			for (int k = 0; k < 3; k++) { a.in_stride[k] = st.in_stride[k]; a.out_stride[k] = st.out_stride[k]; }
			a.in_elements = cur_elements_bytes / bf_kind_byte_size[st.in_kind];
			a.channels = C; a.transmits = A;
			a.in_kind = st.in_kind; a.out_kind = st.out_kind;
			ok &= HIP_OK(bf_launch_hilbert(&a, s));
			cur = a.out; cur_elements_bytes = (int64_t)d.scratch[toggle].size; toggle ^= 1;
		}break;
		case BeamformerShaderKind_Filter:
		case BeamformerShaderKind_Demodulate:{
			bool demod = st.kind == BeamformerShaderKind_Demodulate;
			BfFilterArgs a{};
			a.in = cur; a.out = d.scratch[toggle].ptr;
			a.coefficients   = (const float *)ps->taps[i].ptr;
			a.phasors        = demod ? a.coefficients + st.filter.taps.size() : nullptr;
			a.filter_length  = (uint32_t)st.filter.length;
			a.complex_filter = st.filter.complex_taps;
			a.demodulate     = demod;
			a.decimation     = demod ? plan.decimation : 1;                          /* :846 */
			a.sample_count   = Sd;                                                  /* :845 */
			bool deinterleave = bf_kind_complex[st.in_kind] && !bf_kind_complex[st.out_kind];
			a.batch_sample_count = deinterleave ? C * Sd * A : 0;                   /* :848-851 */
			if (demod) {                                                            /* :870-873 */
				a.demodulation_frequency = bp.demodulation_frequency;
				a.sampling_frequency     = bp.sampling_frequency / 2;
			}
			for (int k = 0; k < 3; k++) { a.in_stride[k] = st.in_stride[k]; a.out_stride[k] = st.out_stride[k]; }
			a.in_elements = cur_elements_bytes / bf_kind_byte_size[st.in_kind];
			a.channels = C; a.transmits = A;
			a.in_kind = st.in_kind; a.out_kind = st.out_kind;
			ok &= HIP_OK(bf_launch_filter(&a, s));
			cur = a.out; cur_elements_bytes = (int64_t)d.scratch[toggle].size; toggle ^= 1;
		}break;
		case BeamformerShaderKind_DAS:{
			uint32_t zfirst = 0, zcount = plan.output_points[2];
			if (pb.shard_z_count) { zfirst = pb.shard_z_first; zcount = pb.shard_z_count; }
			if (c.device_count > 1) { zfirst = d.slab_first; zcount = d.slab_count; }   /* this device's z-slab (push_multi) */
			uint32_t points[3] = {plan.output_points[0], plan.output_points[1], zcount};
			FrameRecord *f = next_frame(points, plan.iq_pipeline, block);
			if (!f) return set_error(BeamformerLibErrorKind_FrameSizeOverflow);
			f->timing_slot = (int)(f->id % kTimingSlots);
			if (zcount == 0) {           /* more devices than planes: this device holds an empty slab of the frame */
				t.das_voxels = 0; t.das_taps = 0; t.das_sample_bytes = 0; t.das_path = 0; t.frame_id = f->id;
				break;
			}

			BfDasArgs a{};
			std::memcpy(a.xdc_transform,   bp.xdc_transform,         sizeof(a.xdc_transform));
			std::memcpy(a.voxel_transform, plan.das_voxel_transform, sizeof(a.voxel_transform));
			a.pitch[0] = bp.xdc_element_pitch[0]; a.pitch[1] = bp.xdc_element_pitch[1];
			a.rf  = cur;
			a.out = (char *)d.ring.ptr + f->offset;
			a.transmits       = (const BfTransmit *)ps->transmits.ptr;
			a.sparse_elements = (const int16_t *)ps->sparse.ptr;
			a.readi_hadamard  = (const uint16_t *)ps->readi_hadamard.ptr;
			switch (bp.acquisition_kind) {                                          /* das.glsl:381-400 */
			case BeamformerAcquisitionKind_FORCES:
			case BeamformerAcquisitionKind_UFORCES:
				a.family = bp.readi_group_count > 1 ? BF_DAS_READI : BF_DAS_FORCES; break;
			case BeamformerAcquisitionKind_HERCULES:
			case BeamformerAcquisitionKind_UHERCULES:
			case BeamformerAcquisitionKind_HERO_PA:
				a.family = BF_DAS_HERCULES; break;
			case BeamformerAcquisitionKind_Flash:
			case BeamformerAcquisitionKind_RCA_TPW:
			case BeamformerAcquisitionKind_RCA_VLS:
				a.family = BF_DAS_RCA; break;
			default: a.family = -1; break;      /* the shader leaves the voxel at zero */
			}
			a.interpolation = (int32_t)bp.interpolation_mode;
			a.complex_data  = plan.iq_pipeline;
			a.coherency_weighting = bp.coherency_weighting != 0;
			a.acquisition_count = (int32_t)A; a.channel_count = (int32_t)C; a.sample_count = (int32_t)Sd;
			a.sparse = plan.das_sparse;
			a.sampling_frequency     = plan.das_sampling_frequency;
			a.inv_sampling_frequency = 1.0f / plan.das_sampling_frequency;
			a.demodulation_frequency = bp.demodulation_frequency;
			a.inv_speed_of_sound     = 1.0f / bp.speed_of_sound;
			a.speed_of_sound         = bp.speed_of_sound;
			a.turns_per_sample       = bp.demodulation_frequency * a.inv_sampling_frequency;
			a.first_transmit_weight  = 1.0f / sqrtf((float)A);
			a.time_offset = plan.das_time_offset;
			a.f_number    = bp.f_number;
			a.size[0] = plan.output_points[0]; a.size[1] = plan.output_points[1]; a.size[2] = plan.output_points[2];
			a.z_first = zfirst; a.z_count = zcount;
			a.readi_group_count = bp.readi_group_count; a.readi_group = bp.readi_group;

			float to_xdc[16];
			if (a.family == BF_DAS_FORCES || a.family == BF_DAS_READI) std::memcpy(to_xdc, plan.das_voxel_transform, sizeof(to_xdc));
			else m4_mul(bp.xdc_transform, plan.das_voxel_transform, to_xdc);
			uint32_t ext[3] = {a.size[0], a.size[1], zcount};
			/* Small frames (real-time 2-D imaging) do not fill 256 CUs with one thread per voxel:
			 * split the channel loop over K waves of a block (wave-level partial sums, combined
			 * through LDS in split order) until the launch has ~16 waves per CU (config 1, us per
			 * frame by target wave count: 2048 -> 19.9, 4096 -> 15.2, 8192 -> 15.1, 16384 -> 17.1). */
			uint64_t voxel_waves = ((uint64_t)ext[0] * ext[1] * ext[2] + 63) / 64;
			a.split_shift = 0;
			while (!(c.das_path_mode & 0x10) && a.split_shift < 4 && (voxel_waves << a.split_shift) < 4096 && (C >> (a.split_shift + 1)) >= 4) a.split_shift++;
			const int depth_axis = choose_tile(to_xdc, a.size, zcount, a.tile_shift, a.split_shift ? 6 : 8);
			for (int k = 0; k < 3; k++) a.blocks[k] = (ext[k] + (1u << a.tile_shift[k]) - 1) >> a.tile_shift[k];
			a.depth_major = tile_walk(depth_axis, zcount, a.blocks[1], a.band_rows);

			if (a.family < 0 || a.interpolation < 0 || a.interpolation > 2) {
				ok &= HIP_OK(hipMemsetAsync(a.out, 0, f->bytes, s));
			} else {
				BfSeparableArgs sep{};
				/* The LDS-table kernel's hand-scheduled loop exists for linear interpolation; for cubic
				 * and nearest its generic loop loses to the factored kernel (200 ch x 33 tx -> 129 x 333 x 21,
				 * cubic: 7.6 ms against 4.8 ms; nearest 2.6 against 2.1), which then goes first. */
				const uint32_t das_mode = c.das_path_mode & 0xF;
				/* (cubic IQ frames with enough transmits try the staged cubic kernel first: it declines -- and the factored
				 * kernel runs -- when the geometry is not separable or the delay spread does not fit a window) */
				const bool staged_cubic = a.interpolation == 2 && plan.iq_pipeline && (das_mode == 3 || (das_mode == 0 && A >= kStagedMinTransmits));
				bool tables_first = a.interpolation == 1 || das_mode == 3 || staged_cubic ||
				                    !factored_applies(a, ps->transmit_table, das_mode);
				if (staged_cubic && das_mode != 3 && tables_first && a.interpolation == 2) {
					BfSeparableArgs probe{};
					if (!(plan_separable(a, ps->transmit_table, bp.xdc_transform, plan.das_voxel_transform, zcount, probe) &&
					      plan_staged(a, ps->transmit_table, bp.xdc_transform, plan.das_voxel_transform, zcount, probe)))
						tables_first = !factored_applies(a, ps->transmit_table, das_mode);
				}
				if (das_mode != 1 && das_mode != 4 && tables_first &&
				    plan_separable(a, ps->transmit_table, bp.xdc_transform, plan.das_voxel_transform, zcount, sep)) {
					/* 64 zero bytes right behind the DAS input (every buffer it can live in is
					 * allocated with that much slack): the gather target of out-of-range lanes */
					uint64_t used = (uint64_t)C * A * Sd * (plan.iq_pipeline ? 8u : 4u);
					sep.zero_offset = (uint32_t)used;
					ok &= HIP_OK(hipMemsetAsync((char *)const_cast<void *>(cur) + used, 0, 64, s));
					/* the LDS-staged kernel pays two block barriers and a window copy per channel: it wins once a channel
					 * carries enough transmits to amortise them (kStagedMinTransmits, measured: tools/staged_threshold.py) */
					const bool want_staged = das_mode == 3 || (das_mode == 0 && A >= kStagedMinTransmits);
					bool staged = want_staged && plan_staged(a, ps->transmit_table, bp.xdc_transform, plan.das_voxel_transform, zcount, sep);
					if (staged && sep.uniform) {
						/* the wave-uniform transmit tables live in global memory: one slice per (lateral tile row, plane), written per frame
						 * (a few MB to 244 MB at 512^3 with 75 transmits; too big or no memory: the shapes with the tables in LDS) */
						const uint64_t table_bytes = (uint64_t)sep.table_stride * sep.tiles[1] * sep.tiles[2];
						if (table_bytes <= (2ull << 30) && d.staged_tables.ensure(table_bytes)) {
							sep.tables = d.staged_tables.ptr;
							ok &= HIP_OK(bf_launch_das_staged_tables(&a, &sep, s));
						} else {
							BfSeparableArgs again = sep;
							again.uniform = 0; again.table_stride = 0; again.tables = nullptr; again.depth_major &= ~4u;
							staged = plan_staged(a, ps->transmit_table, bp.xdc_transform, plan.das_voxel_transform, zcount, again, false);
							if (staged) sep = again;
						}
					}
					if (staged) {
						ok &= HIP_OK(!plan.iq_pipeline ? bf_launch_das_staged_real(&a, &sep, s) :
						             a.interpolation == 2 ? bf_launch_das_staged_cubic(&a, &sep, s) : bf_launch_das_staged(&a, &sep, s));
						das_path = 2;
					} else {
						ok &= HIP_OK(bf_launch_das_separable(&a, &sep, s));
						das_path = 1;
					}
				} else if (BfHerculesArgs hq{}; das_mode != 1 &&
				           plan_hercules(a, ps->transmit_table, bp.xdc_transform, plan.das_voxel_transform, zcount, das_mode == 6, hq) &&
				           d.hercules_table.ensure(((size_t)hq.table_pitch + 2) * a.size[1] * sizeof(float))) {
					uint64_t used = (uint64_t)C * A * Sd * (plan.iq_pipeline ? 8u : 4u);
					hq.zero_offset = (uint32_t)used;            /* as for the gather kernel above */
					ok &= HIP_OK(hipMemsetAsync((char *)const_cast<void *>(cur) + used, 0, 64, s));
					/* linear interpolation of IQ samples reads a {sample, difference} copy of the input (16 bytes per sample, 32-bit byte
					 * offsets: under 4 GiB), built by the launcher; BEAMFORMER_HIP_HERCULES_NOPAIRS: measurement aid */
					hq.pairs = nullptr;
					const uint64_t prepared = used * (a.interpolation == 2 ? 4u : 2u);      /* cubic: four coefficients per sample, 32 bytes */
					/* (not on coarse grids: the copy is 2-4 x the RF, and where every lane reads its own cache line the memory system
					 * pays for the bytes -- the harness's view plane with cubic polynomials: 28.1 ms, 158 GB from beyond L2 per frame;
					 * with the taps gathered from the RF itself 25.2 ms) */
					if (plan.iq_pipeline && (a.interpolation == 1 || a.interpolation == 2) && prepared + 64 < (1ull << 32) &&
					    lane_step_samples(to_xdc, a) < 1.0f &&
					    !std::getenv("BEAMFORMER_HIP_HERCULES_NOPAIRS") && d.hercules_pairs.ensure(prepared + 64)) {
						hq.pairs = d.hercules_pairs.ptr;
						hq.zero_offset = (uint32_t)prepared;
					}
					hq.table    = (float *)d.hercules_table.ptr;
					hq.extremes = hq.table + (size_t)hq.table_pitch * a.size[1];
					ok &= HIP_OK(bf_launch_das_hercules(&a, &hq, s));
					das_path = 5;
				} else if (factored_applies(a, ps->transmit_table, c.das_path_mode & 0xF)) {
					uint64_t used = (uint64_t)C * A * Sd * (plan.iq_pipeline ? 8u : 4u);
					a.zero_offset = (uint32_t)used;             /* as for the gather kernel above */
					/* wave-span staging (das_factored.hip): on COARSE grids -- a voxel step along x of a sample of delay or more, as the
					 * reference harness's 0.23 mm pixels have (tests/throughput.c:20-23) -- the lanes of a gather land in 64 different
					 * places and the per-wave LDS-DMA copy of the span is cheaper (harness frames: 0.90-0.93 of the gather loop's time;
					 * on config 2's fine grid 1.13: not taken there).  das path bit 0x40 forces it wherever the kernel supports it, 0x80
					 * keeps the gather loop (tests: the two frames are bit-identical). */
					const bool span_ok = plan.iq_pipeline && a.interpolation >= 1 && !a.split_shift && Sd >= 128 && used < (1ull << 32);
					if (span_ok && !(c.das_path_mode & 0x80) && ((c.das_path_mode & 0x40) || lane_step_samples(to_xdc, a) >= 1.0f)) {
						a.span_stage = 1;
						/* a wave = 64 voxels along the first lateral axis, the block's four waves stacked along depth */
						uint32_t lat = a.tile_shift[0] >= a.tile_shift[1] ? 0u : 1u;
						if (a.tile_shift[lat] > 6) {
							uint32_t spare = a.tile_shift[lat] - 6;
							a.tile_shift[lat] = 6;
							for (int k = 0; k < 3 && spare; k++) {
								if ((uint32_t)k == lat) continue;
								uint32_t room = ceil_log2(ext[k]) - a.tile_shift[k];
								uint32_t give = room < spare ? room : spare;
								a.tile_shift[k] += give; spare -= give;
							}
							a.tile_shift[lat] += spare;
							for (int k = 0; k < 3; k++) a.blocks[k] = (ext[k] + (1u << a.tile_shift[k]) - 1) >> a.tile_shift[k];
							a.depth_major = tile_walk(depth_axis, zcount, a.blocks[1], a.band_rows);
						}
					}
					ok &= HIP_OK(hipMemsetAsync((char *)const_cast<void *>(cur) + used, 0, 64, s));
					ok &= HIP_OK(bf_launch_das_factored(&a, s));
					das_path = 3;
				} else {
					ok &= HIP_OK(bf_launch_das(&a, s));
				}
				if (c.count_pairs) {
					/* geometry-only recount of the apodization test; its own segment so that it
					 * stays out of the DAS time */
					ok &= d.pair_counter.ensure(sizeof(unsigned long long) * kTimingSlots);
					a.pair_counter = (unsigned long long *)d.pair_counter.ptr + (f->id % kTimingSlots);
					ok &= HIP_OK(hipMemsetAsync(a.pair_counter, 0, sizeof(unsigned long long), s));
					segment((uint32_t)st.kind);
					ok &= HIP_OK(bf_launch_das_count(&a, s));
					segment(kStagePairCount);
					t.counted = true;
					das_segment_done = true;
				}
			}
			t.das_voxels = (uint64_t)ext[0] * ext[1] * ext[2];
			t.das_taps = a.interpolation == 0 ? 1 : a.interpolation == 1 ? 2 : 4;
			t.das_sample_bytes = plan.iq_pipeline ? 8 : 4;
			t.das_path = das_path; t.frame_id = f->id;
		}break;
		case BeamformerShaderKind_CoherencyWeighting:
			/* fused into the DAS epilogue (das.hip); kept in the plan so that the stage list a
			 * client sees through beamformer_compute_timings matches the reference's */
			break;
		default: break;
		}
		if (!(st.kind == BeamformerShaderKind_DAS && das_segment_done)) segment((uint32_t)st.kind);
	}
	if (plan.das_index < 0 && ok) {
		/* no DAS in the pipeline: the frame exists and stays zero (the reference clears it,
		 * beamformer_core.c:1573-1585, and nothing writes it) */
		uint32_t points[3] = {plan.output_points[0], plan.output_points[1], plan.output_points[2]};
		FrameRecord *f = next_frame(points, plan.iq_pipeline, block);
		if (!f) return set_error(BeamformerLibErrorKind_FrameSizeOverflow);
		f->timing_slot = (int)(f->id % kTimingSlots);
		ok &= HIP_OK(hipMemsetAsync((char *)d.ring.ptr + f->offset, 0, f->bytes, s));
		t.das_voxels = 0; t.das_taps = 0; t.das_sample_bytes = 0; t.das_path = 0; t.frame_id = f->id;
	}
	if (!ok) return set_error(BeamformerLibErrorKind_InvalidAccess);
	return true;
}

/* z-slab of device `i` of `n` over `planes` planes starting at `first`: contiguous, sizes differing by
 * at most one -- the rule of ogl_beamforming_amd/sharding.py (one process per GPU), so both ways of
 * spreading a frame over a node cut it at the same planes */
static void device_slab(uint32_t i, uint32_t n, uint32_t first, uint32_t planes, uint32_t &z_first, uint32_t &z_count)
{
	uint32_t begin = (uint32_t)((uint64_t)i * planes / n), end = (uint32_t)((uint64_t)(i + 1) * planes / n);
	z_first = first + begin;
	z_count = end - begin;
}

/* Several devices, one frame: the channel-mapped RF that the ingest device (devices[0])
 * holds at `src` is copied to every peer's RF slot -- hipMemcpyPeerAsync, one stream per destination, so
 * the copies run side by side on their xGMI links and, three RF slots deep, beside the kernels of the
 * previous frame -- and every peer beamforms its own z-slab of the block's grid.  No reduction
 * collective: voxels are independent.  Called with devices[0] current and its ingest already
 * enqueued on its stream; returns with devices[0] current again. */
static bool run_peers(uint32_t block, const void *src, uint64_t rf_size, uint32_t slot)
{
	Context &c = g_context;
	Device  &d0 = c.devices[0];
	const ParameterBlock &pb = c.blocks[block];
	const uint32_t n = c.device_count;
	const uint32_t planes_total = (uint32_t)(pb.parameters.output_points[2] > 1 ? pb.parameters.output_points[2] : 1);
	const uint32_t first  = pb.shard_z_count ? pb.shard_z_first : 0u;
	const uint32_t planes = pb.shard_z_count ? pb.shard_z_count : planes_total;
	for (uint32_t i = 0; i < n; i++) device_slab(i, n, first, planes, c.devices[i].slab_first, c.devices[i].slab_count);

	/* "the mapped RF of this frame is complete on the ingest device" */
	bool ok = HIP_OK(hipEventRecord(d0.rf_landed[slot], d0.stream));
	for (uint32_t i = 1; i < n && ok; i++) {
		Device &p = c.devices[i];
		if (!select_device(i)) { ok = false; break; }
		ok &= p.rf[slot].ensure(round_up(rf_size, 64) + 64);
		ok &= HIP_OK(hipStreamWaitEvent(p.peer_stream, d0.rf_landed[slot], 0));
		/* the frame that read this slot three pushes ago must be done with it */
		if (p.consumed_pending[slot]) ok &= HIP_OK(hipStreamWaitEvent(p.peer_stream, p.rf_consumed[slot], 0));
		if (!ok) break;
		ok &= HIP_OK(hipMemcpyPeerAsync(p.rf[slot].ptr, p.device, src, d0.device, rf_size, p.peer_stream));
		ok &= HIP_OK(hipEventRecord(p.rf_landed[slot], p.peer_stream));
		ok &= HIP_OK(hipStreamWaitEvent(p.stream, p.rf_landed[slot], 0));
		TimingSlot &t = p.timing[p.frame_counter % kTimingSlots];
		t.sampled = true; t.events_slot = (uint32_t)(p.frame_counter % kTimingSlots);
		ok = ok && run_frame(block, p.rf[slot].ptr, (int64_t)p.rf[slot].size, false);
		p.consumed_pending[slot] = ok && HIP_OK(hipEventRecord(p.rf_consumed[slot], p.stream));
	}
	if (!select_device(0)) ok = false;
	return ok || set_error(BeamformerLibErrorKind_InvalidAccess);
}

/* lib .c:491-570 (client copy) + beamformer_core.c:1756-1805 (upload worker) */
bool push_rf_and_compute(uint32_t block, const void *data, uint32_t size, bool data_on_device)
{
	Context &c = g_context;
	Device  &d = *c.cur;
	ParameterBlock &pb = c.blocks[block];
	const BeamformerParameters &bp = pb.parameters;
	hipStream_t s = d.stream;

	const uint64_t bytes   = (uint64_t)bf_kind_byte_size[pb.data_kind];
	const uint64_t out_row = bytes * bp.sample_count * bp.acquisition_count;
	const uint64_t in_row  = bytes * bp.raw_data_dimensions[0];
	const uint64_t rf_size = out_row * bp.channel_count;

	/* the reference copies whatever row the mapping names (lib .c:520-528); on a GPU an
	 * out-of-range row would fault, so it is an error here */
	bool identity = true;
	for (uint32_t ch = 0; ch < bp.channel_count; ch++) {
		uint16_t row = (uint16_t)pb.channel_mapping[ch];
		if (row >= bp.raw_data_dimensions[1]) return set_error(BeamformerLibErrorKind_DataSizeMismatch);
		identity &= row == ch;
	}
	bool a1s2 = bp.contrast_mode == BeamformerContrastMode_A1S2;

	uint32_t slot = (uint32_t)(d.rf_index++ % BeamformerMaxRawDataFramesInFlight);
	if (!d.rf[slot].ensure(round_up(rf_size, 64) + 64)) return set_error(BeamformerLibErrorKind_RFDataSizeOverflow);
	const bool multi = c.device_count > 1;
	if (multi) {
		/* every peer replans before the ingest device does (its commit clears the dirty bits) */
		for (uint32_t i = 1; i < c.device_count; i++) {
			if (!select_device(i) || !commit_block(block)) { select_device(0); return set_error(BeamformerLibErrorKind_InvalidComputeStage); }
		}
		if (!select_device(0)) return set_error(BeamformerLibErrorKind_SharedMemory);
		/* this RF slot was the source of the peer copies three pushes ago: they must have landed before
		 * anything overwrites it (long done by now; waiting on a never-recorded event is a no-op) */
		for (uint32_t i = 1; i < c.device_count; i++) {
			(void)hipStreamWaitEvent(s, c.devices[i].rf_landed[slot], 0);
			(void)hipStreamWaitEvent(d.copy_stream, c.devices[i].rf_landed[slot], 0);
		}
	}

	TimingSlot &t = d.timing[d.frame_counter % kTimingSlots];
	if (!t.created) {
		for (auto &e : t.events) if (!HIP_OK(hipEventCreate(&e))) return set_error(BeamformerLibErrorKind_SharedMemory);
		t.created = true;
	}
	/* sample this frame's per-stage timings?  always for frames that are not small, after a replan,
	 * when pair counting rides along, and every kTimingSamplePeriod-th frame otherwise */
	const bool small = rf_size < kSmallFrameBytes &&
	                   (uint64_t)bp.output_points[0] * (uint64_t)(bp.output_points[1] > 1 ? bp.output_points[1] : 1) *
	                   (uint64_t)(bp.output_points[2] > 1 ? bp.output_points[2] : 1) < (4ull << 20);
	if (!d.have_sample || pb.dirty != 0 || block != d.last_sampled_block) d.replan_frame = d.frame_counter;
	/* the first frames of a plan are all sampled: the very first carries one-off launch costs */
	t.sampled = !small || c.count_pairs || d.frame_counter - d.replan_frame < 3 ||
	            d.frame_counter - d.last_sampled_frame >= kTimingSamplePeriod;
	if (t.sampled) {
		d.have_sample = true; d.last_sampled_frame = d.frame_counter; d.last_sampled_block = block;
		t.events_slot = (uint32_t)(d.frame_counter % kTimingSlots);
		(void)hipEventRecord(t.events[0], s);
	} else {
		t.events_slot = (uint32_t)(d.last_sampled_frame % kTimingSlots);
	}

	UploadSlot &u = d.upload[slot];
	if (!u.copied && (!HIP_OK(hipEventCreateWithFlags(&u.copied, hipEventDisableTiming)) ||
	                  !HIP_OK(hipEventCreateWithFlags(&u.consumed, hipEventDisableTiming))))
		return set_error(BeamformerLibErrorKind_SharedMemory);

	bool ok = true, overlap = false;
	const bool direct = identity && !a1s2 && in_row == out_row;
	const void *raw = data;
	if (!data_on_device) {
		/* Host data: the caller's bytes are copied into a pinned slot (after which the caller
		 * may reuse its buffer, as with the reference's copy into shared memory), the H2D runs
		 * on the copy stream and the compute stream waits for it -- so the upload of frame n+1
		 * overlaps the kernels of frame n, which one stream and pageable memory cannot do.
		 * A copy-engine transfer and each cross-queue dependency cost 40-60 us of latency on
		 * this runtime (tools/h2d_probe.cpp: 0.26 MB pinned H2D + a kernel = 114 us per frame),
		 * more than a small frame's compute, so frames under kOverlapBytes skip the copy engine:
		 * the ingest kernel reads the pinned slot in place over PCIe, in order on the compute
		 * stream. */
		constexpr uint32_t kOverlapBytes = 8u << 20;
		overlap = size >= kOverlapBytes;
		if (u.copy_pending) { (void)hipEventSynchronize(u.copied); u.copy_pending = false; }
		if (u.pinned_size < size) {
			if (u.pinned) (void)hipHostFree(u.pinned);
			u.pinned = nullptr; u.pinned_size = 0;
			if (!HIP_OK(hipHostMalloc(&u.pinned, round_up(size, 4096), hipHostMallocDefault))) {
				u.pinned = nullptr;
				return set_error(BeamformerLibErrorKind_BufferOverflow);
			}
			u.pinned_size = round_up(size, 4096);
		}
		/* one core copies ~37 GB/s into pinned memory here; frames of 64 MiB and more are split
		 * over a few short-lived threads (the copy of a 512 MiB decode-benchmark frame drops from
		 * 14 ms to what the memory system gives) */
		constexpr uint32_t kParallelCopyBytes = 64u << 20;
		if (size >= kParallelCopyBytes) {
			const unsigned parts = 4;
			const size_t   piece = (((size_t)size + parts - 1) / parts + 4095) & ~(size_t)4095;
			std::thread workers[parts - 1];
			for (unsigned i = 1; i < parts; i++) {
				size_t begin = piece * i, end = begin + piece < size ? begin + piece : size;
				workers[i - 1] = std::thread([=] { if (begin < end) std::memcpy((char *)u.pinned + begin, (const char *)data + begin, end - begin); });
			}
			std::memcpy(u.pinned, data, piece < size ? piece : size);
			for (auto &w : workers) w.join();
		} else {
			std::memcpy(u.pinned, data, size);
		}
		if (overlap) {
			void *dst = d.rf[slot].ptr;
			if (!direct) {
				if (!d.raw_staging[slot].ensure(round_up(size, 64) + 64)) return set_error(BeamformerLibErrorKind_BufferOverflow);
				dst = d.raw_staging[slot].ptr;
			}
			/* the device buffers of this slot were last read by the frame three pushes ago; if that
			 * frame recorded no `consumed` event (small or device-resident pushes do not), fence
			 * against everything enqueued so far instead */
			if (u.unfenced_reader) { u.consume_pending = HIP_OK(hipEventRecord(u.consumed, s)); u.unfenced_reader = false; }
			if (u.consume_pending) ok &= HIP_OK(hipStreamWaitEvent(d.copy_stream, u.consumed, 0));
			ok &= HIP_OK(hipMemcpyAsync(dst, u.pinned, direct ? rf_size : (uint64_t)size, hipMemcpyHostToDevice, d.copy_stream));
			ok &= HIP_OK(hipEventRecord(u.copied, d.copy_stream));
			u.copy_pending = true;
			ok &= HIP_OK(hipStreamWaitEvent(s, u.copied, 0));
			raw = dst;
		} else {
			void *mapped = nullptr;
			if (!HIP_OK(hipHostGetDevicePointer(&mapped, u.pinned, 0))) return set_error(BeamformerLibErrorKind_InvalidAccess);
			raw = mapped;
		}
	}
	const bool zero_copy = !data_on_device && !overlap;
	/* Device-resident RF already in the mapped layout is read in place by the first stage (no
	 * copy into the RF ring): the caller keeps it unchanged until the frame has run, which stream
	 * order gives for free when its producer is on the library's stream.  Plans that start with
	 * DAS still copy: the DAS input needs the library's zero block behind it. */
	bool borrowed = false;
	if (data_on_device && direct) {
		PlanState *ps = commit_block(block);
		if (!ps) return set_error(BeamformerLibErrorKind_InvalidComputeStage);
		borrowed = !ps->plan.stages.empty() && ps->plan.stages[0].kind != BeamformerShaderKind_DAS;
	}
	if (direct && !zero_copy) {
		/* the mapped layout is the raw layout: one copy straight into the RF slot */
		if (data_on_device && !borrowed) ok &= HIP_OK(hipMemcpyAsync(d.rf[slot].ptr, data, rf_size, hipMemcpyDeviceToDevice, s));
	} else {
		PlanState *ps = commit_block(block);
		if (!ps) return set_error(BeamformerLibErrorKind_InvalidComputeStage);
		BfIngestArgs a{};
		a.raw = raw; a.out = d.rf[slot].ptr;
		a.channel_mapping = (const int16_t *)ps->mapping.ptr;
		a.in_row_bytes = in_row; a.out_row_bytes = out_row; a.channels = bp.channel_count;
		a.a1s2 = a1s2; a.base = bf_kind_base[pb.data_kind];
		a.a1s2_scalars = bp.sample_count * (uint32_t)bf_kind_element_count[pb.data_kind];
		ok &= HIP_OK(bf_launch_ingest(&a, s));
		if (zero_copy) {                   /* the pinned slot is free again once this kernel has run */
			ok &= HIP_OK(hipEventRecord(u.copied, s));
			u.copy_pending = true;
		}
	}
	if (!ok) return set_error(BeamformerLibErrorKind_InvalidAccess);

	double now = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
	if (c.last_push_time > 0) {
		if (c.rf_time_deltas.size() >= 32) c.rf_time_deltas.erase(c.rf_time_deltas.begin());
		c.rf_time_deltas.push_back((float)(now - c.last_push_time));
	}
	c.last_push_time = now;

	if (multi && !run_peers(block, borrowed ? data : d.rf[slot].ptr, rf_size, slot)) return false;
	bool done = borrowed ? run_frame(block, data, (int64_t)rf_size, true)
	                     : run_frame(block, d.rf[slot].ptr, (int64_t)d.rf[slot].size, true);
	if (overlap) { u.consume_pending = HIP_OK(hipEventRecord(u.consumed, s)); u.unfenced_reader = false; }
	else         { u.consume_pending = false; u.unfenced_reader = true; }
	/* a caller's device buffer read in place: the contract lets the caller overwrite it from work enqueued
	 * later on the library's stream, so that stream also waits for the peer copies out of it */
	if (multi && borrowed)
		for (uint32_t i = 1; i < c.device_count; i++) (void)hipStreamWaitEvent(s, c.devices[i].rf_landed[slot], 0);
	return done;
}

/* the reference waits on futex locks with a timeout (lib .c:192-198, :679);
 * (uint32_t)-1 blocks forever */
bool wait_for_frames(int32_t timeout_ms)
{
	Context &c = g_context;
	if (!c.device_ready) return true;
	auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms < 0 ? 0 : timeout_ms);
	bool ok = true;
	for (uint32_t i = 0; i < c.device_count && ok; i++) {
		Device &d = c.devices[i];
		if (!HIP_OK(hipSetDevice(d.device))) { ok = set_error(BeamformerLibErrorKind_InvalidAccess); break; }
		if (timeout_ms < 0) { ok = HIP_OK(hipStreamSynchronize(d.stream)) || set_error(BeamformerLibErrorKind_InvalidAccess); continue; }
		for (;;) {
			hipError_t e = hipStreamQuery(d.stream);
			if (e == hipSuccess) break;
			if (e != hipErrorNotReady) { ok = set_error(BeamformerLibErrorKind_InvalidAccess); break; }
			if (std::chrono::steady_clock::now() >= deadline) { ok = set_error(BeamformerLibErrorKind_SyncVariable); break; }
			std::this_thread::sleep_for(std::chrono::microseconds(50));
		}
	}
	(void)hipSetDevice(c.devices[0].device);
	c.cur = &c.devices[0];
	return ok;
}

/* BeamformerExportKind_BeamformedData (beamformer_core.c:1474-1494) */
bool export_last_frames(void *out, uint64_t out_size, uint32_t count, int32_t timeout_ms)
{
	Context &c = g_context;
	Device  &d = c.devices[0];
	if (!wait_for_frames(timeout_ms)) return false;
	if (d.frame_counter == 0) return set_error(BeamformerLibErrorKind_InvalidAccess);
	uint64_t req = count < 1 ? 1 : count;
	if (req > d.frame_counter) req = d.frame_counter;
	if (req > d.frames.size()) req = d.frames.size();
	uint64_t index = d.frame_counter - req, exported = 0;
	bool ok = true;
	if (c.device_count == 1) {
		for (uint64_t n = 0; n < req; n++, index++) {
			const FrameRecord &f = d.frames[index % d.frames.size()];
			if (f.bytes && exported + f.bytes <= out_size) {
				ok &= HIP_OK(hipMemcpyAsync((char *)out + exported, (const char *)d.ring.ptr + f.offset, f.bytes,
				                            hipMemcpyDeviceToHost, d.stream));
				exported += f.bytes;
			}
		}
		ok &= HIP_OK(hipStreamSynchronize(d.stream));
		return ok || set_error(BeamformerLibErrorKind_InvalidAccess);
	}
	/* several devices: every frame id exists on each of them as one z-slab (devices run in lockstep);
	 * the caller sees whole frames, slabs stitched in z order, each frame rounded to 64 bytes exactly
	 * as a single device would have exported it */
	for (uint64_t n = 0; n < req; n++, index++) {
		uint64_t voxels = 0, elem = 0; bool valid = true;
		for (uint32_t i = 0; i < c.device_count; i++) {
			const FrameRecord &f = c.devices[i].frames[index % c.devices[i].frames.size()];
			if (f.id != (uint32_t)index) { valid = false; break; }
			uint64_t v = (uint64_t)f.points[0] * f.points[1] * f.points[2];
			if (v && !f.bytes) { valid = false; break; }              /* storage reused by a newer frame */
			voxels += v; if (v) elem = (uint64_t)bf_kind_byte_size[f.data_kind];
		}
		uint64_t whole = round_up(voxels * elem, 64);
		if (!valid || !whole || exported + whole > out_size) continue;
		uint64_t at = exported;
		for (uint32_t i = 0; i < c.device_count; i++) {
			Device &p = c.devices[i];
			const FrameRecord &f = p.frames[index % p.frames.size()];
			uint64_t bytes = (uint64_t)f.points[0] * f.points[1] * f.points[2] * elem;
			if (!bytes) continue;
			ok &= HIP_OK(hipSetDevice(p.device));
			ok &= HIP_OK(hipMemcpyAsync((char *)out + at, (const char *)p.ring.ptr + f.offset, bytes, hipMemcpyDeviceToHost, p.stream));
			at += bytes;
		}
		if (at < exported + whole) std::memset((char *)out + at, 0, exported + whole - at);   /* the rounding tail */
		exported += whole;
	}
	for (uint32_t i = 0; i < c.device_count; i++) {
		ok &= HIP_OK(hipSetDevice(c.devices[i].device));
		ok &= HIP_OK(hipStreamSynchronize(c.devices[i].stream));
	}
	(void)hipSetDevice(d.device);
	return ok || set_error(BeamformerLibErrorKind_InvalidAccess);
}

static bool timings_of(Device &d, BeamformerHipFrameTimings *out)
{
	std::memset(out, 0, sizeof(*out));
	if (d.frame_counter == 0) return set_error(BeamformerLibErrorKind_InvalidAccess);
	if (!HIP_OK(hipSetDevice(d.device)) || !HIP_OK(hipStreamSynchronize(d.stream))) return set_error(BeamformerLibErrorKind_InvalidAccess);
	TimingSlot &t = d.timing[(d.frame_counter - 1) % kTimingSlots];
	TimingSlot &e = d.timing[t.events_slot];       /* t itself, or the newest sampled frame of the same plan */
	out->stage_count = t.count;
	for (uint32_t i = 0; i < t.count; i++) {
		out->stage_kind[i] = t.kinds[i];
		float ms = 0;
		if (HIP_OK(hipEventElapsedTime(&ms, e.events[i], e.events[i + 1]))) out->stage_ms[i] = ms;
	}
	float total = 0;
	if (t.count && HIP_OK(hipEventElapsedTime(&total, e.events[0], e.events[t.count]))) out->frame_ms = total;
	out->das_voxels = t.das_voxels; out->das_taps = t.das_taps;
	out->das_sample_bytes = t.das_sample_bytes; out->das_path = t.das_path;
	if (t.counted && d.pair_counter.ptr) {
		unsigned long long n = 0;
		(void)hipMemcpy(&n, (unsigned long long *)d.pair_counter.ptr + ((d.frame_counter - 1) % kTimingSlots),
		                sizeof(n), hipMemcpyDeviceToHost);
		out->das_pairs = n;
	}
	return true;
}

/* the newest frame as one device saw it (its slab, its events) */
bool device_frame_timings(uint32_t device_index, BeamformerHipFrameTimings *out)
{
	Context &c = g_context;
	std::memset(out, 0, sizeof(*out));
	if (!c.device_ready || device_index >= c.device_count) return set_error(BeamformerLibErrorKind_InvalidAccess);
	bool ok = timings_of(c.devices[device_index], out);
	(void)hipSetDevice(c.devices[0].device);
	return ok;
}

/* the newest frame: stage times of the ingest device; with several devices the voxel and pair counts
 * are those of the whole frame and the frame time is the slowest device's */
bool last_frame_timings(BeamformerHipFrameTimings *out)
{
	Context &c = g_context;
	std::memset(out, 0, sizeof(*out));
	if (!c.device_ready) return set_error(BeamformerLibErrorKind_InvalidAccess);
	if (!timings_of(c.devices[0], out)) return false;
	for (uint32_t i = 1; i < c.device_count; i++) {
		BeamformerHipFrameTimings peer;
		if (!timings_of(c.devices[i], &peer)) { (void)hipSetDevice(c.devices[0].device); return false; }
		out->das_voxels += peer.das_voxels; out->das_pairs += peer.das_pairs;
		if (peer.frame_ms > out->frame_ms) out->frame_ms = peer.frame_ms;
	}
	(void)hipSetDevice(c.devices[0].device);
	return true;
}

/* BeamformerComputeStatsTable (beamformer_compute_stats.c:3-10) as coalesce_timing_table
 * (beamformer_core.c:1683-1747) fills it: seconds per planned stage for the last 32 frames; with
 * several devices each entry is the slowest device's (they run side by side) */
bool fill_stats_table(BeamformerComputeStatsTable *out)
{
	Context &c = g_context;
	std::memset(out, 0, sizeof(*out));
	if (!c.device_ready || c.devices[0].frame_counter == 0) return true;
	for (uint32_t dev = 0; dev < c.device_count; dev++) {
		Device &d = c.devices[dev];
		if (!HIP_OK(hipSetDevice(d.device)) || !HIP_OK(hipStreamSynchronize(d.stream))) {
			(void)hipSetDevice(c.devices[0].device);
			return set_error(BeamformerLibErrorKind_InvalidAccess);
		}
		uint64_t frames = d.frame_counter < kTimingSlots ? d.frame_counter : kTimingSlots;
		for (uint64_t n = 0; n < frames; n++) {
			uint64_t id = d.frame_counter - frames + n;
			TimingSlot &t = d.timing[id % kTimingSlots];
			TimingSlot &e = d.timing[t.events_slot];
			uint32_t col = 0;
			for (uint32_t i = 0; i < t.count; i++) {
				if (t.kinds[i] == kStageIngest || t.kinds[i] == kStagePairCount) continue;
				if (col >= BeamformerMaxComputeShaderStages) break;
				float ms = 0;
				(void)hipEventElapsedTime(&ms, e.events[i], e.events[i + 1]);
				float &cell = out->times[id % 32][col];
				if (ms * 1e-3f > cell) cell = ms * 1e-3f;
				if (dev == 0 && n == frames - 1) out->shader_ids[col] = t.kinds[i];
				col++;
			}
			if (dev == 0 && n == frames - 1) out->shader_count = col;
		}
	}
	(void)hipSetDevice(c.devices[0].device);
	for (size_t i = 0; i < c.rf_time_deltas.size() && i < 32; i++) out->rf_time_deltas[i] = c.rf_time_deltas[i];
	return true;
}

/* byte offset of device `dev`'s slab inside the stitched newest frame, and the frame's total bytes */
static bool newest_layout(Context &c, uint64_t offsets[kMaxDevices], uint64_t per_voxel, uint64_t &total)
{
	total = 0;
	for (uint32_t i = 0; i < c.device_count; i++) {
		Device &p = c.devices[i];
		if (p.frame_counter == 0) return false;
		const FrameRecord &f = p.frames[(p.frame_counter - 1) % p.frames.size()];
		offsets[i] = total;
		total += (uint64_t)f.points[0] * f.points[1] * f.points[2] * per_voxel;
	}
	return true;
}

bool frame_min_max(float out[2])
{
	Context &c = g_context;
	if (!c.device_ready || c.devices[0].frame_counter == 0) return set_error(BeamformerLibErrorKind_InvalidAccess);
	bool ok = true, any = false;
	float lo = 0.f, hi = 0.f;
	for (uint32_t i = 0; i < c.device_count && ok; i++) {
		Device &d = c.devices[i];
		const FrameRecord &f = d.frames[(d.frame_counter - 1) % d.frames.size()];
		uint64_t voxels = (uint64_t)f.points[0] * f.points[1] * f.points[2];
		if (!voxels) continue;
		ok &= HIP_OK(hipSetDevice(d.device));
		if (!ok || !d.minmax_scratch.ensure(sizeof(float) * (2 * 1024 + 2))) { ok = false; break; }
		float *scratch = (float *)d.minmax_scratch.ptr;
		float part[2];
		ok &= HIP_OK(bf_launch_min_max((const char *)d.ring.ptr + f.offset, voxels,
		                               f.data_kind == BeamformerDataKind_Float32Complex, scratch + 2, scratch, d.stream));
		ok &= HIP_OK(hipMemcpyAsync(part, scratch, 2 * sizeof(float), hipMemcpyDeviceToHost, d.stream));
		ok &= HIP_OK(hipStreamSynchronize(d.stream));
		/* the two-float combine across slabs; NaN voxels propagate as in the one-device reduction */
		if (!any) { lo = part[0]; hi = part[1]; any = true; }
		else {
			lo = (part[0] < lo || part[0] != part[0]) ? part[0] : lo;
			hi = (part[1] > hi || part[1] != part[1]) ? part[1] : hi;
		}
	}
	(void)hipSetDevice(c.devices[0].device);
	out[0] = lo; out[1] = hi;
	return (ok && any) || set_error(BeamformerLibErrorKind_InvalidAccess);
}

/* Rolling average of the `count` newest frames as the reference's Sum stage specifies it
 * (beamformer_core.c:1417-1448 + shaders/sum.glsl): cleared output, then one
 * out += (1/count) * frame pass per frame, oldest first.  The reference's planner drops
 * Sum from every pipeline (beamformer_core.c:632-637), so this is reachable only through
 * the extension and never changes what get_last_frames returns.  With several devices each
 * averages its own slab and the slabs are stitched in the caller's buffer. */
bool sum_last_frames(uint32_t count, void *out, uint64_t out_size)
{
	Context &c = g_context;
	if (!c.device_ready || c.devices[0].frame_counter == 0 || count == 0) return set_error(BeamformerLibErrorKind_InvalidAccess);
	uint64_t offsets[kMaxDevices], total = 0;
	{
		const Device &d0 = c.devices[0];
		const FrameRecord &f0 = d0.frames[(d0.frame_counter - 1) % d0.frames.size()];
		if (!newest_layout(c, offsets, (uint64_t)bf_kind_byte_size[f0.data_kind], total)) return set_error(BeamformerLibErrorKind_InvalidAccess);
	}
	if (out_size < round_up(total, 64)) return set_error(BeamformerLibErrorKind_ExportSpaceOverflow);
	bool ok = true;
	for (uint32_t i = 0; i < c.device_count && ok; i++) {
		Device &d = c.devices[i];
		if (count > d.frame_counter || count > d.frames.size()) return set_error(BeamformerLibErrorKind_InvalidAccess);
		const FrameRecord &newest = d.frames[(d.frame_counter - 1) % d.frames.size()];
		uint64_t slab_bytes = (uint64_t)newest.points[0] * newest.points[1] * newest.points[2] * (uint64_t)bf_kind_byte_size[newest.data_kind];
		if (!slab_bytes) continue;
		for (uint64_t id = d.frame_counter - count; id < d.frame_counter; id++) {
			const FrameRecord &f = d.frames[id % d.frames.size()];
			if (!f.bytes || f.bytes != newest.bytes || f.data_kind != newest.data_kind ||
			    f.points[0] != newest.points[0] || f.points[1] != newest.points[1] || f.points[2] != newest.points[2])
				return set_error(BeamformerLibErrorKind_DataSizeMismatch);
		}
		ok &= HIP_OK(hipSetDevice(d.device));
		if (!ok || !d.sum_scratch.ensure(newest.bytes)) { ok = false; break; }
		ok &= HIP_OK(hipMemsetAsync(d.sum_scratch.ptr, 0, newest.bytes, d.stream));
		float prescale = 1.0f / (float)count;
		for (uint64_t id = d.frame_counter - count; ok && id < d.frame_counter; id++) {
			const FrameRecord &f = d.frames[id % d.frames.size()];
			ok &= HIP_OK(bf_launch_sum(d.sum_scratch.ptr, (const char *)d.ring.ptr + f.offset, prescale, f.bytes, d.stream));
		}
		/* one device: the whole 64-byte-rounded frame, as before; several: the slab's own bytes */
		uint64_t copy = c.device_count == 1 ? newest.bytes : slab_bytes;
		ok &= HIP_OK(hipMemcpyAsync((char *)out + offsets[i], d.sum_scratch.ptr, copy, hipMemcpyDeviceToHost, d.stream));
	}
	for (uint32_t i = 0; i < c.device_count; i++) {
		ok &= HIP_OK(hipSetDevice(c.devices[i].device));
		ok &= HIP_OK(hipStreamSynchronize(c.devices[i].stream));
	}
	(void)hipSetDevice(c.devices[0].device);
	return ok || set_error(BeamformerLibErrorKind_InvalidAccess);
}

/* Display intensities of the newest frame (render_3d.frag.glsl:50-73), one float per voxel. */
bool display_last_frame(float threshold_db, float gamma, float db_cutoff, float *out, uint64_t out_floats)
{
	Context &c = g_context;
	if (!c.device_ready || c.devices[0].frame_counter == 0) return set_error(BeamformerLibErrorKind_InvalidAccess);
	uint64_t offsets[kMaxDevices], total = 0;
	if (!newest_layout(c, offsets, 1, total)) return set_error(BeamformerLibErrorKind_InvalidAccess);
	if (out_floats < total) return set_error(BeamformerLibErrorKind_ExportSpaceOverflow);
	bool ok = true;
	for (uint32_t i = 0; i < c.device_count && ok; i++) {
		Device &d = c.devices[i];
		const FrameRecord &f = d.frames[(d.frame_counter - 1) % d.frames.size()];
		uint64_t voxels = (uint64_t)f.points[0] * f.points[1] * f.points[2];
		if (!voxels) continue;
		ok &= HIP_OK(hipSetDevice(d.device));
		if (!ok || !d.sum_scratch.ensure(voxels * sizeof(float))) { ok = false; break; }
		ok &= HIP_OK(bf_launch_display((const char *)d.ring.ptr + f.offset, voxels, f.data_kind == BeamformerDataKind_Float32Complex,
		                               threshold_db, gamma, db_cutoff, (float *)d.sum_scratch.ptr, d.stream));
		ok &= HIP_OK(hipMemcpyAsync(out + offsets[i], d.sum_scratch.ptr, voxels * sizeof(float), hipMemcpyDeviceToHost, d.stream));
	}
	for (uint32_t i = 0; i < c.device_count; i++) {
		ok &= HIP_OK(hipSetDevice(c.devices[i].device));
		ok &= HIP_OK(hipStreamSynchronize(c.devices[i].stream));
	}
	(void)hipSetDevice(c.devices[0].device);
	return ok || set_error(BeamformerLibErrorKind_InvalidAccess);
}

} // namespace bf
