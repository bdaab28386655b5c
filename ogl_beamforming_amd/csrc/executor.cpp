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
		if (index != 0 && (!HIP_OK(hipEventCreate(&d.peer_copy_begin[k])) || !HIP_OK(hipEventCreate(&d.peer_copy_end[k])))) return false;
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
	/* peers copy the RF from the ingest device: ask whether each can reach it directly (xGMI) and enable that; a refusal is not an
	 * error -- hipMemcpyPeerAsync then stages the copy through host memory -- but it is remembered and reported
	 * (beamformer_hip_get_device_info), because a scaling run that silently went through the host explains nothing */
	for (uint32_t i = 0; i < n && ok; i++) c.devices[i].peer_access = 2;
	for (uint32_t i = 1; i < n && ok; i++) {
		if (c.devices[i].device == c.devices[0].device) continue;
		int can = 0;
		bool direct = HIP_OK(hipDeviceCanAccessPeer(&can, c.devices[i].device, c.devices[0].device)) && can;
		if (direct) {
			(void)hipSetDevice(c.devices[i].device);
			hipError_t e = hipDeviceEnablePeerAccess(c.devices[0].device, 0);
			direct = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
			/* and the other way (the ingest device's stream waits on the peer's events; some runtimes want both directions mapped) */
			(void)hipSetDevice(c.devices[0].device);
			hipError_t back = hipDeviceEnablePeerAccess(c.devices[i].device, 0);
			direct = direct && (back == hipSuccess || back == hipErrorPeerAccessAlreadyEnabled);
		}
		(void)hipGetLastError();
		c.devices[i].peer_access = direct ? 1 : 0;
		if (!direct) std::fprintf(stderr, "[beamformer] device %d has no peer access to device %d: RF copies to it are staged through host memory\n",
		                          c.devices[i].device, c.devices[0].device);
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
		if (d.peer_copy_begin[k]) (void)hipEventDestroy(d.peer_copy_begin[k]);
		if (d.peer_copy_end[k])   (void)hipEventDestroy(d.peer_copy_end[k]);
		d.rf_landed[k] = d.rf_consumed[k] = d.peer_copy_begin[k] = d.peer_copy_end[k] = nullptr; d.consumed_pending[k] = false;
	}
	for (auto &b : d.rf) b.release();
	for (auto &b : d.scratch) b.release();
	d.ring.release(); d.pair_counter.release(); d.minmax_scratch.release(); d.sum_scratch.release();
	d.hercules_table.release(); d.hercules_pairs.release(); d.staged_tables.release(); d.staged_violations.release();
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
	d.last_rf = nullptr; d.last_rf_bytes = 0; d.last_rf_slot = 0; d.peer_access = 2;
	d.device = -1;
}

void shutdown_device()
{
	Context &c = g_context;
	if (!c.device_ready) return;
	for (uint32_t i = 0; i < kMaxDevices; i++) release_one_device(c.devices[i]);
	c.device_ready = false; c.device_count = 1; c.cur = &c.devices[0];
	c.push_sequence = 0;
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
	ps.transmit_table = build_transmit_table(pb);
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
	ps.das_parts.clear();
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
	f->data_kind = kind; f->id = (uint32_t)id; f->block = block; f->failed = false;
	d.ring_next_offset += bytes;
	return f;
}

/* A timed HIP event costs ~4 us of stream time on this runtime (measured: a 0.26 MB / 256 x 256
 * frame takes 36.5 us with its five records and 15.7 us without), nothing next to a 3-D volume
 * and more than the kernels of a real-time 2-D frame.  Small frames therefore record their
 * per-stage events on one frame in kTimingSamplePeriod; the frames in between run with no event
 * at all and report the newest sampled timings in the stats table. */
constexpr uint64_t kTimingSamplePeriod = 8;
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
	t.failed = false;
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
	t.failed = false;
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

			/* which kernel, with which geometry: one table of rules (das_select.cpp), computed once per plan / shard / path mode / hook
			 * change and reused by every frame after it.  Usually ONE launch; where a term of the frame can reach an end of its RF row
			 * the z range is cut and the planes concerned go to the kernel behind the staged one (decide_das_parts, das_exact.h). */
			std::vector<DasDecision> &parts = ps->das_parts;
			if (parts.empty() || !parts[0].valid || parts[0].generation != ps->generation || ps->das_z_first != zfirst || ps->das_z_count != zcount ||
			    parts[0].mode_asked != c.das_path_mode || parts[0].hooks_version != hooks().version) {
				decide_das_parts(pb, plan, ps->transmit_table, zfirst, zcount, c.das_path_mode, parts);
				for (DasDecision &dd : parts) { dd.generation = ps->generation; dd.mode_asked = c.das_path_mode; }
				ps->das_z_first = zfirst; ps->das_z_count = zcount;
			}
			const DasDecision &head = main_part(parts);
			const uint32_t ext[3] = {head.a.size[0], head.a.size[1], zcount};
			das_path = (uint32_t)(head.path == DasPath_Zero ? DasPath_General : head.path);
			uint64_t violations_slot = ~0ull;
			uint32_t *frame_counters = nullptr;        /* [0] staged window violations, [1] / [2] das_tile.hip's staged / gathered chunks */
			for (const DasDecision &dd : parts)
				if (dd.path == DasPath_Staged || dd.path == DasPath_Tile) {
					if (d.staged_violations.ensure(sizeof(uint32_t) * 4 * kTimingSlots)) {
						frame_counters = (uint32_t *)d.staged_violations.ptr + 4 * (f->id % kTimingSlots);
						ok &= HIP_OK(hipMemsetAsync(frame_counters, 0, 4 * sizeof(uint32_t), s));
						violations_slot = f->id % kTimingSlots;
					} else ok = false;
					break;
				}
			if (c.count_pairs) {
				ok &= d.pair_counter.ensure(sizeof(unsigned long long) * (kTimingSlots + 2));
				if (ok) ok &= HIP_OK(hipMemsetAsync((unsigned long long *)d.pair_counter.ptr + (f->id % kTimingSlots), 0, sizeof(unsigned long long), s));
			}
			const uint64_t plane_bytes = (uint64_t)head.a.size[0] * head.a.size[1] * (plan.iq_pipeline ? 8u : 4u);

			for (const DasDecision &dd : parts) {
			BfDasArgs a = dd.a;
			a.rf  = cur;
			a.out = (char *)d.ring.ptr + f->offset + (uint64_t)(dd.z_first - zfirst) * plane_bytes;
			a.transmits       = (const BfTransmit *)ps->transmits.ptr;
			a.sparse_elements = (const int16_t *)ps->sparse.ptr;
			a.readi_hadamard  = (const uint16_t *)ps->readi_hadamard.ptr;
			uint32_t part_path = (uint32_t)dd.path;

			if (dd.path == DasPath_Zero) {
				ok &= HIP_OK(hipMemsetAsync(a.out, 0, dd.z_count * plane_bytes, s));
			} else {
				/* 64 zero bytes right behind the DAS input (every buffer it can live in is allocated with that much slack): the
				 * gather target of out-of-range lanes */
				const uint64_t used = dd.das_input_bytes;
				if (dd.path != DasPath_General) ok &= HIP_OK(hipMemsetAsync((char *)const_cast<void *>(cur) + used, 0, 64, s));
				switch (dd.path) {
				case DasPath_Staged:
				case DasPath_Gather:{
					BfSeparableArgs sep = dd.sep;
					bool staged = dd.path == DasPath_Staged;
					if (staged && sep.uniform) {
						/* the wave-uniform transmit tables live in global memory: one slice per (lateral tile row, plane), written per frame;
						 * no memory: the shape with the tables in LDS, else the gather kernel with its own geometry */
						const uint64_t table_bytes = (uint64_t)sep.table_stride * sep.tiles[1] * sep.tiles[2];
						if (d.staged_tables.ensure(table_bytes)) {
							sep.tables = d.staged_tables.ptr;
							ok &= HIP_OK(bf_launch_das_staged_tables(&a, &sep, s));
						} else if (dd.has_lds_tables) {
							sep = dd.sep_lds_tables;
						} else {
							sep = dd.sep_gather; staged = false;
						}
					}
					if (staged) {
						/* window positions outside the staged window are counted (range-checked loop only: STAGED_CHECKED) */
						sep.violations = frame_counters;
						ok &= HIP_OK(!plan.iq_pipeline ? bf_launch_das_staged_real(&a, &sep, s) :
						             a.interpolation == 2 ? bf_launch_das_staged_cubic(&a, &sep, s) : bf_launch_das_staged(&a, &sep, s));
						part_path = DasPath_Staged;
					} else {
						ok &= HIP_OK(bf_launch_das_separable(&a, &sep, s));
						part_path = DasPath_Gather;
					}
				}break;
				case DasPath_Hercules:{
					BfHerculesArgs hq = dd.herc;
					if (d.hercules_table.ensure(((size_t)hq.table_pitch + 2) * a.size[1] * sizeof(float))) {
						hq.pairs = nullptr;
						const uint64_t prepared = used * (a.interpolation == 2 ? 4u : 2u);      /* cubic: four coefficients per sample, 32 bytes */
						if (dd.hercules_prepared && d.hercules_pairs.ensure(prepared + 64)) {
							hq.pairs = d.hercules_pairs.ptr;
							hq.zero_offset = (uint32_t)prepared;
						}
						hq.table    = (float *)d.hercules_table.ptr;
						hq.extremes = hq.table + (size_t)hq.table_pitch * a.size[1];
						ok &= HIP_OK(bf_launch_das_hercules(&a, &hq, s));
					} else {
						ok &= HIP_OK(bf_launch_das(&a, s));                                     /* no memory for the row table: the general kernel */
						part_path = DasPath_General;
					}
				}break;
				case DasPath_Factored:
					ok &= HIP_OK(bf_launch_das_factored(&a, s));
					break;
				case DasPath_Tile:
					/* (block, channel chunk) pairs served from staged windows, and those the kernel sent through its gather loop: words 1, 2 */
					a.tile_counters = frame_counters ? frame_counters + 1 : nullptr;
					ok &= HIP_OK(bf_launch_das_tile(&a, s));
					break;
				default:
					ok &= HIP_OK(bf_launch_das(&a, s));
					break;
				}
			}
			if (&dd == &head && dd.path != DasPath_Zero) das_path = part_path;
			}
			if (c.count_pairs && head.path != DasPath_Zero) {
				/* geometry-only recount of the apodization test; its own segment so that it
				 * stays out of the DAS time */
				segment((uint32_t)st.kind);
				for (const DasDecision &dd : parts) {
					BfDasArgs count = dd.general;              /* the general kernel's own tiles: the specialised kernels reshape them */
					count.rf = cur; count.out = (char *)d.ring.ptr + f->offset + (uint64_t)(dd.z_first - zfirst) * plane_bytes;
					count.transmits = (const BfTransmit *)ps->transmits.ptr; count.sparse_elements = (const int16_t *)ps->sparse.ptr;
					count.readi_hadamard = (const uint16_t *)ps->readi_hadamard.ptr;
					count.pair_counter = (unsigned long long *)d.pair_counter.ptr + (f->id % kTimingSlots);
					ok &= HIP_OK(bf_launch_das_count(&count, s));
				}
				segment(kStagePairCount);
				t.counted = true;
				das_segment_done = true;
			}
			const BfDasArgs &a = head.a;
			t.das_row_end_planes = row_end_planes(parts);
			t.das_voxels = (uint64_t)ext[0] * ext[1] * ext[2];
			t.das_taps = a.interpolation == 0 ? 1 : a.interpolation == 1 ? 2 : 4;
			t.das_sample_bytes = plan.iq_pipeline ? 8 : 4;
			t.das_path = das_path; t.frame_id = f->id;
			t.violations_slot = violations_slot;
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
		/* (several devices: the ingest device holds the whole zero frame, the others an empty slab of it) */
		uint32_t points[3] = {plan.output_points[0], plan.output_points[1], c.device_count > 1 && d.index != 0 ? 0u : plan.output_points[2]};
		FrameRecord *f = next_frame(points, plan.iq_pipeline, block);
		if (!f) return set_error(BeamformerLibErrorKind_FrameSizeOverflow);
		f->timing_slot = (int)(f->id % kTimingSlots);
		if (f->bytes) ok &= HIP_OK(hipMemsetAsync((char *)d.ring.ptr + f->offset, 0, f->bytes, s));
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
		ok &= HIP_OK(hipEventRecord(p.peer_copy_begin[slot], p.peer_stream));
		ok &= HIP_OK(hipMemcpyPeerAsync(p.rf[slot].ptr, p.device, src, d0.device, rf_size, p.peer_stream));
		ok &= HIP_OK(hipEventRecord(p.peer_copy_end[slot], p.peer_stream));
		ok &= HIP_OK(hipEventRecord(p.rf_landed[slot], p.peer_stream));
		p.last_rf = p.rf[slot].ptr; p.last_rf_bytes = rf_size; p.last_rf_slot = slot;
		ok &= HIP_OK(hipStreamWaitEvent(p.stream, p.rf_landed[slot], 0));
		TimingSlot &t = p.timing[p.frame_counter % kTimingSlots];
		t.sampled = true; t.failed = false; t.events_slot = (uint32_t)(p.frame_counter % kTimingSlots);
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
	/* ONE frame id for every device of the set, taken here: whatever fails below -- a peer's replan, a peer copy, the ingest
	 * device's own stages -- the next push starts all devices on the same id again, and beamformer_get_last_frames refuses exactly
	 * the frames that are incomplete instead of skipping every frame from then on */
	const uint64_t sequence = c.push_sequence++;
	for (uint32_t i = 0; i < c.device_count; i++) c.devices[i].frame_counter = sequence;
	/* ... and a push that does not complete leaves a TOMBSTONE under its id on every device (no bytes, no voxels, no stage timings): the readers
	 * below refuse it -- "the newest frame is missing" -- instead of serving whatever record sat in that ring slot
	 * BeamformerMaxBacklogFrames pushes ago */
	struct Lockstep {
		Context &c; uint64_t id; bool complete;
		~Lockstep() {
			for (uint32_t i = 0; i < c.device_count; i++) {
				Device &p = c.devices[i];
				p.frame_counter = id + 1;
				if (complete) continue;
				FrameRecord &f = p.frames[id % p.frames.size()];
				f = FrameRecord{}; f.points[0] = f.points[1] = f.points[2] = 0; f.id = (uint32_t)id; f.failed = true;
				TimingSlot &t = p.timing[id % kTimingSlots];
				t.count = 0; t.counted = false; t.violations_slot = ~0ull; t.das_voxels = 0; t.frame_id = id; t.failed = true;
			}
		}
	} lockstep{c, sequence, false};
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
	t.failed = false;
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

	/* what beamformer_hip_get_device_info checksums later.  A caller's device buffer is only borrowed for the duration of the frame: its
	 * pointer is NOT kept -- with several devices the checksum of what the ingest device read is taken here, on the stream, while the
	 * buffer is guaranteed live; with one device a borrowed frame reports no checksum */
	d.last_rf = borrowed ? nullptr : d.rf[slot].ptr; d.last_rf_bytes = rf_size; d.last_rf_slot = slot; d.last_rf_sum_ready = false;
	if (multi && borrowed && d.pair_counter.ensure(sizeof(unsigned long long) * (kTimingSlots + 2)))
		d.last_rf_sum_ready = HIP_OK(bf_launch_rf_checksum(data, rf_size, (unsigned long long *)d.pair_counter.ptr + kTimingSlots + 1, s));
	if (multi && !run_peers(block, borrowed ? data : d.rf[slot].ptr, rf_size, slot)) return false;
	bool done = borrowed ? run_frame(block, data, (int64_t)rf_size, true)
	                     : run_frame(block, d.rf[slot].ptr, (int64_t)d.rf[slot].size, true);
	if (overlap) { u.consume_pending = HIP_OK(hipEventRecord(u.consumed, s)); u.unfenced_reader = false; }
	else         { u.consume_pending = false; u.unfenced_reader = true; }
	/* a caller's device buffer read in place: the contract lets the caller overwrite it from work enqueued
	 * later on the library's stream, so that stream also waits for the peer copies out of it */
	if (multi && borrowed)
		for (uint32_t i = 1; i < c.device_count; i++) (void)hipStreamWaitEvent(s, c.devices[i].rf_landed[slot], 0);
	lockstep.complete = done;
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

/* The record of device d's newest frame -- or null when the newest push did not complete (its slot holds a tombstone) or the slot still
 * holds an older frame's record: every reader of "the last frame" goes through here and fails instead of serving a stale record. */
const FrameRecord *newest_record(const Device &d)
{
	if (d.frame_counter == 0 || d.frames.empty()) return nullptr;
	const uint64_t id = d.frame_counter - 1;
	const FrameRecord &f = d.frames[id % d.frames.size()];
	return (f.id == (uint32_t)id && !f.failed) ? &f : nullptr;
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
			const bool present = f.id == (uint32_t)index && !f.failed && f.bytes;
			/* older frames that are gone (storage reused, a push that failed) are skipped; the NEWEST one missing is an error, not a
			 * success that leaves the caller's buffer unwritten */
			if (!present) { if (n + 1 == req) ok = false; continue; }
			if (exported + f.bytes <= out_size) {
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
			if (f.id != (uint32_t)index || f.failed) { valid = false; break; }
			uint64_t v = (uint64_t)f.points[0] * f.points[1] * f.points[2];
			if (v && !f.bytes) { valid = false; break; }              /* storage reused by a newer frame */
			voxels += v; if (v) elem = (uint64_t)bf_kind_byte_size[f.data_kind];
		}
		uint64_t whole = round_up(voxels * elem, 64);
		if (!valid || !whole) {
			/* a device of the set holds no complete slab of this frame (a push that failed half way, storage reused): older frames
			 * are skipped as in the one-device export; the NEWEST frame missing is an error, not an unwritten buffer */
			if (n + 1 == req) { ok = false; break; }
			continue;
		}
		if (exported + whole > out_size) continue;
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
	if (t.failed) return set_error(BeamformerLibErrorKind_InvalidAccess);          /* the newest push did not complete: no timings of an older frame in its place */
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
	out->das_row_end_planes = t.das_row_end_planes;
	if (t.violations_slot != ~0ull && d.staged_violations.ptr) {
		uint32_t n[4] = {0, 0, 0, 0};
		(void)hipMemcpy(n, (uint32_t *)d.staged_violations.ptr + 4 * t.violations_slot, sizeof(n), hipMemcpyDeviceToHost);
		out->staged_window_violations = n[0];
		out->tile_staged_chunks = n[1]; out->tile_gather_chunks = n[2];
	}
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

bool device_info(uint32_t device_index, BeamformerHipDeviceInfo *out)
{
	Context &c = g_context;
	std::memset(out, 0, sizeof(*out));
	if (!c.device_ready || device_index >= c.device_count) return set_error(BeamformerLibErrorKind_InvalidAccess);
	Device &d = c.devices[device_index];
	out->ordinal = d.device; out->peer_access = d.peer_access;
	out->slab_first = d.slab_first; out->slab_count = d.slab_count;
	bool ok = HIP_OK(hipSetDevice(d.device)) && HIP_OK(hipStreamSynchronize(d.stream));
	if (ok && d.frame_counter) {
		BeamformerHipFrameTimings t;
		if (timings_of(d, &t)) {
			out->frame_ms = t.frame_ms;
			for (uint32_t i = 0; i < t.stage_count; i++) if (t.stage_kind[i] == (uint32_t)BeamformerShaderKind_DAS) out->das_ms = t.stage_ms[i];
		}
		if (device_index != 0 && d.peer_copy_end[d.last_rf_slot]) {
			float ms = 0;
			if (HIP_OK(hipStreamSynchronize(d.peer_stream)) && HIP_OK(hipEventElapsedTime(&ms, d.peer_copy_begin[d.last_rf_slot], d.peer_copy_end[d.last_rf_slot]))) out->peer_copy_ms = ms;
			(void)hipGetLastError();
		}
		if (d.last_rf && d.last_rf_bytes) {
			ok = d.pair_counter.ensure(sizeof(unsigned long long) * (kTimingSlots + 2));
			unsigned long long *sum = ok ? (unsigned long long *)d.pair_counter.ptr + kTimingSlots : nullptr;
			unsigned long long host = 0;
			ok = ok && HIP_OK(bf_launch_rf_checksum(d.last_rf, d.last_rf_bytes, sum, d.stream)) &&
			     HIP_OK(hipMemcpyAsync(&host, sum, sizeof(host), hipMemcpyDeviceToHost, d.stream)) && HIP_OK(hipStreamSynchronize(d.stream));
			out->rf_checksum = host; out->rf_bytes = d.last_rf_bytes;
		} else if (d.last_rf_sum_ready && d.pair_counter.ptr) {
			/* a borrowed device buffer: summed inside the push (the caller may have freed it since) */
			unsigned long long host = 0;
			ok = HIP_OK(hipMemcpyAsync(&host, (unsigned long long *)d.pair_counter.ptr + kTimingSlots + 1, sizeof(host), hipMemcpyDeviceToHost, d.stream)) &&
			     HIP_OK(hipStreamSynchronize(d.stream));
			out->rf_checksum = host; out->rf_bytes = d.last_rf_bytes;
		}
	}
	(void)hipSetDevice(c.devices[0].device);
	return ok || set_error(BeamformerLibErrorKind_InvalidAccess);
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
		out->staged_window_violations += peer.staged_window_violations;
		out->tile_staged_chunks += peer.tile_staged_chunks; out->tile_gather_chunks += peer.tile_gather_chunks;
		out->das_row_end_planes += peer.das_row_end_planes;
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
			if (t.failed) continue;                    /* a push that did not complete: its row stays zero */
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
		const FrameRecord *fp = newest_record(p);
		if (!fp) return false;
		const FrameRecord &f = *fp;
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
		const FrameRecord *fp = newest_record(d);
		if (!fp) { ok = false; break; }
		const FrameRecord &f = *fp;
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
		const FrameRecord *f0 = newest_record(c.devices[0]);
		if (!f0 || !newest_layout(c, offsets, (uint64_t)bf_kind_byte_size[f0->data_kind], total)) return set_error(BeamformerLibErrorKind_InvalidAccess);
	}
	if (out_size < round_up(total, 64)) return set_error(BeamformerLibErrorKind_ExportSpaceOverflow);
	bool ok = true;
	for (uint32_t i = 0; i < c.device_count && ok; i++) {
		Device &d = c.devices[i];
		if (count > d.frame_counter || count > d.frames.size()) return set_error(BeamformerLibErrorKind_InvalidAccess);
		const FrameRecord &newest = *newest_record(d);          /* (present: newest_layout checked every device) */
		uint64_t slab_bytes = (uint64_t)newest.points[0] * newest.points[1] * newest.points[2] * (uint64_t)bf_kind_byte_size[newest.data_kind];
		if (!slab_bytes) continue;
		for (uint64_t id = d.frame_counter - count; id < d.frame_counter; id++) {
			const FrameRecord &f = d.frames[id % d.frames.size()];
			if (f.id != (uint32_t)id || f.failed || !f.bytes || f.bytes != newest.bytes || f.data_kind != newest.data_kind ||
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
		const FrameRecord &f = *newest_record(d);               /* (present: newest_layout checked every device) */
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
