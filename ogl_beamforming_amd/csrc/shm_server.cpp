/* shm_server.cpp -- headless server for the reference's shared-memory protocol v33.
 *
 * lets the UNMODIFIED reference client library
 * (lib/ogl_beamformer_lib.c, and with it tests/throughput.c, tests/decode.c, the MATLAB and
 * cffi bindings) drive the MI355X backend.  The process creates the region the reference's
 * main_linux.c:189-204 creates ("/ogl_beamformer_shared_memory", 2 GiB), initialises the
 * header as beamformer_init does (beamformer.c:249-263) and then plays both server workers:
 *   upload worker  (beamformer_rf_upload, beamformer_core.c:1756-1805): when the client
 *                  holds UploadRF and has published rf_block_rf_size, copy the channel-mapped
 *                  RF out of the scratch area into a 3-slot device ring, release the locks;
 *   compute worker (complete_queue, beamformer_core.c:1456-1681): pop work items from the
 *                  SPSC queue -- CreateFilter, Compute(Indirect), ExportBuffer -- and serve
 *                  them through this repository's C ABI (libogl_beamformer_lib.so).
 * Layout of the region: beamformer_shared_memory.c:2-166; every offset below is asserted
 * against the compiled reference by tests/test_shm_server.py (tests/golden/shm_layout.txt).
 * Locks are the reference's futex words (util_os.c:5-26, base_linux.c:198-218).
 *
 * No UI, no file watching, no Vulkan: only the hot path's control plane.
 *
 *   ogl_beamformer_server [--name /shm_name] [--size bytes] [--once N]
 * prints one line per event on stdout ("ready", "upload ...", "compute ...", "export ...").
 */
#include <atomic>
#include <cerrno>
#include <cstdarg>
#include <csignal>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <climits>
#include <fcntl.h>
#include <linux/futex.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <hip/hip_runtime_api.h>
#include "../../include/ogl_beamformer_hip.h"

namespace {

/* ---- region layout (beamformer_shared_memory.c:40-166) ---- */
enum { WorkKind_Compute = 0, WorkKind_ComputeIndirect = 1, WorkKind_CreateFilter = 2, WorkKind_ExportBuffer = 3 };
enum { Lock_ScratchSpace = 0, Lock_UploadRF = 1, Lock_ExportSync = 2, Lock_DispatchCompute = 3, Lock_Count = 4 };
enum { Export_BeamformedData = 0, Export_Stats = 1 };
enum { Region_ComputePipeline, Region_ChannelMapping, Region_FocalVectors, Region_Parameters,
       Region_SparseElements, Region_TransmitReceiveOrientations, RegionFlag_NotifyUI };

struct ShmWork {
	int32_t kind;
	int32_t lock;
	union {
		struct { uint32_t view_plane, parameter_block; } compute;
		struct { BeamformerFilterParameters parameters; uint8_t filter_slot, parameter_block; } create_filter;
		struct { uint32_t kind, count; uint64_t size; } export_;
		uint8_t raw[32];
	};
};
struct ShmQueue { std::atomic<uint64_t> queue; ShmWork items[64]; };
struct ShmPipeline {
	int32_t  shaders[BeamformerMaxComputeShaderStages];
	uint8_t  parameters[BeamformerMaxComputeShaderStages];     /* filter slot per stage */
	uint32_t shader_count;
	int32_t  data_kind;
};
struct alignas(16) ShmBlock {
	BeamformerParameters parameters;
	std::atomic<uint32_t> region_update_flags;
	ShmPipeline pipeline;
	alignas(16) int16_t channel_mapping[BeamformerMaxChannelCount];
	alignas(16) int16_t sparse_elements[BeamformerMaxChannelCount];
	alignas(16) uint8_t transmit_receive_orientations[BeamformerMaxChannelCount];
	alignas(16) float   focal_vectors[BeamformerMaxChannelCount][2];
};
struct ShmHeader {
	uint32_t version;
	uint32_t invalid;
	int32_t  locks[Lock_Count + BeamformerMaxParameterBlocks];
	uint32_t reserved_parameter_blocks;
	std::atomic<uint64_t> rf_block_rf_size;
	uint64_t beamformed_frame_buffer_size;
	struct { uint64_t max_rf_data_size; uint8_t cuda, hilbert; } capabilities;
	BeamformerLiveImagingParameters live_imaging_parameters;
	std::atomic<uint32_t> live_imaging_dirty_flags;
	ShmQueue external_work_queue;
};
static_assert(sizeof(ShmWork) == 40 && offsetof(ShmWork, create_filter.filter_slot) == 32 && offsetof(ShmWork, export_.size) == 16, "BeamformWork");
static_assert(sizeof(ShmQueue) == 2568 && offsetof(ShmQueue, items) == 8, "BeamformWorkQueue");
static_assert(sizeof(ShmPipeline) == 88, "BeamformerComputePipeline");
static_assert(sizeof(ShmBlock) == 3696 && offsetof(ShmBlock, region_update_flags) == 264 && offsetof(ShmBlock, pipeline) == 268 &&
              offsetof(ShmBlock, channel_mapping) == 368 && offsetof(ShmBlock, sparse_elements) == 880 &&
              offsetof(ShmBlock, transmit_receive_orientations) == 1392 && offsetof(ShmBlock, focal_vectors) == 1648, "BeamformerParameterBlock");
static_assert(sizeof(ShmHeader) == 2912 && offsetof(ShmHeader, locks) == 8 && offsetof(ShmHeader, reserved_parameter_blocks) == 88 &&
              offsetof(ShmHeader, rf_block_rf_size) == 96 && offsetof(ShmHeader, beamformed_frame_buffer_size) == 104 &&
              offsetof(ShmHeader, capabilities) == 112 && offsetof(ShmHeader, live_imaging_parameters) == 128 &&
              offsetof(ShmHeader, live_imaging_dirty_flags) == 336 && offsetof(ShmHeader, external_work_queue) == 344, "BeamformerSharedMemory");
constexpr uint64_t kArenaHeaderBytes = 96;          /* sizeof(Arena) in the reference (util.c:152-179) */

volatile std::sig_atomic_t g_stop = 0;
void on_signal(int) { g_stop = 1; }

/* ---- futex locks (util_os.c:5-26, base_linux.c:198-218) ---- */
bool wait_on_address(int32_t *value, int32_t current, uint32_t timeout_ms)
{
	timespec ts, *timeout = nullptr;
	if (timeout_ms != (uint32_t)-1) { ts.tv_sec = timeout_ms / 1000; ts.tv_nsec = (timeout_ms % 1000) * 1000000L; timeout = &ts; }
	return syscall(SYS_futex, value, FUTEX_WAIT, current, timeout, 0, 0) == 0;
}
bool take_lock(int32_t *lock, int32_t timeout_ms)
{
	for (;;) {
		int32_t current = 0;
		if (__atomic_compare_exchange_n(lock, &current, 1, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) return true;
		if (!timeout_ms) return false;
		if (!wait_on_address(lock, current, (uint32_t)timeout_ms) && timeout_ms != -1 && errno == ETIMEDOUT) return false;
		if (g_stop) return false;
	}
}
void release_lock(int32_t *lock)
{
	__atomic_store_n(lock, 0, __ATOMIC_SEQ_CST);
	syscall(SYS_futex, lock, FUTEX_WAKE, INT_MAX, 0, 0, 0);
}
/* post_sync_barrier (beamformer_shared_memory.c:305-314): release if held */
void post_sync(ShmHeader *sm, int lock)
{
	if (__atomic_load_n(&sm->locks[lock], __ATOMIC_SEQ_CST)) release_lock(&sm->locks[lock]);
}

struct Server {
	ShmHeader *sm = nullptr;
	uint64_t   size = 0;
	void      *rf_ring[BeamformerMaxRawDataFramesInFlight] = {};
	hipStream_t stream = nullptr;                                /* the library runs on this stream */
	hipEvent_t  read_done[BeamformerMaxRawDataFramesInFlight] = {};   /* the frame that read rf_ring[i] in place has finished */
	bool        read_pending[BeamformerMaxRawDataFramesInFlight] = {};
	uint64_t   rf_ring_bytes = 0, rf_active_size = 0;
	uint64_t   insertion_index = 0, compute_index = 0;
	uint32_t   rf_block = 0;
	bool       multi_device = false;                             /* --devices: frames spread over several GPUs */

	ShmBlock *block(uint32_t i) { return reinterpret_cast<ShmBlock *>(reinterpret_cast<uint8_t *>(sm + 1) + (size_t)i * sizeof(ShmBlock)); }
	/* beamformer_shared_memory_data_pointer (beamformer_shared_memory.c:280-297): the client
	 * re-creates an Arena header behind the last block and aligns the payload to 4 KiB */
	uint64_t payload_offset() const
	{
		uint64_t off = sizeof(ShmHeader) + (uint64_t)sm->reserved_parameter_blocks * sizeof(ShmBlock) + kArenaHeaderBytes;
		return (off + 4095) & ~4095ull;
	}
	/* bytes of the scratch arena a client may have filled / may be handed: every size that
	 * arrives through the region is checked against this before it is used */
	uint64_t payload_capacity() const { uint64_t off = payload_offset(); return off < size ? size - off : 0; }
	uint8_t *payload()
	{
		uint64_t off = sizeof(ShmHeader) + (uint64_t)sm->reserved_parameter_blocks * sizeof(ShmBlock) + kArenaHeaderBytes;
		off = (off + 4095) & ~4095ull;
		return reinterpret_cast<uint8_t *>(sm) + off;
	}
};

void say(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
void say(const char *fmt, ...)
{
	va_list ap; va_start(ap, fmt);
	std::vprintf(fmt, ap);
	va_end(ap);
	std::fputc('\n', stdout);
	std::fflush(stdout);
}

/* beamformer_commit_parameter_block (beamformer_core.c:1191-1287): move a dirty block into
 * the library.  The RF in the scratch area is already channel mapped and contrast reduced by
 * the client (lib .c:519-559), so the library sees dense rows and an identity mapping. */
bool commit_block(Server &s, uint32_t b)
{
	ShmBlock *pb = s.block(b);
	if (!pb->region_update_flags.load()) return true;
	if (!take_lock(&s.sm->locks[Lock_Count + b], -1)) return false;
	ShmBlock copy;
	std::memcpy((void *)&copy, (const void *)pb, sizeof(copy));
	pb->region_update_flags.store(0);
	release_lock(&s.sm->locks[Lock_Count + b]);

	if (s.sm->reserved_parameter_blocks <= BeamformerMaxParameterBlocks)
		beamformer_reserve_parameter_blocks(s.sm->reserved_parameter_blocks);
	BeamformerParameters bp = copy.parameters;
	bp.raw_data_dimensions[0] = bp.sample_count * bp.acquisition_count;
	bp.raw_data_dimensions[1] = bp.channel_count;
	bp.contrast_mode = BeamformerContrastMode_None;
	bool ok = true;
	ok &= beamformer_push_parameters_at(&bp, b) != 0;
	uint32_t stages = copy.pipeline.shader_count <= BeamformerMaxComputeShaderStages ? copy.pipeline.shader_count : BeamformerMaxComputeShaderStages;
	ok &= beamformer_push_pipeline_at(copy.pipeline.shaders, stages, (BeamformerDataKind)copy.pipeline.data_kind, b) != 0;
	for (uint32_t i = 0; i < stages; i++)
		ok &= beamformer_set_pipeline_stage_parameters_at(i, copy.pipeline.parameters[i], b) != 0;
	int16_t identity[BeamformerMaxChannelCount];
	for (int i = 0; i < BeamformerMaxChannelCount; i++) identity[i] = (int16_t)i;
	ok &= beamformer_push_channel_mapping_at(identity, BeamformerMaxChannelCount, b) != 0;
	ok &= beamformer_push_focal_vectors_at(&copy.focal_vectors[0][0], BeamformerMaxChannelCount, b) != 0;
	ok &= beamformer_push_sparse_elements_at(copy.sparse_elements, BeamformerMaxChannelCount, b) != 0;
	ok &= beamformer_push_transmit_receive_orientations_at(copy.transmit_receive_orientations, BeamformerMaxChannelCount, b) != 0;
	if (!ok) say("commit block %u failed: %s", b, beamformer_get_last_error_string());
	return ok;
}

/* beamformer_rf_upload (beamformer_core.c:1756-1805) */
void handle_upload(Server &s, uint64_t rf_block_rf_size)
{
	ShmHeader *sm = s.sm;
	uint32_t block = (uint32_t)(rf_block_rf_size >> 32);
	uint64_t size  = rf_block_rf_size & 0xFFFFFFFFull;
	take_lock(&sm->locks[Lock_ScratchSpace], -1);
	bool ok = size <= s.payload_capacity() && sm->reserved_parameter_blocks <= BeamformerMaxParameterBlocks;
	if (!ok) size = 0;
	uint64_t need = (size + 63) & ~63ull;
	if (need > s.rf_ring_bytes) {
		beamformer_hip_synchronize();
		for (auto &pending : s.read_pending) pending = false;
		for (auto &p : s.rf_ring) { if (p) (void)hipFree(p); p = nullptr; }
		for (auto &p : s.rf_ring) ok &= hipMalloc(&p, need + 64) == hipSuccess;
		s.rf_ring_bytes = ok ? need : 0;
	}
	uint32_t slot = (uint32_t)(s.insertion_index % BeamformerMaxRawDataFramesInFlight);
	/* the slot about to be overwritten was read in place by the frame three uploads ago, which the
	 * library runs asynchronously: wait for exactly that frame */
	if (s.read_pending[slot]) { (void)hipEventSynchronize(s.read_done[slot]); s.read_pending[slot] = false; }
	if (s.multi_device || !s.stream) beamformer_hip_synchronize();       /* no per-slot events without a stream of ours */
	if (ok) ok = hipMemcpy(s.rf_ring[slot], s.payload(), size, hipMemcpyHostToDevice) == hipSuccess;
	s.rf_active_size = size;
	s.rf_block = block;
	release_lock(&sm->locks[Lock_ScratchSpace]);
	post_sync(sm, Lock_UploadRF);
	s.insertion_index++;
	say("upload block %u bytes %llu %s", block, (unsigned long long)size, ok ? "ok" : "failed (no HIP device?)");
}

void handle_work(Server &s, ShmWork *work)
{
	ShmHeader *sm = s.sm;
	switch (work->kind) {
	case WorkKind_CreateFilter:{                                   /* beamformer_core.c:1511-1517 */
		BeamformerFilterParameters fp = work->create_filter.parameters;
		uint32_t ok = beamformer_create_filter(&fp, work->create_filter.filter_slot, work->create_filter.parameter_block);
		say("create_filter block %u slot %u %s", work->create_filter.parameter_block, work->create_filter.filter_slot, ok ? "ok" : "failed");
	}break;
	case WorkKind_Compute:
	case WorkKind_ComputeIndirect:{                                /* beamformer_core.c:1519-1677 */
		uint32_t b = work->compute.parameter_block;
		/* both values live in client-writable memory: bound the block index by the protocol's maximum as
		 * well as by the (client-written) reserved count before it indexes the lock and block arrays */
		uint32_t reserved = sm->reserved_parameter_blocks;
		if (reserved > BeamformerMaxParameterBlocks) reserved = BeamformerMaxParameterBlocks;
		bool ok = b < reserved && commit_block(s, b);
		post_sync(sm, Lock_DispatchCompute);                         /* :1533 */
		if (work->kind == WorkKind_ComputeIndirect) {
			/* :1591-1602: the RF of this frame must have been uploaded */
			while (!g_stop && s.insertion_index <= s.compute_index) {
				uint64_t sz = sm->rf_block_rf_size.exchange(0);
				if (sz) handle_upload(s, sz); else usleep(50);
			}
		}
		/* :1588-1602: indirect work consumes the next uploaded slot, a re-queued Compute re-uses the last */
		uint64_t frame = work->kind == WorkKind_ComputeIndirect ? s.compute_index : (s.compute_index ? s.compute_index - 1 : 0);
		uint32_t slot  = (uint32_t)(frame % BeamformerMaxRawDataFramesInFlight);
		if (ok && s.rf_ring[slot]) {
			ok = beamformer_hip_push_device_data_with_compute(s.rf_ring[slot], (uint32_t)s.rf_active_size, work->compute.view_plane, b) != 0;
			if (s.stream && s.read_done[slot]) s.read_pending[slot] = hipEventRecord(s.read_done[slot], s.stream) == hipSuccess;
		} else ok = false;
		if (work->kind == WorkKind_ComputeIndirect) s.compute_index++;
		say("compute block %u %s%s", b, ok ? "ok" : "failed: ", ok ? "" : beamformer_get_last_error_string());
	}break;
	case WorkKind_ExportBuffer:{                                   /* beamformer_core.c:1468-1509 */
		post_sync(sm, Lock_DispatchCompute);
		if (work->lock < 0 || work->lock >= Lock_Count + BeamformerMaxParameterBlocks) {
			say("export: lock index %d out of range", work->lock);
			post_sync(sm, Lock_ExportSync);                            /* answer the client instead of leaving it blocked */
			break;
		}
		take_lock(&sm->locks[work->lock], -1);
		bool ok = false;
		beamformer_set_global_timeout((uint32_t)-1);
		if (work->export_.size > s.payload_capacity()) {
			ok = false;
		} else if (work->export_.kind == Export_BeamformedData) {
			ok = beamformer_get_last_frames(s.payload(), work->export_.size, work->export_.count) != 0;
		} else if (work->export_.kind == Export_Stats && work->export_.size >= sizeof(BeamformerComputeStatsTable)) {
			ok = beamformer_compute_timings(reinterpret_cast<BeamformerComputeStatsTable *>(s.payload()), -1) != 0;
		}
		release_lock(&sm->locks[work->lock]);
		post_sync(sm, Lock_ExportSync);
		say("export kind %u count %u bytes %llu %s", work->export_.kind, work->export_.count,
		    (unsigned long long)work->export_.size, ok ? "ok" : "failed");
	}break;
	default: say("unknown work kind %d", work->kind); break;
	}
}

} // namespace

int main(int argc, char **argv)
{
	const char *name = "/ogl_beamformer_shared_memory";          /* base_linux.c:5 */
	uint64_t size = 2ull << 30;                                    /* main_linux.c:19 */
	long once = -1;
	bool multi_device = false;
	for (int i = 1; i < argc; i++) {
		if (!std::strcmp(argv[i], "--name") && i + 1 < argc) name = argv[++i];
		else if (!std::strcmp(argv[i], "--size") && i + 1 < argc) size = std::strtoull(argv[++i], nullptr, 0);
		else if (!std::strcmp(argv[i], "--once") && i + 1 < argc) once = std::strtol(argv[++i], nullptr, 0);
		else if (!std::strcmp(argv[i], "--devices") && i + 1 < argc) {
			/* every frame spread over several GPUs, invisible to the client (beamformer_hip_set_devices) */
			int32_t ids[8]; uint32_t n = 0;
			for (const char *c = argv[++i]; *c && n < 8; ) {
				ids[n++] = (int32_t)std::strtol(c, const_cast<char **>(&c), 10);
				if (*c == ',') c++;
			}
			if (!n || !beamformer_hip_set_devices(ids, n)) { std::fprintf(stderr, "bad --devices list\n"); return 2; }
			multi_device = n > 1;
		}
		else { std::fprintf(stderr, "usage: %s [--name /shm] [--size bytes] [--once work_items] [--devices a,b,...]\n", argv[0]); return 2; }
	}
	std::signal(SIGINT, on_signal);
	std::signal(SIGTERM, on_signal);

	/* allocate_shared_memory (main_linux.c:189-204) */
	size = (size + 4095) & ~4095ull;
	int fd = shm_open(name, O_CREAT | O_RDWR, S_IRUSR | S_IWUSR);
	if (fd < 0 || ftruncate(fd, (off_t)size) == -1) { std::perror("shm_open/ftruncate"); return 1; }
	void *region = mmap(nullptr, size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
	close(fd);
	if (region == MAP_FAILED) { std::perror("mmap"); return 1; }

	Server s;
	s.sm = static_cast<ShmHeader *>(region);
	s.size = size;
	/* run the library on a stream of ours so that per-slot completion events can be recorded behind
	 * its frames (no device: the calls fail and every compute request is answered with an error) */
	s.multi_device = multi_device;
	if (multi_device) {
		/* a stream belongs to one device: the library keeps its own streams, and an RF slot is reused only
		 * after beamformer_hip_synchronize (handle_upload) */
		s.stream = nullptr;
	} else if (hipSetDevice(beamformer_hip_get_device() >= 0 ? beamformer_hip_get_device() : 0) == hipSuccess &&
	           hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) == hipSuccess) {
		/* (the stream, the events and the RF slots of this process live on the ONE device the library was told to use --
		 * `--devices 3` -- not on device 0; the library refuses a stream of another device) */
		if (!beamformer_hip_set_stream(s.stream)) { (void)hipStreamDestroy(s.stream); s.stream = nullptr; }
		for (auto &e : s.read_done) if (!s.stream || hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) e = nullptr;
	} else {
		s.stream = nullptr;
	}
	std::memset((void *)s.sm, 0, sizeof(ShmHeader));                /* beamformer.c:249 */
	s.sm->reserved_parameter_blocks    = 1;
	s.sm->beamformed_frame_buffer_size = 4ull << 30;               /* the library's frame ring */
	if (const char *e = std::getenv("BEAMFORMER_HIP_FRAME_RING_BYTES")) {
		unsigned long long v = std::strtoull(e, nullptr, 0);
		if (v >= (1ull << 20)) s.sm->beamformed_frame_buffer_size = (v + 63) & ~63ull;
	}
	s.sm->capabilities.max_rf_data_size = s.sm->beamformed_frame_buffer_size / BeamformerMaxRawDataFramesInFlight;
	s.sm->capabilities.cuda    = 0;
	s.sm->capabilities.hilbert = 0;                                 /* beamformer.c:262-263 */
	__atomic_store_n(&s.sm->version, (uint32_t)BEAMFORMER_SHARED_MEMORY_VERSION, __ATOMIC_SEQ_CST);   /* clients check this first */
	say("ready name %s size %llu version %u", name, (unsigned long long)size, s.sm->version);

	/* The reference's workers sleep on futexes the client posts (lib .c:192-198).  This loop
	 * serves two signalling paths (the upload lock and the work queue) from one thread, so it
	 * polls instead: busily for ~1 ms after the last request -- a streaming client sees
	 * microsecond hand-offs -- then with 100 us sleeps so an idle server costs no CPU. */
	long served = 0;
	auto last_activity = std::chrono::steady_clock::now();
	while (!g_stop && (once < 0 || served < once)) {
		bool idle = true;
		if (__atomic_load_n(&s.sm->locks[Lock_UploadRF], __ATOMIC_SEQ_CST)) {
			uint64_t sz = s.sm->rf_block_rf_size.exchange(0);
			if (sz) { handle_upload(s, sz); idle = false; }
		}
		/* beamform_work_queue_pop / pop_commit (beamformer_shared_memory.c:168-190) */
		for (;;) {
			uint64_t val = s.sm->external_work_queue.queue.load();
			uint64_t widx = val & 63, ridx = (val >> 32) & 63;
			if (ridx == widx) break;
			handle_work(s, &s.sm->external_work_queue.items[ridx]);
			s.sm->external_work_queue.queue.fetch_add(0x100000000ull);
			served++;
			idle = false;
			if (once >= 0 && served >= once) break;
		}
		if (!idle) {
			last_activity = std::chrono::steady_clock::now();
		} else if (std::chrono::steady_clock::now() - last_activity < std::chrono::milliseconds(1)) {
			__builtin_ia32_pause();
		} else {
			usleep(100);
		}
	}

	/* beamformer_terminate (beamformer.c:345-373): make blocked clients fail instead of hang */
	s.sm->invalid = 1;
	post_sync(s.sm, Lock_DispatchCompute);
	post_sync(s.sm, Lock_ExportSync);
	post_sync(s.sm, Lock_UploadRF);
	s.sm->live_imaging_dirty_flags.fetch_or(1u << BeamformerLiveFeedbackFlags_StopImaging);
	for (auto &p : s.rf_ring) if (p) (void)hipFree(p);
	beamformer_hip_shutdown();
	munmap(region, size);
	shm_unlink(name);
	say("stopped after %ld work items", served);
	return 0;
}
