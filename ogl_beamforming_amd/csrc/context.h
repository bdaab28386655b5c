/* context.h -- process-global state of the in-process HIP beamformer.
 *
 * Stands where the reference has a server process: BeamformerCtx / BeamformerComputeContext
 * (beamformer_internal.h:386-470) reached through BeamformerSharedMemory
 * (beamformer_shared_memory.c:133-166).  Here the "server" is this library: parameter
 * blocks are plain host structs, the RF ring, the ping-pong buffers and the frame ring are
 * HIP device allocations, timelines are stream order + HIP events. */
#ifndef BF_CONTEXT_H
#define BF_CONTEXT_H

#include <hip/hip_runtime_api.h>
#include <cstdint>
#include <string>
#include <list>
#include <vector>
#include "planner.h"
#include "bf_kernels.h"
#include "das_select.h"
#include "../../include/ogl_beamformer_hip.h"

namespace bf {

struct DeviceBuffer {
	void  *ptr  = nullptr;
	size_t size = 0;
	bool ensure(size_t bytes);      /* grows (never shrinks); false on allocation failure */
	void release();
};

/* beamformer_internal.h:413-422 (BeamformerFrame / backlog) */
struct FrameRecord {
	uint64_t offset = 0, bytes = 0;
	uint32_t points[3]{1, 1, 1};
	int      data_kind = BeamformerDataKind_Float32;
	uint32_t id = 0, block = 0;
	int      timing_slot = -1;
	bool     failed = false;         /* tombstone of a push that did not complete (executor.cpp, Lockstep) */
};

/* one RF frame in flight on the upload side (beamformer_rf_upload's slot, beamformer_core.c:1756-1805) */
struct UploadSlot {
	void      *pinned = nullptr;         /* hipHostMalloc staging the caller's bytes land in */
	size_t     pinned_size = 0;
	hipEvent_t copied = nullptr;         /* H2D of this slot finished (copy stream) */
	hipEvent_t consumed = nullptr;       /* the frame that read this slot's device buffers finished (compute stream) */
	bool       copy_pending = false, consume_pending = false;
	bool       unfenced_reader = false;  /* a frame read this slot's device buffers without recording `consumed` */
};

struct TimingSlot {
	hipEvent_t events[BEAMFORMER_HIP_MAX_TIMED_STAGES + 1]{};
	uint32_t   kinds[BEAMFORMER_HIP_MAX_TIMED_STAGES]{};
	uint32_t   count = 0;
	bool       created = false;
	bool       sampled = true;       /* false: this frame recorded no events; events_slot names the slot whose events stand in */
	uint32_t   events_slot = 0;
	uint64_t   das_voxels = 0;
	uint32_t   das_taps = 0, das_sample_bytes = 0, das_path = 0;
	bool       counted = false;
	uint64_t   frame_id = 0;
	uint64_t   violations_slot = ~0ull;   /* staged kernels: index of this frame's window-violation counter, or ~0 */
	uint32_t   das_row_end_planes = 0;    /* planes the row-end rule sent to the kernel behind the staged one */
	bool       failed = false;            /* the push that owns this slot did not complete */
};

struct PlanState {
	Plan         plan;
	bool         valid = false;
	DeviceBuffer hadamard_t, hadamard_base, readi_hadamard, transmits, sparse, mapping;
	std::vector<DeviceBuffer> taps;     /* per stage: filter taps (+ demodulation phasors) */
	std::list<std::vector<float>> tap_tables;   /* host copies the async uploads read from */
	uint64_t     generation = 0;          /* bumped by every successful commit (replan) of this block on this device */
	std::vector<BfTransmit>   transmit_table;
	std::vector<uint16_t>     readi_bits;
	std::string  error;
	std::vector<DasDecision> das_parts;   /* the DAS kernel(s) and geometry of this plan's frames (das_select.cpp: decide_das_parts), reused until
	                                         the plan, the shard, the path mode or a hook changes */
	uint32_t     das_z_first = 0, das_z_count = 0;
};

constexpr uint32_t kTimingSlots = 32;    /* beamformer_compute_stats.c: 32-frame table */
constexpr uint32_t kStageIngest    = 0xFFFF;
constexpr uint32_t kStagePairCount = 0xFFFE;

constexpr uint32_t kMaxDevices = 8;      /* one node of MI355X */

/* Everything that lives on one HIP device.  A process normally owns one (devices[0]); after
 * beamformer_hip_set_devices it owns several, each beamforming one z-slab of every frame. */
struct Device {
	int          device = -1;                                  /* HIP ordinal */
	uint32_t     index = 0;                                    /* position in Context::devices */
	hipStream_t  own_stream = nullptr, stream = nullptr;
	PlanState    plans[BeamformerMaxParameterBlocks];
	DeviceBuffer raw_staging[BeamformerMaxRawDataFramesInFlight];
	UploadSlot   upload[BeamformerMaxRawDataFramesInFlight];
	hipStream_t  copy_stream = nullptr;                        /* H2D of frame n+1 overlaps compute of frame n */
	DeviceBuffer rf[BeamformerMaxRawDataFramesInFlight];     /* beamformer.meta:8: 3 in flight */
	uint64_t     rf_index = 0;
	DeviceBuffer scratch[2];                                   /* ping-pong (reference: 3 slots of one buffer) */
	DeviceBuffer ring;                                         /* frame ring ("BeamformedData") */
	uint64_t     ring_next_offset = 0, frame_counter = 0;
	std::vector<FrameRecord> frames;                           /* BeamformerMaxBacklogFrames records */
	TimingSlot   timing[kTimingSlots];
	uint64_t     last_sampled_frame = 0;                       /* frame id whose events are the newest real ones */
	uint32_t     last_sampled_block = 0;
	bool         have_sample = false;
	uint64_t     replan_frame = 0;                             /* first frame of the current plan */
	DeviceBuffer pair_counter, minmax_scratch, sum_scratch;
	DeviceBuffer hercules_pairs;                               /* das_hercules.hip: {sample, difference} copy of the DAS input (IQ, linear) */
	DeviceBuffer staged_tables;        /* das_staged.hip, wave-uniform transmit tables (bf_launch_das_staged_tables) */
	DeviceBuffer staged_violations;    /* das_staged*.hip: one counter per timing slot of window positions outside the staged window */
	DeviceBuffer hercules_table;                               /* das_hercules.hip: per-row lateral table, rebuilt per launch */
	/* multi-device frames (executor.cpp push_multi): the RF of slot k landed on this device / the frame
	 * that read slot k has finished */
	hipStream_t  peer_stream = nullptr;                        /* carries the copies INTO this device */
	hipEvent_t   rf_landed[BeamformerMaxRawDataFramesInFlight]{}, rf_consumed[BeamformerMaxRawDataFramesInFlight]{};
	bool         consumed_pending[BeamformerMaxRawDataFramesInFlight]{};
	uint32_t     slab_first = 0, slab_count = 0;               /* planes of the current multi-device frame */
	int          peer_access = 2;                              /* how RF reaches this device from devices[0]: 2 = it IS that device (or the ingest
	                                                              device itself), 1 = direct peer access over xGMI enabled, 0 = no peer access:
	                                                              hipMemcpyPeerAsync stages the copy through host memory */
	hipEvent_t   peer_copy_begin[BeamformerMaxRawDataFramesInFlight]{}, peer_copy_end[BeamformerMaxRawDataFramesInFlight]{};   /* timed */
	uint32_t     last_rf_slot = 0;                             /* RF slot of the newest frame */
	uint64_t     last_rf_bytes = 0;
	const void  *last_rf = nullptr;                            /* what the newest frame's first stage read: the library's own RF slot, never a caller's pointer */
	bool         last_rf_sum_ready = false;                    /* a borrowed device buffer: its checksum was taken inside the push */
	/* frame graphs (beamformer_hip_enable_frame_graphs): one instantiated hipGraph per parameter block, updated
	 * in place from each frame's capture; graph_generation = the plan generation it was warmed up for */
	hipGraphExec_t frame_exec[BeamformerMaxParameterBlocks]{};
	uint64_t       graph_generation[BeamformerMaxParameterBlocks]{};
};

struct Context {
	/* library-level state that needs no device */
	BeamformerLibErrorKind last_error = BeamformerLibErrorKind_None;
	int32_t        timeout_ms = 0;
	ParameterBlock blocks[BeamformerMaxParameterBlocks];
	uint32_t       reserved_parameter_blocks = 1;              /* beamformer.c:249-263 */
	BeamformerLiveImagingParameters live{};
	uint32_t       live_dirty_flags = 0;
	uint64_t       frame_ring_bytes = 0;                       /* beamformed_frame_buffer_size */
	uint32_t       das_path_mode = 0;
	bool           count_pairs = false;
	bool           hilbert_enabled = false;                    /* beamformer_hip_enable_hilbert */
	bool           frame_graphs = false;                       /* beamformer_hip_enable_frame_graphs */
	uint64_t       graph_frames = 0, graph_instantiations = 0; /* frames replayed from a graph / graphs instantiated */

	/* device state */
	int          requested_devices[kMaxDevices]{-1, -1, -1, -1, -1, -1, -1, -1};
	uint32_t     requested_count = 0;                          /* 0: one device, chosen from the environment */
	bool         device_ready = false;
	Device       devices[kMaxDevices];
	uint32_t     device_count = 1;
	Device      *cur = &devices[0];                            /* the device the executor functions act on */
	std::vector<float> rf_time_deltas;
	double       last_push_time = 0;
	uint64_t     push_sequence = 0;                            /* id of the next frame, on EVERY device: one counter, so that the devices of
	                                                              beamformer_hip_set_devices stay in lockstep even after a push that failed half way */
};

Context &ctx();
bool     set_error(BeamformerLibErrorKind kind);   /* records and returns false */

/* executor.cpp */
bool ensure_device();                               /* SharedMemory error when no HIP device */
uint64_t default_frame_ring_bytes();
bool push_rf_and_compute(uint32_t block, const void *data, uint32_t size, bool data_on_device);
bool wait_for_frames(int32_t timeout_ms);
const FrameRecord *newest_record(const Device &d);     /* null: the newest push did not complete */
bool export_last_frames(void *out, uint64_t out_size, uint32_t count, int32_t timeout_ms);
bool last_frame_timings(BeamformerHipFrameTimings *out);
bool device_frame_timings(uint32_t device_index, BeamformerHipFrameTimings *out);
bool device_info(uint32_t device_index, BeamformerHipDeviceInfo *out);
bool fill_stats_table(BeamformerComputeStatsTable *out);
bool frame_min_max(float out[2]);
bool sum_last_frames(uint32_t count, void *out, uint64_t out_size);
bool display_last_frame(float threshold_db, float gamma, float db_cutoff, float *out, uint64_t out_floats);
void shutdown_device();

} // namespace bf
#endif
