/* lib_api.cpp -- the exported C ABI (include/ogl_beamformer_lib.h, include/ogl_beamformer_hip.h).
 *
 * Every function mirrors the function of the same name in the reference's client library
 * lib/ogl_beamformer_lib.c: same validation, in the same order, with the same error codes.
 * Where the reference writes into a shared-memory parameter block and marks a region dirty
 * (parameter_block_region_upload, lib .c:349-362) this writes into the process-local
 * bf::ParameterBlock and sets the same dirty bit; where it queues work for the server and
 * takes futex locks, this launches on the HIP stream.
 *
 * Parameter functions are pure host state and work without a device; functions that move
 * data or compute need a HIP device and fail with BeamformerLibErrorKind_SharedMemory
 * (the reference's "server unreachable") when there is none.
 */
#include "context.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>

using namespace bf;

namespace {

/* generated/beamformer.c:538-541 */
const uint8_t contrast_mode_samples[] = {1, 3};

template <typename T> T max_of(T a, T b) { return a > b ? a : b; }

bool check(bool condition, BeamformerLibErrorKind kind)          /* lib_error_check, lib .c:140-148 */
{
	if (!condition) set_error(kind);
	return condition;
}

uint64_t frame_ring_bytes()
{
	Context &c = ctx();
	if (!c.frame_ring_bytes) c.frame_ring_bytes = default_frame_ring_bytes();
	return c.frame_ring_bytes;
}

bool valid_parameter_block(uint32_t block)                       /* lib .c:173-182 */
{
	return check(block < ctx().reserved_parameter_blocks, BeamformerLibErrorKind_ParameterBlockUnallocated);
}

/* lib .c:252-277 */
bool validate_parameters(const BeamformerParameters *bp)
{
	if (!check((int)bp->contrast_mode >= 0 && bp->contrast_mode <= BeamformerContrastMode_Count - 1,
	           BeamformerLibErrorKind_InvalidContrastMode))
		return false;

	uint32_t contrast_raw_sample_count = bp->acquisition_count * bp->sample_count * contrast_mode_samples[bp->contrast_mode];
	if (!check(contrast_raw_sample_count <= bp->raw_data_dimensions[0], BeamformerLibErrorKind_DataSizeMismatch))
		return false;

	uint64_t buffer_size = frame_ring_bytes();
	/* three 31-bit extents times 8 can wrap 64 bits (the reference's u64 product does): a wrapped size would
	 * pass the check below and launch over a grid the frame ring cannot hold, so saturate instead */
	const uint64_t px = (uint64_t)max_of(1, bp->output_points[0]), py = (uint64_t)max_of(1, bp->output_points[1]),
	               pz = (uint64_t)max_of(1, bp->output_points[2]);
	uint64_t frame_size = UINT64_MAX;
	if (px * py <= (UINT64_MAX >> 3) / pz) frame_size = px * py * pz * 8u /* Float32Complex */;
	uint64_t incoherent_size = frame_size / 2;
	if (bp->coherency_weighting) buffer_size -= incoherent_size;
	return check(frame_size <= buffer_size, BeamformerLibErrorKind_FrameSizeOverflow);
}

/* lib .c:279-311; capabilities.hilbert is 0 here as in the reference (beamformer.c:262-263) */
bool validate_pipeline(const int32_t *shaders, uint32_t shader_count, BeamformerDataKind data_kind)
{
	if (!check((int)data_kind >= 0 && data_kind <= BeamformerDataKind_Count - 1, BeamformerLibErrorKind_InvalidDataKind))
		return false;
	if (!check(shader_count <= BeamformerMaxComputeShaderStages, BeamformerLibErrorKind_ComputeStageOverflow))
		return false;
	for (uint32_t i = 0; i < shader_count; i++) {
		bool stage_ok = shaders[i] >= BeamformerShaderKind_ComputeFirst && shaders[i] <= BeamformerShaderKind_ComputeLast;
		if (!check(stage_ok, BeamformerLibErrorKind_InvalidComputeStage)) return false;
		if (shaders[i] == BeamformerShaderKind_Hilbert &&
		    !check(ctx().hilbert_enabled, BeamformerLibErrorKind_InvalidComputeStage))     /* capabilities.hilbert */
			return false;
		if (shaders[i] == BeamformerShaderKind_Demodulate &&
		    !check(!bf_kind_complex[data_kind], BeamformerLibErrorKind_InvalidDemodulationDataKind))
			return false;
	}
	/* the reference tests shaders[0] whatever shader_count is (lib .c:305-309): an empty pipeline
	 * passes when the caller's array happens to start with Decode or Demodulate.  Kept as is --
	 * callers pass the 16-entry array of BeamformerSimpleParameters -- and an empty plan is
	 * legal here (no stage runs, the frame stays zero). */
	bool start_ok = shaders && (shaders[0] == BeamformerShaderKind_Demodulate ||
	                            shaders[0] == BeamformerShaderKind_Decode);
	return check(start_ok, BeamformerLibErrorKind_InvalidStartShader);
}

bool push_data_common(const void *data, uint32_t data_size, uint32_t image_plane_tag, uint32_t slot, bool on_device)
{
	Context &c = ctx();
	if (!ensure_device()) return false;
	if (!check(image_plane_tag < BeamformerViewPlaneTag_Count, BeamformerLibErrorKind_InvalidImagePlane)) return false;
	if (!check(slot < c.reserved_parameter_blocks, BeamformerLibErrorKind_ParameterBlockUnallocated)) return false;

	/* beamformer_push_data_base, lib .c:503-511 */
	const ParameterBlock &pb = c.blocks[slot];
	const BeamformerParameters &bp = pb.parameters;
	uint64_t max_rf_size = frame_ring_bytes() / 3;                   /* capabilities.max_rf_data_size */
	/* The reference does this arithmetic in u32 (lib .c:503-511); a product that wraps there passes
	 * its check and then reads far outside the caller's buffer.  Same checks and error codes, in u64:
	 * sizes the reference accepts without wrapping are judged identically, wrapped ones are refused. */
	uint64_t bytes    = (uint64_t)bf_kind_byte_size[pb.data_kind];
	uint64_t rf_size  = (uint64_t)bp.acquisition_count * bp.sample_count * bp.channel_count * bytes;
	uint64_t raw_size = (uint64_t)bp.raw_data_dimensions[0] * bp.raw_data_dimensions[1] * bytes;
	if (!check(data != nullptr, BeamformerLibErrorKind_BufferOverflow)) return false;
	if (!check(rf_size <= max_rf_size && rf_size <= UINT32_MAX, BeamformerLibErrorKind_RFDataSizeOverflow)) return false;
	if (!check(rf_size <= data_size && (uint64_t)data_size == raw_size, BeamformerLibErrorKind_DataSizeMismatch)) return false;
	if (!check(rf_size > 0, BeamformerLibErrorKind_DataSizeMismatch)) return false;
	/* the ingest walks channel_mapping[0 .. channel_count): bound it before anything indexes with it */
	if (!check(bp.channel_count <= BeamformerMaxChannelCount && bp.acquisition_count <= BeamformerMaxEmissionsCount,
	           BeamformerLibErrorKind_DataSizeMismatch)) return false;
	return push_rf_and_compute(slot, data, data_size, on_device);
}

template <typename T>
uint32_t push_array(T *dst, size_t dst_count, const T *src, uint32_t count, uint32_t elements, uint32_t block, uint32_t dirty)
{
	/* X-macro upload functions, lib .c:438-456 */
	if (!check(count <= dst_count, BeamformerLibErrorKind_BufferOverflow)) return 0;
	if (!valid_parameter_block(block)) return 0;
	std::memcpy(dst, src, sizeof(T) * count * elements);
	ctx().blocks[block].dirty |= dirty;
	return 1;
}

} // namespace

extern "C" {

uint32_t beamformer_get_api_version(void) { return BEAMFORMER_SHARED_MEMORY_VERSION; }

const char *beamformer_error_string(BeamformerLibErrorKind kind)
{
	static const char *table[] = {
		"None",
		"host-library version mismatch",
		"library in invalid state",
		"parameter block count overflow",
		"push to unallocated parameter block",
		"compute stage overflow",
		"invalid compute shader stage",
		"starting shader not Decode or Demodulate",
		"data kind for demodulation not Int16 or Float",
		"invalid image plane",
		"invalid filter kind",
		"invalid data kind",
		"invalid contrast mode",
		"passed buffer size exceeds available space",
		"data size doesn't match the size specified in parameters",
		"work queue full",
		"not enough space for data export",
		"failed to open shared memory region",
		"failed to acquire lock within timeout period",
		"maximum frame size exceeded",
		"raw rf size exceeds available GPU space",
		"invalid error kind",
	};
	const unsigned last = sizeof(table) / sizeof(*table) - 1;
	unsigned index = (unsigned)kind;
	return table[index < last ? index : last];
}

BeamformerLibErrorKind beamformer_get_last_error(void) { return ctx().last_error; }
const char *beamformer_get_last_error_string(void)      { return beamformer_error_string(ctx().last_error); }
void beamformer_set_global_timeout(uint32_t timeout_ms) { ctx().timeout_ms = (int32_t)timeout_ms; }

uint32_t beamformer_reserve_parameter_blocks(uint32_t count)
{
	if (!check(count <= BeamformerMaxParameterBlocks, BeamformerLibErrorKind_ParameterBlockOverflow)) return 0;
	ctx().reserved_parameter_blocks = count;
	return 1;
}

uint64_t beamformer_maximum_rf_data_size(void)
{
	if (!ensure_device()) return UINT64_MAX;
	return frame_ring_bytes() / 3;
}

uint64_t beamformer_maximum_frames_for_parameters(BeamformerParameters *bp)
{
	if (!validate_parameters(bp)) return UINT64_MAX;
	uint64_t buffer_size = frame_ring_bytes();
	uint64_t frame_size  = (uint64_t)max_of(1, bp->output_points[0]) * (uint64_t)max_of(1, bp->output_points[1])
	                       * (uint64_t)max_of(1, bp->output_points[2]) * 8u;
	if (bp->coherency_weighting) buffer_size -= frame_size / 2;
	return buffer_size / frame_size;
}

uint64_t beamformer_maximum_frames_for_simple_parameters(BeamformerSimpleParameters *bp)
{
	return beamformer_maximum_frames_for_parameters((BeamformerParameters *)bp);
}

uint32_t beamformer_set_pipeline_stage_parameters_at(uint32_t stage_index, int32_t parameter, uint32_t block)
{
	if (!valid_parameter_block(block)) return 0;
	ParameterBlock &pb = ctx().blocks[block];
	pb.filter_slots[stage_index % BeamformerMaxComputeShaderStages] = (uint8_t)parameter;   /* lib .c:364-374 */
	pb.dirty |= Dirty_ComputePipeline;
	return 1;
}

uint32_t beamformer_set_pipeline_stage_parameters(uint32_t stage_index, int32_t parameter)
{
	return beamformer_set_pipeline_stage_parameters_at(stage_index, parameter, 0);
}

uint32_t beamformer_push_pipeline_at(int32_t *shaders, uint32_t shader_count, BeamformerDataKind data_kind, uint32_t block)
{
	if (!validate_pipeline(shaders, shader_count, data_kind)) return 0;
	if (!valid_parameter_block(block)) return 0;
	ParameterBlock &pb = ctx().blocks[block];
	std::memcpy(pb.shaders, shaders, sizeof(*shaders) * shader_count);
	pb.shader_count = shader_count;
	pb.data_kind    = data_kind;
	pb.dirty |= Dirty_ComputePipeline;
	return 1;
}

uint32_t beamformer_push_pipeline(int32_t *shaders, uint32_t shader_count, BeamformerDataKind data_kind)
{
	return beamformer_push_pipeline_at(shaders, shader_count, data_kind, 0);
}

uint32_t beamformer_create_filter(BeamformerFilterParameters *filter, uint8_t filter_slot, uint8_t parameter_block)
{
	if (!check((int)filter->kind >= 0 && filter->kind < BeamformerFilterKind_Count, BeamformerLibErrorKind_InvalidFilterKind))
		return 0;
	/* lib .c:421-422: slot and block wrap instead of failing */
	ParameterBlock &pb = ctx().blocks[parameter_block % BeamformerMaxParameterBlocks];
	pb.filters[filter_slot % BeamformerFilterSlots] = *filter;
	pb.dirty |= Dirty_Filters;
	return 1;
}

uint32_t beamformer_push_channel_mapping_at(int16_t *mapping, uint32_t count, uint32_t block)
{
	return push_array(ctx().blocks[block % BeamformerMaxParameterBlocks].channel_mapping, (size_t)BeamformerMaxChannelCount,
	                  mapping, count, 1, block, Dirty_ChannelMapping);
}
uint32_t beamformer_push_channel_mapping(int16_t *mapping, uint32_t count) { return beamformer_push_channel_mapping_at(mapping, count, 0); }

uint32_t beamformer_push_sparse_elements_at(int16_t *elements, uint32_t count, uint32_t block)
{
	return push_array(ctx().blocks[block % BeamformerMaxParameterBlocks].sparse_elements, (size_t)BeamformerMaxChannelCount,
	                  elements, count, 1, block, Dirty_SparseElements);
}
uint32_t beamformer_push_sparse_elements(int16_t *elements, uint32_t count) { return beamformer_push_sparse_elements_at(elements, count, 0); }

uint32_t beamformer_push_focal_vectors_at(float *vectors, uint32_t count, uint32_t block)
{
	return push_array(&ctx().blocks[block % BeamformerMaxParameterBlocks].focal_vectors[0][0], (size_t)BeamformerMaxChannelCount,
	                  vectors, count, 2, block, Dirty_FocalVectors);
}
uint32_t beamformer_push_focal_vectors(float *vectors, uint32_t count) { return beamformer_push_focal_vectors_at(vectors, count, 0); }

uint32_t beamformer_push_transmit_receive_orientations_at(uint8_t *values, uint32_t count, uint32_t block)
{
	return push_array(ctx().blocks[block % BeamformerMaxParameterBlocks].transmit_receive_orientations,
	                  (size_t)BeamformerMaxChannelCount, values, count, 1, block, Dirty_Orientations);
}
uint32_t beamformer_push_transmit_receive_orientations(uint8_t *values, uint32_t count)
{
	return beamformer_push_transmit_receive_orientations_at(values, count, 0);
}

uint32_t beamformer_push_parameters_at(BeamformerParameters *bp, uint32_t block)
{
	if (!validate_parameters(bp)) return 0;
	if (!valid_parameter_block(block)) return 0;
	ParameterBlock &pb = ctx().blocks[block];
	std::memcpy(&pb.parameters, bp, sizeof(*bp));
	pb.dirty |= Dirty_Parameters;
	return 1;
}
uint32_t beamformer_push_parameters(BeamformerParameters *bp) { return beamformer_push_parameters_at(bp, 0); }

/* lib .c:620-646 */
uint32_t beamformer_push_simple_parameters_at(BeamformerSimpleParameters *bp, uint32_t block)
{
	float focal_vectors[BeamformerMaxEmissionsCount][2];
	for (uint32_t i = 0; i < BeamformerMaxEmissionsCount; i++) {
		focal_vectors[i][0] = bp->steering_angles[i];
		focal_vectors[i][1] = bp->focal_depths[i];
	}
	uint32_t result = 1;
	result &= beamformer_push_parameters_at((BeamformerParameters *)bp, block);
	result &= beamformer_push_pipeline_at(bp->compute_stages, bp->compute_stages_count, bp->data_kind, block);
	result &= beamformer_push_channel_mapping_at(bp->channel_mapping, bp->channel_count, block);
	result &= beamformer_push_focal_vectors_at(&focal_vectors[0][0], BeamformerMaxEmissionsCount, block);
	result &= beamformer_push_transmit_receive_orientations_at(bp->transmit_receive_orientations, bp->acquisition_count, block);
	if (bp->acquisition_kind == BeamformerAcquisitionKind_UFORCES || bp->acquisition_kind == BeamformerAcquisitionKind_UHERCULES)
		result &= beamformer_push_sparse_elements_at(bp->sparse_elements, bp->acquisition_count, block);
	for (uint32_t stage = 0; stage < bp->compute_stages_count && stage < BeamformerMaxComputeShaderStages; stage++)
		result &= beamformer_set_pipeline_stage_parameters_at(stage, bp->compute_stage_parameters[stage], block);
	return result;
}
uint32_t beamformer_push_simple_parameters(BeamformerSimpleParameters *bp) { return beamformer_push_simple_parameters_at(bp, 0); }

uint32_t beamformer_push_data_with_compute(void *data, uint32_t data_size, uint32_t image_plane_tag, uint32_t parameter_slot)
{
	return push_data_common(data, data_size, image_plane_tag, parameter_slot, false);
}

uint32_t beamformer_get_last_frames(void *out_data, uint64_t out_data_size, uint32_t count)
{
	if (!(out_data && out_data_size && count)) return 0;         /* lib .c:700: fails without an error code */
	if (!ensure_device()) return 0;
	return export_last_frames(out_data, out_data_size, count, ctx().timeout_ms);
}

/* lib .c:704-736.  The reference ignores its timeout_ms argument and waits with the global
 * timeout; a non-zero argument is honoured here (superset). */
uint32_t beamformer_beamform_data(BeamformerSimpleParameters *bp, void *data, uint32_t data_size,
                                  void *out_data, int32_t timeout_ms)
{
	uint32_t result = beamformer_push_simple_parameters(bp);
	if (!result) return 0;

	bool complex_out = false;
	for (uint32_t stage = 0; stage < bp->compute_stages_count && stage < BeamformerMaxComputeShaderStages; stage++)
		complex_out |= bp->compute_stages[stage] == BeamformerShaderKind_Demodulate ||
		               bp->compute_stages[stage] == BeamformerShaderKind_Hilbert;
	uint64_t output_size = (uint64_t)max_of(1, bp->output_points[0]) * (uint64_t)max_of(1, bp->output_points[1])
	                       * (uint64_t)max_of(1, bp->output_points[2]) * sizeof(float);
	if (complex_out) output_size *= 2;

	result = beamformer_push_data_with_compute(data, data_size, 0, 0);
	if (result && out_data) {
		if (!ensure_device()) return 0;
		int32_t wait = timeout_ms != 0 ? timeout_ms : ctx().timeout_ms;
		result = export_last_frames(out_data, output_size, 1, wait);
	}
	return result;
}

uint32_t beamformer_compute_timings(BeamformerComputeStatsTable *output, int32_t timeout_ms)
{
	if (!ensure_device()) return 0;
	if (!wait_for_frames(timeout_ms)) return 0;
	return fill_stats_table(output);
}

int32_t beamformer_live_parameters_get_dirty_flag(void)
{
	Context &c = ctx();
	for (int flag = 0; flag < 32; flag++)
		if (c.live_dirty_flags & (1u << flag)) { c.live_dirty_flags &= ~(1u << flag); return flag; }
	return -1;
}

BeamformerLiveImagingParameters *beamformer_get_live_parameters(void) { return &ctx().live; }

uint32_t beamformer_set_live_parameters(BeamformerLiveImagingParameters *params)
{
	std::memcpy(&ctx().live, params, sizeof(*params));
	return 1;
}

/* ---------------- MI355X extensions (include/ogl_beamformer_hip.h) ---------------- */

uint32_t beamformer_hip_set_device(int32_t device_index)
{
	Context &c = ctx();
	if (c.device_ready) return check(c.device_count == 1 && c.devices[0].device == device_index, BeamformerLibErrorKind_InvalidAccess);
	c.requested_devices[0] = device_index;
	c.requested_count = 1;
	return 1;
}

int32_t beamformer_hip_get_device(void)
{
	Context &c = ctx();
	return c.device_ready ? c.devices[0].device : (c.requested_count ? c.requested_devices[0] : -1);
}

uint32_t beamformer_hip_set_devices(const int32_t *device_indices, uint32_t count)
{
	Context &c = ctx();
	if (!check(device_indices != nullptr && count >= 1 && count <= kMaxDevices, BeamformerLibErrorKind_InvalidAccess)) return 0;
	if (c.device_ready) {
		/* like beamformer_hip_set_device: afterwards only the set already in use is accepted */
		bool same = c.device_count == count;
		for (uint32_t i = 0; same && i < count; i++) same = c.devices[i].device == device_indices[i];
		return check(same, BeamformerLibErrorKind_InvalidAccess);
	}
	for (uint32_t i = 0; i < count; i++) {
		if (!check(device_indices[i] >= 0, BeamformerLibErrorKind_InvalidAccess)) return 0;
		c.requested_devices[i] = device_indices[i];
	}
	c.requested_count = count;
	return 1;
}

uint32_t beamformer_hip_get_device_count(void)
{
	Context &c = ctx();
	return c.device_ready ? c.device_count : (c.requested_count ? c.requested_count : 1u);
}

uint32_t beamformer_hip_get_device_frame_timings(uint32_t device_index, BeamformerHipFrameTimings *out)
{
	if (!ensure_device()) return 0;
	return device_frame_timings(device_index, out);
}

uint32_t beamformer_hip_get_device_info(uint32_t device_index, BeamformerHipDeviceInfo *out)
{
	if (!check(out != nullptr, BeamformerLibErrorKind_InvalidAccess) || !ensure_device()) return 0;
	return device_info(device_index, out);
}

uint32_t beamformer_hip_set_stream(void *hip_stream)
{
	Context &c = ctx();
	/* a stream belongs to one device: with several devices the library keeps to its own streams */
	if (!check(beamformer_hip_get_device_count() == 1 || hip_stream == nullptr, BeamformerLibErrorKind_InvalidAccess)) return 0;
	Device &d = c.devices[0];
	if (hip_stream) {
		/* a stream of another device would have every kernel of the library enqueued against buffers it cannot reach: refuse it.
		 * The device the library runs on: the one in use, else the one asked for, else what ensure_device will pick. */
		int wanted = c.device_ready ? d.device : (c.requested_count ? c.requested_devices[0] : -1);
		if (wanted < 0) {
			const char *e = std::getenv("BEAMFORMER_HIP_DEVICE");
			if (!e) e = std::getenv("LOCAL_RANK");
			wanted = e ? std::atoi(e) : 0;
		}
		int owner = -1;
		if (!check(hipStreamGetDevice((hipStream_t)hip_stream, &owner) == hipSuccess && owner == wanted, BeamformerLibErrorKind_InvalidAccess)) {
			(void)hipGetLastError();
			return 0;
		}
	}
	if (c.device_ready && d.stream) (void)hipStreamSynchronize(d.stream);   /* keep frames ordered across the switch */
	d.stream = hip_stream ? (hipStream_t)hip_stream : d.own_stream;
	return 1;
}

uint32_t beamformer_hip_set_output_shard(uint32_t parameter_slot, uint32_t z_first, uint32_t z_count)
{
	if (!valid_parameter_block(parameter_slot)) return 0;
	ParameterBlock &pb = ctx().blocks[parameter_slot];
	uint32_t z_total = (uint32_t)max_of(1, pb.parameters.output_points[2]);
	if (z_count && !check(z_first < z_total && z_count <= z_total - z_first, BeamformerLibErrorKind_FrameSizeOverflow)) return 0;
	pb.shard_z_first = z_count ? z_first : 0;
	pb.shard_z_count = z_count;
	pb.dirty |= Dirty_Shard;
	return 1;
}

uint32_t beamformer_hip_push_device_data_with_compute(const void *device_data, uint32_t size,
                                                      uint32_t image_plane_tag, uint32_t parameter_slot)
{
	return push_data_common(device_data, size, image_plane_tag, parameter_slot, true);
}

uint32_t beamformer_hip_synchronize(void)
{
	if (!ensure_device()) return 0;
	return wait_for_frames(-1);
}

uint32_t beamformer_hip_get_last_frame_info(BeamformerHipFrameInfo *out)
{
	Context &c = ctx();
	const Device &d = c.devices[0];      /* with several devices: the ingest device's slab */
	const FrameRecord *newest = c.device_ready ? newest_record(d) : nullptr;      /* null after a push that did not complete */
	if (!check(newest != nullptr, BeamformerLibErrorKind_InvalidAccess)) return 0;
	const FrameRecord &f = *newest;
	out->device_pointer = (char *)d.ring.ptr + f.offset;
	out->size_bytes = f.bytes;
	out->points[0] = f.points[0]; out->points[1] = f.points[1]; out->points[2] = f.points[2];
	out->data_kind = (uint32_t)f.data_kind;
	out->frame_id = f.id; out->parameter_block = f.block;
	return 1;
}

uint32_t beamformer_hip_get_last_frame_timings(BeamformerHipFrameTimings *out)
{
	if (!ensure_device()) return 0;
	return last_frame_timings(out);
}

uint32_t beamformer_hip_enable_frame_graphs(uint32_t enable) { ctx().frame_graphs = enable != 0; return 1; }

uint32_t beamformer_hip_frame_graph_counts(uint64_t *frames_replayed, uint64_t *graphs_instantiated)
{
	if (frames_replayed)     *frames_replayed     = ctx().graph_frames;
	if (graphs_instantiated) *graphs_instantiated = ctx().graph_instantiations;
	return 1;
}

uint32_t beamformer_hip_enable_pair_counting(uint32_t enable) { ctx().count_pairs = enable != 0; return 1; }

uint32_t beamformer_hip_frame_min_max(float out_min_max[2])
{
	if (!ensure_device()) return 0;
	return frame_min_max(out_min_max);
}

uint32_t beamformer_hip_sum_last_frames(uint32_t count, void *out, uint64_t out_size)
{
	if (!out) return set_error(BeamformerLibErrorKind_InvalidAccess);
	if (!ensure_device()) return 0;
	return sum_last_frames(count, out, out_size);
}

uint32_t beamformer_hip_display_last_frame(float threshold_db, float gamma, float db_cutoff, float *out, uint64_t out_floats)
{
	if (!out) return set_error(BeamformerLibErrorKind_InvalidAccess);
	if (!ensure_device()) return 0;
	return display_last_frame(threshold_db, gamma, db_cutoff, out, out_floats);
}

uint32_t beamformer_hip_enable_hilbert(uint32_t enable)
{
	Context &c = ctx();
	c.hilbert_enabled = enable != 0;
	for (auto &b : c.blocks) b.dirty |= Dirty_ComputePipeline;
	return 1;
}

uint32_t beamformer_hip_set_das_path(uint32_t mode) { ctx().das_path_mode = mode; return 1; }

void beamformer_hip_host_das_transform(const float min_coordinate[3], const float max_coordinate[3],
                                       int32_t points[3], float out16[16])
{
	das_transform(min_coordinate, max_coordinate, points, out16);
}

uint32_t beamformer_hip_host_hadamard(uint32_t order, float *out)
{
	std::vector<float> h = hadamard_transpose((int)order);
	if (h.empty()) return 0;
	std::memcpy(out, h.data(), sizeof(float) * h.size());
	return 1;
}

int32_t beamformer_hip_host_filter(const BeamformerFilterParameters *filter, float *taps, uint32_t capacity_floats,
                                   float *time_delay, uint32_t *complex_taps)
{
	Filter f;
	if (!filter_create(*filter, f) || f.taps.size() > capacity_floats) return -1;
	std::memcpy(taps, f.taps.data(), sizeof(float) * f.taps.size());
	if (time_delay)   *time_delay   = f.time_delay;
	if (complex_taps) *complex_taps = f.complex_taps;
	return f.length;
}

uint32_t beamformer_hip_describe_plan(uint32_t parameter_slot, BeamformerHipPlan *out)
{
	if (!valid_parameter_block(parameter_slot)) return 0;
	Plan plan;
	std::string error;
	if (!build_plan(ctx().blocks[parameter_slot], plan, error, ctx().hilbert_enabled)) return check(false, BeamformerLibErrorKind_InvalidComputeStage);
	std::memset(out, 0, sizeof(*out));
	out->stage_count = (uint32_t)plan.stages.size();
	for (size_t i = 0; i < plan.stages.size() && i < BeamformerMaxComputeShaderStages; i++) {
		const Stage &st = plan.stages[i];
		out->stages[i].kind = st.kind; out->stages[i].in_kind = st.in_kind; out->stages[i].out_kind = st.out_kind;
		for (int k = 0; k < 3; k++) { out->stages[i].in_stride[k] = st.in_stride[k]; out->stages[i].out_stride[k] = st.out_stride[k]; }
	}
	out->das_samples = plan.das_samples; out->iq_pipeline = plan.iq_pipeline;
	out->das_sampling_frequency = plan.das_sampling_frequency; out->das_time_offset = plan.das_time_offset;
	std::memcpy(out->das_voxel_transform, plan.das_voxel_transform, sizeof(out->das_voxel_transform));
	return 1;
}

uint32_t beamformer_hip_set_hook(const char *name, const char *value)
{
	return check(set_hook(name, value), BeamformerLibErrorKind_InvalidAccess);
}

uint32_t beamformer_hip_describe_das(uint32_t parameter_slot, BeamformerHipDasDescription *out)
{
	if (!valid_parameter_block(parameter_slot) || !check(out != nullptr, BeamformerLibErrorKind_InvalidAccess)) return 0;
	Context &c = ctx();
	const ParameterBlock &pb = c.blocks[parameter_slot];
	Plan plan;
	std::string error;
	if (!build_plan(pb, plan, error, c.hilbert_enabled)) return check(false, BeamformerLibErrorKind_InvalidComputeStage);
	std::memset(out, 0, sizeof(*out));
	if (plan.das_index < 0) { out->path = -1; return 1; }
	uint32_t zfirst = 0, zcount = plan.output_points[2];
	if (pb.shard_z_count) { zfirst = pb.shard_z_first; zcount = pb.shard_z_count; }
	/* the decision for the planes that keep the first choice; where the row-end rule cuts the range (das_select.h: decide_das_parts) the
	 * planes handed to the kernel behind it are counted in row_end_planes */
	std::vector<DasDecision> parts;
	decide_das_parts(pb, plan, build_transmit_table(pb), zfirst, zcount, c.das_path_mode, parts);
	const DasDecision &d = main_part(parts);
	out->row_end_planes = row_end_planes(parts);
	out->path = d.path == DasPath_Zero ? -2 : d.path;
	std::snprintf(out->kernel, sizeof(out->kernel), "%s", das_kernel_name(d.path));
	std::snprintf(out->name, sizeof(out->name), "%s", das_path_name(d.path));
	for (int k = 0; k < DasPath_Count && k < 8; k++) std::snprintf(out->declined[k], sizeof(out->declined[k]), "%s", d.why[k].c_str());
	for (int k = 0; k < 3; k++) { out->tile_shift[k] = d.a.tile_shift[k]; out->blocks[k] = d.a.blocks[k]; }
	out->split_shift = d.a.split_shift; out->tile_walk = d.path == DasPath_Hercules ? d.herc.depth_major : (d.path == DasPath_Gather || d.path == DasPath_Staged) ? d.sep.depth_major : d.a.depth_major;
	out->tile_window_samples = d.path == DasPath_Tile ? 1u << d.a.tile_window_shift : 0u;
	if (d.path == DasPath_Gather || d.path == DasPath_Staged) {
		out->u_axis = d.sep.u_axis; out->u_shift = d.sep.u_shift; out->v_shift = d.sep.v_shift; out->window_samples = d.path == DasPath_Staged ? d.sep.window_samples : 0;
		out->uniform_tables = d.sep.uniform; out->lds_bytes = d.sep.lds_bytes; out->threads = d.sep.threads; out->channel_chunk = d.sep.channel_chunk;
	}
	out->hercules_prepared_copy = d.hercules_prepared;
	out->tile_spread_estimate = d.tile_spread;
	for (int k = 0; k < 3; k++) out->tile_estimate_shift[k] = d.tile_estimate_shift[k];
	out->row_ends = d.a.row_ends;
	return 1;
}

void beamformer_hip_shutdown(void) { shutdown_device(); }

} // extern "C"
