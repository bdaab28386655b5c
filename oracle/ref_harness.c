/* ref_harness.c -- TEST INFRASTRUCTURE, not product code.
 *
 * Thin exported wrappers around the reference's own host DSP math, compiled from the
 * reference sources *where they lie* under /root/reference (nothing is copied into this
 * repository).  Built by oracle/Makefile target `ref` into oracle/_ref/libref_math.so with
 * ROCm clang (gcc 11 lacks _Float16 on x86, which base_types.h:41 needs).
 *
 * Used only by tests/golden/make_golden.py (to emit the committed fixtures that pin
 * oracle/oracle_math.c and the product's host_math.cpp) and by tests that run in the
 * build container.  /root/reference does not exist on the GPU box; the fixtures travel,
 * this library does too (prebuilt), the sources do not.
 *
 * The reference's translation-unit structure: util.h pulls in util.c and math.c
 * (util.h tail), math.c pulls in external/cephes.c (math.c:2).
 */
#define BASE_PLATFORM_NO_MAIN 1
#define BASE_EXPORT static
#define BASE_IMPORT static
#include "/root/reference/util.h"
#include "/root/reference/generated/beamformer.c"
#include "/root/reference/base_linux.c"
#include "/root/reference/util_os.c"
#include "/root/reference/beamformer_compute_stats.c"
#include "/root/reference/beamformer_shared_memory.c"

#define REF_EXPORT __attribute__((visibility("default")))

static Arena *ref_arena(void)
{
	static Arena *a;
	if (!a) a = arena_create(.reserve_size = GB(1), .commit_size = MB(1));
	return a;
}

/* math.c:35-134.  Returns 0 when the reference returns NULL. */
REF_EXPORT int ref_hadamard_transpose(int dim, int row_major, float *out)
{
	Arena *a   = ref_arena();
	Temp   tmp = temp_begin(a);
	f16   *h   = make_hadamard_transpose(a, dim, row_major);
	int result = h != 0;
	if (h) for (int i = 0; i < dim * dim; i++) out[i] = (float)h[i];
	temp_end(tmp);
	return result;
}

/* math.c:750-767 */
REF_EXPORT void ref_kaiser_low_pass(float cutoff, float fs, float beta, int length, float *out)
{
	Arena *a   = ref_arena();
	Temp   tmp = temp_begin(a);
	f32   *h   = kaiser_low_pass_filter(a, cutoff, fs, beta, length);
	for (int i = 0; i < length; i++) out[i] = h[i];
	temp_end(tmp);
}

/* math.c:769-781 */
REF_EXPORT void ref_rf_chirp(float fmin, float fmax, float fs, int length, int reverse, float *out)
{
	Arena *a   = ref_arena();
	Temp   tmp = temp_begin(a);
	f32   *h   = rf_chirp(a, fmin, fmax, fs, length, reverse);
	for (int i = 0; i < length; i++) out[i] = h[i];
	temp_end(tmp);
}

/* math.c:783-797 */
REF_EXPORT void ref_baseband_chirp(float fmin, float fmax, float fs, int length, int reverse,
                                   float scale, float *out /* 2*length */)
{
	Arena *a   = ref_arena();
	Temp   tmp = temp_begin(a);
	v2    *h   = baseband_chirp(a, fmin, fmax, fs, length, reverse, scale);
	for (int i = 0; i < length; i++) { out[2 * i] = h[i].x; out[2 * i + 1] = h[i].y; }
	temp_end(tmp);
}

/* math.c:739-747 */
REF_EXPORT float ref_tukey_window(float t, float tapering) { return tukey_window(t, tapering); }

/* math.c:713-737 */
REF_EXPORT float ref_real_filter_first_moment(float *h, int length, float fs)
{ return real_filter_first_moment(h, length, fs); }
REF_EXPORT float ref_complex_filter_first_moment(float *h, int length, float fs)
{ return complex_filter_first_moment((v2 *)h, length, fs); }

/* external/cephes.c:24-103 */
REF_EXPORT double ref_cephes_i0(double x) { return cephes_i0(x); }

/* math.c:799-920.  points is in/out (das_output_dimension). */
REF_EXPORT void ref_das_transform(const float *min3, const float *max3, int *points3, float *out16)
{
	v3  mn = {{min3[0], min3[1], min3[2]}};
	v3  mx = {{max3[0], max3[1], max3[2]}};
	iv3 p  = {{points3[0], points3[1], points3[2]}};
	m4  r  = das_transform(mn, mx, &p);
	points3[0] = p.x; points3[1] = p.y; points3[2] = p.z;
	for (int i = 0; i < 16; i++) out16[i] = r.E[i];
}

REF_EXPORT void ref_das_transform_2d(int plane /* 0 xz, 1 yz, 2 xy */, const float *min2,
                                     const float *max2, float offset, float *out16)
{
	v2 mn = {{min2[0], min2[1]}};
	v2 mx = {{max2[0], max2[1]}};
	m4 r;
	switch (plane) {
	case 0:  r = das_transform_2d_xz(mn, mx, offset); break;
	case 1:  r = das_transform_2d_yz(mn, mx, offset); break;
	default: r = das_transform_2d_xy(mn, mx, offset); break;
	}
	for (int i = 0; i < 16; i++) out16[i] = r.E[i];
}

REF_EXPORT void ref_das_transform_3d(const float *min3, const float *max3, float *out16)
{
	v3 mn = {{min3[0], min3[1], min3[2]}};
	v3 mx = {{max3[0], max3[1], max3[2]}};
	m4 r  = das_transform_3d(mn, mx);
	for (int i = 0; i < 16; i++) out16[i] = r.E[i];
}

/* math.c:448-458 */
REF_EXPORT void ref_m4_mul(const float *a16, const float *b16, float *out16)
{
	m4 a, b;
	for (int i = 0; i < 16; i++) { a.E[i] = a16[i]; b.E[i] = b16[i]; }
	m4 r = m4_mul(a, b);
	for (int i = 0; i < 16; i++) out16[i] = r.E[i];
}

/* layout facts the C ABI header asserts (generated/beamformer.c:345-467) */
REF_EXPORT int ref_sizeof(int which)
{
	switch (which) {
	case 0: return sizeof(BeamformerParameters);
	case 1: return sizeof(BeamformerSimpleParameters);
	case 2: return sizeof(BeamformerFilterParameters);
	case 3: return sizeof(BeamformerLiveImagingParameters);
	case 4: return sizeof(BeamformerComputeArrayParameters);
	case 5: return sizeof(BeamformerEmissionParameters);
	}
	return -1;
}

#define REF_OFFSET_LIST \
	X(BeamformerParameters, das_voxel_transform) X(BeamformerParameters, xdc_transform) \
	X(BeamformerParameters, xdc_element_pitch) X(BeamformerParameters, raw_data_dimensions) \
	X(BeamformerParameters, focal_vector) X(BeamformerParameters, transmit_receive_orientation) \
	X(BeamformerParameters, sample_count) X(BeamformerParameters, channel_count) \
	X(BeamformerParameters, acquisition_count) X(BeamformerParameters, acquisition_kind) \
	X(BeamformerParameters, decode_mode) X(BeamformerParameters, sampling_mode) \
	X(BeamformerParameters, time_offset) X(BeamformerParameters, single_focus) \
	X(BeamformerParameters, single_orientation) X(BeamformerParameters, output_points) \
	X(BeamformerParameters, sampling_frequency) X(BeamformerParameters, demodulation_frequency) \
	X(BeamformerParameters, speed_of_sound) X(BeamformerParameters, f_number) \
	X(BeamformerParameters, interpolation_mode) X(BeamformerParameters, coherency_weighting) \
	X(BeamformerParameters, decimation_rate) X(BeamformerParameters, contrast_mode) \
	X(BeamformerParameters, emission_parameters) X(BeamformerParameters, readi_group_count) \
	X(BeamformerParameters, readi_group) \
	X(BeamformerSimpleParameters, channel_mapping) X(BeamformerSimpleParameters, sparse_elements) \
	X(BeamformerSimpleParameters, transmit_receive_orientations) \
	X(BeamformerSimpleParameters, steering_angles) X(BeamformerSimpleParameters, focal_depths) \
	X(BeamformerSimpleParameters, compute_stages) X(BeamformerSimpleParameters, compute_stage_parameters) \
	X(BeamformerSimpleParameters, compute_stages_count) X(BeamformerSimpleParameters, data_kind) \
	X(BeamformerFilterParameters, kind) X(BeamformerFilterParameters, sampling_frequency) \
	X(BeamformerFilterParameters, complex) X(BeamformerFilterParameters, kaiser) \
	X(BeamformerLiveImagingParameters, transmit_power) X(BeamformerLiveImagingParameters, image_plane_offsets) \
	X(BeamformerLiveImagingParameters, tgc_control_points) X(BeamformerLiveImagingParameters, save_name_tag_length) \
	X(BeamformerLiveImagingParameters, save_name_tag)

/* writes "Struct.field offset\n" lines; returns bytes written */
REF_EXPORT int ref_describe_offsets(char *out, int cap)
{
	int n = 0;
	#define X(s, f) n += snprintf(out + n, cap - n, #s "." #f " %d\n", (int)offsetof(s, f));
	REF_OFFSET_LIST
	#undef X
	return n;
}

/* shared-memory protocol v33 layout (beamformer_shared_memory.c:2-166) for the headless
 * server (ogl_beamforming_amd/csrc/shm_server.cpp) */
#define REF_SHM_LIST \
	X("sizeof.BeamformerSharedMemory", sizeof(BeamformerSharedMemory)) \
	X("sizeof.BeamformerParameterBlock", sizeof(BeamformerParameterBlock)) \
	X("sizeof.BeamformWork", sizeof(BeamformWork)) \
	X("sizeof.BeamformWorkQueue", sizeof(BeamformWorkQueue)) \
	X("sizeof.BeamformerComputePipeline", sizeof(BeamformerComputePipeline)) \
	X("sizeof.BeamformerComputeStatsTable", sizeof(BeamformerComputeStatsTable)) \
	X("sizeof.Arena", sizeof(Arena)) \
	X("Arena.position", offsetof(Arena, position)) \
	X("Arena.reserved", offsetof(Arena, reserved)) \
	X("shm.version", offsetof(BeamformerSharedMemory, version)) \
	X("shm.invalid", offsetof(BeamformerSharedMemory, invalid)) \
	X("shm.locks", offsetof(BeamformerSharedMemory, locks)) \
	X("shm.reserved_parameter_blocks", offsetof(BeamformerSharedMemory, reserved_parameter_blocks)) \
	X("shm.rf_block_rf_size", offsetof(BeamformerSharedMemory, rf_block_rf_size)) \
	X("shm.beamformed_frame_buffer_size", offsetof(BeamformerSharedMemory, beamformed_frame_buffer_size)) \
	X("shm.capabilities", offsetof(BeamformerSharedMemory, capabilities)) \
	X("shm.capabilities.hilbert", offsetof(BeamformerSharedMemory, capabilities.hilbert)) \
	X("shm.live_imaging_parameters", offsetof(BeamformerSharedMemory, live_imaging_parameters)) \
	X("shm.live_imaging_dirty_flags", offsetof(BeamformerSharedMemory, live_imaging_dirty_flags)) \
	X("shm.external_work_queue", offsetof(BeamformerSharedMemory, external_work_queue)) \
	X("queue.work_items", offsetof(BeamformWorkQueue, work_items)) \
	X("work.kind", offsetof(BeamformWork, kind)) \
	X("work.lock", offsetof(BeamformWork, lock)) \
	X("work.compute_context", offsetof(BeamformWork, compute_context)) \
	X("work.create_filter.filter_slot", offsetof(BeamformWork, create_filter_context.filter_slot)) \
	X("work.create_filter.parameter_block", offsetof(BeamformWork, create_filter_context.parameter_block)) \
	X("work.export.kind", offsetof(BeamformWork, export_context.kind)) \
	X("work.export.count", offsetof(BeamformWork, export_context.count)) \
	X("work.export.size", offsetof(BeamformWork, export_context.size)) \
	X("block.parameters", offsetof(BeamformerParameterBlock, parameters)) \
	X("block.region_update_flags", offsetof(BeamformerParameterBlock, region_update_flags)) \
	X("block.pipeline", offsetof(BeamformerParameterBlock, pipeline)) \
	X("block.pipeline.parameters", offsetof(BeamformerParameterBlock, pipeline.parameters)) \
	X("block.pipeline.shader_count", offsetof(BeamformerParameterBlock, pipeline.shader_count)) \
	X("block.pipeline.data_kind", offsetof(BeamformerParameterBlock, pipeline.data_kind)) \
	X("block.channel_mapping", offsetof(BeamformerParameterBlock, channel_mapping)) \
	X("block.sparse_elements", offsetof(BeamformerParameterBlock, sparse_elements)) \
	X("block.transmit_receive_orientations", offsetof(BeamformerParameterBlock, transmit_receive_orientations)) \
	X("block.focal_vectors", offsetof(BeamformerParameterBlock, focal_vectors)) \
	X("enum.WorkKind_Compute", BeamformerWorkKind_Compute) \
	X("enum.WorkKind_ComputeIndirect", BeamformerWorkKind_ComputeIndirect) \
	X("enum.WorkKind_CreateFilter", BeamformerWorkKind_CreateFilter) \
	X("enum.WorkKind_ExportBuffer", BeamformerWorkKind_ExportBuffer) \
	X("enum.Lock_ScratchSpace", BeamformerSharedMemoryLockKind_ScratchSpace) \
	X("enum.Lock_UploadRF", BeamformerSharedMemoryLockKind_UploadRF) \
	X("enum.Lock_ExportSync", BeamformerSharedMemoryLockKind_ExportSync) \
	X("enum.Lock_DispatchCompute", BeamformerSharedMemoryLockKind_DispatchCompute) \
	X("enum.Lock_Count", BeamformerSharedMemoryLockKind_Count) \
	X("enum.Region_ComputePipeline", BeamformerParameterBlockRegion_ComputePipeline) \
	X("enum.Region_ChannelMapping", BeamformerParameterBlockRegion_ChannelMapping) \
	X("enum.Region_FocalVectors", BeamformerParameterBlockRegion_FocalVectors) \
	X("enum.Region_Parameters", BeamformerParameterBlockRegion_Parameters) \
	X("enum.Region_SparseElements", BeamformerParameterBlockRegion_SparseElements) \
	X("enum.Region_TransmitReceiveOrientations", BeamformerParameterBlockRegion_TransmitReceiveOrientations) \
	X("enum.RegionFlag_NotifyUI", BeamformerParameterRegionFlag_NotifyUI) \
	X("enum.Export_BeamformedData", BeamformerExportKind_BeamformedData) \
	X("enum.Export_Stats", BeamformerExportKind_Stats)

REF_EXPORT int ref_describe_shm_layout(char *out, int cap)
{
	int n = 0;
	#define X(name, value) n += snprintf(out + n, cap - n, "%s %d\n", name, (int)(value));
	REF_SHM_LIST
	#undef X
	return n;
}
