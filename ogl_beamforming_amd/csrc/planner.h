/* planner.h -- host-side pipeline planning (see planner.cpp) */
#ifndef BF_PLANNER_H
#define BF_PLANNER_H
#include <cstdint>
#include <string>
#include <vector>
#include "host_math.h"

namespace bf {

/* Region dirty bits: beamformer_shared_memory.c:65-89 */
enum : uint32_t {
	Dirty_ComputePipeline = 1u << 0,
	Dirty_ChannelMapping  = 1u << 1,
	Dirty_FocalVectors    = 1u << 2,
	Dirty_Parameters      = 1u << 3,
	Dirty_SparseElements  = 1u << 4,
	Dirty_Orientations    = 1u << 5,
	Dirty_Filters         = 1u << 6,   /* create_filter arrived (BeamformerWorkKind_CreateFilter) */
	Dirty_Shard           = 1u << 7,
};

/* What the reference keeps per parameter block in shared memory
 * (BeamformerParameterBlock, beamformer_shared_memory.c:102-123) plus the filter slots of its
 * compute plan (beamformer_core.c:1511-1517). */
struct ParameterBlock {
	BeamformerParameters parameters{};
	int32_t  shaders[BeamformerMaxComputeShaderStages]{};
	uint8_t  filter_slots[BeamformerMaxComputeShaderStages]{};
	uint32_t shader_count = 0;
	int32_t  data_kind    = BeamformerDataKind_Int16;
	int16_t  channel_mapping[BeamformerMaxChannelCount]{};
	int16_t  sparse_elements[BeamformerMaxChannelCount]{};
	uint8_t  transmit_receive_orientations[BeamformerMaxChannelCount]{};
	float    focal_vectors[BeamformerMaxChannelCount][2]{};
	BeamformerFilterParameters filters[BeamformerFilterSlots]{};
	uint32_t dirty = 0;
	uint32_t shard_z_first = 0, shard_z_count = 0;
};

struct Stage {
	int     kind = 0;                    /* BeamformerShaderKind */
	int     in_kind = 0, out_kind = 0;   /* BeamformerDataKind */
	int64_t in_stride[3]{}, out_stride[3]{};   /* sample, channel, transmit; elements */
	int     filter_slot = 0;
	Filter  filter;                      /* Filter / Demodulate stages */
};

struct Plan {
	std::vector<Stage> stages;
	int      das_index = -1;
	bool     iq_pipeline = false;
	int      pipeline_data_kind = 0;     /* kind of the ingested RF as the first stage sees it */
	uint32_t channels = 0, acquisitions = 0, raw_samples = 0;
	uint32_t das_samples = 0;            /* samples per (channel, transmit) row entering DAS */
	float    das_sampling_frequency = 0, das_time_offset = 0;
	uint32_t decimation = 1;
	uint32_t output_points[3]{1, 1, 1};
	float    das_voxel_transform[16]{};
	bool     das_sparse = false;
	std::vector<float> hadamard_t;       /* decode: HtT[T*i + j] */
	std::vector<float> hadamard_base;    /* B (base x base) when HtT == Sylvester (x) B entry for entry; base 1 holds {1} */
	uint32_t           hadamard_base_order = 0;
	std::vector<float> readi_hadamard;   /* G*G, row major, +-1 */
	size_t   intermediate_bytes = 0;     /* largest inter-stage buffer */
};

/* Restates plan_compute_pipeline (beamformer_core.c:553-1013) for a backend that runs every
 * receive channel in one pass: the reference's 16-channel chunk becomes channel_count.
 * Returns false with `error` set when the pipeline cannot run. */
bool build_plan(const ParameterBlock &pb, Plan &plan, std::string &error, bool allow_hilbert = false);

} // namespace bf
#endif
