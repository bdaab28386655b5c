/* planner.cpp -- turns (parameters, user stage list) into the ordered list of kernels a
 * frame runs, with the element kind and strides of every inter-stage buffer.
 *
 * Follows the reference's two-pass graph resolution (plan_compute_pipeline,
 * beamformer_core.c:553-1013): hard layout constraints of Decode and DAS first, then
 * don't-care propagation between neighbours, a Reshape wherever kinds or strides still
 * disagree, and an implicit CoherencyWeighting node after DAS.  Differences, all
 * deliberate:
 *   - there is no 16-channel chunk: "chunk channels" is the whole channel count, so each
 *     stage is one launch per frame (the reference: channel_count/16 launches);
 *   - CoherencyWeighting stays in the plan for bookkeeping but the executor fuses it into the
 *     DAS epilogue;
 *   - cooperative-matrix Decode (a Vulkan device feature) is not modelled; the f16 -> f32
 *     element kinds it would produce are the same as the plain path's;
 *   - a plan that starts with DAS (Float32/Float32Complex data, decode off) is valid here:
 *     DAS reads the ingested RF directly (the reference cannot run it,
 *     beamformer_core.c:1353-1362 vs :1337).
 */
#include "planner.h"
#include "bf_kernels.h"
#include <cstring>

namespace bf {

namespace {

constexpr int kDontCare = BeamformerDataKind_Count;

struct Node {
	int     kind = -1, user_index = -1;
	bool    stand_in = false;          /* DAS node added for resolution only (pipelines without DAS) */
	int     in_kind = kDontCare, out_kind = kDontCare;
	int64_t in_stride[3]{}, out_stride[3]{};
};

bool undecided(const int64_t *s)             { return s[0] == 0 || s[1] == 0 || s[2] == 0; }
bool same(const int64_t *a, const int64_t *b) { return a[0] == b[0] && a[1] == b[1] && a[2] == b[2]; }
void copy3(int64_t *d, const int64_t *s)     { d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; }

} // namespace

bool build_plan(const ParameterBlock &pb, Plan &plan, std::string &error, bool allow_hilbert)
{
	const BeamformerParameters &bp = pb.parameters;
	plan = Plan{};

	bool hilbert = false, demodulate = false;
	for (uint32_t i = 0; i < pb.shader_count; i++) {
		hilbert    |= pb.shaders[i] == BeamformerShaderKind_Hilbert;
		demodulate |= pb.shaders[i] == BeamformerShaderKind_Demodulate;
	}
	if (demodulate) hilbert = false;                                   /* :567 */
	if (hilbert && !allow_hilbert) { error = "Hilbert stage is not available (capabilities.hilbert = 0)"; return false; }

	const uint32_t S = bp.sample_count, A = bp.acquisition_count, C = bp.channel_count;
	if (!S || !A || !C || C > BeamformerMaxChannelCount || A > BeamformerMaxEmissionsCount) {
		error = "sample/channel/acquisition count out of range";
		return false;
	}
	uint32_t D = bp.decimation_rate > 1 ? bp.decimation_rate : 1;
	/* a decimation beyond the row leaves no sample (and 2 * D must not wrap: found by tests/plan_fuzz.cpp) */
	if (D > S) { error = "decimation rate exceeds the sample count"; return false; }

	float    fs      = bp.sampling_frequency;
	uint32_t samples = S;
	int      in_kind = pb.data_kind;
	if (demodulate) {                                                  /* :578-587 */
		if (in_kind == BeamformerDataKind_Int16)   in_kind = BeamformerDataKind_Int16Complex;
		if (in_kind == BeamformerDataKind_Float16) in_kind = BeamformerDataKind_Float16Complex;
		if (in_kind == BeamformerDataKind_Float32) in_kind = BeamformerDataKind_Float32Complex;
		samples  = (uint32_t)(S / (2ull * D));                         /* 64-bit: 2 * D must not wrap */
		fs      /= 2.0f * (float)D;
	}
	/* The DAS kernels test sample indices with unsigned compares against S-1 / S-3 and clamp tap
	 * addresses into the row: a row shorter than the interpolation's support (1 sample nearest,
	 * 2 linear, 4 cubic -- sample_rf, das.glsl:99-124) has no valid index at all and would make those
	 * bounds wrap.  The reference would produce an all-zero frame; refuse the plan instead of
	 * launching it (decimation can take `samples` to 0 or 1). */
	{
		bool has_das = false;
		for (uint32_t i = 0; i < pb.shader_count; i++) has_das |= pb.shaders[i] == BeamformerShaderKind_DAS;
		const uint32_t support = bp.interpolation_mode == BeamformerInterpolationMode_Cubic ? 4u
		                       : bp.interpolation_mode == BeamformerInterpolationMode_Linear ? 2u : 1u;
		if (samples == 0 || (has_das && samples < support)) {
			error = "too few samples per row for the DAS interpolation after demodulation/decimation";
			return false;
		}
	}
	plan.iq_pipeline = bf_kind_complex[in_kind] != 0 || hilbert;       /* :589 */
	const int das_kind = plan.iq_pipeline ? BeamformerDataKind_Float32Complex : BeamformerDataKind_Float32;
	plan.pipeline_data_kind = in_kind;
	plan.channels = C; plan.acquisitions = A; plan.raw_samples = S;
	plan.das_samples = samples; plan.das_sampling_frequency = fs; plan.decimation = D;
	for (int i = 0; i < 3; i++) plan.output_points[i] = bp.output_points[i] > 1 ? (uint32_t)bp.output_points[i] : 1u;

	/* pass 1: nodes with their hard constraints (:609-683) */
	std::vector<Node> nodes;
	Node root;
	root.in_kind = root.out_kind = in_kind;
	root.in_stride[0] = root.out_stride[0] = 1;
	root.in_stride[1] = root.out_stride[1] = (int64_t)S * A;
	root.in_stride[2] = root.out_stride[2] = S;
	nodes.push_back(root);

	for (uint32_t it = 0; it < pb.shader_count; it++) {
		int shader = pb.shaders[it];
		if (shader == BeamformerShaderKind_Hilbert && !hilbert) continue;   /* :625 */
		if (shader == BeamformerShaderKind_Decode && bp.decode_mode == BeamformerDecodeMode_None) continue;
		if (shader == BeamformerShaderKind_Sum || shader == BeamformerShaderKind_MinMax) continue;

		Node n;
		n.kind = shader; n.user_index = (int)it;
		if (shader == BeamformerShaderKind_Decode) {                     /* :645-664 */
			if (bf_kind_element_size[in_kind] < 4 && bf_kind_complex[in_kind])
				n.in_kind = BeamformerDataKind_Float16Complex;
			n.in_stride[0] = (int64_t)C * A;      /* sample */
			n.in_stride[1] = A;                   /* channel */
			n.in_stride[2] = 1;                   /* transmit */
		} else if (shader == BeamformerShaderKind_DAS) {                 /* :666-674 */
			n.in_kind = n.out_kind = das_kind;
			n.in_stride[0]  = 1;
			n.in_stride[1]  = (int64_t)samples * A;
			n.in_stride[2]  = samples;
			n.out_stride[0] = 1;
			n.out_stride[1] = plan.output_points[0];
			n.out_stride[2] = (int64_t)plan.output_points[0] * plan.output_points[1];
		}
		nodes.push_back(n);
		if (shader == BeamformerShaderKind_DAS && bp.coherency_weighting) {   /* :676-678 */
			Node cw; cw.kind = BeamformerShaderKind_CoherencyWeighting;
			nodes.push_back(cw);
		}
	}

	/* A pipeline without DAS (the reference's decode benchmark, tests/decode.c:236-238, pushes
	 * {Decode} alone) is legal: its stages run and the frame stays zero.  The reference leaves the
	 * last stage's output strides unresolved (0, :741-742 only fixes the kind), so that stage
	 * writes every element to index 0; here the last stage gets the layout and element kind DAS
	 * would have asked for -- same work, a defined buffer -- by resolving against a stand-in DAS
	 * node that is not planned. */
	bool has_das = false;
	for (const Node &n : nodes) has_das |= n.kind == BeamformerShaderKind_DAS;
	if (!has_das) {
		Node n;
		n.kind = BeamformerShaderKind_DAS; n.stand_in = true;
		n.in_kind = n.out_kind = das_kind;
		n.in_stride[0] = 1; n.in_stride[1] = (int64_t)samples * A; n.in_stride[2] = samples;
		n.out_stride[0] = 1; n.out_stride[1] = plan.output_points[0];
		n.out_stride[2] = (int64_t)plan.output_points[0] * plan.output_points[1];
		nodes.push_back(n);
	}

	/* pass 2: propagate don't-cares, insert Reshape on mismatch (:685-739) */
	std::vector<Node> order;
	order.push_back(nodes[0]);
	for (size_t i = 1; i < nodes.size(); i++) {
		Node  node = nodes[i];
		Node &prev = order.back();
		bool  reshape = false;
		{
			bool in_dc = undecided(node.in_stride), prev_dc = undecided(prev.out_stride);
			if (prev_dc && !in_dc) copy3(prev.out_stride, node.in_stride);
			if (!prev_dc && in_dc) copy3(node.in_stride, prev.out_stride);
			if (prev_dc && in_dc)  { copy3(prev.out_stride, prev.in_stride); copy3(node.in_stride, prev.in_stride); }
			reshape |= !same(node.in_stride, prev.out_stride);
		}
		{
			bool in_dc = node.in_kind == kDontCare, prev_dc = prev.out_kind == kDontCare;
			if (prev_dc && !in_dc) prev.out_kind = node.in_kind;
			if (!prev_dc && in_dc) node.in_kind  = prev.out_kind;
			if (prev_dc && in_dc)  node.in_kind  = prev.out_kind = prev.in_kind;
			reshape |= node.in_kind != prev.out_kind;
		}
		if (reshape) {
			Node r; r.kind = BeamformerShaderKind_Reshape;
			r.in_kind = prev.out_kind;  copy3(r.in_stride,  prev.out_stride);
			r.out_kind = node.in_kind;  copy3(r.out_stride, node.in_stride);
			order.push_back(r);
		}
		order.push_back(node);
	}
	if (order.back().out_kind == kDontCare) order.back().out_kind = order.back().in_kind;
	if (order.size() - 1 > BeamformerMaxComputeShaderStages) { error = "planned pipeline exceeds 16 stages"; return false; }

	float time_offset = bp.time_offset;
	size_t widest = 0;
	for (size_t i = 1; i < order.size(); i++) {
		const Node &n = order[i];
		if (n.stand_in) continue;
		Stage st;
		st.kind = n.kind; st.in_kind = n.in_kind; st.out_kind = n.out_kind;
		copy3(st.in_stride, n.in_stride); copy3(st.out_stride, n.out_stride);
		st.filter_slot = n.user_index >= 0 ? pb.filter_slots[n.user_index] % BeamformerFilterSlots : 0;

		switch (n.kind) {
		case BeamformerShaderKind_Decode:{
			std::vector<float> h = hadamard_transpose((int)A);           /* :819-823 */
			if (h.empty()) { error = "no Hadamard construction for this acquisition count"; return false; }
			plan.hadamard_t.resize((size_t)A * A);
			for (uint32_t j = 0; j < A; j++)
				for (uint32_t k = 0; k < A; k++)
					plan.hadamard_t[(size_t)k * A + j] = h[(size_t)j * A + k];
			/* math.c:35-134 grows the matrix by Kronecker doubling from a 1x1, 12x12 or 20x20
			 * block, i.e. M = Sylvester(T/b) (x) B with the Sylvester index as the slow one.  When
			 * the uploaded matrix has exactly that form (checked entry for entry, so nothing is
			 * assumed about the construction) Decode may run as a fast Walsh-Hadamard transform
			 * over the slow index plus a dense b x b transform (stages.hip). */
			plan.hadamard_base_order = 0; plan.hadamard_base.clear();
			for (uint32_t b : {1u, 12u, 20u}) {
				if (A % b) continue;
				uint32_t n = A / b;
				if (n & (n - 1)) continue;
				const std::vector<float> &M = plan.hadamard_t;
				bool match = true;
				for (uint32_t i = 0; i < A && match; i++)
					for (uint32_t j = 0; j < A; j++) {
						float sign = (__builtin_popcount((i / b) & (j / b)) & 1) ? -1.0f : 1.0f;
						if (M[(size_t)i * A + j] != sign * M[(size_t)(i % b) * A + (j % b)]) { match = false; break; }
					}
				if (!match) continue;
				plan.hadamard_base_order = b;
				plan.hadamard_base.resize((size_t)b * b);
				for (uint32_t i = 0; i < b; i++)
					for (uint32_t j = 0; j < b; j++) plan.hadamard_base[(size_t)i * b + j] = M[(size_t)i * A + j];
				break;
			}
		}break;
		case BeamformerShaderKind_Filter:
		case BeamformerShaderKind_Demodulate:{
			if (!filter_create(pb.filters[st.filter_slot], st.filter)) {   /* :830 */
				error = "filter slot holds no usable filter";
				return false;
			}
			time_offset += st.filter.time_delay;                         /* :835 */
		}break;
		case BeamformerShaderKind_Hilbert:{
			/* build-defined stage (host_math.cpp hilbert_fir): real input only, 31 samples of delay */
			if (bf_kind_complex[st.in_kind]) { error = "the Hilbert stage needs real input"; return false; }
			st.filter.taps = hilbert_fir();
			st.filter.length = kHilbertLength;
			st.filter.complex_taps = true;
			st.filter.time_delay = (float)((kHilbertLength - 1) / 2) / fs;
			time_offset += st.filter.time_delay;
		}break;
		case BeamformerShaderKind_DAS:{
			plan.das_index = (int)plan.stages.size();
			std::memcpy(plan.das_voxel_transform, bp.das_voxel_transform, sizeof(plan.das_voxel_transform));
			uint32_t id = bp.acquisition_kind;
			if (id == BeamformerAcquisitionKind_UFORCES || id == BeamformerAcquisitionKind_FORCES)   /* :913-915 */
				m4_mul(bp.xdc_transform, bp.das_voxel_transform, plan.das_voxel_transform);
			plan.das_sparse = id == BeamformerAcquisitionKind_UFORCES || id == BeamformerAcquisitionKind_UHERCULES;
			plan.das_time_offset = time_offset;                          /* :888 */
			if (bp.readi_group_count > 1) {                              /* :932-939 */
				/* The kernel reads row readi_group of a readi_group_count^2 matrix and walks
				 * readi_group_count x acquisition_count transmit elements: bound both before anything is sized
				 * by them (an unchecked 65536 asked for a 17 GB matrix: tests/plan_fuzz.cpp) */
				if (bp.readi_group_count > BeamformerMaxEmissionsCount || bp.readi_group >= bp.readi_group_count) {
					error = "readi_group / readi_group_count out of range";
					return false;
				}
				plan.readi_hadamard = hadamard_transpose((int)bp.readi_group_count);
				if (plan.readi_hadamard.empty()) { error = "no Hadamard construction for readi_group_count"; return false; }
			}
		}break;
		default: break;
		}
		if (n.kind != BeamformerShaderKind_DAS && n.kind != BeamformerShaderKind_CoherencyWeighting) {
			/* every pre-image buffer holds channels x transmits x das_samples elements; a
			 * deinterleaving filter writes two real planes of that many elements */
			size_t bytes = (size_t)C * A * samples * 8;
			if (bytes > widest) widest = bytes;
		}
		plan.stages.push_back(std::move(st));
	}
	plan.intermediate_bytes = widest;

	/* the gather kernels index the DAS input with 32-bit byte offsets */
	if ((uint64_t)C * A * samples * (plan.iq_pipeline ? 8 : 4) >= (1ull << 32)) {
		error = "DAS input exceeds 4 GiB";
		return false;
	}
	return true;
}

} // namespace bf
