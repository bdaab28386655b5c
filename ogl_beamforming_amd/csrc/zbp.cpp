/* zbp.cpp -- loader for ZBP acquisition files (".bp" parameter headers + RF payloads).
 *
 * The on-disk layouts are those of the reference's external/zemp_bp.h (header v1 :98-121,
 * header v2 :123-151, emission and acquisition parameter records :153-198); the mapping to
 * BeamformerSimpleParameters follows beamformer_simple_parameters_from_zbp_file
 * (tests/throughput.c:150-374) including its quirks: v1 files are always Int16 + zstd with
 * fd = fs/4 and a 2-cycle sine; v2 RCA_VLS focal depth/origin pairs become (angle, radius)
 * (:347-363); RCA_TPW gets infinite focal depths (:337-338); channel_mapping_offset == -1 means
 * identity.  Unlike the reference, every offset and count read from the file is bounds
 * checked before use -- a short or hostile file is an error, not an out-of-bounds read.
 *
 * zstd payloads are decompressed with the system's libzstd.so.1, bound at run time (the image
 * ships the runtime without headers).  Without it, compressed files fail to load.
 */
#include "../../include/ogl_beamformer_hip.h"
#include "bf_kernels.h"

#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <string>
#include <vector>

namespace {

constexpr uint64_t kZbpMagic = 0x5042504d455afecaULL;      /* zemp_bp.h:23 */

struct HeaderV1 {                                           /* zemp_bp.h:98-121 */
	uint64_t magic;
	uint32_t version;
	int16_t  decode_mode, beamform_mode;
	uint32_t raw_data_dimension[4];
	uint32_t sample_count, channel_count, receive_event_count, frame_count;
	float    element_pitch[2];
	float    transform[16];
	int16_t  channel_mapping[256];
	float    steering_angles[256];
	float    focal_depths[256];
	int16_t  sparse_elements[256];
	int16_t  hadamard_rows[256];
	float    speed_of_sound, demodulation_frequency, sampling_frequency, time_offset;
	uint32_t transmit_mode;
};
static_assert(sizeof(HeaderV1) == 3728 && offsetof(HeaderV1, channel_mapping) == 120 &&
              offsetof(HeaderV1, speed_of_sound) == 3704, "ZBP v1 header layout");

struct HeaderV2 {                                           /* zemp_bp.h:123-151 */
	uint64_t magic;
	uint32_t major, minor;
	uint32_t raw_data_dimension[4];
	int32_t  raw_data_kind, raw_data_offset, raw_data_compression_kind;
	int32_t  decode_mode, sampling_mode;
	float    sampling_frequency, demodulation_frequency, speed_of_sound;
	int32_t  channel_mapping_offset;
	uint32_t sample_count, channel_count, receive_event_count;
	float    transform[16];
	float    element_pitch[2];
	float    time_offset, group_acquisition_time, ensemble_repetition_interval;
	int32_t  acquisition_mode, acquisition_parameters_offset;
	int32_t  contrast_mode, contrast_parameters_offset;
	int32_t  emission_descriptors_offset;
};
static_assert(sizeof(HeaderV2) == 184 && offsetof(HeaderV2, transform) == 80 &&
              offsetof(HeaderV2, acquisition_mode) == 164, "ZBP v2 header layout");

struct EmissionDescriptor { int32_t kind, parameters_offset; };               /* :153-156 */
struct TransmitFocus { float focal_depth, steering_angle, origin_offset; uint32_t orientation; };   /* :169-174 */

enum { Compression_None = 0, Compression_ZSTD = 1 };                          /* :87-91 */
enum { Emission_Sine = 0, Emission_Chirp = 1 };                               /* :66-70 */

thread_local std::string g_error;

bool fail(const char *what) { g_error = what; return false; }

struct Bytes {
	const uint8_t *data; uint64_t size;
	/* pointer to `count` records of `record` bytes at `offset`, or null when outside the file
	 * or not aligned for the record (the format aligns offsets to 4, zemp_bp.h:24) */
	const void *at(int64_t offset, uint64_t record, uint64_t count) const
	{
		if (offset < 0 || (uint64_t)offset > size) return nullptr;
		if ((record >= 4 && offset % 4) || (record == 2 && offset % 2)) return nullptr;
		if (record && count > (size - (uint64_t)offset) / record) return nullptr;
		return data + offset;
	}
};

bool read_file(const std::string &path, std::vector<uint8_t> &out)
{
	FILE *f = std::fopen(path.c_str(), "rb");
	if (!f) return false;
	std::fseek(f, 0, SEEK_END);
	long n = std::ftell(f);
	std::fseek(f, 0, SEEK_SET);
	bool ok = n >= 0;
	if (ok) { out.resize((size_t)n); ok = n == 0 || std::fread(out.data(), 1, (size_t)n, f) == (size_t)n; }
	std::fclose(f);
	return ok;
}

/* ZSTD_getFrameContentSize + ZSTD_decompress of the stable libzstd API (tests/throughput.c:135-148) */
bool zstd_decompress(const uint8_t *src, uint64_t size, void **out, uint64_t *out_size)
{
	typedef unsigned long long (*content_size_fn)(const void *, size_t);
	typedef size_t (*decompress_fn)(void *, size_t, const void *, size_t);
	static void *lib = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
	if (!lib) return fail("libzstd.so.1 is not available: cannot read zstd-compressed RF");
	static content_size_fn content_size = (content_size_fn)dlsym(lib, "ZSTD_getFrameContentSize");
	static decompress_fn   decompress   = (decompress_fn)dlsym(lib, "ZSTD_decompress");
	if (!content_size || !decompress) return fail("libzstd.so.1 lacks ZSTD_getFrameContentSize / ZSTD_decompress");
	unsigned long long want = content_size(src, size);
	if (want == 0ULL - 1 || want == 0ULL - 2 || want > (1ULL << 34)) return fail("zstd frame has no usable content size");
	void *mem = std::malloc(want ? want : 1);
	if (!mem) return fail("out of memory for the decompressed RF");
	size_t got = decompress(mem, want, src, size);
	if (got != want) { std::free(mem); return fail("zstd decompression failed"); }
	*out = mem; *out_size = want;
	return true;
}

/* tests/throughput.c:158-224 */
bool parameters_from_v1(const Bytes &file, BeamformerSimpleParameters *bp, BeamformerHipZbpPayload *payload)
{
	/* the last field ends 4 bytes short of the padded struct size */
	const HeaderV1 *h = (const HeaderV1 *)file.at(0, offsetof(HeaderV1, transmit_mode) + sizeof(uint32_t), 1);
	if (!h) return fail("file shorter than a v1 header");
	if (h->channel_count > BeamformerMaxChannelCount || h->receive_event_count > BeamformerMaxEmissionsCount)
		return fail("v1 header: channel or receive event count out of range");

	bp->sample_count      = h->sample_count;
	bp->channel_count     = h->channel_count;
	bp->acquisition_count = h->receive_event_count;

	bp->sampling_mode          = BeamformerSamplingMode_4X;
	bp->acquisition_kind       = (BeamformerAcquisitionKind)h->beamform_mode;
	bp->decode_mode            = (BeamformerDecodeMode)h->decode_mode;
	bp->sampling_frequency     = h->sampling_frequency;
	bp->demodulation_frequency = h->sampling_frequency / 4;
	bp->speed_of_sound         = h->speed_of_sound;
	bp->time_offset            = h->time_offset;

	std::memcpy(bp->channel_mapping,   h->channel_mapping, sizeof(int16_t) * bp->channel_count);
	std::memcpy(bp->xdc_transform,     h->transform,       sizeof(bp->xdc_transform));
	std::memcpy(bp->xdc_element_pitch, h->element_pitch,   sizeof(bp->xdc_element_pitch));
	bp->raw_data_dimensions[0] = h->raw_data_dimension[0];
	bp->raw_data_dimensions[1] = h->raw_data_dimension[1];

	bp->data_kind             = BeamformerDataKind_Int16;
	payload->data_kind        = BeamformerDataKind_Int16;
	payload->compression_kind = Compression_ZSTD;

	static const uint8_t orientation_for_mode[4] = {0x11, 0x12, 0x21, 0x22};   /* (tx << 4) | rx; Rows = 1, Columns = 2 */
	if (h->transmit_mode >= 4) return fail("v1 header: unknown transmit mode");
	bp->transmit_receive_orientation = orientation_for_mode[h->transmit_mode];

	int kind = h->beamform_mode;
	if (kind == BeamformerAcquisitionKind_FORCES  || kind == BeamformerAcquisitionKind_HERCULES ||
	    kind == BeamformerAcquisitionKind_UFORCES || kind == BeamformerAcquisitionKind_UHERCULES)
	{
		bp->single_focus = 1; bp->single_orientation = 1;
		bp->focal_vector[0] = h->steering_angles[0];
		bp->focal_vector[1] = h->focal_depths[0];
	}
	if (kind == BeamformerAcquisitionKind_UFORCES || kind == BeamformerAcquisitionKind_UHERCULES)
		std::memcpy(bp->sparse_elements, h->sparse_elements, sizeof(int16_t) * bp->acquisition_count);
	if (kind == BeamformerAcquisitionKind_RCA_TPW || kind == BeamformerAcquisitionKind_RCA_VLS) {
		std::memcpy(bp->focal_depths,    h->focal_depths,    sizeof(float) * bp->acquisition_count);
		std::memcpy(bp->steering_angles, h->steering_angles, sizeof(float) * bp->acquisition_count);
		for (uint32_t i = 0; i < bp->acquisition_count; i++)
			bp->transmit_receive_orientations[i] = (uint8_t)bp->transmit_receive_orientation;
	}

	bp->emission_parameters.kind           = BeamformerEmissionKind_Sine;
	bp->emission_parameters.sine.cycles    = 2;
	bp->emission_parameters.sine.frequency = bp->demodulation_frequency;
	return true;
}

/* tests/throughput.c:226-366 */
bool parameters_from_v2(const Bytes &file, BeamformerSimpleParameters *bp, BeamformerHipZbpPayload *payload)
{
	const HeaderV2 *h = (const HeaderV2 *)file.at(0, sizeof(HeaderV2), 1);
	if (!h) return fail("file shorter than a v2 header");
	if (h->channel_count > BeamformerMaxChannelCount || h->receive_event_count > BeamformerMaxEmissionsCount)
		return fail("v2 header: channel or receive event count out of range");
	if ((uint32_t)h->raw_data_kind >= BeamformerDataKind_Count) return fail("v2 header: unknown data kind");
	if ((uint32_t)h->sampling_mode > 1) return fail("v2 header: unknown sampling mode");

	bp->sample_count      = h->sample_count;
	bp->channel_count     = h->channel_count;
	bp->acquisition_count = h->receive_event_count;
	const uint32_t A = bp->acquisition_count;

	/* Standard -> 4X, Bandpass -> 2X (:237-241) */
	bp->sampling_mode = h->sampling_mode == 0 ? BeamformerSamplingMode_4X : BeamformerSamplingMode_2X;

	bp->acquisition_kind       = (BeamformerAcquisitionKind)h->acquisition_mode;
	bp->decode_mode            = (BeamformerDecodeMode)h->decode_mode;
	bp->sampling_frequency     = h->sampling_frequency;
	bp->demodulation_frequency = h->demodulation_frequency;
	bp->speed_of_sound         = h->speed_of_sound;
	bp->time_offset            = h->time_offset;
	bp->contrast_mode          = (BeamformerContrastMode)h->contrast_mode;

	if (h->channel_mapping_offset != -1) {
		const void *map = file.at(h->channel_mapping_offset, sizeof(int16_t), bp->channel_count);
		if (!map) return fail("v2 header: channel mapping lies outside the file");
		std::memcpy(bp->channel_mapping, map, sizeof(int16_t) * bp->channel_count);
	} else {
		for (uint32_t i = 0; i < bp->channel_count; i++) bp->channel_mapping[i] = (int16_t)i;
	}

	std::memcpy(bp->xdc_transform,     h->transform,     sizeof(bp->xdc_transform));
	std::memcpy(bp->xdc_element_pitch, h->element_pitch, sizeof(bp->xdc_element_pitch));
	bp->raw_data_dimensions[0] = h->raw_data_dimension[0];
	bp->raw_data_dimensions[1] = h->raw_data_dimension[1];

	bp->data_kind             = (BeamformerDataKind)h->raw_data_kind;
	payload->data_kind        = (uint32_t)h->raw_data_kind;
	payload->compression_kind = (uint32_t)h->raw_data_compression_kind;

	if (h->raw_data_offset != -1) {
		if (!file.at(h->raw_data_offset, 1, 0)) return fail("v2 header: raw data offset lies outside the file");
		payload->offset = (uint64_t)h->raw_data_offset;
		if (payload->compression_kind == Compression_ZSTD) {
			payload->size = file.size - payload->offset;            /* "limitation in the header format" (:271-272) */
		} else {
			uint64_t n = (uint64_t)h->raw_data_dimension[0] * h->raw_data_dimension[1];
			n *= (uint64_t)h->raw_data_dimension[2] * h->raw_data_dimension[3];
			n *= (uint64_t)bf_kind_byte_size[h->raw_data_kind];
			if (!file.at(h->raw_data_offset, 1, n)) return fail("v2 header: raw data extends past the end of the file");
			payload->size = n;
		}
	}

	/* only the first emission descriptor is looked at (:281-305) */
	{
		const EmissionDescriptor *ed = (const EmissionDescriptor *)file.at(h->emission_descriptors_offset, sizeof(EmissionDescriptor), 1);
		if (!ed) return fail("v2 header: emission descriptor lies outside the file");
		if (ed->kind == Emission_Sine) {
			const float *p = (const float *)file.at(ed->parameters_offset, sizeof(float), 2);
			if (!p) return fail("v2 header: sine parameters lie outside the file");
			bp->emission_parameters.kind           = BeamformerEmissionKind_Sine;
			bp->emission_parameters.sine.cycles    = p[0];
			bp->emission_parameters.sine.frequency = p[1];
		} else if (ed->kind == Emission_Chirp) {
			const float *p = (const float *)file.at(ed->parameters_offset, sizeof(float), 3);
			if (!p) return fail("v2 header: chirp parameters lie outside the file");
			bp->emission_parameters.kind                = BeamformerEmissionKind_Chirp;
			bp->emission_parameters.chirp.duration      = p[0];
			bp->emission_parameters.chirp.min_frequency = p[1];
			bp->emission_parameters.chirp.max_frequency = p[2];
		} else {
			return fail("v2 header: unknown emission kind");
		}
	}

	auto focus = [&](bool sparse) -> bool {                        /* HERCULES / UHERCULES (:310-332) */
		const TransmitFocus *tf = (const TransmitFocus *)file.at(h->acquisition_parameters_offset,
		                                                          sizeof(TransmitFocus) + (sparse ? 4 : 0), 1);
		if (!tf) return fail("v2 header: acquisition parameters lie outside the file");
		bp->transmit_receive_orientation = tf->orientation;
		bp->focal_vector[0] = tf->steering_angle;
		bp->focal_vector[1] = tf->focal_depth;
		bp->single_focus = 1; bp->single_orientation = 1;
		return true;
	};
	auto sparse_elements = [&]() -> bool {                          /* record = focus + int32 offset (:176-195) */
		const uint8_t *rec = (const uint8_t *)file.at(h->acquisition_parameters_offset, sizeof(TransmitFocus) + 4, 1);
		if (!rec) return fail("v2 header: acquisition parameters lie outside the file");
		int32_t offset; std::memcpy(&offset, rec + sizeof(TransmitFocus), 4);
		const void *src = file.at(offset, sizeof(int16_t), A);
		if (!src) return fail("v2 header: sparse elements lie outside the file");
		std::memcpy(bp->sparse_elements, src, sizeof(int16_t) * A);
		return true;
	};

	switch (h->acquisition_mode) {
	case BeamformerAcquisitionKind_FORCES: break;
	case BeamformerAcquisitionKind_HERCULES:  if (!focus(false)) return false; break;
	case BeamformerAcquisitionKind_UFORCES:   if (!sparse_elements()) return false; break;
	case BeamformerAcquisitionKind_UHERCULES: if (!focus(true) || !sparse_elements()) return false; break;
	case BeamformerAcquisitionKind_RCA_TPW:{
		const int32_t *p = (const int32_t *)file.at(h->acquisition_parameters_offset, sizeof(int32_t), 2);
		if (!p) return fail("v2 header: TPW parameters lie outside the file");
		const void *angles = file.at(p[0], sizeof(float), A), *orient = file.at(p[1], 1, A);
		if (!angles || !orient) return fail("v2 header: TPW arrays lie outside the file");
		std::memcpy(bp->transmit_receive_orientations, orient, A);
		std::memcpy(bp->steering_angles, angles, sizeof(float) * A);
		for (uint32_t i = 0; i < A; i++) bp->focal_depths[i] = INFINITY;
	}break;
	case BeamformerAcquisitionKind_RCA_VLS:{
		const int32_t *p = (const int32_t *)file.at(h->acquisition_parameters_offset, sizeof(int32_t), 3);
		if (!p) return fail("v2 header: VLS parameters lie outside the file");
		const float *depths  = (const float *)file.at(p[0], sizeof(float), A);
		const float *origins = (const float *)file.at(p[1], sizeof(float), A);
		const void  *orient  = file.at(p[2], 1, A);
		if (!depths || !origins || !orient) return fail("v2 header: VLS arrays lie outside the file");
		std::memcpy(bp->transmit_receive_orientations, orient, A);
		for (uint32_t i = 0; i < A; i++) {
			float depth, origin;
			std::memcpy(&depth, (const uint8_t *)depths + 4 * i, 4);
			std::memcpy(&origin, (const uint8_t *)origins + 4 * i, 4);
			float sign = depth < 0 ? -1.0f : 1.0f;                  /* Sign(), util.h */
			bp->steering_angles[i] = atan2f(origin, -depth) * 180.0f / 3.14159265358979323846f;
			bp->focal_depths[i]    = sign * sqrtf(depth * depth + origin * origin);
		}
	}break;
	default: return fail("v2 header: acquisition mode the reference's loader does not handle");
	}
	return true;
}

} // namespace

extern "C" {

const char *beamformer_hip_zbp_last_error(void) { return g_error.c_str(); }

uint32_t beamformer_hip_zbp_parameters(const void *file_bytes, uint64_t file_size,
                                       BeamformerSimpleParameters *out, BeamformerHipZbpPayload *payload)
{
	g_error.clear();
	if (!file_bytes || !out || !payload) return fail("null argument");
	std::memset(out, 0, sizeof(*out));
	std::memset(payload, 0, sizeof(*payload));
	Bytes file{(const uint8_t *)file_bytes, file_size};
	const uint64_t *magic = (const uint64_t *)file.at(0, 16, 1);
	if (!magic) return fail("file shorter than a ZBP base header");
	uint64_t m; uint32_t major;
	std::memcpy(&m, file.data, 8); std::memcpy(&major, file.data + 8, 4);
	if (m != kZbpMagic) return fail("not a ZBP file (bad magic)");
	payload->major = major;
	switch (major) {
	case 1: return parameters_from_v1(file, out, payload);
	case 2: return parameters_from_v2(file, out, payload);
	}
	return fail("unsupported ZBP major version");
}

uint32_t beamformer_hip_zbp_load(const char *path, uint32_t frame_number, BeamformerSimpleParameters *out,
                                 void **rf, uint64_t *rf_size)
{
	g_error.clear();
	if (!path || !out || !rf || !rf_size) return fail("null argument");
	*rf = nullptr; *rf_size = 0;
	std::vector<uint8_t> file;
	if (!read_file(path, file)) return fail("cannot read the parameter file");
	BeamformerHipZbpPayload payload;
	if (!beamformer_hip_zbp_parameters(file.data(), file.size(), out, &payload)) return 0;

	if (payload.size == 0) {
		/* data lives beside the header: "<name>.bp" -> "<name>_NN.zst" (tests/throughput.c:499-512) */
		std::string p(path);
		if (p.size() < 3) return fail("parameter file name is too short to derive the data file name");
		p.resize(p.size() - 3);
		char suffix[32];
		std::snprintf(suffix, sizeof(suffix), "_%02u.zst", frame_number);
		p += suffix;
		std::vector<uint8_t> compressed;
		if (!read_file(p, compressed)) return fail("cannot read the frame's .zst file");
		return zstd_decompress(compressed.data(), compressed.size(), rf, rf_size);
	}
	if (payload.compression_kind == Compression_ZSTD)
		return zstd_decompress(file.data() + payload.offset, payload.size, rf, rf_size);
	if (payload.compression_kind != Compression_None) return fail("unknown compression kind");
	void *mem = std::malloc(payload.size ? payload.size : 1);
	if (!mem) return fail("out of memory for the RF");
	std::memcpy(mem, file.data() + payload.offset, payload.size);
	*rf = mem; *rf_size = payload.size;
	return 1;
}

void beamformer_hip_zbp_free(void *rf) { std::free(rf); }

} // extern "C"
