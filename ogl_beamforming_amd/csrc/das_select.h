/* das_select.h -- which DAS kernel a frame runs, and why: ONE table of rules, host only (no HIP call), shared by the executor
 * (which launches what it says), beamformer_hip_describe_das (which reports it, also without a device) and the tests (which ask
 * instead of restating the rules).  Also the library's five diagnostic switches (beamformer_hip_set_hook). */
#ifndef BF_DAS_SELECT_H
#define BF_DAS_SELECT_H

#include "planner.h"
#include "bf_kernels.h"
#include <string>
#include <vector>

namespace bf {

/* BeamformerHipFrameTimings::das_path */
enum DasPath {
	DasPath_General = 0, DasPath_Gather = 1, DasPath_Staged = 2, DasPath_Factored = 3, DasPath_Hercules = 4,
	DasPath_Tile = 5,             /* das_tile.hip: the factored kernel with block-wide LDS staging */
	DasPath_Count = 6,
	DasPath_Zero = 7,             /* a family / interpolation the shader leaves at zero: the frame is cleared, no kernel */
};
const char *das_path_name(int path);      /* "LDS-staged kernel", ... */
const char *das_kernel_name(int path);    /* "das_rca_staged_kernel", ... */

constexpr uint32_t kStagedMinTransmits = 6;      /* das_staged.hip by default from this many transmits per channel (tools/staged_threshold.py,
                                                    profiles/r03_staged_threshold.json: 1.29 of the gather kernel's time at 4 transmits, 1.01 at 6,
                                                    1.0 at 8, 0.88 at 12, 0.75 at 16, 0.69-0.71 at 32-75; round 2's pass: 1.15, -, 0.91, 0.84, 0.75) */

/* Diagnostic switches (none is needed in production, all default off): set through beamformer_hip_set_hook ONLY -- the library
 * reads no environment variable.  They select among code paths that ship anyway (the range-checked loop every boundary wave takes,
 * the LDS-table form every non-64 x 16 tile takes, ...) so that the tests can aim at each of them.  `version` counts changes:
 * cached decisions carry it. */
struct Hooks {
	uint64_t    version = 1;
	int         staged_shape[3] = {0, 0, 0};   /* STAGED_SHAPE="u,v,w": only 2^u x 2^v tiles with 2^w-sample windows */
	bool        staged_shape_set = false;
	bool        staged_checked = false;        /* STAGED_CHECKED: the range-checked loop for every wave (it also counts window violations) */
	bool        staged_nouniform = false;      /* STAGED_NOUNIFORM: transmit tables in LDS also where the wave-uniform form applies */
	uint64_t    staged_table_cap = 2ull << 30; /* STAGED_TABLE_CAP=bytes: largest global transmit table taken (0 forces the fallback) */
	bool        debug = false;                 /* DEBUG: one line per staged plan on stderr */
};
Hooks &hooks();
bool   set_hook(const char *name, const char *value);       /* value null or "" = unset; false: unknown name */
const char *const *hook_names();                             /* null-terminated */

struct DasDecision {
	bool     valid = false;
	uint64_t generation = 0, hooks_version = 0;              /* what it was computed for */
	uint32_t z_first = 0, z_count = 0, mode = 0;
	uint32_t mode_asked = 0;             /* decide_das_parts: the caller's mode (a fallback part is decided under another) */
	BfDasArgs a{};                      /* everything but the device pointers */
	float     tile_spread = 0.f;         /* das_tile.hip: the estimated spread of a tile of 2^tile_estimate_shift voxels (0: not a factored-kernel frame) */
	uint32_t  tile_estimate_shift[3] = {0, 0, 0};
	BfDasArgs general{};                /* the same with the general kernel's tile geometry (no channel split): what the pair count runs with */
	int      path = DasPath_General;
	int      depth_axis = 2;
	BfSeparableArgs sep{};              /* Gather: its geometry; Staged: the staged kernel's */
	BfSeparableArgs sep_gather{};       /* Staged: what the gather kernel would run with -- the fallback when the staged kernel cannot be launched */
	bool            has_lds_tables = false;
	BfSeparableArgs sep_lds_tables{};   /* Staged with wave-uniform (global) tables: the shape with the tables in LDS, used when the table cannot be allocated */
	BfHerculesArgs  herc{};
	bool     hercules_prepared = false; /* Hercules: read the {sample, difference} / polynomial copy of the DAS input */
	uint64_t das_input_bytes = 0;
	std::string why[DasPath_Count];     /* why each kernel was not taken ("" for the one that runs and for kernels not considered) */
	bool     row_end_fallback = false;  /* decide_das_parts: these planes went to the kernel BEHIND the staged one because a term can reach an end of its RF row */
};

/* per-transmit constants of das.glsl:172-202 (host side; the executor uploads them) */
std::vector<BfTransmit> build_transmit_table(const ParameterBlock &pb);

/* Fills `out` for one DAS launch over planes [z_first, z_first + z_count) of the block's grid under das path `mode`
 * (beamformer_hip_set_das_path).  Pure host arithmetic. */
void decide_das(const ParameterBlock &pb, const Plan &plan, const std::vector<BfTransmit> &transmits,
                uint32_t z_first, uint32_t z_count, uint32_t mode, DasDecision &out);

/* The same, cut along z where the ROW-END rule (das_exact.h) asks for it.  The LDS-staged kernels (das_staged*.hip) and the block-staged
 * factored kernel (das_tile.hip) decide sample_rf's range test by their own index and carry no exact evaluation of the terms at the ends
 * of an RF row (their checked loops run at their register limit; every other kernel evaluates those terms itself).  They therefore get only planes on which provably no in-aperture term
 * comes within reach of an end of its row -- a host bound per plane, in double precision, over the plane's corners; on every real
 * acquisition whose rows do not end inside the image that is all of them, and `parts` holds ONE decision.  Otherwise the range is cut into
 * runs of planes: clear runs keep their kernel, the others go to the kernel behind it (gather / factored / general, which evaluate
 * row-end terms exactly).  Parts are contiguous, in z order, and cover [z_first, z_first + z_count). */
void decide_das_parts(const ParameterBlock &pb, const Plan &plan, const std::vector<BfTransmit> &transmits,
                      uint32_t z_first, uint32_t z_count, uint32_t mode, std::vector<DasDecision> &parts);
/* planes of `parts` that took the fallback */
uint32_t row_end_planes(const std::vector<DasDecision> &parts);
/* the part with the most planes (what a frame "ran on" in one word) */
const DasDecision &main_part(const std::vector<DasDecision> &parts);

} // namespace bf
#endif
