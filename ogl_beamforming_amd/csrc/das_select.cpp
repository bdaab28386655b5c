/* das_select.cpp -- which DAS kernel a frame runs, and why (das_select.h).  Host arithmetic only.
 *
 * The rules, in the order they are tried (decide_das):
 *   general     das.hip            every family; forced by path 1; the fallback
 *   gather      das_separable.hip  row-column frames whose receive and transmit delays separate over the tile axes (plan_separable)
 *   staged      das_staged*.hip    ... and whose delay spread provably fits an LDS window (plan_staged), from kStagedMinTransmits
 *   hercules    das_hercules.hip   HERCULES family on array-aligned grids, volumes and view planes (plan_hercules)
 *   factored    das_factored.hip   any frame whose index is a receive term plus a transmit term (factored_applies)
 */
#include "das_select.h"
#include "host_math.h"
#include "../../include/ogl_beamformer_lib.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace bf {

const char *das_path_name(int path)
{
	static const char *names[] = {"general kernel", "separable-delay gather kernel", "separable-delay LDS-staged kernel", "per-voxel factored kernel",
	                              "HERCULES aligned-grid kernel", "per-voxel factored kernel with block-wide LDS staging",
	                              "?", "none: the frame is cleared"};
	return path >= 0 && path <= 7 ? names[path] : "?";
}

const char *das_kernel_name(int path)
{
	static const char *names[] = {"das_kernel", "das_rca_separable_kernel", "das_rca_staged_kernel", "das_factored_kernel",
	                              "das_hercules_kernel", "das_tile_kernel", "?", "(none)"};
	return path >= 0 && path <= 7 ? names[path] : "?";
}

/* ---------------------------------------------------------------- hooks */

static Hooks g_hooks;
static const char *const g_hook_names[] = {"STAGED_SHAPE", "STAGED_CHECKED", "STAGED_NOUNIFORM", "STAGED_TABLE_CAP", "DEBUG", nullptr};
const char *const *hook_names() { return g_hook_names; }

static bool apply_hook(Hooks &h, const char *name, const char *value)
{
	const bool on = value && value[0];
	if (!std::strcmp(name, "STAGED_SHAPE")) {
		unsigned u = 0, v = 0, w = 0;
		h.staged_shape_set = on && std::sscanf(value, "%u,%u,%u", &u, &v, &w) == 3;
		h.staged_shape[0] = (int)u; h.staged_shape[1] = (int)v; h.staged_shape[2] = (int)w;
	}
	else if (!std::strcmp(name, "STAGED_CHECKED"))   h.staged_checked = on;
	else if (!std::strcmp(name, "STAGED_NOUNIFORM")) h.staged_nouniform = on;
	else if (!std::strcmp(name, "STAGED_TABLE_CAP")) h.staged_table_cap = on ? std::strtoull(value, nullptr, 0) : (2ull << 30);
	else if (!std::strcmp(name, "DEBUG"))            h.debug = on;
	else return false;
	return true;
}

Hooks &hooks() { return g_hooks; }

bool set_hook(const char *name, const char *value)
{
	Hooks &h = hooks();
	if (!name || !apply_hook(h, name, value)) return false;
	h.version++;
	return true;
}

/* ---------------------------------------------------------------- geometry rules */

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
	if (depth_axis != 1) return 1u;
	if (zcount != 1) return 2u;
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

/* das_tile.hip: an upper bound of (largest - smallest receive index) + (largest - smallest transmit index) over a tile of
 * 2^shift voxels, in samples -- what a staged window has to hold.  Per voxel axis, from the derivatives of the two distances:
 * the receive distance sqrt(lateral^2 + z^2) moves by at most sin(theta) per unit of lateral step, and the f-number test
 * (|lateral| f / z < 1/2, das.glsl:205-207) keeps sin(theta) under 1 / sqrt(1 + 4 f^2), and by at most 1 per unit of depth; a plane
 * wave's by |sin a| laterally and by at most 1 in depth; a focused or diverging wave's, and a FORCES transmit element's, by at
 * most the length of the step.  (The kernel measures the real spread per block and chunk: the estimate only decides whether the
 * block-staged kernel is worth launching.) */
static float tile_spread_estimate(const BfDasArgs &a, const std::vector<BfTransmit> &tx, const float *voxel_to_xdc, const float *voxel_to_world,
                                  const uint32_t shift[3])
{
	const float sin_rx = 1.0f / std::sqrt(1.0f + 4.0f * a.f_number * a.f_number);
	const float samples_per_metre = a.sampling_frequency * a.inv_speed_of_sound;
	float spread = 2.0f;                                                   /* the two floors */
	for (int k = 0; k < 3; k++) {
		const float n = (float)(a.size[k] > 1 ? a.size[k] - 1 : 1);
		const float xs[3] = {voxel_to_xdc[4 * k + 0] / n, voxel_to_xdc[4 * k + 1] / n, voxel_to_xdc[4 * k + 2] / n};
		const float ws[3] = {voxel_to_world[4 * k + 0] / n, voxel_to_world[4 * k + 1] / n, voxel_to_world[4 * k + 2] / n};
		float receive, transmit = 0.f;
		if (a.family == BF_DAS_RCA) {
			const bool rx_rows = !tx.empty() && (tx[0].flags & BF_RX_ROWS);
			receive = sin_rx * std::fabs(rx_rows ? xs[1] : xs[0]) + std::fabs(xs[2]);
			for (const BfTransmit &t : tx) {
				if (t.flags & BF_TX_NONE) continue;
				const float px = (t.flags & BF_TX_ROWS) ? ws[1] : ws[0];
				const float step = (t.flags & BF_TX_PLANE) ? std::fabs(px * t.sin_a) + std::fabs(ws[2] * t.cos_a) : std::sqrt(px * px + ws[2] * ws[2]);
				transmit = step > transmit ? step : transmit;
			}
		} else {
			receive  = sin_rx * std::fabs(xs[0]) + std::fabs(xs[2]);
			transmit = std::sqrt(xs[0] * xs[0] + xs[1] * xs[1] + xs[2] * xs[2]);
		}
		spread += (receive + transmit) * samples_per_metre * (float)((1u << shift[k]) - 1u);
	}
	return spread;
}

/* das_tile.hip: the spread the kernel itself would measure -- (floor max R_c - floor min R_c over the voxels inside channel c's aperture) +
 * max over the transmits of (floor max T_a - floor min T_a) -- evaluated in double precision over a 5 x 5 x 2 lattice of voxels (the
 * corners among them) of up to 27 tiles: first, middle and last along every axis, where the extremes of an image lie.  Sharper than
 * tile_spread_estimate's derivative bound (config 2: 24 against 27.9), not an upper bound (lattice, sampled tiles): the kernel still
 * decides per block and chunk, and a chunk that does not fit runs its gather loop.  Returns the largest such sum (+ 1 for the lattice). */
static float tile_spread_sampled(const BfDasArgs &a, const std::vector<BfTransmit> &tx, const ParameterBlock &pb, const float *voxel_to_world,
                                 const uint32_t ext[3], const uint32_t shift[3], uint32_t zfirst)
{
	if (tx.empty() && a.family == BF_DAS_RCA) return 0.f;
	const double fs_over_c = (double)a.sampling_frequency * (double)a.inv_speed_of_sound;
	const int C = a.channel_count, A = a.acquisition_count;
	const int first_transmit = a.family == BF_DAS_RCA ? 0 : (a.sparse != 0);
	const bool rx_rows = a.family == BF_DAS_RCA && (tx[0].flags & BF_RX_ROWS);
	auto point = [](const float *m, double x, double y, double z, double o[3]) {
		for (int r = 0; r < 3; r++) o[r] = m[r] * x + m[4 + r] * y + m[8 + r] * z + m[12 + r];
	};
	uint32_t tiles[3][3], tile_count[3];
	for (int k = 0; k < 3; k++) {
		const uint32_t n = (ext[k] + (1u << shift[k]) - 1) >> shift[k];
		tiles[k][0] = 0; tiles[k][1] = n / 2; tiles[k][2] = n - 1;
		tile_count[k] = n >= 3 ? 3 : n == 2 ? 2 : 1;
		if (n == 2) tiles[k][1] = 1;
	}
	double worst = 0;
	std::vector<double> rlo((size_t)C), rhi((size_t)C), tlo((size_t)A), thi((size_t)A);
	for (uint32_t iz = 0; iz < tile_count[2]; iz++) for (uint32_t iy = 0; iy < tile_count[1]; iy++) for (uint32_t ix = 0; ix < tile_count[0]; ix++) {
		const uint32_t t0[3] = {tiles[0][ix] << shift[0], tiles[1][iy] << shift[1], tiles[2][iz] << shift[2]};
		std::fill(rlo.begin(), rlo.end(), 1e30); std::fill(rhi.begin(), rhi.end(), -1e30);
		std::fill(tlo.begin(), tlo.end(), 1e30); std::fill(thi.begin(), thi.end(), -1e30);
		for (int sz = 0; sz < 2; sz++) for (int sy = 0; sy < 5; sy++) for (int sx = 0; sx < 5; sx++) {
			uint32_t v[3];
			const int step[3] = {sx, sy, sz * 4};
			for (int k = 0; k < 3; k++) {
				const uint32_t span = (1u << shift[k]) - 1u;
				v[k] = t0[k] + (uint32_t)((uint64_t)span * (uint32_t)step[k] / 4u);
				if (v[k] >= ext[k]) v[k] = ext[k] - 1;                       /* ragged tiles: threads outside the grid repeat its last voxel */
			}
			const double px = (double)v[0] / std::fmax(1.0, (double)a.size[0] - 1.0), py = (double)v[1] / std::fmax(1.0, (double)a.size[1] - 1.0);
			const double pz = (double)(v[2] + zfirst) / std::fmax(1.0, (double)a.size[2] - 1.0);
			double w[3], x[3];
			point(voxel_to_world, px, py, pz, w);
			if (a.family == BF_DAS_RCA) point(a.xdc_transform, w[0], w[1], w[2], x); else { x[0] = w[0]; x[1] = w[1]; x[2] = w[2]; }
			const double lateral = rx_rows ? x[1] : x[0], pitch = rx_rows ? a.pitch[1] : a.pitch[0];
			for (int c = 0; c < C; c++) {
				const double dx = lateral - c * pitch;
				if (!(std::fabs(dx * a.f_number / std::fabs(x[2])) < 0.5)) continue;
				const double r = std::sqrt(dx * dx + x[2] * x[2]) * fs_over_c;
				rlo[(size_t)c] = std::fmin(rlo[(size_t)c], r); rhi[(size_t)c] = std::fmax(rhi[(size_t)c], r);
			}
			for (int t = first_transmit; t < A; t++) {
				double d = 0;
				if (a.family == BF_DAS_RCA) {
					const BfTransmit &q = tx[(size_t)t];
					if (!(q.flags & BF_TX_NONE)) {
						const double p = (q.flags & BF_TX_ROWS) ? w[1] : w[0];
						d = (q.flags & BF_TX_PLANE) ? p * q.sin_a + w[2] * q.cos_a : std::sqrt((p - q.focus_x) * (p - q.focus_x) + (w[2] - q.focus_z) * (w[2] - q.focus_z));
					}
				} else {
					const double element = a.sparse ? (double)pb.sparse_elements[t - first_transmit] : (double)t;
					const double dy = x[1] - a.pitch[1] * C * 0.5, tdx = x[0] - a.pitch[0] * element;
					d = std::sqrt(dy * dy + x[2] * x[2] + tdx * tdx);
				}
				d *= fs_over_c;
				tlo[(size_t)t] = std::fmin(tlo[(size_t)t], d); thi[(size_t)t] = std::fmax(thi[(size_t)t], d);
			}
		}
		double ts = 0, rs = 0;
		for (int t = first_transmit; t < A; t++) ts = std::fmax(ts, std::floor(thi[(size_t)t]) - std::floor(tlo[(size_t)t]));
		for (int c = 0; c < C; c++) if (rlo[(size_t)c] <= rhi[(size_t)c]) rs = std::fmax(rs, std::floor(rhi[(size_t)c]) - std::floor(rlo[(size_t)c]));
		worst = std::fmax(worst, rs + ts);
	}
	return (float)(worst + 1.0);
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
	q.depth_major = 1u;
	const uint32_t best_u = q.u_shift, best_v = q.v_shift;
	uint32_t nu = a.size[u_axis], nv = a.size[1 - u_axis];
	q.tiles[0] = (nu + (1u << best_u) - 1) >> best_u;
	q.tiles[1] = (nv + (1u << best_v) - 1) >> best_v;
	q.tiles[2] = zcount;
	q.walk_columns = 1u;          /* the gather kernel: one column (four together doubled its HBM-side bytes at config 4, 280 -> 520 GB, time equal) */
	return true;
}

/* Upgrade a separable plan to the LDS-staged kernel (das_staged.hip) when the delay spread of a
 * tile provably fits the staging window.  The receive delay is a distance, so it changes by at
 * most one lateral voxel step (in samples) per voxel along u; the transmit delay likewise along
 * v, scaled by max|sin(angle)| when every transmit is a plane wave. */
static bool plan_staged(const BfDasArgs &a, const std::vector<BfTransmit> &tx, const float *xdc, const float *vox,
                        uint32_t zcount, BfSeparableArgs &q, bool allow_uniform = true)
{
	const Hooks &hk = hooks();
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
			/* hook STAGED_SHAPE="us,vs,ws": only this tile / window shape (testing every template instance;
			 * a window larger than needed is legal, a smaller one is not taken) */
			if (hk.staged_shape_set) {
				const unsigned fu = (unsigned)hk.staged_shape[0], fv = (unsigned)hk.staged_shape[1], fw = (unsigned)hk.staged_shape[2];
				if (fu != us || fv != vs) continue;
				if (fw < ws || fw > 6) continue;
				ws = fw;
			}
			/* complex samples, linear interpolation, x along the receive axis and a 64 x 16 tile: a wave's lanes share one row of the
			 * transmit axis, the transmit tables leave the LDS for a global table read through scalar loads (das_staged.hip, UNI).
			 * Measured faster than every other shape, so it is preferred wherever its window fits.  (An in-between
			 * 48-sample window for this form existed through round 3: 0.7 % faster at config 4 for 4 x the HBM-side traffic; removed.) */
			const bool uniform = allow_uniform && cplx && !cubic && u_axis == 0 && threads_shift == 10 && us == 6 && vs == 4 &&
			                     !hk.staged_nouniform;
			const uint32_t window = 1u << ws;
			/* window elements a thread stages per channel: 4 (complex: registers), 8 (real).  The linear kernels also rest their
			 * tap address on it -- one 16-bit shift of the element index: 4096 x 16 B and 8192 x 8 B both end at 64 KB */
			if (((uint64_t)A4 << ws) > ((uint64_t)(cplx ? 4 : 8) << threads_shift)) continue;
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
				                 (uniform && window == 32 ? 2000u : 0u);
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
	if (hk.staged_checked) best.depth_major |= 2u;       /* test hook: the range-checked loop for every wave */
	/* uniform variant: the two blocks of a CU are neighbours along u in one plane (shared rows of the global transmit table) */
	if (best.uniform && (best.depth_major & 1u)) best.depth_major |= 4u;
	if (hk.debug)
		std::fprintf(stderr, "[beamformer] staged plan: step_u %.3f step_v %.3f waves %u u %u v %u w %u chunk %u lds %u uniform %u\n",
		             step_u, step_v, best_waves, best.u_shift, best.v_shift, best.window_samples, best.channel_chunk, best.lds_bytes, best.uniform);
	if (!best_waves) return false;
	q = best;
	uint32_t nu = a.size[u_axis], nv = a.size[v_axis];
	q.tiles[0] = (nu + (1u << q.u_shift) - 1) >> q.u_shift;
	q.tiles[1] = (nv + (1u << q.v_shift) - 1) >> q.v_shift;
	q.tiles[2] = zcount;
	q.walk_columns = bf_walk_columns(q.tiles[0]);
	return true;
}

/* output rows (waves) per block of das_hercules.hip: they walk the outer elements in step.  Same-box A/B (gpurun_out/r04/ab_hbar2, round 4),
 * no barrier and 4 rows -> barrier and 4 / 8 / 16 rows: config 5 2896 -> 2705 / 2638 / 2888 ms, the reference harness's plane 18.46 -> 17.74 /
 * 17.47 / 19.43 ms, config 3 7.16 -> 7.18 / 7.26 / 7.55 ms */
static const uint32_t kHerculesRows = 8u;

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
	q.rows     = kHerculesRows;
	q.tiles[1] = (a.size[1] + q.rows - 1u) / q.rows;
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
		q.phase_local = a.complex_data && span_turns < 400.0f;         /* (false for a NaN) */
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


/* ---------------------------------------------------------------- the decision */

std::vector<BfTransmit> build_transmit_table(const ParameterBlock &pb)
{
	const BeamformerParameters &bp = pb.parameters;
	const uint32_t A = bp.acquisition_count;
	std::vector<BfTransmit> table(A, BfTransmit{});
	for (uint32_t a = 0; a < A; a++) {
		uint32_t txrx  = bp.single_orientation ? (bp.transmit_receive_orientation & 0xFFu) : pb.transmit_receive_orientations[a];
		float    angle = bp.single_focus ? bp.focal_vector[0] : pb.focal_vectors[a][0];
		float    depth = bp.single_focus ? bp.focal_vector[1] : pb.focal_vectors[a][1];
		uint32_t tx = (txrx >> 4) & 0xF, rx = txrx & 0xF;
		BfTransmit &t = table[a];
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
	return table;
}

void decide_das(const ParameterBlock &pb, const Plan &plan, const std::vector<BfTransmit> &tx,
                uint32_t zfirst, uint32_t zcount, uint32_t mode, DasDecision &out)
{
	const BeamformerParameters &bp = pb.parameters;
	out = DasDecision{};
	out.z_first = zfirst; out.z_count = zcount; out.mode = mode; out.hooks_version = hooks().version;
	const uint32_t C = plan.channels, A = plan.acquisitions, Sd = plan.das_samples;

	BfDasArgs &a = out.a;
	std::memcpy(a.xdc_transform,   bp.xdc_transform,         sizeof(a.xdc_transform));
	std::memcpy(a.voxel_transform, plan.das_voxel_transform, sizeof(a.voxel_transform));
	a.pitch[0] = bp.xdc_element_pitch[0]; a.pitch[1] = bp.xdc_element_pitch[1];
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
	/* das_exact.h: how close to an end of an RF row a kernel may still trust its own index.  The index is a sum of terms of magnitude
	 * up to M = S + |t0 fs| (distance terms and the time offset), each rounded a few times on its way: the kernels' indices and the
	 * shader's differ by a few ulp(M) = M 2^-23 (measured bound: the out-of-sample fuzz, profiles/r04_fuzz.json, runs with this margin).  The
	 * margin is M 2^-19 -- 16 ulp(M): 2^-10 sample for 512-sample rows, 2^-8 for 2048 --, never under 2^-11. */
	{
		const float magnitude = (float)Sd + std::fabs(plan.das_time_offset * plan.das_sampling_frequency);
		a.edge_margin = std::fmax(std::ldexp(magnitude, -19), std::ldexp(1.0f, -11));
		if (!(a.edge_margin < 0.25f)) a.edge_margin = 0.25f;        /* (wild parameters; also a NaN) */
	}
	a.size[0] = plan.output_points[0]; a.size[1] = plan.output_points[1]; a.size[2] = plan.output_points[2];
	a.z_first = zfirst; a.z_count = zcount;
	a.readi_group_count = bp.readi_group_count; a.readi_group = bp.readi_group;
	out.das_input_bytes = (uint64_t)C * A * Sd * (plan.iq_pipeline ? 8u : 4u);

	float to_xdc[16];
	if (a.family == BF_DAS_FORCES || a.family == BF_DAS_READI) std::memcpy(to_xdc, plan.das_voxel_transform, sizeof(to_xdc));
	else m4_mul(bp.xdc_transform, plan.das_voxel_transform, to_xdc);
	const uint32_t ext[3] = {a.size[0], a.size[1], zcount};
	/* Small frames (real-time 2-D imaging) do not fill 256 CUs with one thread per voxel:
	 * split the channel loop over K waves of a block (wave-level partial sums, combined
	 * through LDS in split order) until the launch has ~16 waves per CU (config 1, us per
	 * frame by target wave count: 2048 -> 19.9, 4096 -> 15.2, 8192 -> 15.1, 16384 -> 17.1). */
	const uint64_t voxel_waves = ((uint64_t)ext[0] * ext[1] * ext[2] + 63) / 64;
	a.split_shift = 0;
	const uint64_t split_target = 4096;
	while (!(mode & 0x10) && a.split_shift < 4 && (voxel_waves << a.split_shift) < split_target && (C >> (a.split_shift + 1)) >= 4) a.split_shift++;
	out.depth_axis = choose_tile(to_xdc, a.size, zcount, a.tile_shift, a.split_shift ? 6 : 8);
	for (int k = 0; k < 3; k++) a.blocks[k] = (ext[k] + (1u << a.tile_shift[k]) - 1) >> a.tile_shift[k];
	a.depth_major = tile_walk(out.depth_axis, zcount, a.blocks[1], a.band_rows);
	out.general = a;

	if (a.family < 0 || a.interpolation < 0 || a.interpolation > 2) {
		out.path = DasPath_Zero;
		for (auto &w : out.why) w = "acquisition kind or interpolation mode the shader leaves at zero";
		out.valid = true;
		return;
	}
	const uint32_t das_mode = mode & 0xF;
	const bool factored = factored_applies(a, tx, das_mode);
	auto &why = out.why;
	why[DasPath_Tile] = "only where the factored kernel would run (its block-staged form)";
	if (das_mode == 1) {
		why[DasPath_Gather] = why[DasPath_Staged] = why[DasPath_Hercules] = why[DasPath_Factored] = "das path 1: the general kernel was asked for";
	}
	if (!factored && why[DasPath_Factored].empty())
		why[DasPath_Factored] = a.family == BF_DAS_HERCULES || a.family == BF_DAS_READI ? "the sample index is not a receive term plus a transmit term (HERCULES / READI)"
		                      : "fewer than 3 transmits per channel chunk, or transmits with different receive orientations";

	/* The LDS-table kernels' hand-scheduled loop exists for linear interpolation; for cubic and nearest the gather kernel's
	 * generic loop loses to the factored kernel (200 ch x 33 tx -> 129 x 333 x 21, cubic: 7.6 ms against 4.8 ms; nearest 2.6
	 * against 2.1), which then goes first -- except for cubic IQ frames with enough transmits, which try the staged cubic kernel
	 * (it declines, and the factored kernel runs, when the geometry is not separable or the spread does not fit a window). */
	const bool want_staged  = das_mode == 3 || (das_mode == 0 && A >= kStagedMinTransmits);
	const bool staged_cubic = a.interpolation == 2 && plan.iq_pipeline && want_staged;
	bool tables_first = a.interpolation == 1 || das_mode == 3 || staged_cubic || !factored;
	BfSeparableArgs sep{};
	bool separable = false, staged = false;
	if (das_mode != 1 && das_mode != 4) {
		separable = plan_separable(a, tx, bp.xdc_transform, plan.das_voxel_transform, zcount, sep);
		if (!separable) why[DasPath_Gather] = why[DasPath_Staged] = "not a row-column frame whose receive and transmit delays separate over the voxel x / y axes with depth on z";
	} else if (das_mode == 4) {
		why[DasPath_Gather] = why[DasPath_Staged] = "das path 4: the factored kernel was asked for";
	}
	BfSeparableArgs gather = sep;
	if (separable) {
		sep.zero_offset = gather.zero_offset = (uint32_t)out.das_input_bytes;       /* 64 zero bytes right behind the DAS input */
		if (!want_staged) {
			why[DasPath_Staged] = das_mode == 2 ? "das path 2: no LDS staging" : "fewer than 6 transmits per channel: two barriers and a window copy per channel are not amortised";
		} else {
			staged = plan_staged(a, tx, bp.xdc_transform, plan.das_voxel_transform, zcount, sep);
			if (!staged) {
				sep = gather;
				why[DasPath_Staged] = a.interpolation == 0 ? "nearest interpolation" : (a.interpolation == 2 && !plan.iq_pipeline) ? "cubic interpolation of real samples"
				                    : out.das_input_bytes >= (1ull << 31) ? "DAS input of 2 GiB or more (32-bit staging offsets)"
				                    : "the delay spread of no tile shape fits a 32- / 64-sample window (or the windows do not fit the LDS)";
			}
		}
		if (!staged && staged_cubic && das_mode != 3 && factored) tables_first = false;          /* cubic: the factored kernel, not the gather kernel's generic loop */
		if (!staged && !tables_first) why[DasPath_Gather] = "cubic / nearest interpolation: its generic loop loses to the factored kernel";
	}
	if (separable && tables_first) {
		out.sep = sep; out.sep_gather = gather;
		out.path = staged ? DasPath_Staged : DasPath_Gather;
		if (staged && sep.uniform) {
			/* the wave-uniform transmit tables live in global memory (a few MB to 244 MB at 512^3 with 75 transmits); too big or no
			 * memory at launch: the shape with the tables in LDS, planned here */
			BfSeparableArgs again = gather;
			out.has_lds_tables = plan_staged(a, tx, bp.xdc_transform, plan.das_voxel_transform, zcount, again, false);
			if (out.has_lds_tables) out.sep_lds_tables = again;
			const uint64_t table_bytes = (uint64_t)sep.table_stride * sep.tiles[1] * sep.tiles[2];
			if (table_bytes > hooks().staged_table_cap) {
				if (out.has_lds_tables) out.sep = again;
				else { out.sep = gather; out.path = DasPath_Gather; why[DasPath_Staged] = "global transmit table too large and no LDS-table shape fits"; }
			}
		}
		if (out.path == DasPath_Staged) why[DasPath_Gather] = "superseded by the LDS-staged kernel";
		why[DasPath_Hercules] = "not a HERCULES-family acquisition";
		if (why[DasPath_Factored].empty()) why[DasPath_Factored] = "superseded by the separable-delay kernels";
		why[DasPath_General] = "a specialised kernel applies";
		out.valid = true;
		return;
	}
	if (das_mode != 1) {
		BfHerculesArgs hq{};
		if (plan_hercules(a, tx, bp.xdc_transform, plan.das_voxel_transform, zcount, das_mode == 6, hq)) {
			hq.zero_offset = (uint32_t)out.das_input_bytes;
			/* linear / cubic interpolation of IQ samples may read a prepared copy of the input ({sample, difference}, 16 bytes; cubic: the
			 * four polynomial coefficients of every segment, 32 bytes; 32-bit byte offsets: under 4 GiB).  Not on coarse grids: the copy
			 * is 2-4 x the RF, and where every lane reads its own cache line the memory system pays for the bytes -- the harness's view
			 * plane with cubic polynomials: 28.1 ms, 158 GB from beyond L2 per frame; with the taps gathered from the RF itself 25.2 ms */
			const uint64_t prepared = out.das_input_bytes * (a.interpolation == 2 ? 4u : 2u);
			out.hercules_prepared = plan.iq_pipeline && (a.interpolation == 1 || a.interpolation == 2) && prepared + 64 < (1ull << 32) &&
			                        lane_step_samples(to_xdc, a) < 1.0f;
			out.herc = hq;
			out.path = DasPath_Hercules;
			why[DasPath_General] = "a specialised kernel applies";
			if (why[DasPath_Factored].empty()) why[DasPath_Factored] = "the sample index is not a receive term plus a transmit term (HERCULES)";
			out.valid = true;
			return;
		}
		why[DasPath_Hercules] = a.family != BF_DAS_HERCULES ? "not a HERCULES-family acquisition"
		                      : a.split_shift ? "a small frame: the general kernel's channel split fills the chip better"
		                      : "grid narrower than 32 voxels along x / more than 25 % idle lanes, or no transducer axis is a function of the output row alone";
	}
	if (factored) {
		a.zero_offset = (uint32_t)out.das_input_bytes;
		/* das_tile.hip: cubic IQ frames on FINE grids -- there the four taps of a term are two gather instructions through L1 (32.6 clk
		 * per CU per term: what config 2 waits for) and a 64 x 16-voxel block touches a window of a few dozen samples of every RF row, which
		 * the block stages as cubic polynomials.  Tried when the estimated spread of such a tile fits a 64-sample window (the kernel measures
		 * the real spread per block and chunk and falls back to the gather loop itself, so the estimate only decides whether it is worth
		 * trying); flag 0x100 forces it wherever it is supported, 0x200 forbids it. */
		{
			bool tile_ok = plan.iq_pipeline && a.interpolation == 2 && Sd >= 8 && out.das_input_bytes < (1ull << 31) &&
			               A - (a.family == BF_DAS_FORCES && a.sparse ? 1u : 0u) >= 4u;
			/* tile: 64 voxels along x (a wave), the other 16 along the next axis that has voxels */
			uint32_t shift[3] = {0, 0, 0}, left = 10;
			for (int k = 0; k < 3 && left; k++) {
				uint32_t room = ceil_log2(ext[k]), want = k == 0 ? 6u : left;
				uint32_t give = room < want ? room : want;
				give = give < left ? give : left;
				shift[k] = give; left -= give;
			}
			for (int k = 0; k < 3 && left; k++) { uint32_t room = ceil_log2(ext[k]) - shift[k]; uint32_t give = room < left ? room : left; shift[k] += give; left -= give; }
			/* (left > 0: a frame -- or a device's slab of it -- that does not even span one 1024-voxel tile) */
			/* the derivative bound first (cheap, a true upper bound); where it does not already say "32 samples", what the kernel would measure
			 * on the image's extreme tiles */
			float spread = tile_spread_estimate(a, tx, to_xdc, plan.das_voxel_transform, shift);
			if (left == 0 && spread == spread && spread > 26.f && spread < 400.f) {
				const float sampled = tile_spread_sampled(a, tx, pb, plan.das_voxel_transform, ext, shift, zfirst);
				if (sampled == sampled && sampled < spread) spread = sampled;
			}
			out.tile_spread = left == 0 ? spread : 0.f;
			for (int k = 0; k < 3; k++) out.tile_estimate_shift[k] = shift[k];
			/* frames small enough for the channel split (under 4096 voxel waves): one block per CU, so the block-staged kernel is
			 * only worth it from about three quarters of the CUs (config 2 onto 448 x 448: 196 blocks, 0.66 ms against the split
			 * kernel's 0.82; onto 384 x 384 the split kernel wins: profiles/r03_tile_threshold.json) */
			uint64_t tile_blocks = 1;
			for (int k = 0; k < 3; k++) tile_blocks *= (ext[k] + (1u << shift[k]) - 1) >> shift[k];
			tile_ok = tile_ok && (!a.split_shift || tile_blocks >= 192u);
			const bool forced = (mode & 0x100) != 0;
			if (tile_ok && left == 0 && !(mode & 0x200) && (forced || (spread == spread && spread <= 58.f && lane_step_samples(to_xdc, a) < 1.0f))) {
				a.split_shift = 0;
				for (int k = 0; k < 3; k++) a.tile_shift[k] = shift[k];
				for (int k = 0; k < 3; k++) a.blocks[k] = (ext[k] + (1u << a.tile_shift[k]) - 1) >> a.tile_shift[k];
				a.depth_major = tile_walk(out.depth_axis, zcount, a.blocks[1], a.band_rows);
				a.tile_window_shift = spread <= 26.f ? 5u : 6u;
				out.path = DasPath_Tile;
				why[DasPath_Factored] = "superseded by its block-staged form (das_tile.hip)";
				why[DasPath_General] = "a specialised kernel applies";
				out.valid = true;
				return;
			}
			why[DasPath_Tile] = !plan.iq_pipeline || a.interpolation != 2 ? "cubic interpolation of IQ samples only"
			                  : !tile_ok ? "small frame (channel split: fewer than 192 blocks), fewer than 4 transmits, or a DAS input of 2 GiB or more"
			                  : (mode & 0x200) ? "das path flag 0x200: no block staging"
			                  : "coarse grid or steep delays: a 64 x 16-voxel tile's estimated spread exceeds a 64-sample window";
		}
		out.path = DasPath_Factored;
		why[DasPath_General] = "a specialised kernel applies";
		out.valid = true;
		return;
	}
	out.path = DasPath_General;
	out.valid = true;
}

/* ---------------------------------------------------------------- row ends (das_exact.h), host side */

namespace {

struct Range { double lo, hi; };

/* smallest and largest |v| of values v in [lo, hi] */
Range abs_range(double lo, double hi)
{
	if (lo <= 0 && hi >= 0) return {0.0, std::fmax(-lo, hi)};
	return {std::fmin(std::fabs(lo), std::fabs(hi)), std::fmax(std::fabs(lo), std::fabs(hi))};
}

double point_segment_distance(double px, double py, double ax, double ay, double bx, double by)
{
	const double vx = bx - ax, vy = by - ay, wx = px - ax, wy = py - ay;
	const double vv = vx * vx + vy * vy;
	double t = vv > 0 ? (wx * vx + wy * vy) / vv : 0.0;
	t = t < 0 ? 0 : (t > 1 ? 1 : t);
	const double dx = ax + t * vx - px, dy = ay + t * vy - py;
	return std::sqrt(dx * dx + dy * dy);
}

/* Bounds on the sample index of every IN-APERTURE term of plane z (whole grid coordinates), for the families the staged kernels take
 * (RCA with one receive orientation; FORCES).  Voxel -> world -> transducer is affine, so coordinates take their extremes at the
 * plane's corners; distances to points are convex (largest at a corner) and bounded below by the distance to the corners' hull. */
Range plane_index_bounds(const BfDasArgs &a, const std::vector<BfTransmit> &tx, const float *vox, const float *to_xdc,
                         const int16_t *sparse_elements, uint32_t z)
{
	double world[4][3], xdc[4][3];
	const double pz = (double)z / (double)(a.size[2] > 1 ? a.size[2] - 1 : 1);
	for (int k = 0; k < 4; k++) {
		const double px = (k & 1) && a.size[0] > 1 ? 1.0 : 0.0, py = (k & 2) && a.size[1] > 1 ? 1.0 : 0.0;
		for (int i = 0; i < 3; i++) {
			world[k][i] = vox[i] * px + vox[4 + i] * py + vox[8 + i] * pz + vox[12 + i];
			xdc[k][i]   = to_xdc[i] * px + to_xdc[4 + i] * py + to_xdc[8 + i] * pz + to_xdc[12 + i];
		}
	}
	auto over_corners = [&](auto f) { Range r{1e300, -1e300}; for (int k = 0; k < 4; k++) { double v = f(k); r.lo = std::fmin(r.lo, v); r.hi = std::fmax(r.hi, v); } return r; };
	const bool forces = a.family == BF_DAS_FORCES;
	const int  r_axis = forces ? 0 : ((tx[0].flags & BF_RX_ROWS) ? 1 : 0);
	const Range zr = over_corners([&](int k) { return xdc[k][2]; });
	const Range za = abs_range(zr.lo, zr.hi);
	/* receive: |dx| < |z| / (2 F#) inside the aperture, and never beyond what the array and the plane allow */
	auto reach_along = [&](int axis) {
		const Range lat = over_corners([&](int k) { return xdc[k][axis]; });
		const double last = (double)(a.channel_count - 1) * (double)a.pitch[axis];
		return std::fmax(std::fmax(std::fabs(lat.lo), std::fabs(lat.hi)), std::fmax(std::fabs(lat.lo - last), std::fabs(lat.hi - last)));
	};
	double dx_max = reach_along(r_axis);
	if (!forces && a.family == BF_DAS_RCA)
		for (const BfTransmit &t : tx)                          /* per-transmit receive orientations (general kernel): the larger reach */
			if (((t.flags & BF_RX_ROWS) ? 1 : 0) != r_axis) { dx_max = std::fmax(dx_max, reach_along(1 - r_axis)); break; }
	if (a.f_number > 0) dx_max = std::fmin(dx_max, 0.5 * za.hi / (double)a.f_number);
	const double rx_max = std::sqrt(dx_max * dx_max + za.hi * za.hi), rx_min = za.lo;
	const double c = a.speed_of_sound, fs = a.sampling_frequency, t0 = a.time_offset;
	Range index{1e300, -1e300};
	/* das.glsl:187-202 over the plane: a plane wave's "distance" is affine (extremes at the corners); a focused or diverging wave's is
	 * the distance to the focus: largest at a corner, smallest zero if the focus lies inside the (px, pz) image of the plane, else the
	 * distance to its boundary (corner order 0, 1, 3, 2 walks the parallelogram) */
	auto transmit_range = [&](const BfTransmit &t) {
		Range d{0.0, 0.0};
		if (t.flags & BF_TX_NONE) return d;
		const int w = (t.flags & BF_TX_ROWS) ? 1 : 0;
		if (t.flags & BF_TX_PLANE) return over_corners([&](int k) { return world[k][w] * (double)t.sin_a + world[k][2] * (double)t.cos_a; });
		d = over_corners([&](int k) { return std::hypot(world[k][w] - (double)t.focus_x, world[k][2] - (double)t.focus_z); });
		const int order[4] = {0, 1, 3, 2};
		double nearest = 1e300; int sign = 0; bool inside = true;
		for (int e = 0; e < 4; e++) {
			const int i = order[e], j = order[(e + 1) & 3];
			nearest = std::fmin(nearest, point_segment_distance(t.focus_x, t.focus_z, world[i][w], world[i][2], world[j][w], world[j][2]));
			const double cross = (world[j][w] - world[i][w]) * ((double)t.focus_z - world[i][2]) - (world[j][2] - world[i][2]) * ((double)t.focus_x - world[i][w]);
			if (cross != 0) { const int sg = cross > 0 ? 1 : -1; if (sign && sg != sign) inside = false; sign = sg; }
		}
		d.lo = (inside && sign) ? 0.0 : nearest;
		return d;
	};
	if (a.family == BF_DAS_HERCULES) {
		/* das.glsl:233-286: index = T0(voxel) + sqrt(z^2 + e^2) fs / c over element pairs with e^2 < z^2 / (4 F#^2); without an f-number
		 * every pair passes: e^2 then is bounded by the array (both lateral axes) */
		double e_max;
		if (a.f_number > 0) e_max = 0.5 * za.hi / (double)a.f_number;
		else {
			const Range lx = over_corners([&](int k) { return xdc[k][0]; }), ly = over_corners([&](int k) { return xdc[k][1]; });
			const double n = (double)(a.channel_count > a.acquisition_count ? a.channel_count : a.acquisition_count) + 1.0;
			double far_x = std::fabs(n * a.pitch[0]), far_y = std::fabs(n * a.pitch[1]);
			if (a.sparse) for (int t = 0; t + 1 < a.acquisition_count; t++) { far_x = std::fmax(far_x, std::fabs((double)sparse_elements[t] * a.pitch[0])); far_y = std::fmax(far_y, std::fabs((double)sparse_elements[t] * a.pitch[1])); }
			const double ex = std::fmax(std::fabs(lx.lo), std::fabs(lx.hi)) + far_x, ey = std::fmax(std::fabs(ly.lo), std::fabs(ly.hi)) + far_y;
			e_max = std::sqrt(ex * ex + ey * ey);
		}
		const Range d = transmit_range(tx[0]);
		index.lo = (d.lo / c + t0) * fs + za.lo * fs / c;
		index.hi = (d.hi / c + t0) * fs + std::sqrt(za.hi * za.hi + e_max * e_max) * fs / c;
		return index;
	}
	if (forces) {
		/* das.glsl:288-321: every decoded transmit element e: sqrt((y - pitch_y C / 2)^2 + z^2 + (x - pitch_x e)^2) */
		const double half = (double)a.pitch[1] * (double)a.channel_count / 2.0;
		const Range ty = over_corners([&](int k) { return xdc[k][1] - half; });
		const Range tya = abs_range(ty.lo, ty.hi);
		double e_lo = a.sparse ? 1e300 : 0.0, e_hi = a.sparse ? -1e300 : (double)(a.acquisition_count - 1);
		if (a.sparse) for (int t = 0; t + 1 < a.acquisition_count; t++) { e_lo = std::fmin(e_lo, sparse_elements[t]); e_hi = std::fmax(e_hi, sparse_elements[t]); }
		const Range xs = over_corners([&](int k) { return xdc[k][0]; });
		const Range txa = abs_range(xs.lo - (double)a.pitch[0] * (a.pitch[0] >= 0 ? e_hi : e_lo), xs.hi - (double)a.pitch[0] * (a.pitch[0] >= 0 ? e_lo : e_hi));
		const double tx_max = std::sqrt(tya.hi * tya.hi + za.hi * za.hi + txa.hi * txa.hi);
		const double tx_min = std::sqrt(tya.lo * tya.lo + za.lo * za.lo + txa.lo * txa.lo);
		index.lo = (rx_min / c + t0) * fs + tx_min * fs / c;
		index.hi = (rx_max / c + t0) * fs + tx_max * fs / c;
		return index;
	}
	for (const BfTransmit &t : tx) {
		const Range d = transmit_range(t);
		index.lo = std::fmin(index.lo, ((d.lo + rx_min) / c + t0) * fs);
		index.hi = std::fmax(index.hi, ((d.hi + rx_max) / c + t0) * fs);
	}
	return index;
}

} // namespace

void decide_das_parts(const ParameterBlock &pb, const Plan &plan, const std::vector<BfTransmit> &tx,
                      uint32_t zfirst, uint32_t zcount, uint32_t mode, std::vector<DasDecision> &parts)
{
	parts.clear();
	parts.emplace_back();
	decide_das(pb, plan, tx, zfirst, zcount, mode, parts[0]);
	DasDecision &whole = parts[0];
	whole.a.row_ends = whole.general.row_ends = 1;           /* until shown otherwise */
	if (whole.path == DasPath_Zero || zcount == 0 || tx.empty()) return;
	const BfDasArgs &a = whole.a;
	/* a bound exists for the families whose index is a sum of distances over an aperture the f-number limits (plane_index_bounds);
	 * READI keeps row_ends = 1, and so does nearest interpolation (which has no row-end evaluation: its flips are everywhere) */
	if (a.family == BF_DAS_READI || a.family < 0 || a.interpolation == 0) return;
	float to_xdc[16];
	if (a.family == BF_DAS_FORCES) std::memcpy(to_xdc, plan.das_voxel_transform, sizeof(to_xdc));
	else m4_mul(pb.parameters.xdc_transform, plan.das_voxel_transform, to_xdc);
	const bool   cubic = a.interpolation == 2;
	const double reach = 2.0 * (double)a.edge_margin;        /* the kernels' indices and these bounds differ by far less than the margin */
	const double lo = (cubic ? 1.0 : 0.0) + reach, hi = (double)(a.sample_count - (cubic ? 2 : 1)) - reach;
	std::vector<uint8_t> clear(zcount);
	bool all = true;
	for (uint32_t k = 0; k < zcount; k++) {
		const Range r = plane_index_bounds(a, tx, plan.das_voxel_transform, to_xdc, pb.sparse_elements, zfirst + k);
		clear[k] = r.lo >= lo && r.hi < hi;                  /* (false for a NaN) */
		if (hooks().debug && (k == 0 || k + 1 == zcount))
			std::fprintf(stderr, "[beamformer] row ends: plane %u index within [%.3f, %.3f], clear within [%.3f, %.3f): %s\n", zfirst + k, r.lo, r.hi, lo, hi, clear[k] ? "clear" : "fallback");
		all = all && clear[k];
	}
	if (all) { whole.a.row_ends = whole.general.row_ends = 0; return; }      /* no term of this launch comes near a row end */
	if (whole.path != DasPath_Staged && whole.path != DasPath_Tile) return;    /* every other kernel evaluates such terms itself */
	/* the kernel behind the staged one: "automatic, never staged" with block staging off */
	const uint32_t fallback_mode = (mode & ~0xFu & ~0x100u) | 0x200u | ((mode & 0xFu) == 1u ? 1u : 2u);
	std::vector<DasDecision> cut;
	uint32_t runs = 1;
	for (uint32_t k = 1; k < zcount; k++) runs += clear[k] != clear[k - 1];
	if (runs > 8) std::fill(clear.begin(), clear.end(), (uint8_t)0);         /* (a geometry that alternates: one launch of the fallback) */
	for (uint32_t begin = 0, k = 1; k <= zcount; k++) {
		if (k < zcount && clear[k] == clear[begin]) continue;
		cut.emplace_back();
		DasDecision &d = cut.back();
		decide_das(pb, plan, tx, zfirst + begin, k - begin, clear[begin] ? mode : fallback_mode, d);
		if (!clear[begin] && (d.path == DasPath_Staged || d.path == DasPath_Tile))
			decide_das(pb, plan, tx, zfirst + begin, k - begin, (mode & ~0xFu & ~0x100u) | 0x200u | 1u, d);     /* (cannot happen: the general kernel) */
		d.row_end_fallback = !clear[begin];
		d.a.row_ends = d.general.row_ends = clear[begin] ? 0u : 1u;
		begin = k;
	}
	parts.swap(cut);
}

uint32_t row_end_planes(const std::vector<DasDecision> &parts)
{
	uint32_t n = 0;
	for (const DasDecision &d : parts) if (d.row_end_fallback) n += d.z_count;
	return n;
}

const DasDecision &main_part(const std::vector<DasDecision> &parts)
{
	size_t best = 0;
	for (size_t i = 1; i < parts.size(); i++) if (parts[i].z_count > parts[best].z_count) best = i;
	return parts[best];
}

} // namespace bf
