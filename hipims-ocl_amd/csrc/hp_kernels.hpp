// hp_kernels.hpp -- HIP kernels of the step engine (gfx950).  See DESIGN.md for the kernel graph.
#pragma once

#include "hp_math.hpp"

namespace hp {

// -------------------------------------------------------------------------------------------------
// K0  godunov_basic : one thread per cell, four face solves per cell.
//     Same dataflow as the reference's gts_cacheDisabled (CLSchemeGodunov.clc:164-384); kept as the
//     on-device cross-check for the tuned kernel (HP_KERNEL_BASIC) -- not the performance path.
// -------------------------------------------------------------------------------------------------
template <bool STRICT, typename T>
__global__ __launch_bounds__(256) void godunov_basic(const Params<T> p, const Scalars<T>* __restrict__ sc,
                                                     const T* __restrict__ bed, const State4<T>* __restrict__ src,
                                                     State4<T>* __restrict__ dst, const T* __restrict__ manning)
{
	const long x = (long)blockIdx.x * blockDim.x + threadIdx.x;
	const long y = (long)blockIdx.y * blockDim.y + threadIdx.y;
	if (x >= p.cols - 1 || y >= p.rows - 1 || x <= 0 || y <= 0) return;          // :183-187
	const size_t id = (size_t)y * p.cols + x;
	const T dt = sc->dt;

	if (dt <= T(0)) { dst[id] = src[id]; return; }                                // :201-206

	const State4<T> c = src[id];
	const T zb = bed[id], n = manning[id];
	if (c.zmax <= T(-9999.0) || c.z == T(-9999.0)) { dst[id] = c; return; }       // :214-218

	const size_t iW = id - 1, iE = id + 1, iS = id - p.cols, iN = id + p.cols;
	const State4<T> cW = src[iW], cE = src[iE], cS = src[iS], cN = src[iN];
	const T zW = bed[iW], zE = bed[iE], zS = bed[iS], zN = bed[iN];

	int dry = 0;                                                                  // :248-255
	if (c.z  - zb < p.vs) dry++;
	if (cN.z - zN < p.vs) dry++;
	if (cE.z - zE < p.vs) dry++;
	if (cS.z - zS < p.vs) dry++;
	if (cW.z - zW < p.vs) dry++;
	if (dry >= 5) return;                                                         // dst untouched (Q3)

	const Side<T> sC = make_side(c.z, c.qx, c.qy, zb, p.vs);
	const Side<T> sN = make_side(cN.z, cN.qx, cN.qy, zN, p.vs);
	const Side<T> sE = make_side(cE.z, cE.qx, cE.qy, zE, p.vs);
	const Side<T> sS = make_side(cS.z, cS.qx, cS.qy, zS, p.vs);
	const Side<T> sW = make_side(cW.z, cW.qx, cW.qy, zW, p.vs);

	FaceFlux<T> fN, fE, fS, fW, unused;
	face_solve<AXIS_Y, STRICT, true, false>(sC, sN, p.vs, fN, unused);
	face_solve<AXIS_Y, STRICT, false, true>(sS, sC, p.vs, unused, fS);
	face_solve<AXIS_X, STRICT, true, false>(sC, sE, p.vs, fE, unused);
	face_solve<AXIS_X, STRICT, false, true>(sW, sC, p.vs, unused, fW);

	dst[id] = godunov_update<STRICT>(c, zb, n, dt, fN, fE, fS, fW, p.dx, p.vs, p.friction != 0);
}

// -------------------------------------------------------------------------------------------------
// K3  cfl_reduce : max wave speed over rows [row_lo, row_hi) of one state buffer.
//     tst_Reduce (CLDynamicTimestep.clc:166-249) + the serial max of tst_Advance_Normal (:75-80) as
//     wavefront shuffle -> LDS across the block's waves -> one exact atomic max per block.
// -------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void cfl_reduce(const Params<T> p, const State4<T>* __restrict__ state,
                                                  const T* __restrict__ bed, const long row_lo, const long row_hi,
                                                  T* __restrict__ slot)
{
	__shared__ T wave_part[4];
	const size_t first = (size_t)row_lo * p.cols, last = (size_t)row_hi * p.cols;
	T m = T(0);
	for (size_t i = first + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < last;
	     i += (size_t)gridDim.x * blockDim.x) {
		const State4<T> c = state[i];
		const T s = cfl_speed(c.z, c.zmax, c.qx, c.qy, bed[i], p.qs);
		if (s > m) m = s;
	}
	m = wave_max(m);
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	if (lane == 0) wave_part[wave] = m;
	__syncthreads();
	if (threadIdx.x == 0) {
		for (int w = 1; w < (int)(blockDim.x >> 6); ++w) if (wave_part[w] > m) m = wave_part[w];
		if (m > T(0)) atomic_max_nonneg(slot, m);
	}
}

// -------------------------------------------------------------------------------------------------
// K4  advance_time : tst_Advance_Normal (CLDynamicTimestep.clc:27-146), one lane.
//     `slot` holds the (all-reduced) maximum wave speed; it is cleared for the next accumulation.
//     UPDATE_ONLY = tst_UpdateTimestep (:255-317).
// -------------------------------------------------------------------------------------------------
template <bool UPDATE_ONLY, typename T>
__global__ void advance_time(const Params<T> p, Scalars<T>* __restrict__ sc, T* __restrict__ slot)
{
	if (threadIdx.x != 0 || blockIdx.x != 0) return;
	const T EARLY_LIMIT = T(0.1), EARLY_DURATION = T(60.0), START_MIN = T(1E-10), START_DURATION = T(1.0);
	const T DT_MIN = T(1E-10), DT_MAX = T(15.0), HYDRO = T(1.0);                 // CLDynamicTimestep.clh:24-29
	const T vmax = *slot;
	*slot = T(0);

	T t = sc->t, t_sync = sc->t_sync, batch = sc->batch_dt;
	if (UPDATE_ONLY) {
		const T dt_orig = fabs_(sc->dt);                                          // :264
		T dt = T(0);
		if (p.dynamic_dt) {
			T tmin = p.dx / vmax;
			if (t < START_DURATION && tmin < START_MIN) tmin = START_MIN;
			dt = p.courant * tmin;
		}
		dt = fmin_(dt, dt_orig);                                                  // :297-298
		batch = batch - dt_orig + dt;
		if (t < EARLY_DURATION && dt > EARLY_LIMIT) dt = EARLY_LIMIT;             // :301-302
		if ((t + dt) >= t_sync) dt = fmax_(T(0), t_sync - t);                     // :305-306
		if (dt > DT_MAX) dt = DT_MAX;                                             // :309-310
		sc->dt = dt;
		sc->batch_dt = batch;
		return;
	}

	T dt = fmax_(T(0), sc->dt);                                                   // :42
	T t_hydro = sc->t_hydro;
	uint32_t ok = sc->batch_ok, skipped = sc->batch_skipped;
	t += dt;                                                                      // :50-51
	batch += dt;
	if (dt > T(0)) ok++; else skipped++;                                          // :53-58
	if (t_hydro > HYDRO) t_hydro = dt; else t_hydro += dt;                        // :61-66

	if (p.dynamic_dt) {                                                           // :68-92
		T tmin = p.dx / vmax;
		if (t < START_DURATION && tmin < START_MIN) tmin = START_MIN;
		dt = p.courant * tmin;
	} else {
		dt = p.dt_fixed;                                                          // :93-97
	}
	if (dt > T(0) && dt < DT_MIN) dt = DT_MIN;                                    // :112-113
	if ((t + dt) >= t_sync) {                                                     // :118-124
		const T dt_in = dt;
		if (t_sync - t > p.vs)  dt = t_sync - t;
		if (t_sync - t <= p.vs) dt = -dt_in;
	}
	if (t < EARLY_DURATION && dt > EARLY_LIMIT) dt = EARLY_LIMIT;                 // :128-129
	if ((t + dt) > p.t_end) dt = p.t_end - t;                                     // :132-133
	if (dt > DT_MAX) dt = DT_MAX;                                                 // :136-137

	sc->t = t; sc->dt = dt; sc->t_hydro = t_hydro; sc->batch_dt = batch;          // :140-145
	sc->batch_ok = ok; sc->batch_skipped = skipped;
}

// -------------------------------------------------------------------------------------------------
// K5  boundary source terms applied in place to the source state before the flux kernel
//     bdy_Uniform (Boundaries/CLBoundaries.clc:130-184), bdy_Gridded (:186-246).
//     Rows are addressed globally so ghost rows of a strip receive the same rain as their owner gives them.
// -------------------------------------------------------------------------------------------------
template <typename T> struct UniformBdy { const T* series; uint32_t entries; int definition; T interval, length; };
template <typename T> struct GriddedBdy {
	const T* grids; uint64_t entries, grows, gcols; int definition; T resolution, off_x, off_y, interval;
};

template <typename T>
__device__ __forceinline__ bool bdy_in_range(const Params<T>& p, const long x, const long gy, const bool truncated)
{
	if (x >= p.cols - 1 || gy >= p.global_rows - 1 || x <= 0 || gy <= 0) return false;
	// NDRange = floor(n/8)*8 per axis (CBoundaryUniform.cpp:294-295, CBoundaryGridded.cpp:298-299; Q9)
	if (truncated && (x >= (p.cols / 8) * 8 || gy >= (p.global_rows / 8) * 8)) return false;
	return true;
}

template <typename T>
__global__ __launch_bounds__(256) void bdy_uniform(const Params<T> p, const Scalars<T>* __restrict__ sc,
                                                   const UniformBdy<T> b, State4<T>* __restrict__ state,
                                                   const T* __restrict__ bed, const bool truncated)
{
	const T t = sc->t, dt_real = sc->dt, dt = sc->t_hydro;
	if (dt < T(1.0) || dt_real <= T(0)) return;                                  // :165-166 (uniform over the grid)
	if (t >= b.length) return;                                                    // :168
	const long x = (long)blockIdx.x * blockDim.x + threadIdx.x;
	const long y = (long)blockIdx.y * blockDim.y + threadIdx.y;
	if (y >= p.rows || !bdy_in_range(p, x, y + p.row_offset, truncated)) return;
	const size_t id = (size_t)y * p.cols + x;
	State4<T> c = state[id];
	if (c.zmax <= T(-9999.0)) return;                                             // :168-169
	const unsigned long ts = (unsigned long)floor_(t / b.interval);               // :172-173
	const T rate = b.series[2 * ts + 1];
	if (b.definition == 0) c.z += rate / T(3600000.0) * dt;                       // :176-177
	if (b.definition == 1) c.z = fmax_(bed[id], c.z - rate / T(3600000.0) * dt);  // :179-180
	state[id] = c;
}

template <typename T>
__global__ __launch_bounds__(256) void bdy_gridded(const Params<T> p, const Scalars<T>* __restrict__ sc,
                                                   const GriddedBdy<T> b, State4<T>* __restrict__ state,
                                                   const bool truncated)
{
	const T t = sc->t, dt = sc->t_hydro;
	if (dt < T(1.0)) return;                                                      // :224-225
	const long x = (long)blockIdx.x * blockDim.x + threadIdx.x;
	const long y = (long)blockIdx.y * blockDim.y + threadIdx.y;
	const long gy = y + p.row_offset;
	if (y >= p.rows || !bdy_in_range(p, x, gy, truncated)) return;
	const size_t id = (size_t)y * p.cols + x;
	State4<T> c = state[id];
	if (c.zmax <= T(-9999.0) || c.z == T(-9999.0)) return;                        // :220-221
	unsigned long ts = (unsigned long)floor_(t / b.interval);                     // :228
	if (ts >= b.entries) ts = b.entries - 1;          // reference reads one slice past the end here (:229)
	const T col = floor_((((T)x * p.dx) - b.off_x) / b.resolution);               // :231-232
	const T row = floor_((((T)gy * p.dx) - b.off_y) / b.resolution);
	const unsigned long cell = (b.grows * b.gcols) * ts + (b.gcols * (unsigned long)row) + (unsigned long)col;
	const T rate = b.grids[cell];
	if (b.definition == 0) c.z += rate / T(3600000.0) * dt;                       // :238-239
	if (b.definition == 2) c.z += rate / (p.dx * p.dx) * dt;                      // :241-242
	state[id] = c;
}

} // namespace hp
