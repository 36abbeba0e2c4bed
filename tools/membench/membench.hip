// Memory-pattern probe for the marching kernels (not product code): how fast can the chip move the bytes of one
// Godunov step (read 32 B state + 8 B bed per cell, write 32 B state) with different per-lane access shapes?
//   0  linear float4 copy of the same byte count (the guide's 6.3 TB/s reference shape)
//   1  row march, lane = cell, state as two 16-B loads at stride 32 (what K1/K2/K6 do), 2 rows in flight
//   2  row march, lane-contiguous 16-B loads (two instructions cover 2 KB contiguously), stores likewise
//   3  like 1 with 4 rows in flight
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_linear(const d2* __restrict__ src, const double* __restrict__ bed, d2* __restrict__ dst, size_t cells)
{
	const size_t n2 = cells * 2;                       // 16-B elements of the state
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	const size_t stride = (size_t)gridDim.x * 256;
	double acc = 0;
	for (; i < n2; i += stride) {
		d2 v = src[i];
		if ((i & 3) == 0) acc += bed[i >> 1];          // 8 B per cell, coalesced enough
		v.x += acc * 1e-300;
		dst[i] = v;
	}
}

// Round 3: k_linear above issues ONE 16-B load per loop trip (plus a conditional second one), so a
// wave never has more than a few hundred bytes in flight and the "copy ceiling" it measured was its own latency bound.
// k_copy_u is the guide's float4 copy done properly: UNROLL independent 16-B loads per lane issued back to back, then
// UNROLL stores, grid-stride over the buffer.  `bed_too` adds the 8 B/cell bed read (one 16-B load per two cells).
template <int UNROLL>
__global__ __launch_bounds__(256) void k_copy_u(const d2* __restrict__ src, const d2* __restrict__ bed2, d2* __restrict__ dst,
                                                size_t n2, int bed_too)
{
	const size_t chunk = (size_t)256 * UNROLL;                        // 16-B elements per block trip
	double acc = 0;
	for (size_t base = (size_t)blockIdx.x * chunk; base < n2; base += (size_t)gridDim.x * chunk) {
		d2 v[UNROLL];
		#pragma unroll
		for (int k = 0; k < UNROLL; ++k) {
			const size_t i = base + (size_t)k * 256 + threadIdx.x;
			v[k] = i < n2 ? __builtin_nontemporal_load(&src[i]) : d2{0, 0};
		}
		if (bed_too) {
			// bed is a quarter of the state's bytes: one 16-B load for every four state elements
			#pragma unroll
			for (int k = 0; k < UNROLL; k += 4) {
				const size_t i = (base + (size_t)k * 256) / 4 + threadIdx.x;
				if (i < n2 / 4) { const d2 b = bed2[i]; acc += b.x + b.y; }
			}
		}
		#pragma unroll
		for (int k = 0; k < UNROLL; ++k) {
			const size_t i = base + (size_t)k * 256 + threadIdx.x;
			if (k == 0) v[k].x += acc * 1e-300;
			if (i < n2) dst[i] = v[k];
		}
	}
}
template <int UNROLL>
__global__ __launch_bounds__(256) void k_copy_u_plain(const d2* __restrict__ src, const d2* __restrict__ bed2, d2* __restrict__ dst,
                                                      size_t n2, int bed_too)
{
	const size_t chunk = (size_t)256 * UNROLL;
	double acc = 0;
	for (size_t base = (size_t)blockIdx.x * chunk; base < n2; base += (size_t)gridDim.x * chunk) {
		d2 v[UNROLL];
		#pragma unroll
		for (int k = 0; k < UNROLL; ++k) {
			const size_t i = base + (size_t)k * 256 + threadIdx.x;
			v[k] = i < n2 ? src[i] : d2{0, 0};
		}
		if (bed_too) {
			#pragma unroll
			for (int k = 0; k < UNROLL; k += 4) {
				const size_t i = (base + (size_t)k * 256) / 4 + threadIdx.x;
				if (i < n2 / 4) { const d2 b = bed2[i]; acc += b.x + b.y; }
			}
		}
		#pragma unroll
		for (int k = 0; k < UNROLL; ++k) {
			const size_t i = base + (size_t)k * 256 + threadIdx.x;
			if (k == 0) v[k].x += acc * 1e-300;
			if (i < n2) dst[i] = v[k];
		}
	}
}

// k_march_halo: like k_march<2, false> but every tile also READS one row below and one above its own rows (what the stencil
// kernels do: 2 of 18 rows read twice) -- does that re-read cost time, or does it come out of a cache for free?
__global__ __launch_bounds__(256) void k_march_halo(const d2* __restrict__ src, const double* __restrict__ bed, d2* __restrict__ dst,
                                                    int cols, int rows, int rseg, int groups, int ntiles, int halo)
{
	const unsigned per_xcd = gridDim.x >> 3;
	const unsigned tile = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int strip = (tile % groups) * 4 + wave, seg = tile / groups;
	if (tile >= (unsigned)ntiles) return;
	const int x0 = strip * 64;
	if (x0 >= cols) return;
	const int y0 = seg * rseg, y1 = min(y0 + rseg, rows);
	d2 a[2], b[2]; double z[2];
	auto load = [&](int y, int slot) {
		y = min(max(y, 0), rows - 1);
		const size_t cell = (size_t)y * cols + x0;
		a[slot] = src[(cell + lane) * 2]; b[slot] = src[(cell + lane) * 2 + 1];
		z[slot] = bed[cell + lane];
	};
	double acc = 0;
	// rows y0-halo .. y0-1 are read and folded into acc (never stored)
	for (int y = y0 - halo; y < y0; ++y) { load(y, 0); acc += a[0].x + b[0].y + z[0]; }
	load(y0, 0); load(y0 + 1, 1);
	for (int y = y0; y < y1; y += 2) {
		#pragma unroll
		for (int k = 0; k < 2; ++k) {
			if (y + k >= y1) break;
			d2 va = a[k], vb = b[k]; const double zz = z[k];
			load(y + k + 2 < y1 + halo ? y + k + 2 : y1 + halo - 1, k);        // reads up to `halo` rows beyond the tile
			va.x += (zz + acc) * 1e-300;
			const size_t cell = (size_t)(y + k) * cols + x0;
			dst[(cell + lane) * 2] = va; dst[(cell + lane) * 2 + 1] = vb;
		}
	}
	if (a[0].x + a[1].x == 1.2345e-300) dst[0] = a[0];                             // keep the trailing halo loads alive
}

extern __shared__ char march_lds[];
template <int DEPTH, bool CONTIG>
__global__ __launch_bounds__(256) void k_march(const d2* __restrict__ src, const double* __restrict__ bed, d2* __restrict__ dst,
                                               int cols, int rows, int rseg, int groups, int ntiles)
{
	const unsigned per_xcd = gridDim.x >> 3;
	const unsigned tile = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int strip = (tile % groups) * 4 + wave, seg = tile / groups;
	if (tile >= (unsigned)ntiles) return;
	const int x0 = strip * 64;
	if (x0 >= cols) return;
	const int y0 = seg * rseg, y1 = min(y0 + rseg, rows);
	d2 a[DEPTH], b[DEPTH]; double z[DEPTH];
	auto load = [&](int y, int slot) {
		const size_t cell = (size_t)y * cols + x0;
		if (CONTIG) { a[slot] = src[cell * 2 + lane]; b[slot] = src[cell * 2 + 64 + lane]; }
		else        { a[slot] = src[(cell + lane) * 2]; b[slot] = src[(cell + lane) * 2 + 1]; }
		z[slot] = bed[cell + lane];
	};
	#pragma unroll
	for (int k = 0; k < DEPTH; ++k) load(min(y0 + k, rows - 1), k);
	for (int y = y0; y < y1; y += DEPTH) {
		#pragma unroll
		for (int k = 0; k < DEPTH; ++k) {
			if (y + k >= y1) break;
			d2 va = a[k], vb = b[k]; const double zz = z[k];
			load(min(y + k + DEPTH, rows - 1), k);
			va.x += zz * 1e-300;
			const size_t cell = (size_t)(y + k) * cols + x0;
			if (CONTIG) { dst[cell * 2 + lane] = va; dst[cell * 2 + 64 + lane] = vb; }
			else        { dst[(cell + lane) * 2] = va; dst[(cell + lane) * 2 + 1] = vb; }
		}
	}
}


// round 4: the march copy in K1's exact shape, one ingredient at a time.  WIN62: a wave's window starts at column 62 * strip (every
// other window is 64 B off the 128-B lines) and the windows overlap by two columns; EDGE: lanes 0 and 63 do not store (the stores
// of a row leave the first and the last 128-B line partly written, the neighbouring wave writes the rest); HALO: one row below and
// one above the tile are read as well.
template <bool WIN62, bool EDGE, bool HALO>
__global__ __launch_bounds__(256) void k_march_k1(const d2* __restrict__ src, const double* __restrict__ bed, d2* __restrict__ dst,
                                                  int cols, int rows, int rseg, int groups, int ntiles)
{
	const unsigned per_xcd = gridDim.x >> 3;
	const unsigned tile = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int strip = (tile % groups) * 4 + wave, seg = tile / groups;
	if (tile >= (unsigned)ntiles) return;
	const int x0 = strip * (WIN62 ? 62 : 64);
	if (x0 >= cols) return;
	const int x = min(x0 + lane, cols - 1);
	const bool st = (!EDGE || (lane >= 1 && lane <= 62)) && x0 + lane < cols;
	const int y0 = seg * rseg, y1 = min(y0 + rseg, rows);
	d2 a[2], b[2]; double z[2];
	auto load = [&](int y, int slot) {
		y = min(max(y, 0), rows - 1);
		const size_t cell = (size_t)y * cols + x;
		a[slot] = src[cell * 2]; b[slot] = src[cell * 2 + 1];
		z[slot] = bed[cell];
	};
	double acc = 0;
	if (HALO) { load(y0 - 1, 0); acc += a[0].x + b[0].y + z[0]; }
	load(y0, 0); load(y0 + 1, 1);
	for (int y = y0; y < y1; y += 2) {
		#pragma unroll
		for (int k = 0; k < 2; ++k) {
			if (y + k >= y1) break;
			d2 va = a[k], vb = b[k]; const double zz = z[k];
			load(y + k + 2 < y1 + (HALO ? 1 : 0) ? y + k + 2 : y1 + (HALO ? 1 : 0) - 1, k);
			va.x += (zz + acc) * 1e-300;
			const size_t cell = (size_t)(y + k) * cols + x;
			if (st) { dst[cell * 2] = va; dst[cell * 2 + 1] = vb; }
		}
	}
	if (a[0].x + a[1].x == 1.2345e-300) dst[0] = a[0];
}


// round 5 (VERDICT r04 item 5 i): the shape of a kernel that does TWO iterations per pass -- quirk Q1 makes every other timestep known
// in advance, so one pass could read a tile with a two-cell halo (60 useful columns of 64, two rows below and two above) and store
// it once per two steps.  Bytes per step are halved; what is left of that once the shape's own costs are paid (wider overlap of the
// windows, four silent lanes, four halo rows per tile, fewer waves per SIMD for the second step's registers) is what this measures.
template <int WIN, int EDGE, int HROWS>
__global__ __launch_bounds__(256) void k_march_shape(const d2* __restrict__ src, const double* __restrict__ bed, d2* __restrict__ dst,
                                                     int cols, int rows, int rseg, int groups, int ntiles)
{
	const unsigned per_xcd = gridDim.x >> 3;
	const unsigned tile = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int strip = (tile % groups) * 4 + wave, seg = tile / groups;
	if (tile >= (unsigned)ntiles) return;
	const int x0 = strip * WIN;
	if (x0 >= cols) return;
	const int x = min(x0 + lane, cols - 1);
	const bool st = lane >= EDGE && lane < 64 - EDGE && x0 + lane < cols;
	const int y0 = seg * rseg, y1 = min(y0 + rseg, rows);
	d2 a[2], b[2]; double z[2];
	auto load = [&](int y, int slot) {
		y = min(max(y, 0), rows - 1);
		const size_t cell = (size_t)y * cols + x;
		a[slot] = src[cell * 2]; b[slot] = src[cell * 2 + 1];
		z[slot] = bed[cell];
	};
	double acc = 0;
	for (int h = HROWS; h >= 1; --h) { load(y0 - h, 0); acc += a[0].x + b[0].y + z[0]; }
	load(y0, 0); load(y0 + 1, 1);
	const int y_last = y1 + HROWS - 1;                 // last row that is read
	for (int y = y0; y < y1; y += 2) {
		#pragma unroll
		for (int k = 0; k < 2; ++k) {
			if (y + k >= y1) break;
			d2 va = a[k], vb = b[k]; const double zz = z[k];
			load(y + k + 2 <= y_last ? y + k + 2 : y_last, k);
			va.x += (zz + acc) * 1e-300;
			const size_t cell = (size_t)(y + k) * cols + x;
			if (st) { dst[cell * 2] = va; dst[cell * 2 + 1] = vb; }
		}
	}
	for (int h = 2; h < HROWS; ++h) { load(y1 + h, 0); acc += a[0].x; }     // (rows y1, y1 + 1 were read by the loop's prefetch)
	if (a[0].x + a[1].x + acc == 1.2345e-300) dst[0] = a[0];
}

// round 4: does keeping the four waves of a block on the SAME row (a barrier per row: 8 KB contiguous per row and block instead of
// four drifting 2-KB streams) help the march?  WAVES: waves per block (4 or 8: 256 or 512 columns per block).
template <int WAVES, bool SYNC>
__global__ __launch_bounds__(WAVES * 64) void k_march_sync(const d2* __restrict__ src, const double* __restrict__ bed, d2* __restrict__ dst,
                                                            int cols, int rows, int rseg, int groups, int ntiles)
{
	const unsigned per_xcd = gridDim.x >> 3;
	const unsigned tile = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int strip = (tile % groups) * WAVES + wave, seg = tile / groups;
	if (tile >= (unsigned)ntiles) return;
	const int x0 = strip * 64;
	const bool live = x0 < cols;
	const int x = min(x0 + lane, cols - 1);
	const int y0 = seg * rseg, y1 = min(y0 + rseg, rows);
	d2 a[2], b[2]; double z[2];
	auto load = [&](int y, int slot) {
		y = min(max(y, 0), rows - 1);
		const size_t cell = (size_t)y * cols + x;
		a[slot] = src[cell * 2]; b[slot] = src[cell * 2 + 1];
		z[slot] = bed[cell];
	};
	load(y0, 0);
	for (int y = y0; y < y1; ++y) {
		const int k = (y - y0) & 1;
		if (SYNC) __syncthreads();
		load(y + 1 < y1 ? y + 1 : y1 - 1, k ^ 1);
		d2 va = a[k], vb = b[k]; const double zz = z[k];
		va.x += zz * 1e-300;
		const size_t cell = (size_t)y * cols + x;
		if (live) { dst[cell * 2] = va; dst[cell * 2 + 1] = vb; }
	}
}


// round 4: the march over a TILED layout -- a wave's tile (64 columns x rseg rows) contiguous in memory, row after row: what a
// device-side layout change would buy the marching kernels (the bytes and the march are the same; only the addresses differ)
__global__ __launch_bounds__(256) void k_march_tiled(const d2* __restrict__ src, const double* __restrict__ bed, d2* __restrict__ dst,
                                                     int cols, int rows, int rseg, int groups, int ntiles)
{
	const unsigned per_xcd = gridDim.x >> 3;
	const unsigned tile = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	if (tile >= (unsigned)ntiles) return;
	const size_t wtile = (size_t)tile * 4 + wave;                  // this wave's tile
	const size_t base = wtile * (size_t)rseg * 64;                // in cells
	if (base + (size_t)rseg * 64 > (size_t)cols * rows) return;
	d2 a[2], b[2]; double z[2];
	auto load = [&](int r, int slot) {
		const size_t cell = base + (size_t)min(r, rseg - 1) * 64 + lane;
		a[slot] = src[cell * 2]; b[slot] = src[cell * 2 + 1];
		z[slot] = bed[cell];
	};
	load(0, 0);
	for (int r = 0; r < rseg; ++r) {
		const int k = r & 1;
		load(r + 1, k ^ 1);
		d2 va = a[k], vb = b[k]; const double zz = z[k];
		va.x += zz * 1e-300;
		const size_t cell = base + (size_t)r * 64 + lane;
		dst[cell * 2] = va; dst[cell * 2 + 1] = vb;
	}
}


// round 4: the row-major march with the XCDs' column order SKEWED (XCD k starts k/8 of the way along the row): do the eight bands,
// whose rows lie 64 MB apart, step on each other's channels when they walk the same columns at the same time?
template <int SKEW>
__global__ __launch_bounds__(256) void k_march_skew(const d2* __restrict__ src, const double* __restrict__ bed, d2* __restrict__ dst,
                                                    int cols, int rows, int rseg, int groups, int ntiles)
{
	const unsigned per_xcd = gridDim.x >> 3;
	const unsigned xcd = blockIdx.x & 7u;
	const unsigned tile = xcd * per_xcd + (blockIdx.x >> 3);
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	if (tile >= (unsigned)ntiles) return;
	int g = tile % groups;
	if (SKEW == 1) g = (g + (int)xcd * groups / 8) % groups;
	if (SKEW == 2) g = (g + (int)xcd * 3) % groups;
	const int strip = g * 4 + wave, seg = tile / groups;
	const int x0 = strip * 64;
	if (x0 >= cols) return;
	const int y0 = seg * rseg, y1 = min(y0 + rseg, rows);
	d2 a[2], b[2]; double z[2];
	auto load = [&](int y, int slot) {
		const size_t cell = (size_t)min(y, rows - 1) * cols + x0 + lane;
		a[slot] = src[cell * 2]; b[slot] = src[cell * 2 + 1];
		z[slot] = bed[cell];
	};
	load(y0, 0);
	for (int y = y0; y < y1; ++y) {
		const int k = (y - y0) & 1;
		load(y + 1, k ^ 1);
		d2 va = a[k], vb = b[k]; const double zz = z[k];
		va.x += zz * 1e-300;
		const size_t cell = (size_t)y * cols + x0 + lane;
		dst[cell * 2] = va; dst[cell * 2 + 1] = vb;
	}
}

int main(int argc, char** argv)
{
	const int cols = 4096, rows = argc > 1 ? atoi(argv[1]) : 4096;
	const size_t cells = (size_t)cols * rows;
	d2 *src, *dst; double* bed;
	CK(hipMalloc(&src, cells * 32)); CK(hipMalloc(&dst, cells * 32)); CK(hipMalloc(&bed, cells * 8));
	CK(hipMemset(src, 0, cells * 32)); CK(hipMemset(dst, 0, cells * 32)); CK(hipMemset(bed, 0, cells * 8));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	const double bytes = (double)cells * 72;
	auto timeit = [&](const char* name, auto launch) {
		for (int i = 0; i < 5; ++i) launch();
		CK(hipEventRecord(e0));
		const int n = 50;
		for (int i = 0; i < n; ++i) launch();
		CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
		float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= n;
		printf("%-44s %.4f ms  %.2f TB/s\n", name, ms, bytes / ms / 1e9);
	};
	for (int blocks : {2048, 4096, 8192, 16384})
		timeit(("linear float4-shaped copy, blocks=" + std::to_string(blocks)).c_str(),
		       [&] { hipLaunchKernelGGL(k_linear, dim3(blocks), dim3(256), 0, 0, src, bed, dst, cells); });
	// the guide's shape: many independent 16-B loads in flight per lane.  72 B/cell with the bed, 64 B/cell without.
	for (int bed_too : {1, 0}) {
		const double b = (double)cells * (bed_too ? 72 : 64);
		auto timeit2 = [&](const char* name, auto launch) {
			for (int i = 0; i < 5; ++i) launch();
			CK(hipEventRecord(e0));
			const int n = 50;
			for (int i = 0; i < n; ++i) launch();
			CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
			float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= n;
			printf("%-60s %.4f ms  %.2f TB/s\n", name, ms, b / ms / 1e9);
		};
		for (int blocks : {1024, 2048, 4096, 8192, 65536}) {
			char nm[160];
			snprintf(nm, sizeof nm, "unrolled x4 copy (nt loads)%s, blocks=%d", bed_too ? " + bed" : "", blocks);
			timeit2(nm, [&] { hipLaunchKernelGGL((k_copy_u<4>), dim3(blocks), dim3(256), 0, 0, src, (const d2*)bed, dst, cells * 2, bed_too); });
			snprintf(nm, sizeof nm, "unrolled x8 copy (nt loads)%s, blocks=%d", bed_too ? " + bed" : "", blocks);
			timeit2(nm, [&] { hipLaunchKernelGGL((k_copy_u<8>), dim3(blocks), dim3(256), 0, 0, src, (const d2*)bed, dst, cells * 2, bed_too); });
			snprintf(nm, sizeof nm, "unrolled x4 copy (plain loads)%s, blocks=%d", bed_too ? " + bed" : "", blocks);
			timeit2(nm, [&] { hipLaunchKernelGGL((k_copy_u_plain<4>), dim3(blocks), dim3(256), 0, 0, src, (const d2*)bed, dst, cells * 2, bed_too); });
			snprintf(nm, sizeof nm, "unrolled x8 copy (plain loads)%s, blocks=%d", bed_too ? " + bed" : "", blocks);
			timeit2(nm, [&] { hipLaunchKernelGGL((k_copy_u_plain<8>), dim3(blocks), dim3(256), 0, 0, src, (const d2*)bed, dst, cells * 2, bed_too); });
		}
	}
	{
		// hipMemcpyAsync device-to-device of the state alone, for reference (64 B/cell)
		const double b = (double)cells * 64;
		for (int i = 0; i < 3; ++i) CK(hipMemcpyAsync(dst, src, cells * 32, hipMemcpyDeviceToDevice, 0));
		CK(hipEventRecord(e0));
		for (int i = 0; i < 20; ++i) CK(hipMemcpyAsync(dst, src, cells * 32, hipMemcpyDeviceToDevice, 0));
		CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
		float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
		printf("%-60s %.4f ms  %.2f TB/s\n", "hipMemcpyAsync D2D (state only)", ms, b / ms / 1e9);
	}
	for (int rep = 0; rep < 2; ++rep) for (int halo : {0, 1}) {
		const int rseg = 18;
		const int groups = cols / 64 / 4, nsegs = (rows + rseg - 1) / rseg, ntiles = groups * nsegs;
		const unsigned blocks = (ntiles + 7) / 8 * 8;
		char nm[128];
		snprintf(nm, sizeof nm, "march stride-32 pairs, depth 2, rseg=18, %d halo row(s) each side", halo);
		timeit(nm, [&] { hipLaunchKernelGGL(k_march_halo, dim3(blocks), dim3(256), 0, 0, src, bed, dst, cols, rows, rseg, groups, ntiles, halo); });
	}
	for (int rseg : {16, 32, 64}) {
		const int groups = cols / 64 / 4, nsegs = (rows + rseg - 1) / rseg, ntiles = groups * nsegs;
		const unsigned blocks = (ntiles + 7) / 8 * 8;
		char nm[128];
		snprintf(nm, sizeof nm, "march stride-32 pairs, depth 2, rseg=%d", rseg);
		timeit(nm, [&] { hipLaunchKernelGGL((k_march<2, false>), dim3(blocks), dim3(256), 0, 0, src, bed, dst, cols, rows, rseg, groups, ntiles); });
		snprintf(nm, sizeof nm, "march lane-contiguous, depth 2, rseg=%d", rseg);
		timeit(nm, [&] { hipLaunchKernelGGL((k_march<2, true>), dim3(blocks), dim3(256), 0, 0, src, bed, dst, cols, rows, rseg, groups, ntiles); });
		snprintf(nm, sizeof nm, "march stride-32 pairs, depth 4, rseg=%d", rseg);
		timeit(nm, [&] { hipLaunchKernelGGL((k_march<4, false>), dim3(blocks), dim3(256), 0, 0, src, bed, dst, cols, rows, rseg, groups, ntiles); });
		snprintf(nm, sizeof nm, "march lane-contiguous, depth 4, rseg=%d", rseg);
		timeit(nm, [&] { hipLaunchKernelGGL((k_march<4, true>), dim3(blocks), dim3(256), 0, 0, src, bed, dst, cols, rows, rseg, groups, ntiles); });
	}
	// round 4: the march shape at K1's occupancy.  K1 runs 3 waves per SIMD (147 VGPRs) with ONE row in flight per wave while it
	// computes; the kernels above run 8 waves per SIMD.  Dynamic LDS caps the blocks per CU (a block = 4 waves = one per SIMD):
	// 52 KiB -> 3 blocks, 36 KiB -> 4, 28 KiB -> 5.  How much of the march copy's rate needs more bytes in flight than that?
	{
		CK(hipFuncSetAttribute((const void*)k_march<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
		CK(hipFuncSetAttribute((const void*)k_march<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
		CK(hipFuncSetAttribute((const void*)k_march<3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
		CK(hipFuncSetAttribute((const void*)k_march<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
		for (int lds_kib : {52, 36, 28, 0}) for (int rseg : {16, 18}) {
			const int groups = cols / 64 / 4, nsegs = (rows + rseg - 1) / rseg, ntiles = groups * nsegs;
			const unsigned blocks = (ntiles + 7) / 8 * 8;
			const int waves = lds_kib == 52 ? 3 : lds_kib == 36 ? 4 : lds_kib == 28 ? 5 : 8;
			char nm[160];
			snprintf(nm, sizeof nm, "march, %d waves/SIMD, rows in flight 1, rseg=%d", waves, rseg);
			timeit(nm, [&] { hipLaunchKernelGGL((k_march<1, false>), dim3(blocks), dim3(256), lds_kib * 1024, 0, src, bed, dst, cols, rows, rseg, groups, ntiles); });
			snprintf(nm, sizeof nm, "march, %d waves/SIMD, rows in flight 2, rseg=%d", waves, rseg);
			timeit(nm, [&] { hipLaunchKernelGGL((k_march<2, false>), dim3(blocks), dim3(256), lds_kib * 1024, 0, src, bed, dst, cols, rows, rseg, groups, ntiles); });
			snprintf(nm, sizeof nm, "march, %d waves/SIMD, rows in flight 3, rseg=%d", waves, rseg);
			timeit(nm, [&] { hipLaunchKernelGGL((k_march<3, false>), dim3(blocks), dim3(256), lds_kib * 1024, 0, src, bed, dst, cols, rows, rseg, groups, ntiles); });
			snprintf(nm, sizeof nm, "march, %d waves/SIMD, rows in flight 4, rseg=%d", waves, rseg);
			timeit(nm, [&] { hipLaunchKernelGGL((k_march<4, false>), dim3(blocks), dim3(256), lds_kib * 1024, 0, src, bed, dst, cols, rows, rseg, groups, ntiles); });
		}
	}
	{
		// K1's shape, ingredient by ingredient (3 waves per SIMD through 52 KiB of dynamic LDS, 18-row tiles)
		const int rseg = 18, lds = 52 * 1024;
		auto run = [&](const char* what, auto kern, int width) {
			CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
			const int strips = (cols + width - 1) / width, groups = (strips + 3) / 4, nsegs = (rows + rseg - 1) / rseg, ntiles = groups * nsegs;
			const unsigned blocks = (ntiles + 7) / 8 * 8;
			timeit(what, [&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, src, bed, dst, cols, rows, rseg, groups, ntiles); });
		};
		auto run_sync = [&](const char* what, auto kern, int waves, int lds_bytes) {
			CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
			const int strips = cols / 64, groups = (strips + waves - 1) / waves, nsegs = (rows + 16 - 1) / 16, ntiles = groups * nsegs;
			const unsigned blocks = (ntiles + 7) / 8 * 8;
			timeit(what, [&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(waves * 64), lds_bytes, 0, src, bed, dst, cols, rows, 16, groups, ntiles); });
		};
		for (int rep = 0; rep < 3; ++rep) {
			const int rseg = 16, groups = cols / 64 / 4, nsegs = rows / rseg, ntiles = groups * nsegs;
			const unsigned blocks = (ntiles + 7) / 8 * 8;
			CK(hipFuncSetAttribute((const void*)k_march_skew<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
			CK(hipFuncSetAttribute((const void*)k_march_skew<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
			CK(hipFuncSetAttribute((const void*)k_march_skew<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
			timeit("march 16 rows, XCD column order: same", [&] { hipLaunchKernelGGL(k_march_skew<0>, dim3(blocks), dim3(256), 52 * 1024, 0, src, bed, dst, cols, rows, rseg, groups, ntiles); });
			timeit("march 16 rows, XCD column order: skewed by 1/8 row", [&] { hipLaunchKernelGGL(k_march_skew<1>, dim3(blocks), dim3(256), 52 * 1024, 0, src, bed, dst, cols, rows, rseg, groups, ntiles); });
			timeit("march 16 rows, XCD column order: skewed by 3 groups", [&] { hipLaunchKernelGGL(k_march_skew<2>, dim3(blocks), dim3(256), 52 * 1024, 0, src, bed, dst, cols, rows, rseg, groups, ntiles); });
		}
		for (int rep = 0; rep < 2; ++rep) for (int rseg : {16, 18, 32}) {
			CK(hipFuncSetAttribute((const void*)k_march_tiled, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
			const int groups = cols / 64 / 4, nsegs = rows / rseg, ntiles = groups * nsegs;
			const unsigned blocks = (ntiles + 7) / 8 * 8;
			char nm[128];
			snprintf(nm, sizeof nm, "march over a TILED layout, 3 waves/SIMD, rseg=%d", rseg);
			timeit(nm, [&] { hipLaunchKernelGGL(k_march_tiled, dim3(blocks), dim3(256), 52 * 1024, 0, src, bed, dst, cols, rows, rseg, groups, ntiles); });
		}
		for (int rep = 0; rep < 2; ++rep) {
			run_sync("march 16 rows, 3 waves/SIMD, 4-wave blocks, free", k_march_sync<4, false>, 4, 52 * 1024);
			run_sync("march 16 rows, 3 waves/SIMD, 4-wave blocks, barrier per row", k_march_sync<4, true>, 4, 52 * 1024);
			run_sync("march 16 rows, 8-wave blocks (2 per CU = 4 waves/SIMD), free", k_march_sync<8, false>, 8, 64 * 1024);
			run_sync("march 16 rows, 8-wave blocks, barrier per row", k_march_sync<8, true>, 8, 64 * 1024);
		}
		// round 5: the two-iterations-per-pass shape against K1's own (one pass moves two steps' worth: ms per STEP = ms / 2)
		auto run_shape = [&](const char* what, auto kern, int width, int rseg2, int lds_bytes, double steps) {
			CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 / 2));
			const int strips = (cols + width - 1) / width, groups = (strips + 3) / 4, nsegs = (rows + rseg2 - 1) / rseg2, ntiles = groups * nsegs;
			const unsigned blocks = (ntiles + 7) / 8 * 8;
			float ms = 0; hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
			for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds_bytes, 0, src, bed, dst, cols, rows, rseg2, groups, ntiles);
			CK(hipEventRecord(e0)); const int reps = 20;
			for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds_bytes, 0, src, bed, dst, cols, rows, rseg2, groups, ntiles);
			CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
			const hipError_t le = hipGetLastError();
			if (le != hipSuccess) { printf("%-78s launch failed: %s\n", what, hipGetErrorString(le)); return; }
			printf("%-78s %.4f ms per pass  %.4f ms per step\n", what, ms, ms / steps);
		};
		for (int rep = 0; rep < 2; ++rep) {
			run_shape("one step : 62-col, 1 edge lane, 1 halo row, rseg 18, 3 waves/SIMD (K1 today)", k_march_shape<62, 1, 1>, 62, 18, 52 * 1024, 1);
			run_shape("two steps: 60-col, 2 edge lanes, 2 halo rows, rseg 18, 3 waves/SIMD", k_march_shape<60, 2, 2>, 60, 18, 52 * 1024, 2);
			run_shape("two steps: 60-col, 2 edge lanes, 2 halo rows, rseg 32, 3 waves/SIMD", k_march_shape<60, 2, 2>, 60, 32, 52 * 1024, 2);
			run_shape("two steps: 60-col, 2 edge lanes, 2 halo rows, rseg 18, 2 waves/SIMD", k_march_shape<60, 2, 2>, 60, 18, 80 * 1024, 2);
			run_shape("two steps: 60-col, 2 edge lanes, 2 halo rows, rseg 32, 2 waves/SIMD", k_march_shape<60, 2, 2>, 60, 32, 80 * 1024, 2);
			run_shape("two steps: 60-col, 2 edge lanes, 2 halo rows, rseg 64, 2 waves/SIMD", k_march_shape<60, 2, 2>, 60, 64, 80 * 1024, 2);
			run_shape("64 useful columns: 64-col aligned windows, 1 halo row, rseg 18, 3 waves/SIMD", k_march_shape<64, 0, 1>, 64, 18, 52 * 1024, 1);
		}
		for (int rep = 0; rep < 2; ++rep) {
			run("K1 shape: 64-col windows", k_march_k1<false, false, false>, 64);
			run("K1 shape: 64-col windows + halo rows", k_march_k1<false, false, true>, 64);
			run("K1 shape: 62-col windows (all lanes store)", k_march_k1<true, false, false>, 62);
			run("K1 shape: 62-col windows, edge lanes silent", k_march_k1<true, true, false>, 62);
			run("K1 shape: 62-col, edge silent, halo rows", k_march_k1<true, true, true>, 62);
		}
	}
	return 0;
}
