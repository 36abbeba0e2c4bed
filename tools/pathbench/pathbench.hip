// pathbench -- what the pieces of a row step COST as the engine's workloads execute them (not product code).
//
// Every wavefront takes 64 consecutive cells of a real state snapshot (tools/pathbench/snap.py: S-RAIN thin films, S-ROUGH
// wet/dry terrain, the S-DAM front) and runs ONE piece of the row step -- a face solve, the cell update with friction,
// the MUSCL predictor ... -- `iters` times on it, with exactly three such waves per SIMD (what K1/K2 run with) on every CU.
// An empty asm on the inputs makes every trip recompute everything.  Reported: nanoseconds per call per wave and the
// implied SIMD issue cycles per call (t * clock / 3), i.e. what the piece adds to a row step's issue time.
// The arithmetic comes from the engine's own header; build twice with -DHP_MATH="\"variant.hpp\"" to compare variants,
// `--dump file` writes every piece's outputs for a bit-level / tolerance comparison between them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#ifndef PB_STRICT
#define PB_STRICT false          // -DPB_STRICT=true: the STRICT flavour of every piece (round 4's price list of the exact mode)
#endif
#ifndef HP_KERNELS
#define HP_KERNELS "../../hipims-ocl_amd/csrc/hp_kernels.hpp"
#endif
#include HP_KERNELS
using namespace hp;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Snap { int cols, rows; const double* state; const double* bed; double dt; };

__device__ __forceinline__ void opaque(double& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void opaque(Side<double>& s) { opaque(s.eta); opaque(s.zb); opaque(s.qx); opaque(s.qy); opaque(s.u0); opaque(s.v0); }

__device__ __forceinline__ void cell(const Snap& s, long x, long y, State4<double>& c, double& zb)
{
	x = x < 0 ? 0 : (x >= s.cols ? s.cols - 1 : x);
	y = y < 0 ? 0 : (y >= s.rows ? s.rows - 1 : y);
	const size_t id = (size_t)y * s.cols + x;
	c.z = s.state[4 * id]; c.zmax = s.state[4 * id + 1]; c.qx = s.state[4 * id + 2]; c.qy = s.state[4 * id + 3];
	zb = s.bed[id];
}

// wave -> (row, first column): waves walk the snapshot's interior in 62-column strips like K1
__device__ __forceinline__ void place(const Snap& s, long& x, long& y)
{
	const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
	const long strips = (s.cols - 2) / 62;
	y = 1 + (wave / strips) % (s.rows - 2);
	x = (wave % strips) * 62 + lane;
}

enum { P_FACE_X = 0, P_FACE_Y, P_UPDATE, P_MAKE_SIDE, P_CFL, P_PREDICT, P_SIDE_FROM_FACE, P_MUSCL_FACE_X, P_MUSCL_FACE_Y, P_EMPTY, P_COUNT };
static const char* NAMES[P_COUNT] = {"face_solve x (K1 sides)", "face_solve y (K1 sides)", "godunov_update + friction", "make_side", "cfl_speed",
                                     "muscl_predict", "4 x side_from_face", "face_solve x (MUSCL face states)", "face_solve y (MUSCL face states)", "empty loop"};

template <int PIECE>
__global__ __launch_bounds__(256) void k_piece(const Snap s, const int iters, double* __restrict__ out, const int dump)
{
	extern __shared__ char lds_pad[];                       // sized by the host so that exactly 3 blocks fit a CU
	long x, y;
	place(s, x, y);
	const double vs = 1e-10, dx = 2.0, inv_dx = 0.5, n = 0.03;
	State4<double> c, cn, ce, cs_, cw; double zb, zbn, zbe, zbs, zbw;
	cell(s, x, y, c, zb); cell(s, x, y + 1, cn, zbn); cell(s, x + 1, y, ce, zbe); cell(s, x, y - 1, cs_, zbs); cell(s, x - 1, y, cw, zbw);
	double dt = s.dt;
	double acc = 0;
	Side<double> sC = make_side<PB_STRICT>(c.z, c.qx, c.qy, zb, vs), sN = make_side<PB_STRICT>(cn.z, cn.qx, cn.qy, zbn, vs);
	Side<double> sE = make_side<PB_STRICT>(ce.z, ce.qx, ce.qy, zbe, vs), sS = make_side<PB_STRICT>(cs_.z, cs_.qx, cs_.qy, zbs, vs);
	Side<double> sW = make_side<PB_STRICT>(cw.z, cw.qx, cw.qy, zbw, vs);
	FaceFlux<double> fN = face_solve<AXIS_Y, PB_STRICT, true, true>(sC, sN, vs).forL, fS = face_solve<AXIS_Y, PB_STRICT, true, true>(sS, sC, vs).forR;
	FaceFlux<double> fE = face_solve<AXIS_X, PB_STRICT, true, true>(sC, sE, vs).forL, fW = face_solve<AXIS_X, PB_STRICT, true, true>(sW, sC, vs).forR;
	// MUSCL face states of this cell and its east / north neighbour (predictor output)
	Raw<double> rc{c.z, c.zmax, c.qx, c.qy, zb}, rn{cn.z, cn.zmax, cn.qx, cn.qy, zbn}, re{ce.z, ce.zmax, ce.qx, ce.qy, zbe},
	            rs{cs_.z, cs_.zmax, cs_.qx, cs_.qy, zbs}, rw{cw.z, cw.zmax, cw.qx, cw.qy, zbw};
	bool q0, q1;
	Faces<double> pc = muscl_predict<PB_STRICT>(rc, rn, re, rs, rw, dt, dx, inv_dx, vs, true, q0, q1);
	Side<double> mE = side_from_face<PB_STRICT>(pc.e, c.qx, c.qy, vs), mN = side_from_face<PB_STRICT>(pc.n, c.qx, c.qy, vs);
	Side<double> mEnb, mNnb;
	{
		// neighbours' opposite faces: the east neighbour's W face, the north neighbour's S face (their own predictors)
		State4<double> t; double tz;
		Raw<double> r2[5];
		for (int k = 0; k < 5; ++k) {
			const long ox = (k == 2) - (k == 4), oy = (k == 1) - (k == 3);
			cell(s, x + 1 + ox, y + oy, t, tz); r2[k] = Raw<double>{t.z, t.zmax, t.qx, t.qy, tz};
		}
		bool a, b;
		Faces<double> pe = muscl_predict<PB_STRICT>(r2[0], r2[1], r2[2], r2[3], r2[4], dt, dx, inv_dx, vs, true, a, b);
		mEnb = side_from_face<PB_STRICT>(pe.w, ce.qx, ce.qy, vs);
		for (int k = 0; k < 5; ++k) {
			const long ox = (k == 2) - (k == 4), oy = (k == 1) - (k == 3);
			cell(s, x + ox, y + 1 + oy, t, tz); r2[k] = Raw<double>{t.z, t.zmax, t.qx, t.qy, tz};
		}
		Faces<double> pn = muscl_predict<PB_STRICT>(r2[0], r2[1], r2[2], r2[3], r2[4], dt, dx, inv_dx, vs, true, a, b);
		mNnb = side_from_face<PB_STRICT>(pn.s, cn.qx, cn.qy, vs);
	}

	for (int it = 0; it < iters; ++it) {
		if (PIECE == P_FACE_X || PIECE == P_FACE_Y || PIECE == P_MUSCL_FACE_X || PIECE == P_MUSCL_FACE_Y) {
			Side<double> L = PIECE == P_FACE_X ? sC : PIECE == P_FACE_Y ? sC : PIECE == P_MUSCL_FACE_X ? mE : mN;
			Side<double> R = PIECE == P_FACE_X ? sE : PIECE == P_FACE_Y ? sN : PIECE == P_MUSCL_FACE_X ? mEnb : mNnb;
			opaque(L); opaque(R);
			const FacePair<double> f = (PIECE == P_FACE_X || PIECE == P_MUSCL_FACE_X) ? face_solve<AXIS_X, PB_STRICT, true, true>(L, R, vs)
			                                                                             : face_solve<AXIS_Y, PB_STRICT, true, true>(L, R, vs);
			acc += f.forL.f0 + f.forL.fx + f.forL.fy + f.forL.eta_nb + f.forL.zb_nb + (f.forL.stop ? 1.0 : 0.0)
			     + f.forR.f0 + f.forR.fx + f.forR.fy + f.forR.eta_nb + f.forR.zb_nb + (f.forR.stop ? 2.0 : 0.0);
		} else if (PIECE == P_UPDATE) {
			State4<double> cc = c; opaque(cc.z); opaque(cc.qx); opaque(cc.qy); opaque(cc.zmax);
			FaceFlux<double> a = fN, b = fE, d = fS, e = fW;
			opaque(a.f0); opaque(b.f0); opaque(d.f0); opaque(e.f0); opaque(a.fx); opaque(b.fy); opaque(d.fx); opaque(e.fy);
			opaque(a.fy); opaque(b.fx); opaque(d.fy); opaque(e.fx); opaque(a.eta_nb); opaque(b.eta_nb); opaque(d.zb_nb); opaque(e.zb_nb);
			const State4<double> u = godunov_update<PB_STRICT>(cc, zb, n, dt, a, b, d, e, dx, inv_dx, vs, true);
			acc += u.z + u.zmax + u.qx + u.qy;
		} else if (PIECE == P_MAKE_SIDE) {
			State4<double> cc = c; opaque(cc.z); opaque(cc.qx); opaque(cc.qy);
			const Side<double> t = make_side<PB_STRICT>(cc.z, cc.qx, cc.qy, zb, vs);
			acc += t.u0 + t.v0;
		} else if (PIECE == P_CFL) {
			State4<double> cc = c; opaque(cc.z); opaque(cc.qx); opaque(cc.qy);
			acc += cfl_speed<PB_STRICT>(cc.z, cc.zmax, cc.qx, cc.qy, zb, 1e-9);
		} else if (PIECE == P_PREDICT) {
			Raw<double> a = rc; opaque(a.z); opaque(a.qx); opaque(a.qy);
			Raw<double> b = rn; opaque(b.z); opaque(b.qx);
			bool qa, qb;
			const Faces<double> p = muscl_predict<PB_STRICT>(a, b, re, rs, rw, dt, dx, inv_dx, vs, true, qa, qb);
			acc += p.n.z + p.n.h + p.n.qx + p.n.qy + p.e.z + p.e.h + p.e.qx + p.e.qy + p.s.z + p.s.h + p.s.qx + p.s.qy + p.w.z + p.w.h + p.w.qx + p.w.qy;
		} else if (PIECE == P_SIDE_FROM_FACE) {
			Faces<double> p = pc; opaque(p.n.z); opaque(p.n.h); opaque(p.e.z); opaque(p.e.h); opaque(p.s.z); opaque(p.s.h); opaque(p.w.z); opaque(p.w.h);
			opaque(p.n.qx); opaque(p.e.qx); opaque(p.s.qx); opaque(p.w.qx);
			const Side<double> a = side_from_face<PB_STRICT>(p.n, c.qx, c.qy, vs), b = side_from_face<PB_STRICT>(p.e, c.qx, c.qy, vs),
			                   d = side_from_face<PB_STRICT>(p.s, c.qx, c.qy, vs), e = side_from_face<PB_STRICT>(p.w, c.qx, c.qy, vs);
			acc += a.u0 + a.v0 + a.zb + b.u0 + b.v0 + b.zb + d.u0 + d.v0 + d.zb + e.u0 + e.v0 + e.zb;
		} else {
			opaque(acc);
		}
	}
	const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (dump || acc == 1.2345e-300) out[gid] = acc;
}

template <int PIECE> void run(const Snap& s, int iters, double* out, int blocks, size_t lds, double clock_ghz, int dump, FILE* df)
{
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	hipLaunchKernelGGL(k_piece<PIECE>, dim3(blocks), dim3(256), lds, 0, s, iters / 10 + 1, out, 0);
	CK(hipEventRecord(e0));
	hipLaunchKernelGGL(k_piece<PIECE>, dim3(blocks), dim3(256), lds, 0, s, iters, out, dump);
	CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
	float ms; CK(hipEventElapsedTime(&ms, e0, e1));
	const double ns = ms * 1e6 / iters;
	printf("  %-36s %8.1f ns/call/wave   %7.1f SIMD cycles/call\n", NAMES[PIECE], ns, ns * clock_ghz / 3.0);
	if (dump && df) {
		std::vector<double> h((size_t)blocks * 256);
		CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
		fwrite(h.data(), 8, h.size(), df);
	}
}

int main(int argc, char** argv)
{
	int iters = 2000; const char* dump = nullptr;
	std::vector<std::string> snaps;
	for (int i = 1; i < argc; ++i) {
		if (!strcmp(argv[i], "--iters")) iters = atoi(argv[++i]);
		else if (!strcmp(argv[i], "--dump")) dump = argv[++i];
		else snaps.push_back(argv[i]);
	}
	hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount;
	const double clock_ghz = prop.clockRate / 1e6;
	const int blocks = cus * 3;
	const size_t lds = 50 * 1024;                           // 3 x 50 KiB of 160: never a fourth block on a CU
	CK(hipFuncSetAttribute((const void*)k_piece<P_EMPTY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	double* out; CK(hipMalloc(&out, (size_t)blocks * 256 * 8));
	FILE* df = dump ? fopen(dump, "wb") : nullptr;
	printf("pathbench: %d CUs, %.2f GHz, %d blocks x 4 waves (3 waves per SIMD), %d calls per wave\n", cus, clock_ghz, blocks, iters);
	for (const std::string& path : snaps) {
		FILE* f = fopen(path.c_str(), "rb");
		if (!f) { printf("cannot open %s\n", path.c_str()); continue; }
		int dims[2]; if (fread(dims, 4, 2, f) != 2) return 1;
		const size_t cells = (size_t)dims[0] * dims[1];
		std::vector<double> st(cells * 4), bed(cells); double dt;
		if (fread(st.data(), 8, cells * 4, f) != cells * 4 || fread(bed.data(), 8, cells, f) != cells || fread(&dt, 8, 1, f) != 1) return 1;
		fclose(f);
		double *dst, *dbed; CK(hipMalloc(&dst, cells * 32)); CK(hipMalloc(&dbed, cells * 8));
		CK(hipMemcpy(dst, st.data(), cells * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(dbed, bed.data(), cells * 8, hipMemcpyHostToDevice));
		Snap s{dims[0], dims[1], dst, dbed, dt > 0 ? dt : 0.05};
		printf("%s (%d x %d, dt %.4g)\n", path.c_str(), dims[0], dims[1], s.dt);
#define RUN(P) CK(hipFuncSetAttribute((const void*)k_piece<P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); run<P>(s, iters, out, blocks, lds, clock_ghz, dump != nullptr, df);
		RUN(P_EMPTY) RUN(P_FACE_X) RUN(P_FACE_Y) RUN(P_UPDATE) RUN(P_MAKE_SIDE) RUN(P_CFL) RUN(P_PREDICT) RUN(P_SIDE_FROM_FACE) RUN(P_MUSCL_FACE_X) RUN(P_MUSCL_FACE_Y)
		CK(hipFree(dst)); CK(hipFree(dbed));
	}
	if (df) fclose(df);
	return 0;
}
