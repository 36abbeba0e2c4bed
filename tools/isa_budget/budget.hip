// Per-phase VALU instruction budget of the FAST fp64 kernels: each phase of a MUSCL row step as a kernel of its own,
// inputs from memory, outputs to memory; tools/isa_budget/count.py counts the v_* instructions in the ISA.
#include "../../hipims-ocl_amd/csrc/hp_math.hpp"
using namespace hp;
typedef double T;
#define LD(i) in[(i) * 64 + threadIdx.x]
__global__ void phase_predict(const T* in, T* out)
{
	Raw<T> c{LD(0), LD(1), LD(2), LD(3), LD(4)}, n{LD(5), LD(6), LD(7), LD(8), LD(9)}, e{LD(10), LD(11), LD(12), LD(13), LD(14)},
	       s{LD(15), LD(16), LD(17), LD(18), LD(19)}, w{LD(20), LD(21), LD(22), LD(23), LD(24)};
	bool quiet, same; const Faces<T> f = muscl_predict<false>(c, n, e, s, w, LD(25), LD(26), LD(27), LD(28), true, quiet, same);
	T* o = out + threadIdx.x;
	o[0] = f.n.z; o[64] = f.n.h; o[128] = f.n.qx; o[192] = f.n.qy; o[256] = f.e.z; o[320] = f.e.h; o[384] = f.e.qx; o[448] = f.e.qy;
	o[512] = f.s.z; o[576] = f.s.h; o[640] = f.s.qx; o[704] = f.s.qy; o[768] = f.w.z; o[832] = f.w.h; o[896] = f.w.qx; o[960] = f.w.qy;
}
__global__ void phase_sides4(const T* in, T* out)
{
	T acc = 0;
	for (int k = 0; k < 4; ++k) {
		Face4<T> f{LD(4 * k), LD(4 * k + 1), LD(4 * k + 2), LD(4 * k + 3)};
		const Side<T> s = side_from_face<false>(f, LD(20), LD(21), LD(22));
		acc += s.eta + s.zb + s.u0 + s.v0 + s.qx + s.qy;
	}
	out[threadIdx.x] = acc;
}
template <int AXIS> __device__ void face(const T* in, T* out)
{
	Side<T> L{LD(0), LD(1), LD(2), LD(3), LD(4), LD(5)}, R{LD(6), LD(7), LD(8), LD(9), LD(10), LD(11)};
	const FacePair<T> p = face_solve<AXIS, false, true, true>(L, R, LD(12));
	T* o = out + threadIdx.x;
	o[0] = p.forL.f0; o[64] = p.forL.fx; o[128] = p.forL.fy; o[192] = p.forL.eta_nb; o[256] = p.forL.zb_nb; o[320] = p.forL.stop;
	o[384] = p.forR.f0; o[448] = p.forR.fx; o[512] = p.forR.fy; o[576] = p.forR.eta_nb; o[640] = p.forR.zb_nb; o[704] = p.forR.stop;
}
__global__ void phase_face_x(const T* in, T* out) { face<AXIS_X>(in, out); }
__global__ void phase_face_y(const T* in, T* out) { face<AXIS_Y>(in, out); }
__global__ void phase_update(const T* in, T* out)
{
	State4<T> c{LD(0), LD(1), LD(2), LD(3)};
	FaceFlux<T> f[4];
	for (int k = 0; k < 4; ++k) f[k] = FaceFlux<T>{LD(6 + 6 * k), LD(7 + 6 * k), LD(8 + 6 * k), LD(9 + 6 * k), LD(10 + 6 * k), LD(11 + 6 * k) > 0};
	const State4<T> u = godunov_update<false, true>(c, LD(4), LD(5), LD(30), f[0], f[1], f[2], f[3], LD(31), LD(32), LD(33), true);
	T* o = out + threadIdx.x;
	o[0] = u.z; o[64] = u.zmax; o[128] = u.qx; o[192] = u.qy;
}
__global__ void phase_cfl(const T* in, T* out) { out[threadIdx.x] = cfl_speed<false>(LD(0), LD(1), LD(2), LD(3), LD(4), LD(5)); }
__global__ void phase_empty(const T* in, T* out) { out[threadIdx.x] = LD(0); }
