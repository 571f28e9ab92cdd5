// pk_rate.hip -- issue rate of the packed fp32 VALU forms k_pyramid_rows is built from (gfx950), timing experiment.
//   hipcc --offload-arch=gfx950 -O2 -o scripts/ubench/pk_rate scripts/ubench/pk_rate.hip && scripts/ubench/pk_rate
// Every kernel runs REPS x 64 instructions of one form in 8 independent (or deliberately dependent) chains per wavefront
// and reports cycles per instruction per SIMD at 1, 2 and 4 wavefronts per SIMD (clock64 around the loop, wave 0).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define REPS 2000

#define BODY8(INS) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)
template <int FORM>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* cyc, float s0, float s1) {
	f32x2 a[8], b[8];
	for (int i = 0; i < 8; ++i) { a[i] = f32x2{1.f + threadIdx.x * 1e-3f + i, 2.f + i}; b[i] = f32x2{0.5f + i, 0.25f}; }
	const f32x2 sp = f32x2{s0, s1};      // uniform: lands in an SGPR pair
	float x[8]; for (int i = 0; i < 8; ++i) x[i] = 1.f + i + threadIdx.x * 1e-3f;
	__syncthreads();
	const unsigned long long t0 = clock64();
	for (int r = 0; r < REPS; ++r) {
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			if (FORM == 0) {          // v_pk_mul_f32 vgpr, vgpr: 8 independent chains
#define I0(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
				BODY8(I0)
			} else if (FORM == 1) {   // v_pk_mul_f32 vgpr, sgpr pair
#define I1(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "s"(sp));
				BODY8(I1)
			} else if (FORM == 2) {   // the column-pass form: low half broadcast (op_sel_hi:[0,1]) x sgpr pair
#define I2(i) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[0,1]" : "+v"(a[i]) : "s"(sp));
				BODY8(I2)
			} else if (FORM == 3) {   // mul into a temporary, dependent add right behind it (4 chains x 2 instructions)
#define I3(i) asm volatile("v_pk_mul_f32 %1, %2, %3 op_sel_hi:[0,1]\n\tv_pk_add_f32 %0, %0, %1" : "+v"(a[i]), "=&v"(b[i + 4]) : "v"(b[i]), "s"(sp));
				I3(0) I3(1) I3(2) I3(3)
			} else if (FORM == 4) {   // plain v_mul_f32: 8 chains
#define I4(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b[i].x));
				BODY8(I4)
			} else if (FORM == 5) {   // v_pk_fma_f32
#define I5(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
				BODY8(I5)
			} else if (FORM == 6) {   // v_pk_add_f32 vgpr, vgpr
#define I6(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
				BODY8(I6)
			} else if (FORM == 7) {   // mul + dependent add, mul's result consumed two instructions later (software-interleaved pairs)
				asm volatile("v_pk_mul_f32 %4, %8, %12 op_sel_hi:[0,1]\n\tv_pk_mul_f32 %5, %9, %12 op_sel_hi:[0,1]\n\tv_pk_mul_f32 %6, %10, %12 op_sel_hi:[0,1]\n\tv_pk_mul_f32 %7, %11, %12 op_sel_hi:[0,1]\n\t"
						"v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %5\n\tv_pk_add_f32 %2, %2, %6\n\tv_pk_add_f32 %3, %3, %7"
						: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "=&v"(b[4]), "=&v"(b[5]), "=&v"(b[6]), "=&v"(b[7])
						: "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "s"(sp));
			}
		}
	}
	const unsigned long long t1 = clock64();
	float s = 0; for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y + x[i] + b[i].x;
	out[blockIdx.x * 256 + threadIdx.x] = s;
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int FORM> void run(const char* name, int per_body) {
	float* out; unsigned long long* cyc;
	hipMalloc(&out, sizeof(float) * 256 * 2048); hipMalloc(&cyc, sizeof(unsigned long long) * 2048);
	for (int wg_per_cu = 1; wg_per_cu <= 4; wg_per_cu *= 2) {       // 256 threads = 4 wavefronts = 1 per SIMD
		const int blocks = 256 * wg_per_cu;
		hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0001f, 0.9999f);
		hipDeviceSynchronize();
		hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
		hipEventRecord(e0); hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0001f, 0.9999f); hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		std::vector<unsigned long long> h(blocks); hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
		double avg = 0; for (auto v : h) avg += (double)v; avg /= blocks;
		const double ninst = (double)REPS * 8 * per_body;
		// clock64 ticks at a fixed 100 MHz reference on this part: the event time is what counts
		printf("%-58s %d wave/SIMD  %.3f ms  %.2f ns per instruction per SIMD (x2.1 GHz = %.2f cycles)  clock64/inst %.3f\n", name, wg_per_cu, ms,
				ms * 1e6 / (ninst * wg_per_cu), ms * 1e6 / (ninst * wg_per_cu) * 2.1, avg / ninst);
	}
	hipFree(out); hipFree(cyc);
}
int main() {
	run<4>("v_mul_f32 v,v (8 chains)", 8);
	run<0>("v_pk_mul_f32 v,v (8 chains)", 8);
	run<6>("v_pk_add_f32 v,v (8 chains)", 8);
	run<5>("v_pk_fma_f32 v,v,v (8 chains)", 8);
	run<1>("v_pk_mul_f32 v,s (8 chains)", 8);
	run<2>("v_pk_mul_f32 v,s op_sel_hi:[0,1] (8 chains)", 8);
	run<3>("pk_mul(op_sel, sgpr) + dependent pk_add back to back (4 pairs)", 8);
	run<7>("4 pk_mul(op_sel, sgpr) then their 4 pk_add (interleaved)", 8);
	return 0;
}
