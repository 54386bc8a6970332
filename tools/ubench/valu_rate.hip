// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction per SIMD at 1..8 waves/SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int KIND>
__global__ void k(float* out, int iters, float seed)
{
	float a0 = seed + threadIdx.x, a1 = a0 * 1.1f, a2 = a0 * 1.2f, a3 = a0 * 1.3f, a4 = a0 * 1.4f, a5 = a0 * 1.5f, a6 = a0 * 1.6f, a7 = a0 * 1.7f;
	const float c = seed * 0.5f;
	for (int i = 0; i < iters; i++) {
#pragma unroll
		for (int u = 0; u < 8; u++) {
			if (KIND == 0) { a0 += c; a1 += c; a2 += c; a3 += c; a4 += c; a5 += c; a6 += c; a7 += c; }
			if (KIND == 1) { a0 = fmaxf(a0, c); a1 = fminf(a1, c); a2 = fmaxf(a2, c); a3 = fminf(a3, c); a4 = fmaxf(a4, c); a5 = fminf(a5, c); a6 = fmaxf(a6, c); a7 = fminf(a7, c); }
			if (KIND == 2) { a0 = (float)(int)a0; a1 = (float)(int)a1; a2 = (float)(int)a2; a3 = (float)(int)a3; a4 = (float)(int)a4; a5 = (float)(int)a5; a6 = (float)(int)a6; a7 = (float)(int)a7; }
			if (KIND == 3) { a0 = fminf(fabsf(a0 - a1), c); a2 = fminf(fabsf(a2 - a3), c); a4 = fminf(fabsf(a4 - a5), c); a6 = fminf(fabsf(a6 - a7), c); a1 += c; a3 += c; a5 += c; a7 += c; }
			if (KIND == 4) { // packed adds: two floats per instruction
				typedef float f2 __attribute__((ext_vector_type(2)));
				f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, cc = {c, c};
				p0 += cc; p1 += cc; p2 += cc; p3 += cc;
				a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
			}
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int KIND>
void run(const char* name, int ops_per_iter)
{
	float* d; CHK(hipMalloc(&d, 256 * 8 * 1024 * 4));
	hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	const int iters = 20000;
	for (int wps = 1; wps <= 8; wps *= 2) {
		const int threads = 256 * wps > 1024 ? 1024 : 256 * wps;   // waves per SIMD = wps (one block per CU, 4*wps waves)
		const int blocks = 256 * (256 * wps / threads);
		hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, 10, 1.0f);
		CHK(hipDeviceSynchronize());
		CHK(hipEventRecord(e0));
		hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f);
		CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
		float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
		const double insts_per_simd = (double)iters * 8 * ops_per_iter * wps;   // wave-instructions issued on each SIMD
		printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (= %.2f cycles at 2.4 GHz)\n", name, wps, ms,
		       ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4);
	}
	CHK(hipFree(d));
}

int main()
{
	run<0>("v_add_f32 x8", 8);
	run<1>("v_max/min_f32 x8", 8);
	run<2>("cvt_i32_f32+cvt_f32_i32 x8", 16);
	run<3>("sub+min|.| x4, add x4", 12);
	run<4>("v_pk_add_f32 x4 (8 adds)", 4);
	return 0;
}
