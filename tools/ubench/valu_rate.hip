// VALU issue-rate microbenchmark for gfx950: ns per wave64 instruction per SIMD at 1..8 waves/SIMD.
// Every instruction is written as inline asm so the compiler can neither SLP-pack scalar adds into v_pk_add_f32
// (it does under plain -O3, which makes a "v_add_f32" loop look twice as fast as it is) nor fold anything away.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

#define OP1(ins, r) asm volatile(ins " %0, %0, %1" : "+v"(r) : "v"(c))
#define OPC(ins, r) asm volatile(ins " %0, %0" : "+v"(r))
#define OP8(ins) OP1(ins, a0); OP1(ins, a1); OP1(ins, a2); OP1(ins, a3); OP1(ins, a4); OP1(ins, a5); OP1(ins, a6); OP1(ins, a7)

template <int KIND>
__global__ void k(float* out, int iters, float seed)
{
	float a0 = seed + threadIdx.x, a1 = a0 * 1.1f, a2 = a0 * 1.2f, a3 = a0 * 1.3f, a4 = a0 * 1.4f, a5 = a0 * 1.5f, a6 = a0 * 1.6f, a7 = a0 * 1.7f;
	float c = seed * 0.5f;
	unsigned long long mask = __ballot(threadIdx.x & 1);
	float sc = __builtin_amdgcn_readfirstlane(__float_as_int(seed)) * 0.25f;
	__shared__ float lds[16384];
	for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i;
	__syncthreads();
	unsigned addr = ((threadIdx.x * 2654435761u) >> 18) * 4u;   // pseudo-random dword in the table
	f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = p0 * 2.0f, p5 = p1 * 2.0f, p6 = p2 * 2.0f, p7 = p3 * 2.0f, cc = {c, c};
	for (int i = 0; i < iters; i++) {
#pragma unroll
		for (int u = 0; u < 8; u++) {
			if (KIND == 0) { OP8("v_add_f32"); }
			if (KIND == 1) { OP8("v_max_f32"); }
			if (KIND == 2) { OP8("v_mul_f32"); }
			if (KIND == 3) { OPC("v_cvt_i32_f32", a0); OPC("v_cvt_i32_f32", a1); OPC("v_cvt_i32_f32", a2); OPC("v_cvt_i32_f32", a3); OPC("v_cvt_i32_f32", a4); OPC("v_cvt_i32_f32", a5); OPC("v_cvt_i32_f32", a6); OPC("v_cvt_i32_f32", a7); }
			if (KIND == 4) {
#define PK(ins, r) asm volatile(ins " %0, %0, %1" : "+v"(r) : "v"(cc))
				PK("v_pk_add_f32", p0); PK("v_pk_add_f32", p1); PK("v_pk_add_f32", p2); PK("v_pk_add_f32", p3);
				PK("v_pk_add_f32", p4); PK("v_pk_add_f32", p5); PK("v_pk_add_f32", p6); PK("v_pk_add_f32", p7);
			}
			if (KIND == 5) {
				PK("v_pk_mul_f32", p0); PK("v_pk_mul_f32", p1); PK("v_pk_mul_f32", p2); PK("v_pk_mul_f32", p3);
				PK("v_pk_mul_f32", p4); PK("v_pk_mul_f32", p5); PK("v_pk_mul_f32", p6); PK("v_pk_mul_f32", p7);
			}
			if (KIND == 6) { // one dependent chain per wave: latency of back-to-back dependent v_add_f32
				OP1("v_add_f32", a0); OP1("v_add_f32", a0); OP1("v_add_f32", a0); OP1("v_add_f32", a0);
				OP1("v_add_f32", a0); OP1("v_add_f32", a0); OP1("v_add_f32", a0); OP1("v_add_f32", a0);
			}
			if (KIND == 8) { OP8("v_min_f32"); }
			if (KIND == 9) { OP8("v_sub_f32"); }
			if (KIND == 10) { // shift by a constant
#define SH(r) asm volatile("v_lshlrev_b32 %0, 2, %0" : "+v"(r))
				SH(a0); SH(a1); SH(a2); SH(a3); SH(a4); SH(a5); SH(a6); SH(a7);
			}
			if (KIND == 11) { // select on an SGPR-pair mask (VOP3)
#define CM(r) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r) : "v"(c), "s"(mask))
				CM(a0); CM(a1); CM(a2); CM(a3); CM(a4); CM(a5); CM(a6); CM(a7);
			}
			if (KIND == 12) { // VOP3 min with |.| modifier
#define MA(r) asm volatile("v_min_f32_e64 %0, |%0|, %1" : "+v"(r) : "v"(c))
				MA(a0); MA(a1); MA(a2); MA(a3); MA(a4); MA(a5); MA(a6); MA(a7);
			}
			if (KIND == 13) { // add with a 32-bit literal (8-byte encoding)
#define AL(r) asm volatile("v_add_f32 %0, 0x40490fdb, %0" : "+v"(r))
				AL(a0); AL(a1); AL(a2); AL(a3); AL(a4); AL(a5); AL(a6); AL(a7);
			}
			if (KIND == 14) { // add with an SGPR operand
#define AS(r) asm volatile("v_add_f32 %0, %1, %0" : "+v"(r) : "s"(sc))
				AS(a0); AS(a1); AS(a2); AS(a3); AS(a4); AS(a5); AS(a6); AS(a7);
			}
			if (KIND == 15) { OP8("v_and_b32"); }
			if (KIND == 16) { OP8("v_add_u32"); }
			if (KIND == 17) { // compare into an SGPR pair
#define CP(r) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(mask) : "v"(r), "v"(c))
				CP(a0); CP(a1); CP(a2); CP(a3); CP(a4); CP(a5); CP(a6); CP(a7);
			}
			if (KIND == 18) { // LDS gather, lane-dependent addresses (as the logsum table look-up)
#define LD(r) asm volatile("ds_read_b32 %0, %1" : "=v"(r) : "v"(addr))
				float t0, t1, t2, t3, t4, t5, t6, t7;
				LD(t0); LD(t1); LD(t2); LD(t3); LD(t4); LD(t5); LD(t6); LD(t7);
				asm volatile("s_waitcnt lgkmcnt(0)");
				asm volatile("" :: "v"(t0), "v"(t1), "v"(t2), "v"(t3), "v"(t4), "v"(t5), "v"(t6), "v"(t7));
			}
			if (KIND == 19) { OP8("v_max_f32"); OP8("v_add_f32"); } // does a slow op leave room for a fast one?
			if (KIND == 20) { // v_med3 / 3-operand
#define M3(r) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(r) : "v"(c), "v"(a7))
				M3(a0); M3(a1); M3(a2); M3(a3); M3(a4); M3(a5); M3(a6); M3(a0);
			}
			if (KIND == 21) { // fma
#define FM(r) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r) : "v"(c))
				FM(a0); FM(a1); FM(a2); FM(a3); FM(a4); FM(a5); FM(a6); FM(a7);
			}
			if (KIND == 7) { // v_cndmask with an SGPR-pair condition
				asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a0) : "v"(c)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a1) : "v"(c));
				asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a2) : "v"(c)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a3) : "v"(c));
				asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a4) : "v"(c)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a5) : "v"(c));
				asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a6) : "v"(c)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a7) : "v"(c));
			}
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)mask + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
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
		printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD\n", name, wps, ms, ms * 1e6 / insts_per_simd);
	}
	CHK(hipFree(d));
}

int main()
{
	run<0>("v_add_f32 x8", 8);
	run<1>("v_max_f32 x8", 8);
	run<2>("v_mul_f32 x8", 8);
	run<3>("v_cvt_i32_f32 x8", 8);
	run<4>("v_pk_add_f32 x8 (16 adds)", 8);
	run<5>("v_pk_mul_f32 x8 (16 muls)", 8);
	run<6>("v_add_f32 dependent x8", 8);
	run<7>("v_cndmask_b32 vcc x8", 8);
	run<8>("v_min_f32 x8", 8);
	run<9>("v_sub_f32 x8", 8);
	run<10>("v_lshlrev_b32 x8", 8);
	run<11>("v_cndmask_b32_e64 sgpr x8", 8);
	run<12>("v_min_f32_e64 |a| x8", 8);
	run<13>("v_add_f32 literal x8", 8);
	run<14>("v_add_f32 sgpr x8", 8);
	run<15>("v_and_b32 x8", 8);
	run<16>("v_add_u32 x8", 8);
	run<17>("v_cmp_lt_f32_e64 ->sgpr x8", 8);
	run<18>("ds_read_b32 gather x8", 8);
	run<19>("v_max_f32 x8 + v_add_f32 x8", 16);
	run<20>("v_med3_f32 x8", 8);
	run<21>("v_fma_f32 x8", 8);
	return 0;
}
