// How often does (float)exp((double)x) differ between the device (ocml) and the host's libm (what the reference uses,
// scaledprob2prob src/misc.c:98-105)?  Same for the Q formula's log10.  Exhaustive over a dense float range.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void k_exp(const float* x, float* y, size_t n)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) y[i] = (float)exp((double)x[i]);
}
__global__ void k_q(const float* p, float* y, size_t n)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) y[i] = (float)(-10.0 * log10((double)p[i]));
}

int main()
{
	const size_t n = 1u << 26; // 67 M values per sweep
	std::vector<float> hx(n), hy(n);
	float *dx, *dy;
	CHK(hipMalloc(&dx, n * 4)); CHK(hipMalloc(&dy, n * 4));
	size_t bad_exp = 0, bad_q = 0, total = 0;
	for (int sweep = 0; sweep < 4; sweep++) {
		// consecutive float bit patterns starting at -104 (sweep 0), -20, -1, -1e-3: covers the whole live range densely
		const float starts[4] = { -104.0f, -20.0f, -1.0f, -1e-3f };
		uint32_t bits; memcpy(&bits, &starts[sweep], 4);
		for (size_t i = 0; i < n; i++) { uint32_t b = bits - (uint32_t)i; memcpy(&hx[i], &b, 4); } // towards zero
		CHK(hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice));
		hipLaunchKernelGGL(k_exp, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dx, dy, n);
		CHK(hipMemcpy(hy.data(), dy, n * 4, hipMemcpyDeviceToHost));
		for (size_t i = 0; i < n; i++) { const float want = (float)exp((double)hx[i]); if (memcmp(&want, &hy[i], 4)) bad_exp++; }
		total += n;
	}
	for (int sweep = 0; sweep < 2; sweep++) {
		const float starts[2] = { 1e-6f, 0.5f };
		uint32_t bits; memcpy(&bits, &starts[sweep], 4);
		for (size_t i = 0; i < n; i++) { uint32_t b = bits + (uint32_t)i; memcpy(&hx[i], &b, 4); if (!(hx[i] < 1.0f)) hx[i] = 0.999f; }
		CHK(hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice));
		hipLaunchKernelGGL(k_q, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dx, dy, n);
		CHK(hipMemcpy(hy.data(), dy, n * 4, hipMemcpyDeviceToHost));
		for (size_t i = 0; i < n; i++) { const float want = (float)(-10.0 * log10((double)hx[i])); if (memcmp(&want, &hy[i], 4)) bad_q++; }
	}
	printf("exp: %zu of %zu float results differ from libm\n", bad_exp, total);
	printf("-10*log10: %zu of %zu float results differ from libm\n", bad_q, 2 * n);
	return 0;
}
