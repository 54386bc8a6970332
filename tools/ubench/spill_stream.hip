// HBM streaming rate for the decode kernel's spill pattern: every resident wave writes its own multi-megabyte
// region in 1 KiB pieces (64 lanes x 16 B), then reads it back -- versus one flat copy over the same bytes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int NT>
__global__ void __launch_bounds__(512) per_wave(float4* ws, size_t slot_f4, int pieces, float4* sink)
{
	const int lane = threadIdx.x & 63;
	const size_t slot = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	float4* p = ws + slot * slot_f4;
	float4 v = make_float4(lane, 1.f, 2.f, 3.f);
	for (int i = 0; i < pieces; i++) {
		if (NT) __builtin_nontemporal_store(v.x, &p[(size_t)i * 64 + lane].x), __builtin_nontemporal_store(v.y, &p[(size_t)i * 64 + lane].y),
		        __builtin_nontemporal_store(v.z, &p[(size_t)i * 64 + lane].z), __builtin_nontemporal_store(v.w, &p[(size_t)i * 64 + lane].w);
		else p[(size_t)i * 64 + lane] = v;
		v.x += 1.0f;
	}
	float4 acc = make_float4(0, 0, 0, 0);
	for (int i = 0; i < pieces; i++) {
		const float4 q = p[(size_t)i * 64 + lane];
		acc.x += q.x; acc.y += q.y; acc.z += q.z; acc.w += q.w;
	}
	if (acc.x == -1.0f) sink[0] = acc;
}

// the kernel's actual pattern: a group of NH HMMs, each with its own region of the slot; per position the wave writes
// (then, in the second sweep, reads) PIECES KiB-pieces into each of the NH regions
template <int NH, int PIECES>
__global__ void __launch_bounds__(512) per_wave_streams(float4* ws, size_t slot_f4, int positions, float4* sink)
{
	const int lane = threadIdx.x & 63;
	const size_t slot = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	float4* p = ws + slot * slot_f4;
	const size_t region = slot_f4 / NH;
	float4 v = make_float4(lane, 1.f, 2.f, 3.f);
	for (int i = positions - 1; i >= 0; i--)
		for (int h = 0; h < NH; h++)
			for (int k = 0; k < PIECES; k++) { p[h * region + ((size_t)i * PIECES + k) * 64 + lane] = v; v.x += 1.0f; }
	float4 acc = make_float4(0, 0, 0, 0);
	for (int i = 0; i < positions; i++)
		for (int h = 0; h < NH; h++)
			for (int k = 0; k < PIECES; k++) { const float4 q = p[h * region + ((size_t)i * PIECES + k) * 64 + lane]; acc.x += q.x; acc.y += q.y; acc.z += q.z; acc.w += q.w; }
	if (acc.x == -1.0f) sink[0] = acc;
}

__global__ void flat_copy(const float4* a, float4* b, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

int main()
{
	const int waves = 4096, pieces = 4800;            // 4.7 MB per wave, 19.3 GB in all (x2: written, then read)
	const size_t slot_f4 = (size_t)pieces * 64;
	float4 *ws, *sink;
	CHK(hipMalloc(&ws, slot_f4 * waves * sizeof(float4)));
	CHK(hipMalloc(&sink, 64));
	hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	const double bytes = 2.0 * (double)slot_f4 * waves * sizeof(float4);
	for (int rep = 0; rep < 3; rep++) {
		CHK(hipEventRecord(e0));
		hipLaunchKernelGGL(per_wave<0>, dim3(waves / 8), dim3(512), 0, 0, ws, slot_f4, pieces, sink);
		CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
		float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
		printf("per-wave regions, write then read: %.2f ms  %.2f TB/s\n", ms, bytes / ms / 1e9);
	}
	for (int rep = 0; rep < 2; rep++) {
		CHK(hipEventRecord(e0));
		hipLaunchKernelGGL((per_wave_streams<8, 3>), dim3(waves / 8), dim3(512), 0, 0, ws, slot_f4, pieces / 24, sink);
		CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
		float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
		printf("8 streams x 3 KiB per position, backward then forward: %.2f ms  %.2f TB/s\n", ms, bytes / ms / 1e9);
	}
	for (int rep = 0; rep < 2; rep++) {
		CHK(hipEventRecord(e0));
		hipLaunchKernelGGL((per_wave_streams<1, 24>), dim3(waves / 8), dim3(512), 0, 0, ws, slot_f4, pieces / 24, sink);
		CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
		float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
		printf("1 stream x 24 KiB per position, backward then forward:  %.2f ms  %.2f TB/s\n", ms, bytes / ms / 1e9);
	}
	{
		const size_t n = slot_f4 * waves / 2;
		for (int rep = 0; rep < 3; rep++) {
			CHK(hipEventRecord(e0));
			hipLaunchKernelGGL(flat_copy, dim3(256 * 16), dim3(256), 0, 0, ws, ws + n, n);
			CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
			float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
			printf("flat copy of the same bytes:        %.2f ms  %.2f TB/s\n", ms, (double)n * 32 / ms / 1e9);
		}
	}
	return 0;
}
