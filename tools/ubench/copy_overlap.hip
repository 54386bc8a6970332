// copy_overlap.hip -- does an H2D copy queued on its own stream run while a long kernel occupies the GPU?
// Mimics the pipelined batches of libtagdust_hip: stream A = compute (long kernel), B = uploads, C = downloads.
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench/copy_overlap.hip -o tools/ubench/copy_overlap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void spin(long long cycles, int* sink)
{
	const long long t0 = wall_clock64();
	while (wall_clock64() - t0 < cycles) { }
	if (sink && threadIdx.x == 0 && blockIdx.x == 0) *sink = 1;
}

// fills the register file and most of the LDS like the decode kernel does (128 VGPRs x 4 waves per SIMD, 66.5 KB LDS per
// 512-thread workgroup, two workgroups per CU): no other wave fits on a CU while it runs
__global__ __launch_bounds__(512, 4) void spin_full(long long cycles, float* out)
{
	__shared__ float lds[16640];
	float r[96];
#pragma unroll
	for (int k = 0; k < 96; k++) r[k] = (float)(threadIdx.x + k);
	lds[threadIdx.x] = r[5];
	__syncthreads();
	const long long t0 = wall_clock64();
	while (wall_clock64() - t0 < cycles) {
#pragma unroll
		for (int k = 0; k < 96; k++) r[k] = r[k] * 1.0001f + lds[(threadIdx.x + k) & 16383];
	}
	float s = 0;
#pragma unroll
	for (int k = 0; k < 96; k++) s += r[k];
	if (s == 12345.678f) out[0] = s;
}

// streams through HBM for `cycles`: 4096 waves each reading and writing its own 4 MB region
__global__ __launch_bounds__(512, 4) void spin_hbm(long long cycles, float4* buf)
{
	const int wave = (blockIdx.x * 512 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
	float4* p = buf + (size_t)wave * (4u << 20) / 16;
	const long long t0 = wall_clock64();
	float4 acc = {0, 0, 0, 0};
	while (wall_clock64() - t0 < cycles) {
		for (int k = 0; k < 4096; k++) {
			float4 v = p[k * 64 + lane];
			acc.x += v.x;
			p[k * 64 + lane] = acc;
		}
	}
}

static double now_ms()
{
	return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv)
{
	const int variant = argc > 1 ? atoi(argv[1]) : 0;
	const size_t bytes = 157u << 20;
	hipStream_t A, B, C;
	CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
	CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
	CK(hipStreamCreateWithFlags(&C, hipStreamNonBlocking));
	char *h_in, *h_out, *d_in, *d_out;
	int* d_sink;
	CK(hipHostMalloc((void**)&h_in, bytes, hipHostMallocDefault));
	CK(hipHostMalloc((void**)&h_out, bytes, hipHostMallocDefault));
	CK(hipMalloc((void**)&d_in, bytes));
	CK(hipMalloc((void**)&d_out, bytes));
	CK(hipMalloc((void**)&d_sink, 4));
	char* d_big = nullptr;
	if (variant & 16) CK(hipMalloc((void**)&d_big, (size_t)4096 * (4u << 20)));
	hipEvent_t eK, eUp, eDown, eDone;
	CK(hipEventCreate(&eK)); CK(hipEventCreate(&eUp)); CK(hipEventCreate(&eDown)); CK(hipEventCreate(&eDone));
	// 100 MHz wall clock: 40 ms = 4e6 ticks
	const long long ticks = 4000000;
	const int blocks = (variant & 1) ? 64 : 512;      // bit 0: a kernel that leaves most CUs free
	for (int rep = 0; rep < 3; rep++) {
		const double t0 = now_ms();
		CK(hipEventRecord(eDone, A));   // time zero on the device
		if (variant & 8) hipLaunchKernelGGL(spin_full, dim3(512), dim3(512), 0, A, ticks, (float*)d_sink);
		else if (variant & 16) hipLaunchKernelGGL(spin_hbm, dim3(512), dim3(512), 0, A, ticks, (float4*)d_big);
		else hipLaunchKernelGGL(spin, dim3(blocks), dim3(512), 0, A, ticks, d_sink);
		CK(hipEventRecord(eK, A));
		if (variant & 2) {                           // bit 1: a download on C queued behind the kernel (stream wait)
			CK(hipStreamWaitEvent(C, eK, 0));
			CK(hipMemcpyAsync(h_out, d_out, bytes, hipMemcpyDeviceToHost, C));
			CK(hipEventRecord(eDown, C));
		}
		std::this_thread::sleep_for(std::chrono::milliseconds(5));
		const double t1 = now_ms();
		CK(hipMemcpyAsync(d_in, h_in, bytes, hipMemcpyHostToDevice, B));
		CK(hipEventRecord(eUp, B));
		if (variant & 4) {                           // bit 2: the next kernel on A queued behind the upload
			CK(hipStreamWaitEvent(A, eUp, 0));
			hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, A, 1000LL, d_sink);
		}
		const double t2 = now_ms();
		if (variant & 32) {   // the host blocks on the download (like td_wait) instead of polling
			CK(hipEventSynchronize((variant & 2) ? eDown : eK));
			CK(hipDeviceSynchronize());
			float up = 0, k = 0;
			CK(hipEventElapsedTime(&up, eDone, eUp));
			CK(hipEventElapsedTime(&k, eDone, eK));
			printf("variant %2d rep %d (host blocked): kernel done at %.1f ms, upload queued at %.1f, upload done at %.1f ms (device clock)\n", variant, rep, k, t1 - t0, up);
			continue;
		}
		double up_done = -1, k_done = -1, down_done = (variant & 2) ? -1 : 0;
		while (up_done < 0 || k_done < 0 || down_done < 0) {
			if (up_done < 0 && hipEventQuery(eUp) == hipSuccess) up_done = now_ms();
			if (k_done < 0 && hipEventQuery(eK) == hipSuccess) k_done = now_ms();
			if (down_done < 0 && hipEventQuery(eDown) == hipSuccess) down_done = now_ms();
		}
		CK(hipDeviceSynchronize());
		printf("variant %2d rep %d: kernel done at %.1f ms, upload queued at %.1f (call took %.2f), upload done at %.1f ms, download done at %.1f\n", variant, rep,
		       k_done - t0, t1 - t0, t2 - t1, up_done - t0, down_done - t0);
	}
	return 0;
}
