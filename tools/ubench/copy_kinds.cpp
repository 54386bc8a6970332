// copy_kinds: does a device<->host copy issued while a kernel owns every CU proceed beside it (SDMA engines) or behind it
// (a blit kernel that waits for CU slots)?  Times a pinned D2H / H2D copy alone, then under a spin kernel that fills the
// register files for ~45 ms (four rounds of ~11 ms workgroups), for several kinds of copy: plain and high-priority streams,
// hipHostMalloc / Portable / hipHostRegister memory, behind an event wait, and small sizes.  A copy that takes ~1.2 ms per
// 64 MB went through SDMA; one that takes about as long as a round of the spin kernel was a blit kernel waiting for a CU.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/copy_kinds.cpp -o /tmp/copy_kinds && /tmp/copy_kinds
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void __launch_bounds__(256) spin(long long cycles, float* sink)
{
	// 128 live floats per lane: with __launch_bounds__(256) and this pressure the kernel sits at a few waves per SIMD like the decode kernel
	float a[122];
	for (int k = 0; k < 122; k++) a[k] = (float)(threadIdx.x + k);
	const long long t0 = wall_clock64();
	while (wall_clock64() - t0 < cycles) { for (int k = 0; k < 122; k++) a[k] = a[k] * 1.0001f + 0.5f; }
	float s = 0; for (int k = 0; k < 122; k++) s += a[k];
	if (s == 12345.0f) *sink = s;
}

int main()
{
	const size_t bytes = 64u << 20;
	void *d = nullptr, *h = nullptr; float* sink = nullptr;
	CK(hipMalloc(&d, bytes)); CK(hipHostMalloc(&h, bytes, hipHostMallocDefault)); CK(hipMalloc((void**)&sink, 4));
	hipStream_t sk, sc;
	CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
	hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
	const int cus = p.multiProcessorCount;
	const long long cyc = 40LL * 100000;   // wall_clock64 ticks at 100 MHz: ~40 ms
	auto copy_ms = [&](bool d2h) {
		const double t0 = now_ms();
		if (d2h) CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, sc)); else CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, sc));
		CK(hipStreamSynchronize(sc));
		return now_ms() - t0;
	};
	for (int w = 0; w < 2; w++) { copy_ms(true); copy_ms(false); }
	printf("CUs %d   alone: D2H %.2f ms  H2D %.2f ms (64 MB)\n", cus, copy_ms(true), copy_ms(false));
	// which copies go to the SDMA engines (done in ~1.2 ms whatever the CUs do) and which become blit kernels (wait for the spin kernel)?
	hipStream_t sp; int plo = 0, phi = 0;
	CK(hipDeviceGetStreamPriorityRange(&plo, &phi));
	CK(hipStreamCreateWithPriority(&sp, hipStreamNonBlocking, phi));
	void* hp = nullptr; CK(hipHostMalloc(&hp, bytes, hipHostMallocPortable));
	void* hr = malloc(bytes); CK(hipHostRegister(hr, bytes, hipHostRegisterDefault));
	hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
	struct V { const char* name; hipStream_t st; void* host; bool after_event; size_t n; } vs[] = {
		{ "plain stream, hipHostMallocDefault", sc, h, false, bytes },
		{ "high-priority stream", sp, h, false, bytes },
		{ "hipHostMallocPortable memory", sc, hp, false, bytes },
		{ "hipHostRegister memory", sc, hr, false, bytes },
		{ "behind a (completed) event wait", sc, h, true, bytes },
		{ "4 bytes", sc, h, false, 4 },
		{ "1 MB", sc, h, false, 1u << 20 },
	};
	for (const V& v : vs) {
		for (int d2h = 1; d2h >= 0; d2h--) {
			const double t0 = now_ms();
			hipLaunchKernelGGL(spin, dim3(cus * 16), dim3(256), 0, sk, cyc / 4, sink);
			CK(hipGetLastError());
			while (now_ms() - t0 < 3.0) {}
			if (v.after_event) { CK(hipEventRecord(ev, sp)); CK(hipStreamWaitEvent(v.st, ev, 0)); }
			const double c0 = now_ms();
			if (d2h) CK(hipMemcpyAsync(v.host, d, v.n, hipMemcpyDeviceToHost, v.st)); else CK(hipMemcpyAsync(d, v.host, v.n, hipMemcpyHostToDevice, v.st));
			CK(hipStreamSynchronize(v.st));
			const double c = now_ms() - c0;
			CK(hipStreamSynchronize(sk));
			printf("%-40s %s %.2f ms under a kernel of %.1f ms\n", v.name, d2h ? "D2H" : "H2D", c, now_ms() - t0);
		}
	}
	return 0;
}
