// td_stage.hip -- device-side ingest and egress around the decode kernels (gfx950).
//
// What the reference does on the host around run_pHMM()'s thread fan-out (src/barcode_hmm.c:1895-2029), done here by three
// small bandwidth-bound kernels so that the host only moves bytes:
//   ingest : base coding of the sequence text (init_nuc_code, src/nuc_code.c:46-74), a stable sort of the reads by length
//            (so that the 64 reads of a tile have nearly one length), 2-bit + N-mask packing, lane-interleaved per tile;
//   egress : the per-read record (struct read_info's mapq / read_type / barcode / fingerprint, src/io.h:76-91), the
//            sequence as make_extracted_read() rewrites it (src/barcode_hmm.c:3325-3356: non-read positions become byte
//            65) and ri->labels, all back in the caller's order and contiguous per read, so that the download is a plain copy.
// Integer / byte work only; nothing here touches the arithmetic contract.
#include <hip/hip_runtime.h>
#include <string.h>
#include <rocprim/device/device_radix_sort.hpp>

#include "td_device.h"
#include "td_stage.h"

#define STAGE_BLOCK 256

// read lengths as sort keys + identity values
__global__ __launch_bounds__(STAGE_BLOCK) void td_len_iota_kernel(const int64_t* __restrict__ offs, int64_t n,
                                                                    uint32_t* __restrict__ keys, int32_t* __restrict__ vals)
{
	const int64_t i = (int64_t)blockIdx.x * STAGE_BLOCK + threadIdx.x;
	if (i >= n) return;
	keys[i] = (uint32_t)(offs[i + 1] - offs[i]);
	vals[i] = (int32_t)i;
}

static int key_bits(int lmax)
{
	int b = 1;
	while (b < 32 && ((int64_t)1 << b) <= (int64_t)lmax) b++;
	return b;
}

size_t td_stage_sort_temp_bytes(int64_t n_reads, int lmax)
{
	size_t bytes = 0;
	(void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const int32_t*)nullptr,
	                                (int32_t*)nullptr, (size_t)n_reads, 0u, (unsigned)key_bits(lmax), (hipStream_t)0);
	return bytes ? bytes : 256;
}

hipError_t td_stage_sort(const int64_t* offs, int64_t n, int lmax, int32_t* read_at, uint32_t* keys, uint32_t* keys_alt,
                         int32_t* vals_alt, void* temp, size_t temp_bytes, hipStream_t stream)
{
	if (n <= 0) return hipSuccess;
	const unsigned blocks = (unsigned)((n + STAGE_BLOCK - 1) / STAGE_BLOCK);
	hipLaunchKernelGGL(td_len_iota_kernel, dim3(blocks), dim3(STAGE_BLOCK), 0, stream, offs, n, keys, vals_alt);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) return e;
	// LSD radix sort is stable: reads of one length keep the caller's order
	return rocprim::radix_sort_pairs(temp, temp_bytes, (const uint32_t*)keys, keys_alt, (const int32_t*)vals_alt, read_at, (size_t)n,
	                                 0u, (unsigned)key_bits(lmax), stream);
}

// init_nuc_code(), src/nuc_code.c:46-74: A C G T/U in either case -> 0 1 2 3, everything else 4; base codes above 4 are 4
__device__ __forceinline__ uint32_t base_code(uint32_t ch, int is_ascii)
{
	if (!is_ascii) return ch > 4u ? 4u : ch;
	const uint32_t u = ch | 0x20u;
	uint32_t c = 4u;
	c = (u == 'a') ? 0u : c;
	c = (u == 'c') ? 1u : c;
	c = (u == 'g') ? 2u : c;
	c = (u == 't' || u == 'u') ? 3u : c;
	return c;
}

// One workgroup per tile: 4 waves, each lane owns a read (so every packed word is one coalesced 256-byte store per wave) and
// the waves share the tile's 32-base chunks between them.
__global__ __launch_bounds__(STAGE_BLOCK) void td_pack_kernel(const TdStageBatch b)
{
	const int tile = blockIdx.x;
	const int lane = threadIdx.x & (TD_WAVE - 1);
	const int wv = threadIdx.x >> 6;
	const int64_t k = (int64_t)tile * TD_WAVE + lane;
	int64_t i = -1;
	int len = 0;
	const uint8_t* src = nullptr;
	if (k < b.n_reads) {
		i = b.read_at ? (int64_t)b.read_at[k] : k;
		const int64_t o = b.offs[i];
		len = (int)(b.offs[i + 1] - o);
		src = b.raw + o;
	}
	uint32_t* pk = b.packed + (int64_t)tile * (b.nw2 + b.nw1) * TD_WAVE;
	for (int c = wv; c < b.nw1; c += STAGE_BLOCK / TD_WAVE) {
		uint32_t w2a = 0, w2b = 0, w1 = 0;
		const int p0 = c * 32;
		const int e = len - p0 < 32 ? len - p0 : 32;
		for (int q = 0; q < e; q++) {
			const uint32_t cd = base_code(src[p0 + q], b.is_ascii);
			if (cd == 4u) w1 |= 1u << q;
			else if (q < 16) w2a |= cd << (2 * q);
			else w2b |= cd << (2 * (q - 16));
		}
		pk[(2 * c) * TD_WAVE + lane] = w2a;
		if (2 * c + 1 < b.nw2) pk[(2 * c + 1) * TD_WAVE + lane] = w2b;
		pk[(b.nw2 + c) * TD_WAVE + lane] = w1;
	}
	if (wv == 0) b.lens[k] = len;
}

// -ref artifact filter: match_to_reference() takes the reads of each thread range [t*interval, ...) in fours and gives the
// (range length mod 4) left-over reads to another routine (src/barcode_hmm.c:2495-2575, ranges as in run_pHMM :1911-1922).
__global__ __launch_bounds__(STAGE_BLOCK) void td_art_left_kernel(const TdStageBatch b)
{
	const int64_t k = (int64_t)blockIdx.x * STAGE_BLOCK + threadIdx.x;
	if (k >= (int64_t)b.n_tiles * TD_WAVE) return;
	uint8_t left = 0;
	if (k < b.n_reads && b.art_threads > 0) {
		const int64_t i = b.art_first + (b.read_at ? (int64_t)b.read_at[k] : k);
		const int64_t n = b.art_total > 0 ? b.art_total : b.n_reads, T = b.art_threads, interval = n / T;
		int64_t t = interval > 0 ? i / interval : T - 1;
		if (t > T - 1) t = T - 1;
		const int64_t start = t * interval, end = (t == T - 1) ? n : (t + 1) * interval;
		left = i >= start + (end - start) / 4 * 4;
	}
	b.art_left[k] = left;
}

hipError_t td_stage_art_left(const TdStageBatch& b, hipStream_t stream)
{
	if (b.n_tiles <= 0) return hipSuccess;
	const unsigned blocks = (unsigned)(((int64_t)b.n_tiles * TD_WAVE + STAGE_BLOCK - 1) / STAGE_BLOCK);
	hipLaunchKernelGGL(td_art_left_kernel, dim3(blocks), dim3(STAGE_BLOCK), 0, stream, b);
	return hipGetLastError();
}

hipError_t td_stage_pack(const TdStageBatch& b, hipStream_t stream)
{
	if (b.n_tiles <= 0) return hipSuccess;
	hipLaunchKernelGGL(td_pack_kernel, dim3((unsigned)b.n_tiles), dim3(STAGE_BLOCK), 0, stream, b);
	return hipGetLastError();
}

// One workgroup per tile (device order).  Per-read records first (one lane per read), then the tile's (read, position)
// pairs are spread over the threads: consecutive threads write consecutive bytes of one read.
__global__ __launch_bounds__(STAGE_BLOCK) void td_finish_kernel(const TdStageBatch b)
{
	const int tile = blockIdx.x;
	__shared__ int64_t s_off[TD_WAVE];
	__shared__ int64_t s_idx[TD_WAVE];
	__shared__ int32_t s_len[TD_WAVE];
	if (threadIdx.x < TD_WAVE) {
		const int lane = threadIdx.x;
		const int64_t k = (int64_t)tile * TD_WAVE + lane;
		int64_t i = -1, o = 0;
		int len = -1;
		if (k < b.n_reads) {
			i = b.read_at ? (int64_t)b.read_at[k] : k;
			o = b.offs[i];
			len = (int)(b.offs[i + 1] - o);
			if (b.res) {
				uint32_t v[8];
#pragma unroll
				for (int a = 0; a < 8; a++) v[a] = *(const uint32_t*)(b.out_soa + a * b.soa_stride + k * 4);
				uint4* dst = (uint4*)(b.res + i * 32);
				dst[0] = make_uint4(v[0], v[1], v[2], v[3]);
				dst[1] = make_uint4(v[4], v[5], v[6], v[7]);
			}
		}
		s_off[lane] = o; s_idx[lane] = i; s_len[lane] = len;
		if (b.keep_out && k < b.n_reads) {
			const uint32_t* kw = b.keep + (int64_t)tile * b.nw1 * TD_WAVE;
			for (int w = 0; w < b.nw1; w++) b.keep_out[i * b.nw1 + w] = kw[w * TD_WAVE + lane];
		}
		if (b.rle_out && b.runs) {      // the decode kernel left the runs: into the caller's order
			if (b.runs_overflow && blockIdx.x == 0 && threadIdx.x == 0 && *b.runs_overflow) atomicOr(b.rle_overflow, 1);
			if (k < b.n_reads) {
				const uint32_t* rs = b.runs + (int64_t)tile * b.rle_cap * TD_WAVE + lane;
				for (int j = 0; j < b.rle_cap; j++) b.rle_out[i * b.rle_cap + j] = rs[j * TD_WAVE];
			}
		} else if (b.rle_out) {
			// the label path of a read is a handful of runs (it moves through the segments in order): (length, label) pairs
			const int8_t* lb = b.labels + (int64_t)tile * (b.lmax + 1) * TD_WAVE;
			uint32_t* dst = (k < b.n_reads) ? b.rle_out + i * b.rle_cap : nullptr;
			int nr = 0, run = 0, cur = 0;
			int lmx = len;
			for (int o2 = 32; o2 >= 1; o2 >>= 1) { const int t2 = __shfl_xor(lmx, o2); lmx = t2 > lmx ? t2 : lmx; }
			for (int p = 0; p <= lmx; p++) {
				const int v = (len >= 1 && p <= len) ? (int)lb[p * TD_WAVE + lane] : 0;   // (a read without bases has the one label 0)
				if (p <= len || (p == 0 && len < 1)) {
					if (p == 0) { cur = v; run = 1; }
					else if (v == cur) run++;
					else { if (dst && nr < b.rle_cap) dst[nr] = ((uint32_t)run << 8) | (uint32_t)(cur & 0xFF); nr++; cur = v; run = 1; }
				}
			}
			if (dst) {
				if (nr < b.rle_cap) dst[nr] = ((uint32_t)run << 8) | (uint32_t)(cur & 0xFF);
				nr++;
				for (int j = nr; j < b.rle_cap; j++) dst[j] = 0u;
				if (nr > b.rle_cap) atomicOr(b.rle_overflow, 1);
			}
		}
	}
	__syncthreads();
	if (b.seq_out) {
		const uint32_t* kw = b.keep + (int64_t)tile * b.nw1 * TD_WAVE;
		const int span = b.lmax;
		const int total = TD_WAVE * span;
		for (int t = threadIdx.x; t < total; t += STAGE_BLOCK) {
			const int r = t / span, p = t - r * span;
			if (p < s_len[r]) {
				const int64_t at = s_off[r] + p;
				const uint32_t cd = base_code(b.raw[at], b.is_ascii);
				const uint32_t w = kw[(p >> 5) * TD_WAVE + r];
				b.seq_out[at] = ((w >> (p & 31)) & 1u) ? (uint8_t)cd : (uint8_t)65;   // spacer byte, barcode_hmm.c:3348
			}
		}
	}
	if (b.labels_out) {
		const int8_t* lb = b.labels + (int64_t)tile * (b.lmax + 1) * TD_WAVE;
		const int span = b.lmax + 1;
		const int total = TD_WAVE * span;
		for (int t = threadIdx.x; t < total; t += STAGE_BLOCK) {
			const int r = t / span, p = t - r * span;
			if (p <= s_len[r]) b.labels_out[s_off[r] + s_idx[r] + p] = s_len[r] >= 1 ? lb[p * TD_WAVE + r] : (int8_t)0;
		}
	}
}

hipError_t td_stage_finish(const TdStageBatch& b, hipStream_t stream)
{
	if (b.n_tiles <= 0) return hipSuccess;
	hipLaunchKernelGGL(td_finish_kernel, dim3((unsigned)b.n_tiles), dim3(STAGE_BLOCK), 0, stream, b);
	return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Workspace placement probe.  The decode kernel streams every wave slot of the workspace to HBM and back; how fast that
// goes depends on where the driver placed the allocation (one box: 36.9 ... 40.4 ms for the same batch, following the
// allocation, not the time -- tools/bimodal_probe.py).  This kernel does the memory side alone on a candidate allocation:
// every wave writes and reads back `pieces` runs of 16 KiB spread evenly over its slot (so the whole footprint and its
// address translation are touched), 16 B per lane like the spill.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void td_ws_probe_kernel(uint8_t* __restrict__ ws, int64_t slot_bytes, int n_slots, int pieces)
{
	const int lane = threadIdx.x & (TD_WAVE - 1);
	const int slot = blockIdx.x * (512 / TD_WAVE) + (threadIdx.x >> 6);
	if (slot >= n_slots) return;
	uint8_t* base = ws + (int64_t)slot * slot_bytes;
	const int64_t step = ((slot_bytes - 16384) / pieces) & ~(int64_t)255;
	float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	for (int p = 0; p < pieces; p++) {
		float4* q = (float4*)(base + p * step);
		for (int k = 0; k < 16; k++) q[k * TD_WAVE + lane] = make_float4((float)p, (float)k, (float)lane, 1.0f);
	}
	for (int p = 0; p < pieces; p++) {
		const float4* q = (const float4*)(base + p * step);
		for (int k = 0; k < 16; k++) { const float4 v = q[k * TD_WAVE + lane]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
	}
	if (acc.x == -1.0f) ((float4*)base)[lane] = acc;   // keep the loads
}

// best of three timed passes, in milliseconds (a warm-up pass first)
hipError_t td_ws_probe(uint8_t* ws, int64_t slot_bytes, int n_slots, hipStream_t stream, float* ms)
{
	*ms = 0.0f;
	if (n_slots <= 0 || slot_bytes < (1 << 20)) return hipSuccess;
	hipEvent_t e0, e1;
	hipError_t e = hipEventCreate(&e0);
	if (e != hipSuccess) return e;
	e = hipEventCreate(&e1);
	if (e != hipSuccess) { (void)hipEventDestroy(e0); return e; }
	const int pieces = 48;
	const unsigned blocks = (unsigned)((n_slots + 7) / 8);
	float best = 1e30f;
	for (int rep = 0; rep < 4 && e == hipSuccess; rep++) {
		(void)hipEventRecord(e0, stream);
		hipLaunchKernelGGL(td_ws_probe_kernel, dim3(blocks), dim3(512), 0, stream, ws, slot_bytes, n_slots, pieces);
		(void)hipEventRecord(e1, stream);
		e = hipEventSynchronize(e1);
		float t = 0.0f;
		if (e == hipSuccess) e = hipEventElapsedTime(&t, e0, e1);
		if (rep > 0 && t < best) best = t;
	}
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
	*ms = best;
	return e;
}
