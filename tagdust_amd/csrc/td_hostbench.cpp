// td_hostbench.cpp -- the host halves of td_submit / td_wait WITHOUT a device (diagnostic entry point, tools/host_scale.py).
//
// At 13-15 ms per 2^20-read step a rank's host work -- staging 157 MB into page-locked memory, rebuilding 157 MB of rewritten
// sequences from keep bits, expanding 158 MB of labels from runs, copying 32 MB of records -- is of the order of the decode kernel
// itself, and eight ranks do it on one host.  This runs exactly those routines (td_host_inner.h: the code td_api.hip calls around
// its device calls) on plain memory, back to back, so that N processes side by side show what N ranks' host halves cost the
// machine: the 8-GPU ceiling the host sets, measured without eight GPUs.
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "../../include/tagdust_hip.h"
#include "td_host_inner.h"

static double now_s()
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + (double)ts.tv_nsec * 1e-9;
}

// One "batch" = n_reads reads of read_len bases.  mode 0: pageable caller buffers (staging copy in, records copied out);
// mode 1: page-locked caller buffers under "stable_input" (no staging copy, records land where they belong).  Both: sequences
// rebuilt from keep bits, labels expanded from runs (the compact egress).  out[0] = seconds per batch, out[1] = host bytes read +
// written per batch (what the routines touch, counted from the sizes), out[2] = batches timed.
extern "C" int td_host_halves_bench(int64_t n_reads, int32_t read_len, int32_t n_threads, int32_t iters, int32_t mode, double* out)
{
	if (n_reads <= 0 || read_len <= 0 || n_threads < 1 || iters < 1 || !out) return TD_FAIL;
	const int64_t n = n_reads, L = read_len;
	const int nw1 = (int)((L + 31) / 32), cap = 6;
	const size_t nb = (size_t)(n * L);
	uint8_t* user_in = (uint8_t*)malloc(nb);
	uint8_t* staged = (uint8_t*)malloc(nb);
	uint8_t* seq_out = (uint8_t*)malloc(nb);
	int8_t* labels = (int8_t*)malloc(nb + (size_t)n);
	int64_t* offs_user = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n + 1));
	int64_t* offs = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n + 1));
	uint32_t* keep = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n * nw1));
	uint32_t* runs = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n * cap));
	td_read_result* res_pinned = (td_read_result*)malloc(sizeof(td_read_result) * (size_t)n);
	td_read_result* res_user = (td_read_result*)malloc(sizeof(td_read_result) * (size_t)n);
	if (!user_in || !staged || !seq_out || !labels || !offs_user || !offs || !keep || !runs || !res_pinned || !res_user) return TD_FAIL;
	// plausible contents: a read keeps bases [9, L - 6), its labels are four runs (barcode, spacer, read, adapter)
	uint64_t x = 88172645463325252ULL;
	for (size_t k = 0; k < nb; k++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; user_in[k] = (uint8_t)(x & 3); }
	for (int64_t i = 0; i <= n; i++) offs_user[i] = i * L;
	for (int64_t i = 0; i < n; i++) {
		for (int w = 0; w < nw1; w++) {
			uint32_t m = 0;
			for (int b = 0; b < 32; b++) { const int p = w * 32 + b; if (p >= 9 && p < L - 6) m |= 1u << b; }
			keep[i * nw1 + w] = m;
		}
		uint32_t* r = runs + i * cap;
		r[0] = (7u << 8) | 2u; r[1] = (3u << 8) | 9u; r[2] = ((uint32_t)(L - 15) << 8) | 11u; r[3] = (6u << 8) | 12u; r[4] = 0; r[5] = 0;
	}
	memset(res_pinned, 1, sizeof(td_read_result) * (size_t)n);
	CopyPool pool;
	auto one = [&]() {
		// td_submit's half: the offsets pass (longest / shortest read, relative offsets into page-locked memory), the staging copy
		int lmax = 1;
		offs[0] = 0;
		for (int64_t i = 0; i < n; i++) { const int64_t l = offs_user[i + 1] - offs_user[i]; offs[i + 1] = offs_user[i + 1] - offs_user[0]; if (l > lmax) lmax = (int)l; }
		const uint8_t* raw = user_in;
		if (mode == 0) { pool.copy(staged, user_in, nb, n_threads); raw = staged; }
		// td_wait's half
		if (mode == 0) pool.copy(res_user, res_pinned, sizeof(td_read_result) * (size_t)n, n_threads);
		td_host_rebuild_sequences(pool, n_threads, n, raw, offs, keep, nw1, 0, seq_out);
		td_host_expand_labels(pool, n_threads, n, offs, runs, cap, labels);
		return lmax;
	};
	volatile int sink = one();   // pages touched, pool started
	const double t0 = now_s();
	for (int k = 0; k < iters; k++) sink += one();
	const double dt = now_s() - t0;
	(void)sink;
	double bytes = 16.0 * (double)(n + 1);                                             // offsets read + written
	if (mode == 0) bytes += 2.0 * (double)nb + 2.0 * 32.0 * (double)n;                 // staging copy, records copy (read + write each)
	bytes += (double)nb + 4.0 * (double)(n * nw1) + (double)nb;                        // rebuild: bases + keep bits read, sequences written
	bytes += 4.0 * (double)(n * cap) + (double)(nb + (size_t)n);                       // runs read, labels written
	out[0] = dt / iters; out[1] = bytes; out[2] = iters;
	free(user_in); free(staged); free(seq_out); free(labels); free(offs_user); free(offs); free(keep); free(runs); free(res_pinned); free(res_user);
	return TD_OK;
}
