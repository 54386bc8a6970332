// td_host_inner.h -- the HOST halves of td_submit / td_wait that touch every byte of a batch: the staging copy into page-locked
// memory, the rebuilding of the rewritten sequences from keep bits, the expansion of the label runs.  No HIP in here: td_api.hip
// calls these around its device calls, and csrc/td_hostbench.cpp runs them alone (tools/host_scale.py: what do N ranks' host
// halves cost one host when they run side by side?).
#pragma once
#include <stdint.h>
#include <string.h>

#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

// The copy threads of one context: started once, reused by every batch (the calling thread takes a share of each copy itself).
struct CopyPool {
	struct Job { char* dst; const char* src; size_t bytes; const std::function<void(int64_t, int64_t)>* fn; int64_t lo, hi; };
	std::vector<std::thread> th;
	std::mutex mu;
	std::condition_variable cv_job, cv_done;
	std::deque<Job> q;
	int pending = 0;
	bool stop = false;

	void start(int n_workers)
	{
		for (int k = (int)th.size(); k < n_workers; k++)
			th.emplace_back([this] {
				for (;;) {
					Job j;
					{
						std::unique_lock<std::mutex> lk(mu);
						cv_job.wait(lk, [this] { return stop || !q.empty(); });
						if (q.empty()) return;   // stop
						j = q.front(); q.pop_front();
					}
					if (j.fn) (*j.fn)(j.lo, j.hi); else memcpy(j.dst, j.src, j.bytes);
					{
						std::lock_guard<std::mutex> lk(mu);
						if (--pending == 0) cv_done.notify_all();
					}
				}
			});
	}
	// memcpy(dst, src, bytes) on nt threads (this one included)
	void copy(void* dst, const void* src, size_t bytes, int nt)
	{
		const size_t chunk = (size_t)1 << 20;
		const size_t nchunks = (bytes + chunk - 1) / chunk;
		if ((size_t)nt > nchunks) nt = (int)nchunks;
		if (nchunks <= 4 || nt <= 1) { memcpy(dst, src, bytes); return; }
		start(nt - 1);
		const size_t per = (nchunks + (size_t)nt - 1) / (size_t)nt * chunk;
		{
			std::lock_guard<std::mutex> lk(mu);
			for (size_t lo = per; lo < bytes; lo += per) {
				q.push_back(Job{ (char*)dst + lo, (const char*)src + lo, lo + per < bytes ? per : bytes - lo, nullptr, 0, 0 });
				pending++;
			}
		}
		cv_job.notify_all();
		memcpy(dst, src, per < bytes ? per : bytes);
		std::unique_lock<std::mutex> lk(mu);
		cv_done.wait(lk, [this] { return pending == 0; });
	}
	// fn(lo, hi) over [0, n) in contiguous ranges on nt threads (this one included)
	void ranges(int64_t n, int nt, const std::function<void(int64_t, int64_t)>& fn)
	{
		if (n <= 0) return;
		if (nt > n / 4096) nt = (int)(n / 4096);
		if (nt <= 1) { fn(0, n); return; }
		start(nt - 1);
		const int64_t per = (n + nt - 1) / nt;
		{
			std::lock_guard<std::mutex> lk(mu);
			for (int64_t lo = per; lo < n; lo += per) { q.push_back(Job{ nullptr, nullptr, 0, &fn, lo, lo + per < n ? lo + per : n }); pending++; }
		}
		cv_job.notify_all();
		fn(0, per < n ? per : n);
		std::unique_lock<std::mutex> lk(mu);
		cv_done.wait(lk, [this] { return pending == 0; });
	}
	~CopyPool()
	{
		{ std::lock_guard<std::mutex> lk(mu); stop = true; }
		cv_job.notify_all();
		for (auto& t : th) t.join();
	}
};


// make_extracted_read(), barcode_hmm.c:3343-3350, from the keep bits and the batch's bases on the host (base codes as the device
// sees them: init_nuc_code for sequence text, anything above 4 is 4): out[offs[i] + p] = kept ? code : 65.
static inline void td_host_rebuild_sequences(CopyPool& pool, int host_threads, int64_t n, const uint8_t* raw, const int64_t* offs,
                                             const uint32_t* kb, int nw1, int ascii, uint8_t* out)
{
	static const struct Lut { uint8_t t[256]; Lut() { for (int k = 0; k < 256; k++) t[k] = 4; t['A'] = t['a'] = 0; t['C'] = t['c'] = 1; t['G'] = t['g'] = 2; t['T'] = t['t'] = t['U'] = t['u'] = 3; } } lut;
	const std::function<void(int64_t, int64_t)> fn = [=](int64_t lo, int64_t hi) {
		for (int64_t i = lo; i < hi; i++) {
			const int64_t o = offs[i];
			const int len = (int)(offs[i + 1] - o);
			const uint32_t* kw = kb + i * nw1;
			for (int p0 = 0; p0 < len; p0 += 32) {
				const uint32_t w = kw[p0 >> 5];
				const int e = len - p0 < 32 ? len - p0 : 32;
				const uint8_t* src = raw + o + p0; uint8_t* dst = out + o + p0;
				if (w == 0u) { memset(dst, 65, (size_t)e); continue; }
				if (ascii) { for (int q = 0; q < e; q++) dst[q] = ((w >> q) & 1u) ? lut.t[src[q]] : (uint8_t)65; }
				else {
					// eight bases per step: the keep bits of the group spread into byte masks, kept bytes taken as they are
					// (a group holding a code above 4 goes byte by byte)
					int q = 0;
					for (; q + 8 <= e; q += 8) {
						uint64_t v;
						memcpy(&v, src + q, 8);
						if (((v + 0x7B7B7B7B7B7B7B7BULL) | v) & 0x8080808080808080ULL) break;
						uint64_t m = ((uint64_t)((w >> q) & 0xFFu) * 0x0101010101010101ULL) & 0x8040201008040201ULL;
						m = (((m + 0x7F7F7F7F7F7F7F7FULL) & 0x8080808080808080ULL) >> 7) * 0xFFULL;
						v = (v & m) | (0x4141414141414141ULL & ~m);
						memcpy(dst + q, &v, 8);
					}
					for (; q < e; q++) { const uint8_t cd = src[q] > 4 ? (uint8_t)4 : src[q]; dst[q] = ((w >> q) & 1u) ? cd : (uint8_t)65; }
				}
			}
		}
	};
	pool.ranges(n, host_threads, fn);
}

// ri->labels (barcode_hmm.c:4503-4514) from (length << 8 | label) runs: read i's len + 1 labels at out + offs[i] + i
static inline void td_host_expand_labels(CopyPool& pool, int host_threads, int64_t n, const int64_t* offs, const uint32_t* rl, int cap, int8_t* out)
{
	const std::function<void(int64_t, int64_t)> fn = [=](int64_t lo, int64_t hi) {
		for (int64_t i = lo; i < hi; i++) {
			int8_t* p = out + offs[i] + i;
			int8_t* const end = out + offs[i + 1] + i + 1;   // the read's len + 1 labels
			const uint32_t* r = rl + i * cap;
			for (int j = 0; j < cap && r[j]; j++) {
				// a run in 8-byte stores that may run over into the next run of the same read (written after it), never
				// past the read's own labels
				const size_t len = r[j] >> 8;
				const uint64_t pat = (uint64_t)(r[j] & 0xFF) * 0x0101010101010101ULL;
				size_t k = 0;
				for (; k < len && p + k + 8 <= end; k += 8) memcpy(p + k, &pat, 8);
				for (; k < len; k++) p[k] = (int8_t)(r[j] & 0xFF);
				p += len;
			}
		}
	};
	pool.ranges(n, host_threads, fn);
}
