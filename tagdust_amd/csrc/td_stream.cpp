// td_stream.cpp -- one input file of any size through the decode path as a pipeline (include/tagdust_io.h, td_stream_run):
//
//   reader / parser thread            caller's thread                     writer thread
//   block k+1: read or map, find      batch k: td_submit ... td_wait      batch k-1: format per output file, append
//   records, base-code them into      (several batches in flight on the
//   the next batch's pinned buffers   device)
//
// It replaces the reference's batch loop around run_pHMM for one file (src/barcode_hmm.c:244-385): read_fasta_fastq() of
// <= 1 000 001 records (io.c:1684-1815, through popen("cat|zcat|bzcat"), io.c:382-608) -> run_pHMM -> print_all() appending
// to the per-barcode files (io.c:757-1016) -- with the three steps of consecutive batches running side by side, each of the
// two host steps on several threads.  Batches hold exactly `batch_reads` records like the reference's (the -ref artifact
// filter's per-thread read ranges are taken over a batch, barcode_hmm.c:2478-2583, so the boundaries are part of the result).
#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <math.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include <chrono>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/tagdust_io.h"
#include "td_io_internal.h"

namespace {

double now_s()
{
	return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---- a few persistent threads that run task(k) for k in [0, n) ----
class Pool {
public:
	explicit Pool(int n_threads)
	{
		for (int t = 1; t < n_threads; t++) th_.emplace_back([this] { worker(); });
	}
	~Pool()
	{
		{ std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
		cv_.notify_all();
		for (auto& t : th_) t.join();
	}
	int size() const { return (int)th_.size() + 1; }
	void run(int64_t n, const std::function<void(int64_t)>& task)
	{
		if (n <= 0) return;
		if (th_.empty() || n == 1) { for (int64_t k = 0; k < n; k++) task(k); return; }
		{
			std::lock_guard<std::mutex> lk(mu_);
			task_ = &task; next_ = 0; n_ = n; left_ = n; gen_++;
		}
		cv_.notify_all();
		drain();                               // the calling thread takes tasks too
		std::unique_lock<std::mutex> lk(mu_);
		done_.wait(lk, [this] { return left_ == 0; });
		task_ = nullptr;
	}

private:
	void drain()
	{
		for (;;) {
			int64_t k;
			const std::function<void(int64_t)>* t;
			{
				std::lock_guard<std::mutex> lk(mu_);
				if (!task_ || next_ >= n_) return;
				k = next_++; t = task_;
			}
			(*t)(k);
			std::lock_guard<std::mutex> lk(mu_);
			if (--left_ == 0) done_.notify_all();
		}
	}
	void worker()
	{
		uint64_t seen = 0;
		for (;;) {
			{
				std::unique_lock<std::mutex> lk(mu_);
				cv_.wait(lk, [&] { return stop_ || (gen_ != seen && task_); });
				if (stop_) return;
				seen = gen_;
			}
			drain();
		}
	}
	std::vector<std::thread> th_;
	std::mutex mu_;
	std::condition_variable cv_, done_;
	const std::function<void(int64_t)>* task_ = nullptr;
	int64_t next_ = 0, n_ = 0, left_ = 0;
	uint64_t gen_ = 0;
	bool stop_ = false;
};

// ---- bounded FIFO between the stages ----
template <typename T>
class Queue {
public:
	explicit Queue(size_t cap) : cap_(cap) {}
	bool push(T v)          // false: the pipeline was aborted
	{
		std::unique_lock<std::mutex> lk(mu_);
		cv_.wait(lk, [&] { return abort_ || q_.size() < cap_; });
		if (abort_) return false;
		q_.push_back(std::move(v));
		cv_.notify_all();
		return true;
	}
	bool pop(T& v)          // false: closed and empty, or aborted
	{
		std::unique_lock<std::mutex> lk(mu_);
		cv_.wait(lk, [&] { return abort_ || closed_ || !q_.empty(); });
		if (abort_ || q_.empty()) return false;
		v = std::move(q_.front());
		q_.pop_front();
		cv_.notify_all();
		return true;
	}
	int try_pop(T& v)       // 1: got one; 0: nothing there right now; -1: closed and empty, or aborted
	{
		std::lock_guard<std::mutex> lk(mu_);
		if (abort_) return -1;
		if (q_.empty()) return closed_ ? -1 : 0;
		v = std::move(q_.front());
		q_.pop_front();
		cv_.notify_all();
		return 1;
	}
	void close() { std::lock_guard<std::mutex> lk(mu_); closed_ = true; cv_.notify_all(); }
	void abort() { std::lock_guard<std::mutex> lk(mu_); abort_ = true; cv_.notify_all(); }

private:
	std::mutex mu_;
	std::condition_variable cv_;
	std::deque<T> q_;
	size_t cap_;
	bool closed_ = false, abort_ = false;
};

// ---- input text: one mapping of a plain file, or blocks read from a pipe ----
struct Block {
	const char* data = nullptr;
	int64_t len = 0;
	char* owned = nullptr;          // malloc'ed (pipe input); a mapped file is owned by the Source
	~Block() { free(owned); }
};

class Source {
public:
	~Source()
	{
		if (map_ && map_ != MAP_FAILED) munmap(map_, (size_t)map_len_);
		if (pipe_) pclose(pipe_);
		if (fd_ >= 0) close(fd_);
	}
	bool open(const char* path, int64_t block_bytes, std::string& err)
	{
		block_ = block_bytes;
		const std::string p = path;
		auto ends = [&](const char* sfx) { const size_t n = strlen(sfx); return p.size() >= n && p.compare(p.size() - n, n, sfx) == 0; };
		struct stat st;
		if (p != "-" && stat(path, &st) != 0) { err = "td_stream_run: cannot find input file " + p; return false; }
		std::string q;
		for (char ch : p) { if (ch == '\'') q += "'\\''"; else q += ch; }
		const bool sam = ends(".sam") || ends(".sam.gz"), bam = ends(".bam") || ends(".bam.gz");
		if (sam || bam) {
			// io_handler(), io.c:467-575: alignments come as text from `samtools view` (-S for SAM text), secondary and
			// QC-failed records dropped (-F 768); a .gz of either goes through zcat first.  read_sam_chunk (io.c:1498-1660) takes
			// QNAME, SEQ and QUAL of every line that does not start with '@'.
			const std::string view = std::string("samtools view ") + (sam ? "-SF 768 " : "-F 768 ");
			cmd_ = ends(".gz") ? "zcat -- '" + q + "' | " + view + "-" : view + "'" + q + "'";
			if (p == "-") { err = "td_stream_run: SAM/BAM input from stdin is not supported"; return false; }
			pipe_ = popen(cmd_.c_str(), "r");
			if (!pipe_) { err = "td_stream_run: cannot start " + cmd_; return false; }
			fd_in_ = fileno(pipe_);
			sam_ = true;
			return true;
		}
		if (ends(".gz") || ends(".bz2")) {             // io_handler(), io.c:382-608: zcat / bzcat through popen
			cmd_ = std::string(ends(".gz") ? "zcat" : "bzcat") + " -- '" + q + "'";
			pipe_ = popen(cmd_.c_str(), "r");
			if (!pipe_) { err = "td_stream_run: cannot start " + cmd_; return false; }
			fd_in_ = fileno(pipe_);
			return true;
		}
		if (p == "-") { fd_in_ = 0; return true; }
		fd_ = ::open(path, O_RDONLY);
		if (fd_ < 0) { err = "td_stream_run: cannot open " + p + ": " + strerror(errno); return false; }
		if (S_ISREG(st.st_mode)) {
			map_len_ = st.st_size;
			if (map_len_ == 0) { eof_ = true; return true; }
			map_ = mmap(nullptr, (size_t)map_len_, PROT_READ, MAP_PRIVATE, fd_, 0);
			if (map_ != MAP_FAILED) {
				(void)madvise(map_, (size_t)map_len_, MADV_SEQUENTIAL);
				fasta_ = ((const char*)map_)[0] == '>';
				return true;
			}
			map_ = nullptr;                            // fall back to read()
		}
		fd_in_ = fd_;
		return true;
	}
	// the next block of whole records (nullptr at the end); *read_s += time spent waiting for bytes
	std::shared_ptr<Block> next(double* read_s, std::string& err)
	{
		if (eof_) return nullptr;
		auto b = std::make_shared<Block>();
		if (map_) {
			const char* text = (const char*)map_;
			int64_t end = map_pos_ + block_;
			end = end >= map_len_ ? map_len_ : td_next_record_start(text, map_len_, end, fasta_);
			b->data = text + map_pos_; b->len = end - map_pos_;
			(void)madvise((void*)((uintptr_t)(text + map_pos_) & ~(uintptr_t)4095), (size_t)(b->len + 4096), MADV_WILLNEED);
			map_pos_ = end;
			if (map_pos_ >= map_len_) eof_ = true;
			return b;
		}
		// pipe / stdin: fill a buffer, cut it at the last record start found in its tail, carry the rest over
		const double t0 = now_s();
		const int64_t cap = block_ + (int64_t)carry_.size() + 1;
		b->owned = (char*)malloc((size_t)cap);
		if (!b->owned) { err = "td_stream_run: out of memory"; return nullptr; }
		int64_t have = (int64_t)carry_.size();
		if (have) memcpy(b->owned, carry_.data(), (size_t)have);
		carry_.clear();
		bool end_of_input = false;
		while (have < cap - 1) {
			const ssize_t r = read(fd_in_, b->owned + have, (size_t)(cap - 1 - have));
			if (r < 0) { if (errno == EINTR) continue; err = std::string("td_stream_run: read failed: ") + strerror(errno); return nullptr; }
			if (r == 0) { end_of_input = true; break; }
			have += r;
		}
		if (end_of_input && pipe_) {
			// the decompressor's verdict: a truncated or corrupt .gz / .bz2 ends the stream early with a non-zero status -- an error,
			// not the end of the input (io_handler's pclose, io.c:382-608, ignores it; a partial set of output files helps nobody)
			const int status = pclose(pipe_);
			pipe_ = nullptr;
			if (status != 0) {
				err = "td_stream_run: `" + cmd_ + "` failed (status " + std::to_string(status) + "): truncated or corrupt input" + (sam_ ? ", or no samtools on PATH" : "");
				return nullptr;
			}
		}
		*read_s += now_s() - t0;
		if (sam_) {
			std::shared_ptr<Block> r = sam_block(b, have, end_of_input, err);
			if (r && r->len == 0) return eof_ ? nullptr : next(read_s, err);   // header lines only
			return r;
		}
		if (first_block_) { fasta_ = have > 0 && b->owned[0] == '>'; first_block_ = false; }
		int64_t cut = have;
		if (!end_of_input) {
			// the last record start in the buffer: search forward from ever earlier points of the tail
			cut = -1;
			for (int64_t back = 1 << 16; cut < 0; back *= 4) {
				int64_t from = have - back;
				if (from < 1) from = 1;
				int64_t p = td_next_record_start(b->owned, have, from, fasta_), last = -1;
				while (p < have) { last = p; p = td_next_record_start(b->owned, have, p + 1, fasta_); }
				if (last > 0) cut = last;
				else if (from == 1) { err = "td_stream_run: no record boundary within a block of input (raise block_bytes)"; return nullptr; }
			}
			carry_.assign(b->owned + cut, b->owned + have);
		} else {
			eof_ = true;
		}
		b->data = b->owned; b->len = cut;
		if (cut == 0 && eof_) return nullptr;
		return b;
	}

private:
	// Alignment text -> the four-line records the parser takes: '@' QNAME, SEQ, '+', QUAL (fields 1, 10 and 11 of every line that
	// is not a header line, read_sam_chunk, io.c:1498-1660).  Whole lines only; the rest is carried over to the next block.
	std::shared_ptr<Block> sam_block(std::shared_ptr<Block> b, int64_t have, const bool end_of_input, std::string& err)
	{
		int64_t cut = have;
		if (!end_of_input) {
			while (cut > 0 && b->owned[cut - 1] != '\n') --cut;
			if (cut == 0) { err = "td_stream_run: no line end within a block of SAM input (raise block_bytes)"; return nullptr; }
			carry_.assign(b->owned + cut, b->owned + have);
		} else {
			eof_ = true;
		}
		auto out = std::make_shared<Block>();
		out->owned = (char*)malloc((size_t)cut + 8);      // a record shrinks: 11+ tab-separated fields against 3 of them and 5 bytes
		if (!out->owned) { err = "td_stream_run: out of memory"; return nullptr; }
		char* o = out->owned;
		const char* t = b->owned;
		const char* const end = t + cut;
		auto blank = [](const char ch) { return ch == ' ' || ch == '\t' || ch == '\r' || ch == '\v' || ch == '\f'; };
		while (t < end) {
			const char* nl = (const char*)memchr(t, '\n', (size_t)(end - t));
			const char* le = nl ? nl : end;
			if (le > t && t[0] != '@') {
				const char* f[12];
				int nf = 0;
				const char* c = t;
				f[nf++] = c;
				while (c < le && nf < 12) { if (blank(*c)) f[nf++] = c + 1; ++c; }
				if (nf < 11) { err = "td_stream_run: SAM line with fewer than 11 fields: " + std::string(t, (size_t)std::min<int64_t>(le - t, 60)); return nullptr; }
				auto field_end = [&](const char* s) { while (s < le && !blank(*s)) ++s; return s; };
				const char* n1 = field_end(f[0]);
				const char* s1 = field_end(f[9]);
				const char* q1 = field_end(f[10]);
				if (q1 - f[10] != s1 - f[9]) { err = "td_stream_run: SAM record without one quality per base: " + std::string(f[0], (size_t)(n1 - f[0])); return nullptr; }
				*o++ = '@'; memcpy(o, f[0], (size_t)(n1 - f[0])); o += n1 - f[0]; *o++ = '\n';
				memcpy(o, f[9], (size_t)(s1 - f[9])); o += s1 - f[9]; *o++ = '\n';
				*o++ = '+'; *o++ = '\n';
				memcpy(o, f[10], (size_t)(q1 - f[10])); o += q1 - f[10]; *o++ = '\n';
			}
			t = nl ? nl + 1 : end;
		}
		out->data = out->owned; out->len = o - out->owned;
		return out;
	}

	int fd_ = -1, fd_in_ = -1;
	std::string cmd_;
	bool sam_ = false;
	FILE* pipe_ = nullptr;
	void* map_ = nullptr;
	int64_t map_len_ = 0, map_pos_ = 0, block_ = 0;
	bool eof_ = false, fasta_ = false, first_block_ = true;
	std::vector<char> carry_;
};

// ---- a decode batch: exactly batch_reads records (fewer at the end), device-ready buffers in page-locked memory ----
struct Piece {                      // records [lo, hi) of one parsed chunk of a block
	std::shared_ptr<Block> blk;
	std::shared_ptr<std::vector<TdRec>> recs;
	int64_t lo = 0, hi = 0;
	int64_t first = 0;              // index of record lo within the batch
};

struct Batch {
	int64_t n = 0, n_bases = 0;
	uint8_t* codes = nullptr;  size_t cap_codes = 0;
	uint8_t* seq_out = nullptr; size_t cap_seq = 0;
	int64_t* offs = nullptr;
	td_read_result* res = nullptr;
	std::vector<Piece> pieces;
	int64_t ticket = 0;
	bool last = false;
	int64_t cap_reads = 0;      // offs / res hold this many reads
	const uint8_t* seq_src = nullptr;   // what the writer prints: seq_out, or codes for a file that is not decoded (run_rna_dust)
};

// Page-locked batch buffers outlive a run: page-locking runs at about 1 GB/s, which is most of what a short file costs.  The
// buffers of the last run (up to 1 GiB) wait here for the next one of the same process; td_stream_release frees them.
std::mutex g_cache_mu;
std::vector<Batch*> g_cache;
const size_t kCacheBytes = (size_t)1 << 30;

size_t batch_bytes(const Batch* b)
{
	return b->cap_codes + b->cap_seq + (size_t)b->cap_reads * (sizeof(int64_t) + sizeof(td_read_result));
}

// batch buffers: page-locked for the device; plain memory for a parse-only run (no GPU runtime needed then)
void* buf_alloc(bool dry, size_t bytes) { return dry ? malloc(bytes ? bytes : 1) : td_host_alloc(bytes); }
void buf_free(bool dry, void* p) { if (dry) free(p); else td_host_free(p); }

bool grow_buf(bool dry, uint8_t** p, size_t* cap, size_t need, size_t keep, size_t hint)
{
	if (*cap >= need) return true;
	size_t want = need + need / 4 + 4096;
	if (want < hint) want = hint;          // what a whole batch is expected to need: allocated once, reused by later batches
	uint8_t* q = (uint8_t*)buf_alloc(dry, want);
	if (!q) return false;
	if (*p && keep) memcpy(q, *p, keep);
	if (*p) buf_free(dry, *p);
	*p = q; *cap = want;
	return true;
}

// a growing byte buffer without the zero-fill of std::string::resize
struct Bytes {
	char* p = nullptr;
	size_t n = 0, cap = 0;
	Bytes() = default;
	Bytes(const Bytes&) = delete;
	Bytes& operator=(const Bytes&) = delete;
	Bytes(Bytes&& o) noexcept : p(o.p), n(o.n), cap(o.cap) { o.p = nullptr; o.n = o.cap = 0; }
	~Bytes() { free(p); }
	char* room(size_t more)
	{
		if (n + more > cap) {
			size_t c = cap ? cap * 2 : (size_t)1 << 16;
			while (c < n + more) c *= 2;
			p = (char*)realloc(p, c);
			cap = c;
		}
		return p + n;
	}
};

// "%d"
inline int put_int(char* w, int v)
{
	char tmp[16];
	int k = 0;
	unsigned u = v < 0 ? 0u - (unsigned)v : (unsigned)v;
	do { tmp[k++] = (char)('0' + u % 10); u /= 10; } while (u);
	int o = 0;
	if (v < 0) w[o++] = '-';
	while (k) w[o++] = tmp[--k];
	return o;
}

// "%0.2f" of a float, digit for digit what printf prints: the float times 100 is exact in double (24 x 7 significant bits),
// nearbyint rounds it half-to-even like printf rounds the exact decimal expansion; anything unusual goes to snprintf
inline int put_q(char* w, float q)
{
	const double x = (double)q * 100.0;
	if (!(x >= 0.0 && x < 1.0e15) || (x == 0.0 && std::signbit(q))) return snprintf(w, 48, "%0.2f", (double)q);   // (at most 42 characters)
	unsigned long long u = (unsigned long long)nearbyint(x);
	char tmp[24];
	int k = 0;
	tmp[k++] = (char)('0' + u % 10); u /= 10;
	tmp[k++] = (char)('0' + u % 10); u /= 10;
	tmp[k++] = '.';
	do { tmp[k++] = (char)('0' + u % 10); u /= 10; } while (u);
	int o = 0;
	while (k) w[o++] = tmp[--k];
	return o;
}

// formats the records of one piece sub-range into one buffer per output file (print_all(), io.c:917-1001)
struct OutBufs { std::vector<Bytes> file; };

// The rewritten sequence of a read is base codes 0..4 with removed positions as 65: the stretches of kept bases become the
// records.  Sixteen bytes at a time where the host has SSE2 (every x86-64 has): the write stage is the slower host stage of the
// pipeline, and these two loops were half of its formatting time.
#if defined(__SSE2__)
// number of leading bytes of s[0, n) that are >= 5 (GE = true) / < 5 (GE = false)
template <bool GE>
inline int64_t span_of(const uint8_t* s, int64_t n)
{
	int64_t g = 0;
	const __m128i five = _mm_set1_epi8(5);
	for (; g + 16 <= n; g += 16) {
		const __m128i v = _mm_loadu_si128((const __m128i*)(s + g));
		const unsigned ge = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_max_epu8(v, five), v));   // bit k: s[g + k] >= 5
		const unsigned stop = GE ? (~ge & 0xFFFFu) : ge;   // first byte that ends the span
		if (stop) return g + __builtin_ctz(stop);
	}
	while (g < n && ((s[g] >= 5) == GE)) g++;
	return g;
}
// w[k] = "ACGTN"[s[k]] for codes 0..4
inline void codes_to_text(char* w, const uint8_t* s, size_t n)
{
	size_t k = 0;
	const __m128i a = _mm_set1_epi8('A'), c1 = _mm_set1_epi8(1), c2 = _mm_set1_epi8(2), c3 = _mm_set1_epi8(3), c4 = _mm_set1_epi8(4);
	const __m128i d1 = _mm_set1_epi8('C' - 'A'), d2 = _mm_set1_epi8('G' - 'A'), d3 = _mm_set1_epi8('T' - 'A'), d4 = _mm_set1_epi8('N' - 'A');
	for (; k + 16 <= n; k += 16) {
		const __m128i v = _mm_loadu_si128((const __m128i*)(s + k));
		__m128i o = a;
		o = _mm_add_epi8(o, _mm_and_si128(_mm_cmpeq_epi8(v, c1), d1));
		o = _mm_add_epi8(o, _mm_and_si128(_mm_cmpeq_epi8(v, c2), d2));
		o = _mm_add_epi8(o, _mm_and_si128(_mm_cmpeq_epi8(v, c3), d3));
		o = _mm_add_epi8(o, _mm_and_si128(_mm_cmpeq_epi8(v, c4), d4));
		_mm_storeu_si128((__m128i*)(w + k), o);
	}
	static const char alphabet[] = "ACGTNN";
	for (; k < n; k++) w[k] = alphabet[s[k]];
}
#else
template <bool GE>
inline int64_t span_of(const uint8_t* s, int64_t n) { int64_t g = 0; while (g < n && ((s[g] >= 5) == GE)) g++; return g; }
inline void codes_to_text(char* w, const uint8_t* s, size_t n) { static const char alphabet[] = "ACGTNN"; for (size_t k = 0; k < n; k++) w[k] = alphabet[s[k]]; }
#endif

// ctype / cbar: read_type and barcode of the record as the controller combines them over the input files of a run
// (barcode_hmm.c:329-351; NULL: this file's own); file_base: index of this input file's first output file (io.c:917-1001: c)
void format_records(const Batch& b, const Piece& pc, int64_t lo, int64_t hi, int num_alternatives, OutBufs& out,
                    const int32_t* ctype = nullptr, const int32_t* cbar = nullptr, size_t file_base = 0)
{
	const char* text = pc.blk->data;
	const std::vector<TdRec>& recs = *pc.recs;
	char head[96];
	const size_t n_files = out.file.size();
	for (int64_t r = lo; r < hi; r++) {
		const TdRec& rec = recs[(size_t)r];
		const int64_t i = pc.first + (r - pc.lo);             // index in the batch
		const td_read_result& rr = b.res[i];
		size_t f;                                              // io.c:923-934
		const int32_t rtype = ctype ? ctype[i] : rr.read_type, rbar = cbar ? cbar[i] : rr.barcode;
		if (rtype == TD_EXTRACT_SUCCESS) f = (rbar != -1) ? (size_t)(rbar & 0xFF) : 0;
		else f = (size_t)num_alternatives - 1;
		f += file_base;
		const uint8_t* s = (b.seq_src ? b.seq_src : b.seq_out) + b.offs[i];
		const int64_t len = b.offs[i + 1] - b.offs[i];
		const char* q = rec.qual_off >= 0 ? text + rec.qual_off : nullptr;
		int head_len = -1;
		int64_t g = 0;
		while (g < len) {
			g += span_of<true>(s + g, len - g);                // removed positions (65) split the read into records
			const int64_t g0 = g;
			g += span_of<false>(s + g, len - g);
			if (g == g0) break;
			if (f < n_files) {                                 // io.c:955-975: "@<name>[;FP:%d];RQ:%0.2f"
				if (head_len < 0) {
					int k = 0;
					if (rr.fingerprint != -1) { memcpy(head, ";FP:", 4); k = 4 + put_int(head + 4, rr.fingerprint); }
					memcpy(head + k, ";RQ:", 4); k += 4;
					k += put_q(head + k, rr.mapq);
					head[k++] = '\n';
					head_len = k;
				}
				const size_t run = (size_t)(g - g0);
				Bytes& o = out.file[f];
				const size_t total = 1 + (size_t)rec.name_len + (size_t)head_len + run + 3 + run + 1;
				char* w = o.room(total);
				*w++ = '@';
				memcpy(w, text + rec.name_off, (size_t)rec.name_len); w += rec.name_len;
				memcpy(w, head, (size_t)head_len); w += head_len;
				codes_to_text(w, s + g0, run);
				w += run;
				*w++ = '\n'; *w++ = '+'; *w++ = '\n';
				if (q) memcpy(w, q + g0, run); else memset(w, '.', run);
				w += run;
				*w++ = '\n';
				o.n += total;
			}
			f += (size_t)num_alternatives;
		}
	}
}

struct Pipeline {
	td_ctx* ctx = nullptr;
	const td_arch* arch = nullptr;
	td_stream_opts o{};
	td_stream_stats st{};
	Source src;
	std::string err;
	std::mutex err_mu;
	std::unique_ptr<Pool> parse_pool, write_pool;
	// Appends run beside the formatting of the next batch: every append is a pwrite at an offset fixed when the batch was
	// formatted, so the order in which they reach the files is free.  Two sets of formatted text; a set is reused once its
	// appends have been written (append_wait).
	struct Wr { size_t file, k0, k1; int64_t at; };
	struct AppendJob { int set; std::vector<Wr> wr; };
	std::unique_ptr<Pool> append_pool;
	std::thread appender;
	std::mutex ap_mu;
	std::condition_variable ap_cv;
	std::deque<AppendJob> ap_jobs;
	bool ap_busy[2] = { false, false }, ap_stop = false, ap_started = false;
	int ap_errno = 0, ap_set = 0;
	std::vector<OutBufs> ap_bufs[2];
	std::unique_ptr<Queue<Batch*>> ready, done, free_list;
	std::vector<Batch*> all;
	std::vector<int> fds;
	std::vector<int64_t> file_off;
	int num_alternatives = 2;
	bool dry = false;
	uint64_t fnv = 1469598103934665603ULL;
	double dbg_pass1 = 0, dbg_grow = 0, dbg_encode = 0, dbg_format = 0, dbg_pwrite = 0;
	size_t batch_hint = 0;
	std::mutex all_mu;
	std::condition_variable hint_cv;
	bool hint_ready = false, stop_alloc = false;

	void fail(const std::string& m)
	{
		{ std::lock_guard<std::mutex> lk(err_mu); if (err.empty()) err = m; }
		ready->abort(); done->abort(); free_list->abort();
	}
	bool failed() { std::lock_guard<std::mutex> lk(err_mu); return !err.empty(); }

	Batch* new_batch()
	{
		if (!dry) {   // one the last run left behind?
			std::lock_guard<std::mutex> lk(g_cache_mu);
			for (size_t k = 0; k < g_cache.size(); k++)
				if (g_cache[k]->cap_reads >= o.batch_reads) {
					Batch* b = g_cache[k];
					g_cache.erase(g_cache.begin() + (long)k);
					b->offs[0] = 0;
					std::lock_guard<std::mutex> lk2(all_mu);
					all.push_back(b);
					return b;
				}
		}
		Batch* b = new Batch();
		b->cap_reads = o.batch_reads;
		b->offs = (int64_t*)buf_alloc(dry, sizeof(int64_t) * ((size_t)o.batch_reads + 1));
		b->res = (td_read_result*)buf_alloc(dry, sizeof(td_read_result) * (size_t)o.batch_reads);
		if (!b->offs || !b->res) { buf_free(dry, b->offs); buf_free(dry, b->res); delete b; return nullptr; }
		b->offs[0] = 0;
		{ std::lock_guard<std::mutex> lk(all_mu); all.push_back(b); }
		return b;
	}

	// The batch buffers are page-locked, and page-locking runs at about 1 GB/s: the pipeline starts with two batches and this
	// thread adds the others, sized for a whole batch (known after the first block), while the first ones are already at work.
	void allocator(int n_more)
	{
		size_t hint = 0;
		{
			std::unique_lock<std::mutex> lk(all_mu);
			hint_cv.wait(lk, [&] { return hint_ready || stop_alloc; });
			if (stop_alloc) return;
			hint = batch_hint;
		}
		for (int k = 0; k < n_more; k++) {
			{ std::lock_guard<std::mutex> lk(all_mu); if (stop_alloc) return; }
			Batch* b = new_batch();
			if (!b || !grow_buf(dry, &b->codes, &b->cap_codes, hint, 0, hint) || !grow_buf(dry, &b->seq_out, &b->cap_seq, hint, 0, hint)) return;   // (the pipeline runs with what it has)
			if (!free_list->push(b)) return;
		}
	}

	// ---- stage 1: read / map a block, find its records, hand them out to batches, base-code them ----
	void producer()
	{
		Batch* cur = nullptr;
		const int P = parse_pool->size();
		for (;;) {
			std::string e;
			std::shared_ptr<Block> blk = src.next(&st.read_s, e);
			if (!blk) { if (!e.empty()) { fail(e); return; } break; }
			const double t0 = now_s();
			st.bytes_in += blk->len;
			// records: P chunks cut at record starts, each parsed by the reference's line state machine
			const bool fasta = blk->len > 0 && blk->data[0] == '>';
			std::vector<int64_t> cut(1, 0);
			const int nchunk_want = blk->len < (1 << 20) ? 1 : P * 4;
			for (int t = 1; t < nchunk_want; t++) {
				const int64_t p = td_next_record_start(blk->data, blk->len, blk->len / nchunk_want * t, fasta);
				if (p < blk->len && p > cut.back()) cut.push_back(p);
			}
			cut.push_back(blk->len);
			const int nchunk = (int)cut.size() - 1;
			std::vector<std::shared_ptr<std::vector<TdRec>>> recs((size_t)nchunk);
			for (auto& r : recs) r = std::make_shared<std::vector<TdRec>>();
			std::vector<int64_t> bad((size_t)nchunk, -1);
			parse_pool->run(nchunk, [&](int64_t k) {
				std::vector<TdRec>& v = *recs[(size_t)k];
				v.reserve((size_t)((cut[(size_t)k + 1] - cut[(size_t)k]) / 200 + 16));
				td_parse_range(blk->data, cut[(size_t)k], cut[(size_t)k + 1], v);
				// "Length of sequence and base qualities differ" ends the reference's run (io.c:1776-1781)
				for (size_t i = 0; i < v.size(); i++)
					if (v[i].qual_off >= 0 && v[i].qual_len != (v[i].seq_off >= 0 ? v[i].seq_len : 0)) { bad[(size_t)k] = (int64_t)i; break; }
			});
			dbg_pass1 += now_s() - t0;
			for (int k = 0; k < nchunk; k++)
				if (bad[(size_t)k] >= 0) {
					const TdRec& q = (*recs[(size_t)k])[(size_t)bad[(size_t)k]];
					char msg[256];
					snprintf(msg, sizeof msg, "td_stream_run: record \"%.*s\": sequence has %d characters, base qualities %d",
					         q.name_len < 60 ? q.name_len : 60, blk->data + q.name_off, q.seq_off >= 0 ? q.seq_len : 0, q.qual_len);
					fail(msg);
					return;
				}
			{   // bases a full batch of reads like this block's will hold (+6 %)
				int64_t nrec = 0, nb = 0;
				for (auto& r : recs) { nrec += (int64_t)r->size(); for (const TdRec& q : *r) nb += q.seq_off >= 0 ? q.seq_len : 0; }
				if (nrec > 0) {
					const size_t h = (size_t)((double)nb / (double)nrec * (double)o.batch_reads * 1.06) + 4096;
					std::lock_guard<std::mutex> lk(all_mu);
					if (h > batch_hint) batch_hint = h;
					hint_ready = true;
					hint_cv.notify_all();
				}
			}
			// hand the records out to batches of exactly batch_reads, in order (offsets by a running sum); the pieces handed out
			// are base-coded (init_nuc_code) in parallel and full batches passed on whenever two are waiting, and before this
			// thread might have to wait for a free batch
			struct Job { Batch* b; size_t piece; };
			std::vector<Job> jobs;
			std::vector<Batch*> full;
			double t_seg = t0;
			auto flush = [&]() -> bool {
				const double tf0 = now_s();
				std::vector<Batch*> touched = full;
				if (cur) touched.push_back(cur);
				for (Batch* b : touched) {
					// page-locked room for the codes going in and the rewritten sequences coming back (bases of earlier pieces of
					// a batch are already encoded: keep them)
					bool mine = false;
					size_t keep = 0;
					for (const Job& j : jobs) if (j.b == b) { keep = (size_t)b->offs[b->pieces[j.piece].first]; mine = true; break; }
					if (!mine) continue;
					if (!grow_buf(dry, &b->codes, &b->cap_codes, (size_t)b->n_bases + 1, keep, batch_hint) ||
					    !grow_buf(dry, &b->seq_out, &b->cap_seq, (size_t)b->n_bases + 1, 0, batch_hint)) { fail("td_stream_run: page-locked memory exhausted"); return false; }
				}
				dbg_grow += now_s() - tf0;
				const double tf1 = now_s();
				struct Sub { Batch* b; size_t piece; int64_t lo, hi; };
				std::vector<Sub> subs;
				for (const Job& j : jobs) {
					const Piece& pc = j.b->pieces[j.piece];
					for (int64_t a = pc.lo; a < pc.hi; a += 16384) subs.push_back(Sub{ j.b, j.piece, a, std::min<int64_t>(a + 16384, pc.hi) });
				}
				parse_pool->run((int64_t)subs.size(), [&](int64_t k) {
					const Sub& sb = subs[(size_t)k];
					const Piece& pc = sb.b->pieces[sb.piece];
					const std::vector<TdRec>& v = *pc.recs;
					for (int64_t r = sb.lo; r < sb.hi; r++) {
						const TdRec& q = v[(size_t)r];
						if (q.seq_off < 0) continue;
						uint8_t* dst = sb.b->codes + sb.b->offs[pc.first + (r - pc.lo)];
						const unsigned char* sq = (const unsigned char*)pc.blk->data + q.seq_off;
						td_encode_bases(sq, dst, q.seq_len);
					}
				});
				jobs.clear();
				dbg_encode += now_s() - tf1;
				st.parse_s += now_s() - t_seg;                          // (time spent waiting for a free batch is not parsing)
				for (Batch* b : full) if (!ready->push(b)) return false;
				full.clear();
				t_seg = now_s();
				return true;
			};
			for (int k = 0; k < nchunk; k++) {
				const std::vector<TdRec>& v = *recs[(size_t)k];
				int64_t lo = 0;
				while (lo < (int64_t)v.size()) {
					if (!cur) {
						if (!full.empty() && !flush()) return;
						st.parse_s += now_s() - t_seg;
						if (!free_list->pop(cur)) return;                  // aborted
						t_seg = now_s();
						cur->n = 0; cur->n_bases = 0; cur->pieces.clear(); cur->last = false; cur->ticket = 0;
					}
					const int64_t take = std::min<int64_t>((int64_t)v.size() - lo, (int64_t)o.batch_reads - cur->n);
					Piece pc; pc.blk = blk; pc.recs = recs[(size_t)k]; pc.lo = lo; pc.hi = lo + take; pc.first = cur->n;
					int64_t total = cur->n_bases;
					for (int64_t r = lo; r < lo + take; r++) {
						total += v[(size_t)r].seq_off >= 0 ? v[(size_t)r].seq_len : 0;
						cur->offs[cur->n + (r - lo) + 1] = total;
					}
					cur->n += take; cur->n_bases = total;
					cur->pieces.push_back(pc);
					jobs.push_back(Job{ cur, cur->pieces.size() - 1 });
					lo += take;
					if (cur->n == o.batch_reads) { full.push_back(cur); cur = nullptr; }
				}
			}
			if (!flush()) return;
		}
		if (getenv("TD_STREAM_DEBUG")) fprintf(stderr, "td_stream: records %.3f s, buffers %.3f s, base codes %.3f s (parse stage %.3f s)\n", dbg_pass1, dbg_grow, dbg_encode, st.parse_s);
		if (cur && cur->n > 0) { cur->last = true; if (!ready->push(cur)) return; }
		else if (cur) free_list->push(cur);
		ready->close();
	}

	void append_loop()
	{
		for (;;) {
			AppendJob job;
			{
				std::unique_lock<std::mutex> lk(ap_mu);
				ap_cv.wait(lk, [&] { return ap_stop || !ap_jobs.empty(); });
				if (ap_jobs.empty()) return;
				job = std::move(ap_jobs.front());
				ap_jobs.pop_front();
			}
			const double t0 = now_s();
			const std::vector<OutBufs>& bufs = ap_bufs[job.set];
			std::vector<int> wrc(job.wr.size(), 0);
			append_pool->run((int64_t)job.wr.size(), [&](int64_t t) {
				const Wr& w = job.wr[(size_t)t];
				int64_t at = w.at;
				for (size_t k = w.k0; k < w.k1; k++) {
					const Bytes& sb = bufs[k].file[w.file];
					size_t off = 0;
					while (off < sb.n) {
						const ssize_t r = pwrite(fds[w.file], sb.p + off, sb.n - off, (off_t)(at + (int64_t)off));
						if (r < 0) { if (errno == EINTR) continue; wrc[(size_t)t] = errno ? errno : EIO; return; }
						off += (size_t)r;
					}
					at += (int64_t)sb.n;
				}
			});
			std::lock_guard<std::mutex> lk(ap_mu);
			dbg_pwrite += now_s() - t0;
			for (int e : wrc) if (e && !ap_errno) ap_errno = e;
			ap_busy[job.set] = false;
			ap_cv.notify_all();
		}
	}
	// waits until the appends of `set` (-1: of every set) are in the files; false if one of them failed
	bool append_wait(int set)
	{
		std::unique_lock<std::mutex> lk(ap_mu);
		ap_cv.wait(lk, [&] { return set < 0 ? (!ap_busy[0] && !ap_busy[1]) : !ap_busy[set]; });
		if (ap_errno) { const int e = ap_errno; lk.unlock(); fail(std::string("td_stream_run: write failed: ") + strerror(e)); return false; }
		return true;
	}
	void append_stop()
	{
		{ std::lock_guard<std::mutex> lk(ap_mu); ap_stop = true; }
		ap_cv.notify_all();
		if (appender.joinable()) appender.join();
	}
	~Pipeline() { append_stop(); }

	// format the records of one batch per output file (on the write pool) and hand the appends (in input order per file) to the
	// appender thread; append_wait(-1) before the files are closed
	bool write_batch(Batch* b, const int32_t* ctype, const int32_t* cbar, size_t file_base)
	{
		if (!ap_started) {
			ap_started = true;
			append_pool.reset(new Pool(write_pool->size()));
			appender = std::thread([this] { append_loop(); });
		}
		const int set = ap_set;
		ap_set ^= 1;
		if (!append_wait(set)) return false;
		std::vector<OutBufs>& bufs = ap_bufs[set];
		const double t0 = now_s();
		const int W = write_pool->size();
		struct Sub { size_t piece; int64_t lo, hi; };
		std::vector<Sub> subs;
		const int64_t step = std::max<int64_t>(4096, (b->n + W * 4 - 1) / (W * 4));
		for (size_t p = 0; p < b->pieces.size(); p++)
			for (int64_t a = b->pieces[p].lo; a < b->pieces[p].hi; a += step)
				subs.push_back(Sub{ p, a, std::min<int64_t>(a + step, b->pieces[p].hi) });
		if (bufs.size() < subs.size()) bufs.resize(subs.size());
		for (size_t k = 0; k < subs.size(); k++) {
			if (bufs[k].file.size() != fds.size()) bufs[k].file.resize(fds.size());
			for (auto& s : bufs[k].file) s.n = 0;
		}
		write_pool->run((int64_t)subs.size(), [&](int64_t k) {
			const Sub& sb = subs[(size_t)k];
			format_records(*b, b->pieces[sb.piece], sb.lo, sb.hi, num_alternatives, bufs[(size_t)k], ctype, cbar, file_base);
		});
		dbg_format += now_s() - t0;
		// Appends.  Buffered writes to one file serialise on its inode lock (8 threads on one file: 8 GB/s; one thread on each of
		// nine files: 56 GB/s, tools/ubench/file_write.cpp), so a file gets one task that appends its share of every sub-range
		// in order -- or a few tasks over runs of sub-ranges when it takes most of the bytes (no barcode segment: two files)
		std::vector<Wr> wr;
		int64_t total_bytes = 0;
		std::vector<int64_t> per_file(fds.size(), 0);
		for (size_t f = 0; f < fds.size(); f++) {
			for (size_t k = 0; k < subs.size(); k++) per_file[f] += (int64_t)bufs[k].file[f].n;
			total_bytes += per_file[f];
		}
		for (size_t f = 0; f < fds.size(); f++) {
			if (!per_file[f]) continue;
			int parts = (int)((double)per_file[f] / (double)total_bytes * (double)W + 0.5);
			if (parts > 4) parts = 4;
			if (parts < 1) parts = 1;
			const int64_t target = (per_file[f] + parts - 1) / parts;
			int64_t at = file_off[f], acc = 0;
			size_t k0 = 0;
			for (size_t k = 0; k < subs.size(); k++) {
				acc += (int64_t)bufs[k].file[f].n;
				if (acc >= target || k + 1 == subs.size()) {
					if (acc > 0) wr.push_back(Wr{ f, k0, k + 1, at });
					at += acc; acc = 0; k0 = k + 1;
				}
			}
			file_off[f] += per_file[f];
			st.bytes_out += per_file[f];
		}
		{
			std::lock_guard<std::mutex> lk(ap_mu);
			ap_busy[set] = true;
			ap_jobs.push_back(AppendJob{ set, std::move(wr) });
		}
		ap_cv.notify_all();
		static const bool sync_appends = getenv("TD_STREAM_SYNC_APPENDS") && atoi(getenv("TD_STREAM_SYNC_APPENDS")) != 0;   // A/B: round 3's order
		if (sync_appends && !append_wait(set)) return false;
		return true;
	}

	// ---- stage 3: format the records per output file and append them in input order ----
	void consumer()
	{
		Batch* b = nullptr;
		while (done->pop(b)) {
			const double t0 = now_s();
			if (!dry) {
				if (!write_batch(b, nullptr, nullptr, 0)) return;
			} else {
				// parse-only run: a checksum over what would have gone to the device (lengths and codes, in order)
				for (int64_t i = 0; i < b->n; i++) {
					const int64_t l = b->offs[i + 1] - b->offs[i];
					fnv = (fnv ^ (uint64_t)l) * 1099511628211ULL;
					const uint8_t* s = b->codes + b->offs[i];
					for (int64_t j = 0; j < l; j++) fnv = (fnv ^ s[j]) * 1099511628211ULL;
				}
			}
			st.n_reads += b->n; st.n_batches++;
			b->pieces.clear();                 // releases the blocks
			st.write_s += now_s() - t0;
			if (!free_list->push(b)) return;
		}
		{
			const double t0 = now_s();
			const bool ok = append_wait(-1);
			st.write_s += now_s() - t0;
			if (!ok) return;
		}
		if (getenv("TD_STREAM_DEBUG")) fprintf(stderr, "td_stream: formatting %.3f s, appends %.3f s beside it (write stage %.3f s)\n", dbg_format, dbg_pwrite, st.write_s);
	}
};

// the batches of a finished run: kept for the next one (page-locked, within the cache's size) or freed
static void release_batches(Pipeline& p, bool keep)
{
	std::lock_guard<std::mutex> lk(g_cache_mu);
	size_t held = 0;
	for (const Batch* q : g_cache) held += batch_bytes(q);
	for (Batch* q : p.all) {
		if (!p.dry && keep && q->codes && held + batch_bytes(q) <= kCacheBytes) {
			q->pieces.clear(); q->n = 0; q->n_bases = 0; q->ticket = 0; q->seq_src = nullptr;
			held += batch_bytes(q);
			g_cache.push_back(q);
			continue;
		}
		buf_free(p.dry, q->codes); buf_free(p.dry, q->seq_out); buf_free(p.dry, q->offs); buf_free(p.dry, q->res);
		delete q;
	}
	p.all.clear();
}

} // namespace

extern "C" void td_stream_release(void)
{
	std::lock_guard<std::mutex> lk(g_cache_mu);
	for (Batch* q : g_cache) { td_host_free(q->codes); td_host_free(q->seq_out); td_host_free(q->offs); td_host_free(q->res); delete q; }
	g_cache.clear();
}

// the writer's "%0.2f" (tests compare it with printf digit for digit); returns the number of characters written to buf[48]
extern "C" int td_format_q(float q, char* buf) { return put_q(buf, q); }

extern "C" int td_stream_run(td_ctx* ctx, const char* in_path, const td_arch* arch, const char* out_prefix,
                             const td_stream_opts* opts, td_stream_stats* stats)
{
	if (!in_path) { td_io_set_error("td_stream_run: NULL input path"); return TD_FAIL; }
	if (ctx && (!arch || !out_prefix)) { td_io_set_error("td_stream_run: a decoding run needs the architecture and an output prefix"); return TD_FAIL; }
	Pipeline p;
	p.ctx = ctx; p.arch = arch; p.dry = ctx == nullptr;
	if (opts) p.o = *opts;
	if (p.o.batch_reads <= 0) {
		// The reference reads 1 000 001 records at a time (param->num_query, barcode_hmm.c:172).  Per-read results do not depend on
		// how a file is cut into batches -- except through the -ref artifact filter, whose per-thread read ranges are taken over a
		// batch: with a filter set the batches are the reference's, without one they are 2^18 reads (one tile per wave slot of
		// the device; a quarter of the page-locked memory and a pipeline that fills four times sooner).
		int32_t art = 0;
		if (ctx) (void)td_get_option(ctx, "artifacts_active", &art);
		p.o.batch_reads = (!ctx || art) ? 1000001 : (1 << 18);
	}
	if (p.o.block_bytes <= 0) p.o.block_bytes = (int64_t)64 << 20;
	if (p.o.block_bytes < 4096) p.o.block_bytes = 4096;
	int hw = (int)std::thread::hardware_concurrency();
	if (hw < 1) hw = 1;
	if (p.o.n_threads <= 0) p.o.n_threads = hw >= 16 ? 8 : (hw >= 4 ? hw / 2 : 1);
	if (p.o.n_threads > 32) p.o.n_threads = 32;
	int32_t depth = 3;
	if (ctx) (void)td_get_option(ctx, "pipeline_depth", &depth);
	const double t_start = now_s();
	std::string err;
	if (!p.src.open(in_path, p.o.block_bytes, err)) { td_io_set_error(err); return TD_FAIL; }
	if (!p.dry) {
		std::vector<std::string> names;
		td_writer_file_names(out_prefix, arch, names, &p.num_alternatives);
		for (auto& nm : names) {
			const int fd = open(nm.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
			if (fd < 0) { for (int g : p.fds) close(g); td_io_set_error("td_stream_run: cannot create " + nm + ": " + strerror(errno)); return TD_FAIL; }
			p.fds.push_back(fd);
		}
		p.file_off.assign(p.fds.size(), 0);
	}
	p.parse_pool.reset(new Pool(p.o.n_threads));
	p.write_pool.reset(new Pool(p.o.n_threads));
	// batches: `depth` on the device, one being filled, one being written, one spare on either side
	const int n_batches = depth + 4;
	p.ready.reset(new Queue<Batch*>((size_t)n_batches));
	p.done.reset(new Queue<Batch*>((size_t)n_batches));
	p.free_list.reset(new Queue<Batch*>((size_t)n_batches));
	bool ok = true;
	for (int k = 0; k < 2 && ok; k++) {
		Batch* b = p.new_batch();
		if (!b) { ok = false; break; }
		p.free_list->push(b);
	}
	int rc = TD_OK;
	if (!ok) { p.fail("td_stream_run: page-locked memory exhausted"); rc = TD_FAIL; }
	std::thread t_alloc([&] { p.allocator(n_batches - 2); });
	std::thread t_prod([&] { p.producer(); });
	std::thread t_cons([&] { p.consumer(); });
	// ---- stage 2, on the caller's thread (a context is driven from one thread): submit, keep `depth` in flight, wait in order ----
	std::deque<Batch*> flying;
	auto retire = [&]() -> bool {
		Batch* b = flying.front();
		flying.pop_front();
		const double t0 = now_s();
		if (ctx && td_wait(ctx, b->ticket) != TD_OK) { p.fail(std::string("td_stream_run: ") + td_last_error(ctx)); return false; }
		p.st.decode_s += now_s() - t0;
		return p.done->push(b);
	};
	// With nothing ready to submit and batches in flight the oldest one is retired instead of waiting: the reader may be waiting
	// for exactly that batch's buffers.  (The allocator thread "runs with what it has" when page-locking fails part-way -- a
	// memlock limit, a container -- and with fewer than depth + 2 batches in all a loop that only retires at full depth leaves
	// the reader waiting for a free batch, this thread for a ready one and the writer for a finished one, forever.)
	while (rc == TD_OK) {
		Batch* b = nullptr;
		const int got = flying.empty() ? (p.ready->pop(b) ? 1 : -1) : p.ready->try_pop(b);
		if (got < 0) break;
		if (got == 0) { if (!retire()) break; continue; }
		if ((int)flying.size() >= depth && !retire()) break;
		const double t0 = now_s();
		if (ctx && td_submit(ctx, b->codes, 0, b->offs, b->n, TD_MODE_GET_LABEL, b->res, nullptr, b->seq_out, &b->ticket) != TD_OK) {
			p.fail(std::string("td_stream_run: ") + td_last_error(ctx));
			break;
		}
		p.st.decode_s += now_s() - t0;
		flying.push_back(b);
	}
	while (!flying.empty() && !p.failed()) if (!retire()) break;
	if (p.failed() && ctx) for (Batch* f : flying) (void)td_wait(ctx, f->ticket);   // nothing of ours may stay queued on the device
	p.done->close();
	t_prod.join();
	t_cons.join();
	p.append_stop();                       // (after a failure appends may still be queued: none may outlive the descriptors)
	{ std::lock_guard<std::mutex> lk(p.all_mu); p.stop_alloc = true; }
	p.hint_cv.notify_all();
	p.free_list->abort();          // (an allocator waiting to hand over a batch)
	t_alloc.join();
	for (int fd : p.fds) if (close(fd) != 0 && !p.failed()) p.fail(std::string("td_stream_run: close failed: ") + strerror(errno));
	release_batches(p, !p.failed());
	p.st.wall_s = now_s() - t_start;
	p.st.codes_fnv = p.dry ? p.fnv : 0;
	if (stats) *stats = p.st;
	if (p.failed()) { td_io_set_error(p.err); return TD_FAIL; }
	return rc;
}

// ---------------------------------------------------------------------------------------------------------
// Several input files of one run, several devices: the controller's loop for paired / three-read data
// (hmm_controller_multiple, src/barcode_hmm.c:244-385).
// ---------------------------------------------------------------------------------------------------------
namespace {

// dust_sequences(), src/barcode_hmm.c:2407-2467, on a read that nothing was removed from (run_rna_dust: the read as it was read)
bool rna_dust_low(const uint8_t* s, int64_t len, int dust_cut)
{
	if (len < 1) return false;
	int trip[64];
	for (int j = 0; j < 64; j++) trip[j] = 0;
	auto at = [&](int64_t k) -> unsigned { return k < len ? (unsigned)s[k] : 0u; };   // (the reference's sequences end in a 0 byte)
	unsigned key = ((at(0) & 3u) << 2) | (at(1) & 3u);
	const int64_t n = len > 64 ? 64 : len;
	int64_t c = 2;
	for (int64_t j = 2; j < n; j++) {
		key = ((key << 2) | (at(j) & 3u)) & 0x3Fu;
		trip[key]++;
		c++;
	}
	double sc = 0.0;
	for (int j = 0; j < 64; j++) sc += (double)trip[j] * ((double)trip[j] - 1.0) / 2.0;
	sc = sc / (double)(c - 3) * 10.0;
	return sc > (double)dust_cut;
}

// compare_read_names(), src/io.c:2128-2393: the same place on the flow cell (CASAVA 1.8 / <= 1.7 names), else the same name up
// to the first blank or ';'.  The format is taken from the first name seen (`detected`, like the reference's static).
bool names_differ(const std::string& a, const std::string& b, int& detected)
{
	char i1[100], f1[100], i2[100], f2[100];
	int r1 = 0, l1 = 0, t1 = 0, x1 = 0, y1 = 0, r2 = 0, l2 = 0, t2 = 0, x2 = 0, y2 = 0;
	i1[0] = f1[0] = i2[0] = f2[0] = 0;
	if (detected == -1) {
		if (sscanf(a.c_str(), "%99[^:]:%d:%99[^:]:%d:%d:%d:%d ", i1, &r1, f1, &l1, &t1, &x1, &y1) == 7) detected = 1;
		else if (sscanf(a.c_str(), "%99[^:]:%d:%d:%d:%d", i1, &l1, &t1, &x1, &y1) == 5) detected = 2;
		else detected = 1000;
	}
	if (detected == 1) {
		if (sscanf(a.c_str(), "%99[^:]:%d:%99[^:]:%d:%d:%d:%d ", i1, &r1, f1, &l1, &t1, &x1, &y1) != 7) return true;
		if (sscanf(b.c_str(), "%99[^:]:%d:%99[^:]:%d:%d:%d:%d ", i2, &r2, f2, &l2, &t2, &x2, &y2) != 7) return true;
		return y1 != y2 || x1 != x2 || t1 != t2 || l1 != l2 || strcmp(f1, f2) != 0 || r1 != r2 || strcmp(i1, i2) != 0;
	}
	if (detected == 2) {
		if (sscanf(a.c_str(), "%99[^:]:%d:%d:%d:%d", i1, &l1, &t1, &x1, &y1) != 5) return true;
		if (sscanf(b.c_str(), "%99[^:]:%d:%d:%d:%d", i2, &l2, &t2, &x2, &y2) != 5) return true;
		return y1 != y2 || x1 != x2 || t1 != t2 || l1 != l2 || strcmp(i1, i2) != 0;
	}
	for (size_t i = 0; i < a.size(); i++) {
		if (isspace((unsigned char)a[i]) || a[i] == ';') break;
		if (i >= b.size() || a[i] != b[i]) return true;
	}
	return false;
}

// name of record i of a batch
std::string record_name(const Batch& b, int64_t i)
{
	for (const Piece& pc : b.pieces)
		if (i >= pc.first && i < pc.first + (pc.hi - pc.lo)) {
			const TdRec& r = (*pc.recs)[(size_t)(pc.lo + (i - pc.first))];
			return std::string(pc.blk->data + r.name_off, (size_t)r.name_len);
		}
	return std::string();
}

struct Tuple {                        // record range [k * batch_reads, ...) of every input file
	std::vector<Batch*> b;
	std::vector<std::vector<int64_t>> tickets;   // [file][device]
};

} // namespace

extern "C" int td_stream_run_multi(const td_stream_file* files, int32_t n_files, int32_t n_devices, const char* out_prefix, int32_t dust,
                                   const td_stream_opts* opts, td_stream_stats* stats, int64_t* counts)
{
	if (!files || n_files < 1 || n_files > 8 || !out_prefix) { td_io_set_error("td_stream_run_multi: bad arguments (1..8 input files, an output prefix)"); return TD_FAIL; }
	if (n_devices < 1) n_devices = 1;
	const int K = n_files, N = n_devices;
	// barcode_hmm.c:105-153: which file holds the barcode (at most one may), how many read segments every file contributes
	int bar_file = -1, num_out_reads = 0;
	std::vector<int> read_present((size_t)K, 0);
	for (int k = 0; k < K; k++) {
		const td_arch* a = files[k].arch;
		if (!files[k].path || !a || a->n_segments < 1) { td_io_set_error("td_stream_run_multi: every input file needs a path and an architecture"); return TD_FAIL; }
		bool has_bar = false;
		for (int j = 0; j < a->n_segments; j++) { if (a->type[j] == 'B') has_bar = true; if (a->type[j] == 'R') read_present[(size_t)k]++; }
		if (has_bar) {
			if (bar_file >= 0) { td_io_set_error("td_stream_run_multi: barcodes seem to be in both architectures (barcode_hmm.c:140-145)"); return TD_FAIL; }
			bar_file = k;
		}
		num_out_reads += read_present[(size_t)k];
		if (!files[k].ctx) {
			// run_rna_dust (barcode_hmm.c:315-319) is what the controller runs instead of the HMM for an architecture that is one read segment
			if (!(a->n_segments == 1 && a->type[0] == 'R')) { td_io_set_error("td_stream_run_multi: a file without contexts must have the architecture R:N"); return TD_FAIL; }
		} else {
			for (int d = 0; d < N; d++) if (!files[k].ctx[d]) { td_io_set_error("td_stream_run_multi: NULL context"); return TD_FAIL; }
			int32_t art = 0;
			(void)td_get_option(files[k].ctx[0], "artifacts_active", &art);
			if (art) { td_io_set_error("td_stream_run_multi: a -ref artifact filter is not supported here (its thread ranges belong to the reference's batches; use td_stream_run / td_multi_decode)"); return TD_FAIL; }
		}
	}
	if (num_out_reads == 0) { td_io_set_error("td_stream_run_multi: no read segment in any architecture: no output files to create (io.c:846-852)"); return TD_FAIL; }
	// print_all() names its files after param->read_structure: the barcode file's architecture, else the last file's
	const td_arch* print_arch = files[bar_file >= 0 ? bar_file : K - 1].arch;
	td_stream_opts o{};
	if (opts) o = *opts;
	if (o.batch_reads <= 0) o.batch_reads = 1 << 18;
	if (o.block_bytes <= 0) o.block_bytes = (int64_t)64 << 20;
	if (o.block_bytes < 4096) o.block_bytes = 4096;
	int hw = (int)std::thread::hardware_concurrency();
	if (hw < 1) hw = 1;
	if (o.n_threads <= 0) o.n_threads = hw >= 16 ? 8 : (hw >= 4 ? hw / 2 : 1);
	if (o.n_threads > 32) o.n_threads = 32;
	int32_t depth = 3;
	for (int k = 0; k < K; k++) if (files[k].ctx) { (void)td_get_option(files[k].ctx[0], "pipeline_depth", &depth); break; }
	const double t_start = now_s();

	// the writer's side lives in one Pipeline object (output files, write pool, statistics), the readers' in one per input file
	Pipeline pw;
	pw.o = o; pw.dry = false;
	std::vector<std::unique_ptr<Pipeline>> pf;
	for (int k = 0; k < K; k++) pf.emplace_back(new Pipeline());
	const int n_batches = depth + 4;
	auto fail_all = [&](const std::string& m) { pw.fail(m); for (auto& q : pf) q->fail(m); };
	pw.ready.reset(new Queue<Batch*>(1)); pw.done.reset(new Queue<Batch*>(1)); pw.free_list.reset(new Queue<Batch*>(1));
	{
		std::vector<std::string> names;
		td_writer_file_names_n(out_prefix, print_arch, num_out_reads, names, &pw.num_alternatives);
		for (auto& nm : names) {
			const int fd = open(nm.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
			if (fd < 0) { for (int g : pw.fds) close(g); td_io_set_error("td_stream_run_multi: cannot create " + nm + ": " + strerror(errno)); return TD_FAIL; }
			pw.fds.push_back(fd);
		}
		pw.file_off.assign(pw.fds.size(), 0);
	}
	std::vector<size_t> file_base((size_t)K, 0);      // io.c:917-1001: c
	{ size_t c = 0; for (int k = 0; k < K; k++) { file_base[(size_t)k] = c; c += (size_t)pw.num_alternatives * (size_t)read_present[(size_t)k]; } }
	pw.write_pool.reset(new Pool(o.n_threads));
	const int parse_threads = std::max(1, o.n_threads / K);
	int rc = TD_OK;
	for (int k = 0; k < K && rc == TD_OK; k++) {
		Pipeline& p = *pf[(size_t)k];
		p.o = o; p.dry = files[k].ctx == nullptr;       // (a file that is not decoded needs no page-locked buffers)
		p.ready.reset(new Queue<Batch*>((size_t)n_batches));
		p.done.reset(new Queue<Batch*>((size_t)n_batches));
		p.free_list.reset(new Queue<Batch*>((size_t)n_batches));
		std::string err;
		if (!p.src.open(files[k].path, o.block_bytes, err)) { td_io_set_error(err); rc = TD_FAIL; break; }
		p.parse_pool.reset(new Pool(parse_threads));
		for (int b = 0; b < 2; b++) {
			Batch* q = p.new_batch();
			if (!q) { rc = TD_FAIL; td_io_set_error("td_stream_run_multi: page-locked memory exhausted"); break; }
			p.free_list->push(q);
		}
	}
	if (rc != TD_OK) {
		for (int g : pw.fds) close(g);
		for (auto& q : pf) release_batches(*q, false);
		return TD_FAIL;
	}
	std::vector<std::thread> th_alloc, th_prod;
	for (int k = 0; k < K; k++) {
		Pipeline* p = pf[(size_t)k].get();
		th_alloc.emplace_back([p, n_batches] { p->allocator(n_batches - 2); });
		th_prod.emplace_back([p] { p->producer(); });
	}
	Queue<Tuple*> done_t((size_t)n_batches);
	int64_t cnt[TD_NUM_COUNTERS];
	for (int q = 0; q < TD_NUM_COUNTERS; q++) cnt[q] = 0;
	// ---- writer: run_rna_dust for the files that are not decoded, the per-record combination, print_all ----
	std::thread t_write([&] {
		Tuple* t = nullptr;
		std::vector<int32_t> ctype, cbar;
		while (done_t.pop(t)) {
			const double t0 = now_s();
			const int64_t n = t->b[0]->n;
			for (int k = 0; k < K; k++) {
				Batch* b = t->b[(size_t)k];
				if (files[k].ctx) { b->seq_src = nullptr; continue; }
				b->seq_src = b->codes;
				const int64_t chunk = 16384, nch = (n + chunk - 1) / chunk;
				pw.write_pool->run(nch, [&](int64_t c) {   // do_rna_dust, barcode_hmm.c:2370-2395 (no -ref filter here)
					for (int64_t i = c * chunk; i < std::min(n, (c + 1) * chunk); i++) {
						td_read_result& r = b->res[i];
						memset(&r, 0, sizeof r);
						r.mapq = -1.0f; r.barcode = -1; r.fingerprint = -1;        // read_fasta_fastq's defaults, io.c:1698-1702
						r.read_type = TD_EXTRACT_SUCCESS;
						if (dust && rna_dust_low(b->codes + b->offs[i], b->offs[i + 1] - b->offs[i], dust)) r.read_type = TD_EXTRACT_FAIL_LOW_COMPLEXITY;
					}
				});
			}
			ctype.resize((size_t)n); cbar.resize((size_t)n);
			for (int64_t i = 0; i < n; i++) {          // barcode_hmm.c:329-351
				int32_t c = -100000;
				for (int k = 0; k < K; k++) c = std::max(c, t->b[(size_t)k]->res[i].read_type);
				ctype[(size_t)i] = c;
				cbar[(size_t)i] = t->b[(size_t)(bar_file >= 0 ? bar_file : 0)]->res[i].barcode;
				cnt[c & (TD_NUM_OUTCOME_SLOTS - 1)]++;                                      // the controller's counting, :354-384
				if (c == TD_EXTRACT_SUCCESS && cbar[(size_t)i] >= 0) cnt[TD_NUM_OUTCOME_SLOTS + (cbar[(size_t)i] & 0xFF)]++;
			}
			bool ok = true;
			for (int k = 0; k < K && ok; k++)
				if (read_present[(size_t)k] > 0) ok = pw.write_batch(t->b[(size_t)k], ctype.data(), cbar.data(), file_base[(size_t)k]);
			pw.st.n_reads += n; pw.st.n_batches++;
			pw.st.write_s += now_s() - t0;
			for (int k = 0; k < K; k++) { t->b[(size_t)k]->pieces.clear(); t->b[(size_t)k]->seq_src = nullptr; (void)pf[(size_t)k]->free_list->push(t->b[(size_t)k]); }
			delete t;
			if (!ok) return;
		}
		const double t0 = now_s();
		(void)pw.append_wait(-1);
		pw.st.write_s += now_s() - t0;
	});
	// ---- the calling thread: one record range of every file at a time, `depth` of them on the devices ----
	std::deque<Tuple*> flying;
	auto retire = [&]() -> bool {
		Tuple* t = flying.front();
		flying.pop_front();
		const double t0 = now_s();
		bool ok = true;
		for (int k = 0; k < K; k++)
			for (int d = 0; d < N; d++)
				if (files[k].ctx && t->tickets[(size_t)k][(size_t)d] && td_wait(files[k].ctx[d], t->tickets[(size_t)k][(size_t)d]) != TD_OK) {
					if (ok) fail_all(std::string("td_stream_run_multi: ") + td_last_error(files[k].ctx[d]));
					ok = false;
				}
		pw.st.decode_s += now_s() - t0;
		if (!ok) { delete t; return false; }
		return done_t.push(t);
	};
	bool first = true;
	int name_format = -1;
	while (!pw.failed()) {
		Tuple* t = new Tuple();
		t->b.assign((size_t)K, nullptr);
		t->tickets.assign((size_t)K, std::vector<int64_t>((size_t)N, 0));
		int n_closed = 0;
		bool bad = false;
		for (int k = 0; k < K && !bad; k++) {
			for (;;) {   // (nothing ready from this file and record ranges in flight: retire the oldest -- its buffers may be what the reader waits for)
				Batch* b = nullptr;
				const int got = flying.empty() ? (pf[(size_t)k]->ready->pop(b) ? 1 : -1) : pf[(size_t)k]->ready->try_pop(b);
				if (got == 1) { t->b[(size_t)k] = b; break; }
				if (got < 0) { n_closed++; break; }
				if (!retire()) { bad = true; break; }
			}
		}
		if (bad || n_closed == K) { for (Batch* b : t->b) if (b) (void)b; delete t; break; }
		bool same_n = n_closed == 0;
		for (int k = 1; k < K && same_n; k++) same_n = t->b[(size_t)k]->n == t->b[0]->n;
		if (!same_n) {   // barcode_hmm.c:257-268
			fail_all("td_stream_run_multi: the input files differ in their number of records");
			delete t;
			break;
		}
		const int64_t n = t->b[0]->n;
		if (first) {     // the first 1000 names of every pair of files name the same reads (barcode_hmm.c:272-289)
			first = false;
			bool differ = false;
			std::string na, nb;
			for (int64_t i = 0; i < std::min<int64_t>(1000, n) && !differ; i++) {
				na = record_name(*t->b[0], i);
				for (int k = 1; k < K && !differ; k++) { nb = record_name(*t->b[(size_t)k], i); differ = names_differ(na, nb, name_format); }
			}
			if (differ) { fail_all("td_stream_run_multi: the input files seem to contain reads in different order: " + na + " / " + nb); delete t; break; }
		}
		if ((int)flying.size() >= depth && !retire()) { delete t; break; }
		const double t0 = now_s();
		bool ok = true;
		for (int k = 0; k < K && ok; k++) {
			if (!files[k].ctx) continue;
			Batch* b = t->b[(size_t)k];
			for (int d = 0; d < N && ok; d++) {      // run_pHMM's contiguous ranges over the devices (barcode_hmm.c:1911-1922)
				const int64_t interval = n / N, lo = (int64_t)d * interval, hi = (d == N - 1) ? n : lo + interval;
				if (hi <= lo) continue;
				if (td_submit(files[k].ctx[d], b->codes, 0, b->offs + lo, hi - lo, TD_MODE_GET_LABEL, b->res + lo, nullptr, b->seq_out + b->offs[lo],
				              &t->tickets[(size_t)k][(size_t)d]) != TD_OK) {
					fail_all(std::string("td_stream_run_multi: ") + td_last_error(files[k].ctx[d]));
					ok = false;
				}
			}
		}
		pw.st.decode_s += now_s() - t0;
		flying.push_back(t);
		if (!ok) break;
	}
	while (!flying.empty() && !pw.failed()) if (!retire()) break;
	if (pw.failed()) {   // nothing of ours may stay queued on the devices
		for (Tuple* t : flying) {
			for (int k = 0; k < K; k++) for (int d = 0; d < N; d++) if (files[k].ctx && t->tickets[(size_t)k][(size_t)d]) (void)td_wait(files[k].ctx[d], t->tickets[(size_t)k][(size_t)d]);
			delete t;
		}
		flying.clear();
		done_t.abort();
	}
	done_t.close();
	t_write.join();
	pw.append_stop();
	for (int k = 0; k < K; k++) {
		Pipeline& p = *pf[(size_t)k];
		if (pw.failed()) p.fail(pw.err);
		th_prod[(size_t)k].join();
		{ std::lock_guard<std::mutex> lk(p.all_mu); p.stop_alloc = true; }
		p.hint_cv.notify_all();
		p.free_list->abort();
		th_alloc[(size_t)k].join();
		pw.st.bytes_in += p.st.bytes_in; pw.st.parse_s += p.st.parse_s; pw.st.read_s += p.st.read_s;
		if (p.failed() && !pw.failed()) pw.fail(p.err);
	}
	for (int fd : pw.fds) if (close(fd) != 0 && !pw.failed()) pw.fail(std::string("td_stream_run_multi: close failed: ") + strerror(errno));
	for (auto& q : pf) release_batches(*q, !pw.failed());
	pw.st.wall_s = now_s() - t_start;
	if (stats) *stats = pw.st;
	if (counts) for (int q = 0; q < TD_NUM_COUNTERS; q++) counts[q] = cnt[q];
	if (pw.failed()) { td_io_set_error(pw.err); return TD_FAIL; }
	return TD_OK;
}
