// td_fastq.cpp -- chunk-parallel FASTQ/FASTA parsing and buffered demultiplexed FASTQ writing (include/tagdust_io.h).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <thread>
#include <vector>

#include "../../include/tagdust_io.h"
#include "td_io_internal.h"

namespace {

inline bool is_cntrl(unsigned char ch) { return ch < 32 || ch == 127; } // iscntrl() in the C locale

struct Code {
	uint8_t t[256];
	Code()
	{ // init_nuc_code(), src/nuc_code.c:46-74
		for (int k = 0; k < 256; k++) t[k] = 4;
		t['.'] = 5;
		t['A'] = t['a'] = 0; t['C'] = t['c'] = 1; t['G'] = t['g'] = 2; t['T'] = t['t'] = 3; t['U'] = t['u'] = 3;
	}
};
const Code kCode;

typedef TdRec Rec;

thread_local std::string g_io_error;

// length of the field starting at p: up to the first control character, at most n bytes.  Lines almost never hold one before
// their end, so the bytes are tested sixteen at a time (the loop body vectorises) and walked one by one only where one is.
inline int32_t field_len_fast(const char* p, int64_t n)
{
	int64_t q = 0;
	for (; q + 16 <= n; q += 16) {
		unsigned any = 0;
		for (int k = 0; k < 16; k++) { const unsigned char ch = (unsigned char)p[q + k]; any |= (unsigned)(ch < 32) | (unsigned)(ch == 127); }
		if (any) break;
	}
	while (q < n && !is_cntrl((unsigned char)p[q])) q++;
	return (int32_t)q;
}

} // namespace

void td_io_set_error(const std::string& msg) { g_io_error = msg; }

// read_fasta_fastq()'s line state machine (io.c:1697-1799) over text[lo, hi); lo must be the start of a line
void td_parse_range(const char* text, int64_t lo, int64_t hi, std::vector<TdRec>& out)
{
	bool set = false, seq_p = false;
	int64_t p = lo;
	while (p < hi) {
		const char* nl = (const char*)memchr(text + p, '\n', (size_t)(hi - p));
		const int64_t e = nl ? (nl - text) : hi; // line = [p, e)
		const char first = (p < e) ? text[p] : '\n';
		auto field_len = [&](int64_t from) { return field_len_fast(text + from, e - from); };
		if ((first == '@' || first == '>') && !set) {
			Rec r; r.name_off = p + 1; r.name_len = field_len(p + 1); r.seq_off = -1; r.seq_len = 0; r.qual_off = -1; r.qual_len = 0;
			out.push_back(r);
			seq_p = true; set = true;
		} else if (first == '+' && !set) {
			seq_p = false; set = true;
		} else {
			if (set && !out.empty()) {
				if (seq_p) { out.back().seq_off = p; out.back().seq_len = field_len(p); }
				else { out.back().qual_off = p; out.back().qual_len = field_len(p); }
			}
			set = false;
		}
		p = e + 1;
	}
}

extern const uint8_t* const td_nuc_code_ptr;
const uint8_t* const td_nuc_code_ptr = kCode.t;

int64_t td_next_record_start(const char* text, int64_t len, int64_t from, bool fasta)
{
	// FASTQ: a line starting with '@' whose line after next starts with '+' (a quality line may start with '@', but then the
	// line after next is a sequence, which cannot start with '+').  FASTA: any line starting with '>'.
	int64_t p = from;
	if (p <= 0) return 0;
	if (text[p - 1] != '\n') {   // to the start of the next line
		const char* nl = (const char*)memchr(text + p, '\n', (size_t)(len - p));
		if (!nl) return len;
		p = nl - text + 1;
	}
	while (p < len) {
		const char* nl1 = (const char*)memchr(text + p, '\n', (size_t)(len - p));
		if (!nl1) return len;
		if (fasta && text[p] == '>') return p;
		if (!fasta && text[p] == '@') {
			const char* nl2 = (const char*)memchr(nl1 + 1, '\n', (size_t)(len - (nl1 + 1 - text)));
			if (nl2 && nl2 + 1 < text + len && nl2[1] == '+') return p;
		}
		p = nl1 - text + 1;
	}
	return len;
}

extern "C" int td_reads_parse(const char* text, int64_t len, int32_t n_threads, td_reads** out)
{
	if (!text || len < 0 || !out) return TD_FAIL;
	if (n_threads <= 0) { n_threads = (int)std::thread::hardware_concurrency(); if (n_threads > 16) n_threads = 16; }
	if (n_threads < 1 || len < (1 << 22)) n_threads = 1;
	// chunk boundaries at record starts.  FASTQ: a line starting with '@' whose line after next starts with '+' (a quality
	// line may start with '@', but then the line after next is a sequence, which cannot start with '+').  FASTA (text
	// starts with '>'): any line starting with '>'.  The first chunk starts at 0.
	const bool fasta = len > 0 && text[0] == '>';
	std::vector<int64_t> cut(1, 0);
	for (int t = 1; t < n_threads; t++) {
		const int64_t p = td_next_record_start(text, len, len / n_threads * t, fasta);
		if (p < len && p > cut.back()) cut.push_back(p);
	}
	cut.push_back(len);
	const int nchunk = (int)cut.size() - 1;
	std::vector<std::vector<Rec>> recs((size_t)nchunk);
	{
		std::vector<std::thread> th;
		for (int k = 1; k < nchunk; k++) th.emplace_back(td_parse_range, text, cut[(size_t)k], cut[(size_t)k + 1], std::ref(recs[(size_t)k]));
		td_parse_range(text, cut[0], cut[1], recs[0]);
		for (auto& t : th) t.join();
	}
	int64_t n = 0;
	for (auto& v : recs) {
		// "Length of sequence and base qualities differ" ends the reference's run (io.c:1776-1781): a short or cut-off
		// quality line must not reach the writer, which reads one quality character per base
		for (const Rec& q : v)
			if (q.qual_off >= 0 && q.qual_len != (q.seq_off >= 0 ? q.seq_len : 0)) {
				char msg[256];
				snprintf(msg, sizeof msg, "td_reads_parse: record %lld (\"%.*s\"): sequence has %d characters, base qualities %d",
				         (long long)(n + (&q - v.data())), q.name_len < 60 ? q.name_len : 60, text + q.name_off, q.seq_off >= 0 ? q.seq_len : 0, q.qual_len);
				g_io_error = msg;
				return TD_FAIL;
			}
		n += (int64_t)v.size();
	}
	td_reads* r = (td_reads*)calloc(1, sizeof(td_reads));
	if (!r) return TD_FAIL;
	r->n_reads = n; r->text = text;
	r->name_off = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n + 1)); r->name_len = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1));
	r->qual_off = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n + 1)); r->offs = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n + 1));
	if (!r->name_off || !r->name_len || !r->qual_off || !r->offs) { td_reads_free(r); return TD_FAIL; }
	std::vector<int64_t> seq_off((size_t)n);
	int64_t i = 0, total = 0;
	r->offs[0] = 0;
	for (auto& v : recs)
		for (const Rec& q : v) {
			r->name_off[i] = q.name_off; r->name_len[i] = q.name_len; r->qual_off[i] = q.qual_off;
			seq_off[(size_t)i] = q.seq_off;
			total += q.seq_off >= 0 ? q.seq_len : 0;
			r->offs[++i] = total;
		}
	r->codes = (uint8_t*)malloc((size_t)total + 1);
	if (!r->codes) { td_reads_free(r); return TD_FAIL; }
	auto encode = [&](int64_t lo, int64_t hi) {
		for (int64_t k = lo; k < hi; k++) {
			const int64_t so = seq_off[(size_t)k];
			uint8_t* dst = r->codes + r->offs[k];
			const int64_t l = r->offs[k + 1] - r->offs[k];
			td_encode_bases((const unsigned char*)text + so, dst, l);
		}
	};
	if (n_threads == 1 || n < 65536) encode(0, n);
	else {
		std::vector<std::thread> th;
		const int64_t per = (n + n_threads - 1) / n_threads;
		for (int t = 0; t < n_threads; t++) { const int64_t lo = t * per, hi = lo + per < n ? lo + per : n; if (lo < hi) th.emplace_back(encode, lo, hi); }
		for (auto& t : th) t.join();
	}
	*out = r;
	return TD_OK;
}

extern "C" const char* td_io_last_error(void) { return g_io_error.c_str(); }

extern "C" void td_reads_free(td_reads* r)
{
	if (!r) return;
	free(r->name_off); free(r->name_len); free(r->qual_off); free(r->offs); free(r->codes);
	free(r);
}

// ---------------------------------------------------------------------------------------------------------
// print_all(), io.c:757-1016, for one input file
// ---------------------------------------------------------------------------------------------------------
struct td_writer {
	std::vector<FILE*> files;
	int num_alternatives = 2;
	int num_out_reads = 1;
};

void td_writer_file_names(const char* prefix, const td_arch* a, std::vector<std::string>& names, int* num_alternatives)
{
	td_writer_file_names_n(prefix, a, -1, names, num_alternatives);
}

// n_out_reads: read segments over ALL input files of the run (io.c:795-797: num_out_reads); < 0: those of this architecture
void td_writer_file_names_n(const char* prefix, const td_arch* a, int n_out_reads, std::vector<std::string>& names, int* num_alternatives)
{
	int barsegment = -1, n_r = 0;
	for (int i = 0; i < a->n_segments; i++) {
		if (a->type[i] == 'B' && barsegment < 0) barsegment = i;
		if (a->type[i] == 'R') n_r++;
	}
	if (n_out_reads >= 0) n_r = n_out_reads;
	const int alt = barsegment >= 0 ? a->n_seq[barsegment] : 2;
	if (num_alternatives) *num_alternatives = alt;
	names.clear();
	for (int i = 0; i < n_r; i++) { // io.c:859-915
		const std::string rd = n_r > 1 ? "_READ" + std::to_string(i + 1) : "";
		if (barsegment >= 0) {
			for (int j = 0; j < alt - 1; j++) names.push_back(std::string(prefix) + "_BC_" + a->seqs[barsegment][j] + rd + ".fq");
		} else {
			names.push_back(std::string(prefix) + rd + ".fq");
		}
		names.push_back(std::string(prefix) + "_un" + rd + ".fq");
	}
}

extern "C" int td_writer_open(const char* prefix, const td_arch* a, td_writer** out)
{
	if (!prefix || !a || !out) return TD_FAIL;
	td_writer* w = new td_writer();
	std::vector<std::string> names;
	td_writer_file_names(prefix, a, names, &w->num_alternatives);
	for (auto& nm : names) {
		FILE* f = fopen(nm.c_str(), "w");
		if (!f) { for (FILE* g : w->files) fclose(g); delete w; return TD_FAIL; }
		w->files.push_back(f);
	}
	*out = w;
	return TD_OK;
}

// formats reads [lo, hi) into one buffer per output file (io.c:917-1001)
static void format_range(const td_writer* w, const td_reads* rd, const td_read_result* res, const uint8_t* seq_out,
                         int64_t lo, int64_t hi, std::vector<std::string>* bufs)
{
	static const char alphabet[] = "ACGTNN";
	char head[64];
	std::string seq, qual;
	for (int64_t i = lo; i < hi; i++) {
		size_t f; // io.c:923-934
		if (res[i].read_type == TD_EXTRACT_SUCCESS) f = (res[i].barcode != -1) ? (size_t)(res[i].barcode & 0xFF) : 0;
		else f = (size_t)w->num_alternatives - 1;
		const uint8_t* s = seq_out + rd->offs[i];
		const int64_t len = rd->offs[i + 1] - rd->offs[i];
		const char* q = rd->qual_off[i] >= 0 ? rd->text + rd->qual_off[i] : nullptr;
		auto emit = [&]() { // io.c:955-975
			if (f >= bufs->size()) return;
			std::string& b = (*bufs)[f];
			b += '@'; b.append(rd->text + rd->name_off[i], (size_t)rd->name_len[i]);
			if (res[i].fingerprint != -1) { snprintf(head, sizeof head, ";FP:%d", res[i].fingerprint); b += head; }
			snprintf(head, sizeof head, ";RQ:%0.2f\n", (double)res[i].mapq); b += head;
			b += seq; b += "\n+\n"; b += qual; b += '\n';
		};
		seq.clear(); qual.clear();
		for (int64_t g = 0; g < len; g++) {
			if (s[g] < 5) {
				seq += alphabet[s[g]];
				qual += q ? q[g] : '.';
			} else if (!seq.empty()) {
				emit();
				f += (size_t)w->num_alternatives;
				seq.clear(); qual.clear();
			}
		}
		if (!seq.empty()) emit();
	}
}

extern "C" int td_writer_write(td_writer* w, const td_reads* rd, const td_read_result* res, const uint8_t* seq_out)
{
	if (!w || !rd || !res || !seq_out) return TD_FAIL;
	if (w->files.empty()) return TD_OK;
	// records are formatted by several threads over contiguous read ranges and appended range by range, so every file
	// keeps the input order, like the reference's single loop
	int nt = (int)std::thread::hardware_concurrency();
	if (const char* e = getenv("TD_HOST_THREADS")) nt = atoi(e);
	if (nt > 16) nt = 16;
	if (nt < 1 || rd->n_reads < 65536) nt = 1;
	std::vector<std::vector<std::string>> parts((size_t)nt, std::vector<std::string>(w->files.size()));
	const int64_t per = (rd->n_reads + nt - 1) / nt;
	std::vector<std::thread> th;
	for (int t = 1; t < nt; t++) {
		const int64_t lo = t * per, hi = lo + per < rd->n_reads ? lo + per : rd->n_reads;
		if (lo < hi) th.emplace_back(format_range, w, rd, res, seq_out, lo, hi, &parts[(size_t)t]);
	}
	format_range(w, rd, res, seq_out, 0, per < rd->n_reads ? per : rd->n_reads, &parts[0]);
	for (auto& t : th) t.join();
	for (int t = 0; t < nt; t++)
		for (size_t f = 0; f < w->files.size(); f++) {
			const std::string& b = parts[(size_t)t][f];
			if (!b.empty() && fwrite(b.data(), 1, b.size(), w->files[f]) != b.size()) {   // e.g. a full disk
				g_io_error = "td_writer_write: short write";
				return TD_FAIL;
			}
		}
	return TD_OK;
}

extern "C" int td_writer_close(td_writer* w)
{
	if (!w) return TD_FAIL;
	int rc = TD_OK;
	for (size_t f = 0; f < w->files.size(); f++) if (fclose(w->files[f]) != 0) rc = TD_FAIL;
	delete w;
	return rc;
}

// ---------------------------------------------------------------------------------------------------------
// -ref artifact sequences: read_fasta(), src/io.c:1912-2001
// ---------------------------------------------------------------------------------------------------------
extern "C" int td_fasta_parse(const char* text, int64_t len, td_fasta** out)
{
	if (!text || !out || len < 0) return TD_FAIL;
	td_fasta* f = (td_fasta*)calloc(1, sizeof(td_fasta));
	if (!f) return TD_FAIL;
	// the reference works on a NUL-terminated copy: stop at an embedded NUL like strlen() (:1923)
	int64_t nbytes = 0;
	while (nbytes < len && text[nbytes]) nbytes++;
	int32_t nseq = 0;
	{   // :1930-1939: a '>' counts once per line, wherever it stands
		bool stop = false;
		for (int64_t i = 0; i < nbytes; i++) {
			const char ch = text[i];
			if (ch == '>' && !stop) { nseq++; stop = true; }
			else if (ch == '\n') stop = false;
		}
	}
	f->string = (uint8_t*)malloc((size_t)nbytes + 2);
	f->s_index = (int32_t*)calloc((size_t)nseq + 1, sizeof(int32_t));
	f->names = (char**)calloc((size_t)(nseq > 0 ? nseq : 1), sizeof(char*));
	if (!f->string || !f->s_index || !f->names) { td_fasta_free(f); return TD_FAIL; }
	auto at = [&](int64_t i) -> char { const char ch = i < nbytes ? text[i] : '\n'; return ch == '\r' ? '\n' : ch; }; // :1937
	int64_t n = 0;
	int32_t c = 0;
	for (int64_t i = 0; i < nbytes; i++) {
		const char ch = at(i);
		if (ch == '>') {  // :1954-1984 (every '>' starts a record here; headers that hold a second one are not expected)
			int64_t j = i + 1;
			while (at(j) != '\n') j++;
			if (c >= nseq) break;
			char* name = (char*)malloc((size_t)(j - i));
			int32_t l = 0;
			for (int64_t k = i + 1; k < j; k++) name[l++] = isspace((unsigned char)text[k]) ? '_' : text[k];
			name[l] = 0;
			f->names[c] = name;
			f->s_index[c] = (int32_t)n;
			f->string[n++] = 'X';
			i = j;
			c++;
		} else if (isalnum((unsigned char)ch)) {
			f->string[n++] = kCode.t[(unsigned char)ch];
		}
	}
	f->n_seq = c;
	f->s_index[c] = (int32_t)n;
	f->string[n] = 'X';
	*out = f;
	return TD_OK;
}

extern "C" void td_fasta_free(td_fasta* f)
{
	if (!f) return;
	if (f->names) for (int32_t j = 0; j < f->n_seq; j++) free(f->names[j]);
	free(f->names); free(f->string); free(f->s_index); free(f);
}
