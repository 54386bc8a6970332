// td_io_internal.h -- shared between td_fastq.cpp (whole-text parser / writer) and td_stream.cpp (streaming pipeline)
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/tagdust_io.h"
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

// one record as read_fasta_fastq()'s state machine (src/io.c:1697-1799) sees it; offsets into the text it was parsed from
struct TdRec { int64_t name_off; int32_t name_len; int64_t seq_off; int32_t seq_len; int64_t qual_off; int32_t qual_len; };

// the state machine over text[lo, hi); lo must be the start of a line
void td_parse_range(const char* text, int64_t lo, int64_t hi, std::vector<TdRec>& out);
// next record start at or after byte `from` of text[0, len) (a line starting with '@' whose line after next starts with '+';
// for FASTA text any line starting with '>'); len when there is none
int64_t td_next_record_start(const char* text, int64_t len, int64_t from, bool fasta);
// init_nuc_code(), src/nuc_code.c:46-74
extern const uint8_t* const td_nuc_code_ptr;   // [256]
// dst[j] = nuc_code[s[j]] (init_nuc_code, src/nuc_code.c:46-74: A C G T/U in either case -> 0 1 2 3, '.' -> 5, anything else 4),
// sixteen letters at a time: the table walk was two thirds of the parse stage (TD_STREAM_DEBUG).
inline void td_encode_bases(const unsigned char* s, uint8_t* dst, const int64_t n)
{
#if defined(__SSE2__)
	if (n >= 16) {
		const __m128i lc = _mm_set1_epi8(0x20), four = _mm_set1_epi8(4), three = _mm_set1_epi8(3), two = _mm_set1_epi8(2), one = _mm_set1_epi8(1);
		const __m128i la = _mm_set1_epi8('a'), lcc = _mm_set1_epi8('c'), lg = _mm_set1_epi8('g'), lt = _mm_set1_epi8('t'), lu = _mm_set1_epi8('u'), dot = _mm_set1_epi8('.');
		auto block = [&](const int64_t k) {
			const __m128i v = _mm_loadu_si128((const __m128i*)(s + k));
			const __m128i l = _mm_or_si128(v, lc);        // 'A' and 'a' are the only bytes that give 'a', and so on
			__m128i o = four;
			o = _mm_sub_epi8(o, _mm_and_si128(_mm_cmpeq_epi8(l, la), four));
			o = _mm_sub_epi8(o, _mm_and_si128(_mm_cmpeq_epi8(l, lcc), three));
			o = _mm_sub_epi8(o, _mm_and_si128(_mm_cmpeq_epi8(l, lg), two));
			o = _mm_sub_epi8(o, _mm_and_si128(_mm_or_si128(_mm_cmpeq_epi8(l, lt), _mm_cmpeq_epi8(l, lu)), one));
			o = _mm_add_epi8(o, _mm_and_si128(_mm_cmpeq_epi8(v, dot), one));
			_mm_storeu_si128((__m128i*)(dst + k), o);
		};
		int64_t k = 0;
		for (; k + 16 <= n; k += 16) block(k);
		if (k < n) block(n - 16);                         // the last sixteen again, overlapping
		return;
	}
#endif
	for (int64_t j = 0; j < n; j++) dst[j] = td_nuc_code_ptr[s[j]];
}

// print_all()'s file set for one input file (src/io.c:859-915): names in file-index order; *num_alternatives as io.c:923-934 uses it
void td_writer_file_names(const char* prefix, const td_arch* a, std::vector<std::string>& names, int* num_alternatives);
void td_writer_file_names_n(const char* prefix, const td_arch* a, int n_out_reads, std::vector<std::string>& names, int* num_alternatives);
// "td_..." message of the last failure on this thread (td_io_last_error)
void td_io_set_error(const std::string& msg);
