// td_io_internal.h -- shared between td_fastq.cpp (whole-text parser / writer) and td_stream.cpp (streaming pipeline)
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/tagdust_io.h"

// one record as read_fasta_fastq()'s state machine (src/io.c:1697-1799) sees it; offsets into the text it was parsed from
struct TdRec { int64_t name_off; int32_t name_len; int64_t seq_off; int32_t seq_len; int64_t qual_off; int32_t qual_len; };

// the state machine over text[lo, hi); lo must be the start of a line
void td_parse_range(const char* text, int64_t lo, int64_t hi, std::vector<TdRec>& out);
// next record start at or after byte `from` of text[0, len) (a line starting with '@' whose line after next starts with '+';
// for FASTA text any line starting with '>'); len when there is none
int64_t td_next_record_start(const char* text, int64_t len, int64_t from, bool fasta);
// init_nuc_code(), src/nuc_code.c:46-74
extern const uint8_t* const td_nuc_code_ptr;   // [256]
// print_all()'s file set for one input file (src/io.c:859-915): names in file-index order; *num_alternatives as io.c:923-934 uses it
void td_writer_file_names(const char* prefix, const td_arch* a, std::vector<std::string>& names, int* num_alternatives);
void td_writer_file_names_n(const char* prefix, const td_arch* a, int n_out_reads, std::vector<std::string>& names, int* num_alternatives);
// "td_..." message of the last failure on this thread (td_io_last_error)
void td_io_set_error(const std::string& msg);
