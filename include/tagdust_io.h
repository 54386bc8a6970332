/*
 * tagdust_io.h -- FASTQ/FASTA ingest and demultiplexed FASTQ egress around the decode path (part of
 * libtagdust_hip.so, plain C, host only).  SURVEY.md 8(f2): at GPU decode rates the reference's fgets()/fprintf()
 * loops are the end-to-end limiter, so these are chunk-parallel and buffered.  They mirror
 *
 *   read_fasta_fastq()   src/io.c:1684-1815   (record state machine, name up to the first control character,
 *                                              base codes through nuc_code)            -> td_reads_parse
 *   print_all()          src/io.c:757-1016    (file set, ";FP:%d" / ";RQ:%0.2f" header tags, one record per run of
 *                                              kept bases, unextracted reads to "_un")  -> td_writer_*
 * for one input file, and
 *   get_fasta() / read_fasta()   src/io.c:1826-2001   (-ref artifact sequences)        -> td_fasta_parse
 */
#ifndef TAGDUST_IO_H
#define TAGDUST_IO_H

#include <stdint.h>
#include "tagdust_hip.h"
#include "tagdust_model.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct td_reads {
	int64_t  n_reads;
	const char* text;      /* the caller's buffer; names and qualities are referenced in place */
	int64_t* name_off;     /* [n] offset of the name (after '@' / '>') in text */
	int32_t* name_len;     /* [n] */
	int64_t* qual_off;     /* [n] offset of the quality string, -1 for FASTA records */
	int64_t* offs;         /* [n+1] read boundaries in codes (ready for td_batch_upload) */
	uint8_t* codes;        /* base codes 0..4 ('.' -> 5 like the reference) */
} td_reads;

/* Parse a whole FASTQ/FASTA text held in memory with n_threads host threads (<= 0: pick).  TD_FAIL (with a message in
 * td_io_last_error) when a record's quality line is not as long as its sequence -- the reference ends the run there
 * (src/io.c:1776-1781). */
int  td_reads_parse(const char* text, int64_t len, int32_t n_threads, td_reads** out);
void td_reads_free(td_reads* reads);
/* message of the last failed td_reads_parse / td_writer_write of this thread */
const char* td_io_last_error(void);

/* struct fasta (src/io.h:59-71) as read_fasta() leaves it: ready for td_set_artifacts */
typedef struct td_fasta {
	int32_t  n_seq;
	uint8_t* string;       /* per sequence one 'X' byte, then nuc_code of every alphanumeric character */
	int32_t* s_index;      /* [n_seq+1]; sequence j = string[s_index[j] .. s_index[j+1]) */
	char**   names;        /* [n_seq] header line without '>', white space -> '_' */
} td_fasta;
int  td_fasta_parse(const char* text, int64_t len, td_fasta** out);
void td_fasta_free(td_fasta* fasta);

typedef struct td_writer td_writer;
/* Open print_all()'s file set for one input file: "<prefix>_BC_<barcode>.fq" per barcode + "<prefix>_un.fq", or
 * "<prefix>.fq" + "<prefix>_un.fq" without a barcode segment ("_READ<k>" suffixes when the architecture has several
 * R segments).  Existing files are overwritten. */
int  td_writer_open(const char* out_prefix, const td_arch* arch, td_writer** out);
/* Append one decoded batch: res / seq_out exactly as td_batch_download returns them. */
int  td_writer_write(td_writer* w, const td_reads* reads, const td_read_result* res, const uint8_t* seq_out);
int  td_writer_close(td_writer* w);

/* ---- one input file of any size as a pipeline: the reference's batch loop around run_pHMM (src/barcode_hmm.c:244-385) ----
 * read_fasta_fastq() of <= 1 000 001 records (io.c:1684-1815; plain files, or zcat / bzcat through popen like io_handler(),
 * io.c:382-608, by the file name's suffix; "-" = stdin) -> td_submit / td_wait on `ctx` (model, parameters, artifact filter
 * and window as set by the caller; TD_MODE_GET_LABEL) -> print_all()'s per-barcode appends (io.c:757-1016), with the three
 * steps of consecutive batches side by side: a reader/parser thread fills the next batch's page-locked buffers (records
 * found and base-coded by n_threads threads), the calling thread drives the device ("pipeline_depth" batches in flight), a
 * writer thread formats and appends finished batches (n_threads threads; every output file keeps input order).  A plain
 * file is mapped, not read.  Batches hold exactly batch_reads records, whatever the block size.
 * The output files are those td_writer_open names, byte for byte what td_reads_parse / td_writer_write give for the whole
 * text at once.  ctx == NULL: a parse-only run (no GPU, nothing written) that fills stats, codes_fnv included.
 * Alignment files (.sam, .bam, .sam.gz, .bam.gz) come as text from `samtools view -SF 768 <file>` / `samtools view -F 768 <file>`
 * (a .gz through zcat first) exactly as io_handler() starts it (io.c:467-575), and a record is QNAME, SEQ and QUAL of a line
 * that does not start with '@' (read_sam_chunk, io.c:1498-1660); samtools must be on PATH, as for the reference.  A line with
 * fewer than 11 fields or without one quality per base fails the run (the reference reads past the field it did not find).
 * Errors: a decompressor or samtools that ends with a non-zero status (truncated / corrupt .gz, .bz2; no samtools) fails the run -- the reference
 * ignores pclose()'s status and writes a partial file set.  A mapped plain file must not shrink while the call runs (the parse
 * threads read the mapping; the kernel answers a truncated mapping with SIGBUS, as for any mmap reader).  When page-locked
 * memory runs short the pipeline runs with the batches it could get (at least one), more slowly, and fails with "page-locked
 * memory exhausted" only if not even a single batch fits. */
typedef struct td_stream_opts {
	int32_t batch_reads;   /* records per batch; 0 = 1 000 001 (param->num_query, src/barcode_hmm.c:172) when the context has a
	                          -ref artifact filter -- its per-thread read ranges are taken over a batch, so the boundaries are part
	                          of the result -- and 2^18 otherwise (results do not depend on the batching then) */
	int32_t n_threads;     /* host threads of the parse stage and of the write stage, each; 0 = pick (<= 8) */
	int64_t block_bytes;   /* bytes of input taken at a time; 0 = 64 MiB */
} td_stream_opts;
typedef struct td_stream_stats {
	int64_t n_reads, n_batches, bytes_in, bytes_out;
	double  wall_s;        /* the whole call */
	double  read_s;        /* waiting for input bytes (pipes; a mapped file is paged in by the parse threads) */
	double  parse_s;       /* parse stage busy: records, batch assembly, base coding */
	double  decode_s;      /* calling thread inside td_submit / td_wait */
	double  write_s;       /* write stage busy: formatting + appends */
	uint64_t codes_fnv;    /* parse-only runs: FNV-1a over (read length, base codes) of all reads in order; else 0 */
} td_stream_stats;
/* ";RQ:%0.2f" of print_all() (io.c:963-992) without printf: the characters "%0.2f" gives for q, into buf[48]; returns their
 * number.  Exposed so that tests can hold it against printf. */
int td_format_q(float q, char* buf);
int td_stream_run(td_ctx* ctx, const char* in_path, const td_arch* arch, const char* out_prefix,
                  const td_stream_opts* opts, td_stream_stats* stats);
/* ---- several input files of one run, several devices (BASELINE configs[3]: the CASAVA three-read shape) ----
 * The controller's loop for paired / multi-read data (hmm_controller_multiple, src/barcode_hmm.c:244-385): the next batch_reads
 * records of EVERY input file (the files must hold the same reads in the same order: equal record counts, :257-268, and the first
 * 1000 names compared like compare_read_names(), io.c:2128-2393) -> each file's records through that file's own model --
 * run_pHMM's contiguous ranges (:1911-1922) over the n_devices contexts the caller holds for it, every range a td_submit /
 * td_wait on its device, results in input order -- or, for a file whose architecture is a single read segment, through
 * run_rna_dust (:315-319, :2370-2395: no HMM; DUST with `dust`, src/barcode_hmm.c:2407-2467) -> the per-record combination
 * (:329-351: the outcome is the maximum over the files, the barcode that of the one file that has a barcode segment) ->
 * print_all() (io.c:757-1016): every file that has read segments writes its records to its own "_READ<k>" files, named after the
 * barcode file's architecture, by the combined outcome and barcode.  The same three-stage pipeline as td_stream_run, one
 * reader / parser per input file.  counts (may be NULL) receives the controller's serial counting over the combined records
 * (TD_NUM_COUNTERS words: outcome slots, then per-barcode bins).  A -ref artifact filter is not supported here. */
typedef struct td_stream_file {
	const char*    path;   /* one input file (plain, .gz, .bz2) */
	const td_arch* arch;   /* its architecture: param->read_structures[i], barcode_hmm.c:105-137 */
	td_ctx* const* ctx;    /* n_devices contexts holding this file's model, threshold, minlen and dust (one per device, in device
	                          order); NULL for an architecture that is one read segment ("R:N"): that file is not decoded */
} td_stream_file;
int td_stream_run_multi(const td_stream_file* files, int32_t n_files, int32_t n_devices, const char* out_prefix, int32_t dust,
                        const td_stream_opts* opts, td_stream_stats* stats, int64_t* counts /* [TD_NUM_COUNTERS] */);
/* td_stream_run keeps the page-locked batch buffers of its last run (at most 1 GiB) for the next run of the process --
 * page-locking runs at about 1 GB/s, most of what a short file costs; this frees them. */
void td_stream_release(void);

#ifdef __cplusplus
}
#endif
#endif
