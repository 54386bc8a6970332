/*
 * td_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, own code) of TagDust2's per-read HMM decoding path, used as the
 * parity checker for the HIP kernels and as bench.py's `cpu_baseline` ("port").  Nothing in the
 * product path (tagdust_amd/) may include, link or call this.
 *
 * Parity status: PINNED -- checked bit-for-bit against the reference itself (oracle/_ref/ref_dump,
 * built from /root/reference/src by oracle/Makefile) through the committed fixtures under
 * tests/golden/ (tests/test_oracle_golden.py), which also cover the reference's own
 * dev/bar_read_test.sh scenarios.
 *
 * Each function cites the reference file:line it follows (paths relative to /root/reference/src).
 */
#ifndef TD_ORACLE_H
#define TD_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TDO_LOGSUM_SIZE 16000          /* misc.h:45 */
#define TDO_MAX_SEG 64                 /* barcode_hmm.h MAX_NUM_SUB_MODELS */

/* transition indices, barcode_hmm.h:87-96 */
enum { TDO_MM = 0, TDO_MI, TDO_MD, TDO_II, TDO_IM, TDO_DD, TDO_DM, TDO_MSKIP, TDO_ISKIP };

/* extraction outcomes, io.h:40-46 */
enum {
	TDO_EXTRACT_SUCCESS = 0,
	TDO_FAIL_ARCHITECTURE_MISMATCH = 1,
	TDO_FAIL_READ_TOO_SHORT = 2,
	TDO_FAIL_BAR_FINGER_NOT_FOUND = 3,
	TDO_FAIL_MATCHES_ARTIFACTS = 5,
	TDO_FAIL_LOW_COMPLEXITY = 6
};

/* Flattened read-architecture HMM (the tables struct model_bag holds, barcode_hmm.h:187-272).
 * Column index of (segment j, hmm f, column g) = col_off[j] + f*n_col[j] + g. */
typedef struct tdo_model {
	int32_t S;                 /* segments (mb->num_models) */
	int32_t H;                 /* total HMMs (mb->total_hmm_num) */
	int32_t C;                 /* total columns */
	int32_t avg_len;           /* mb->average_raw_length */
	float   bg[5];             /* model[0]->background_nuc_frequency */
	int32_t n_hmm[TDO_MAX_SEG];
	int32_t n_col[TDO_MAX_SEG];
	int32_t col_off[TDO_MAX_SEG];
	int32_t hmm_off[TDO_MAX_SEG];
	float   skip[TDO_MAX_SEG];
	int8_t  type[TDO_MAX_SEG]; /* 'B','R','P','F','S','O','G' (read_structure->type) */
	int32_t finger_len[TDO_MAX_SEG]; /* strlen(sequence_matrix[j][0]) for 'F' segments, else 0 */
	const float*   trans;      /* [C][9]  */
	const float*   eM;         /* [C][5]  */
	const float*   eI;         /* [C][5]  */
	const float*   sM;         /* [C] silent_to_M[f][g] */
	const float*   sI;         /* [C] silent_to_I[f][g] */
	const int32_t* label;      /* [H] (f<<16)|j | skippable<<31 */
	const float*   A;          /* [H][H] 0/1 label transition matrix */
} tdo_model;

typedef struct tdo_params {
	float   threshold;         /* param->confidence_threshold in effect */
	int32_t minlen;            /* param->minlen */
	int32_t dust;              /* param->dust (0 = off) */
	int32_t matchstart;        /* param->matchstart / matchend (-start / -end): the read is decoded on seq + matchstart for */
	int32_t matchend;          /* matchend - matchstart bases (barcode_hmm.c:2290-2296); both <= 0 (or -1): whole reads    */
} tdo_params;

/* per-read results */
typedef struct tdo_result {
	float  b_score, f_score, r_score;
	float  bar_prob;           /* the float the reference stores into the double ri->bar_prob */
	float  Q;                  /* ri->mapq */
	int32_t read_type, barcode, fingerprint;
} tdo_result;

typedef struct tdo_workspace tdo_workspace;

void  tdo_init_logsum(void);                          /* misc.c:57-63 */
float tdo_logsum(float a, float b);                   /* misc.c:72-78 */
const float* tdo_logsum_table(void);

tdo_workspace* tdo_workspace_new(const tdo_model* m, int max_len);
void  tdo_workspace_free(tdo_workspace* ws);

/* backward(), barcode_hmm.c:3439-3640; returns b_score */
float tdo_backward(const tdo_model* m, tdo_workspace* ws, const uint8_t* seq, int len);
/* forward_max_posterior_decoding(), barcode_hmm.c:4128-4525 (needs tdo_backward first) */
void  tdo_forward_decode(const tdo_model* m, tdo_workspace* ws, const uint8_t* seq, int len,
                         float b_score, float* f_score, float* r_score, float* bar_prob,
                         int8_t* labels /* len+1 */);
/* Q value, do_label_thread barcode_hmm.c:2320-2338 */
float tdo_qvalue(float f_score, float r_score, float bar_prob);
/* extract_reads + make_extracted_read, barcode_hmm.c:3172-3356.  seq/qual (len bytes) rewritten in place
 * on success exactly like the reference (non-read positions -> 65). qual may be NULL. */
void  tdo_extract(const tdo_model* m, const tdo_params* p, uint8_t* seq, uint8_t* qual, int len,
                  const int8_t* labels, float Q, int32_t* read_type, int32_t* barcode, int32_t* fingerprint);
/* the same with a -start/-end window (extract_reads :3189-3193): labels[1..wlen] describe seq[woff .. woff+wlen); the
 * rewrite (make_extracted_read :3325-3356) still walks the whole read with labels[j+1] on position j, and beyond the
 * window meets the zeros read_fasta_fastq() put into ri->labels (io.c:1755-1764) */
void  tdo_extract_window(const tdo_model* m, const tdo_params* p, uint8_t* seq, uint8_t* qual, int len, int woff, int wlen,
                         const int8_t* labels, float Q, int32_t* read_type, int32_t* barcode, int32_t* fingerprint);
/* dust_sequences, barcode_hmm.c:2407-2467 (one read); returns 1 if low complexity */
int   tdo_dust(const uint8_t* seq, int len, int dust_cut);

/* -ref artifacts: struct fasta as read_fasta() leaves it (io.c:1912-2001): per sequence one 'X' byte followed by the
 * base codes, s_index[n_seq+1] (sequence j = string[s_index[j] .. s_index[j+1]), the 'X' included). */
typedef struct tdo_artifacts {
	const uint8_t* string;
	const int32_t* s_index;
	int32_t n_seq;
	int32_t filter_error;      /* param->filter_error */
} tdo_artifacts;

/* match_to_reference(), barcode_hmm.c:2478-2583, over one thread's range [start, end) of already extracted reads:
 * reads are taken in fours from `start` (bmp_single, misc.c:718-765: best hit over all sequences and both strands),
 * the up-to-3 left-over reads go through bpm_check_error (misc.c:581-640: first hit).  seqs are rewritten and restored
 * by the in-place reverse complement exactly as in the reference. */
void  tdo_match_artifacts(const tdo_artifacts* a, uint8_t* seqs, const int64_t* offs, tdo_result* res,
                          int64_t start, int64_t end);

/* whole per-read path of do_label_thread (barcode_hmm.c:2269-2360) for one read */
void  tdo_label_read(const tdo_model* m, const tdo_params* p, tdo_workspace* ws,
                     uint8_t* seq, uint8_t* qual, int len, int8_t* labels, tdo_result* res);

/* run_pHMM(MODE_GET_LABEL) analogue (barcode_hmm.c:1895-2029): static contiguous split over
 * n_threads pthreads.  seqs: concatenated codes, offs[n+1]; labels laid out at offs[i]+i. */
int   tdo_label_batch(const tdo_model* m, const tdo_params* p, int n_threads,
                      uint8_t* seqs, const int64_t* offs, int64_t n_reads,
                      int8_t* labels, tdo_result* res);
/* the same with artifact matching between extraction and DUST (do_label_thread :2349-2355); art may be NULL */
int   tdo_label_batch_art(const tdo_model* m, const tdo_params* p, const tdo_artifacts* art, int n_threads,
                          uint8_t* seqs, const int64_t* offs, int64_t n_reads,
                          int8_t* labels, tdo_result* res);

/* test support for the device kernel's position pruning: smallest margins by which the host's bound tables dominate
 * the DP values of the first n_seg segments over a batch (see td_oracle.c) */
int   tdo_bound_margins(const tdo_model* m, const uint8_t* seqs, const int64_t* offs, int64_t n_reads, int n_seg, int sfx_first,
                        const float* tab, int stride, double* margins);

#ifdef __cplusplus
}
#endif
/* tools/cut_slack.py: last position with a leading-segment posterior term above floor_ per read, and the largest term per position */
int   tdo_lead_class_profile(const tdo_model* m, const uint8_t* seqs, const int64_t* offs, int64_t n_reads, int n_seg, float* out, int prof_len, int max_cls);
int   tdo_lead_profile(const tdo_model* m, const uint8_t* seqs, const int64_t* offs, int64_t n_reads, int n_seg, float floor_,
                       int32_t* last_pos, float* prof, int prof_len);
#endif
