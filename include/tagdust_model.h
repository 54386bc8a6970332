/*
 * tagdust_model.h -- host-side construction of the read-architecture HMM (part of libtagdust_hip.so, plain C,
 * no GPU needed): the boundary *input* of include/tagdust_hip.h for hosts that do not carry TagDust2's own
 * struct model_bag.  Mirrors, table for table,
 *
 *   assign_segment_sequences()              src/interface.c:489-598   -> td_arch_parse
 *   get_sequence_stats()                    src/io.c:52-300           -> td_sequence_stats
 *   init_model_bag()                        src/barcode_hmm.c:5760-6011
 *     init_model_according_to_read_structure()             :4689-5084
 *     set_hmm_transition_parameters()                      :1710-1881 -> td_model_build
 *
 * (checked bit-for-bit against the reference's own tables in tests/test_model_builder.py).
 */
#ifndef TAGDUST_MODEL_H
#define TAGDUST_MODEL_H

#include <stdint.h>
#include "tagdust_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* struct read_structure (src/interface.h:82-88) */
typedef struct td_arch {
	int32_t n_segments;
	int8_t  type[TD_MAX_SEGMENTS];        /* 'B','R','P','F','S','O','G' */
	int32_t n_seq[TD_MAX_SEGMENTS];       /* numseq_in_segment, incl. the all-N decoy of B and S segments */
	int32_t seq_len[TD_MAX_SEGMENTS];     /* strlen(sequence_matrix[j][0]) */
	char**  seqs[TD_MAX_SEGMENTS];        /* sequence_matrix[j][f] */
} td_arch;

/* struct sequence_stats_info (src/io.h:97-108) */
typedef struct td_seq_stats {
	double background[5];                 /* log frequencies (float-valued, io.c:263-270) */
	double expected_5_len, expected_3_len;
	double mean_5_len, stdev_5_len, mean_3_len, stdev_3_len;
	double average_length;
	int32_t max_seq_len;
} td_seq_stats;

/* owning container of the flattened tables; desc points into it */
typedef struct td_model_tables {
	td_model_desc desc;
	void* storage;
} td_model_tables;

/* segments[k] is the argument of option -(k+1), e.g. "B:ACGT,TTGA", "R:N", "P:AGATCGGAAGAGC" */
int  td_arch_parse(const char* const* segments, int32_t n_segments, td_arch** out);
void td_arch_free(td_arch* arch);

/* Statistics over the first <= 1 000 001 reads of a file (base codes 0..4, offsets like td_batch_upload). */
int  td_sequence_stats(const td_arch* arch, const uint8_t* codes, const int64_t* offs, int64_t n_reads, td_seq_stats* out);
/* the same for a run with -start / -end (param->matchstart / matchend; -1, -1 = none): the average length becomes the
 * window's (io.c:258-260) */
int  td_sequence_stats_window(const td_arch* arch, const uint8_t* codes, const int64_t* offs, int64_t n_reads,
                              int32_t matchstart, int32_t matchend, td_seq_stats* out);

/* sequencer_error_rate = param->sequencer_error_rate (-e, default 0.05; forced to 0.05 by calibration, calibrateQ.c:65),
 * indel_frequency = param->indel_frequency (-i, default 0.1) */
int  td_model_build(const td_arch* arch, const td_seq_stats* stats, float sequencer_error_rate, float indel_frequency,
                    td_model_tables** out);
void td_model_tables_free(td_model_tables* tables);

/* ---- threshold calibration, estimateQthreshold() src/calibrateQ.c:17-235 ----
 * The reference simulates 2*(n/4) reads from the model (emit_read_sequence, src/barcode_hmm.c:2696-3046) and
 * 2*(n/4) reads from the background (emit_random_sequence, :2599-2680), scores them with run_pHMM(MODE_GET_PROB),
 * sorts by Q and picks the Q that maximises sensitivity + specificity (capped at 20).  Emission is bound to the
 * C library's rand() sequence, so it stays on the host; scoring is TD_MODE_GET_PROB on the device. */
typedef struct td_calibration {
	int64_t  n_reads;
	uint8_t* codes;               /* emitted base codes, concatenated */
	int64_t* offs;                /* [n_reads+1] */
	uint8_t* is_random;           /* [n_reads] 0 = emitted from the model, 1 = from the background */
	td_model_tables* scoring;     /* the model the reads are scored with (sequencer_error_rate forced to 0.05, calibrateQ.c:117) */
} td_calibration;

/* n_reads: 400000 in the reference (4000 in its -DRTEST builds).  rng: 0 = srand(seed)/rand() of the C library,
 * 1 = the reference's private LCG of its -DRTEST builds (src/misc.c:878-887, RAND_MAX taken as 32768).
 * With rng 0 the process-wide rand() state is reseeded, as estimateQthreshold() does (calibrateQ.c:33); on glibc the
 * numbers come from an inline copy of its generator that is checked against srand()/rand() once per process. */
int   td_calibration_emit(const td_arch* arch, const td_seq_stats* stats, float indel_frequency, uint32_t seed,
                          int32_t n_reads, int32_t rng, td_calibration** out);
/* the sort + sweep of calibrateQ.c:146-212 on the per-read Q values */
float td_calibration_select(const float* mapq, const uint8_t* is_random, int64_t n_reads);
void  td_calibration_free(td_calibration* cal);
/* emit -> td_model_upload(scoring) -> td_run(TD_MODE_GET_PROB) -> select.  Replaces the context's model and resident
 * batch: the scoring model and the calibration reads are what it holds afterwards. */
int   td_estimate_threshold(td_ctx* ctx, const td_arch* arch, const td_seq_stats* stats, float indel_frequency,
                            uint32_t seed, int32_t n_reads, int32_t rng, float* threshold);

/* ---- synthetic reads: the reference's simreads (src/simulate_reads.c:28-470, mutate() :480-560) ----
 * The bench and test inputs of the architectures simreads can emit ([5' linker][barcode] mutated + uniform read +
 * [3' linker] mutated, optional end loss, a share of fully random reads appended at the end; names "@READ<i>;SEQ:<read>;
 * RBC:<barcode>;BARNUM:<k>", qualities all 'I') on the same rand() sequence: with rng 0 the C library's (glibc: an inline
 * copy checked against srand()/rand()), with rng 1 the -DRTEST generator (misc.c:878-887).  The text equals the file
 * "simreads <tags> -seed S -sim_... -o file" writes, byte for byte.  Free it with td_text_free. */
typedef struct td_sim_params {
	uint32_t seed;            /* -seed                                   */
	int32_t  rng;             /* 0 = rand(), 1 = the RTEST generator      */
	int32_t  barnum;          /* -sim_barnum (first barnum tags are used) */
	int32_t  barlen;          /* -sim_barlen (length of the random reads only) */
	int32_t  readlen;         /* -sim_readlen                             */
	int32_t  readlen_mod;     /* -sim_readlen_mod                         */
	int32_t  numseq;          /* -sim_numseq                              */
	int32_t  end_loss;        /* -sim_endloss                             */
	float    random_frac;     /* -sim_random_frac                         */
	float    error_rate;      /* -sim_error_rate                          */
	float    indel_frac;      /* -sim_InDel_frac                          */
	const char* seq5;         /* -sim_5seq or NULL                        */
	const char* seq3;         /* -sim_3seq or NULL                        */
} td_sim_params;
int   td_simreads(const td_sim_params* p, const char* const* barcodes, int32_t n_barcodes, char** fastq_out, int64_t* len_out);
void  td_text_free(char* text);

/* ---- architecture selection, test_architectures() src/test_architectures.c:20-289 ----
 * Every candidate gets its own sequence statistics and model (error rate e, indel frequency d); the first <= 100 000
 * reads are scored with backward() alone for all candidates in one launch (td_arch_scores: generic kernel, no per-candidate
 * compile, reads staged once) and the
 * per-candidate float sums are formed over the reference's n_threads contiguous ranges in read order, then over the
 * ranges (barcode_hmm.c:2111-2148, :1995-2016) -- the float result depends on that order.  posterior[k] is the
 * normalised probability the reference logs as "Confidence"; *best the index it selects.  The context's model and
 * options are untouched; its resident batch is replaced. */
int   td_compare_architectures(td_ctx* ctx, const td_arch* const* archs, int32_t n_arch, const uint8_t* codes,
                               const int64_t* offs, int64_t n_reads, float sequencer_error_rate, float indel_frequency,
                               int32_t n_threads, float* posterior, int32_t* best);

#ifdef __cplusplus
}
#endif
#endif
