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

/* sequencer_error_rate = param->sequencer_error_rate (-e, default 0.05; forced to 0.05 by calibration, calibrateQ.c:65),
 * indel_frequency = param->indel_frequency (-i, default 0.1) */
int  td_model_build(const td_arch* arch, const td_seq_stats* stats, float sequencer_error_rate, float indel_frequency,
                    td_model_tables** out);
void td_model_tables_free(td_model_tables* tables);

#ifdef __cplusplus
}
#endif
#endif
