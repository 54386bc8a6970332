/*
 * tagdust_hip.h -- C-ABI of libtagdust_hip.so: TagDust2's per-read HMM decoding path on MI355X (gfx950).
 *
 * This is the drop-in boundary for the reference's
 *
 *     int run_pHMM(struct arch_bag* ab, struct model_bag* mb, struct read_info** ri,
 *                  struct parameters* param, struct fasta* reference_fasta, int numseq, int mode);
 *                                                   (src/barcode_hmm.h:342, src/barcode_hmm.c:1895-2029)
 *
 * i.e. the pthread fan-out over do_label_thread / do_probability_estimation (barcode_hmm.c:2174-2360), whose
 * per-read body is backward() (:3439) -> forward_max_posterior_decoding() (:4128) -> Q value (:2320-2338) ->
 * extract_reads() (:3172) -> match_to_reference() (:2478, with -ref) -> dust_sequences() (:2407).  Plain C types only; every entry point returns
 * TD_OK (0) / TD_FAIL (1) like the reference's kslOK / kslFAIL (src/kslib.h:12-16) and never calls exit().
 * How a reference maintainer binds it is shown in INTEGRATION.md.
 *
 * Flow:  td_ctx_create -> td_model_upload -> td_set_params [-> td_set_artifacts]
 *        per batch: td_batch_upload (host reads) -> td_run -> td_batch_download          (one batch resident at a time)
 *               or: td_submit (reads in, result buffers named) ... td_wait               (pipelined: the copies of the
 *                   neighbouring batches overlap the decode kernel of the current one)
 *        end of run: td_counts_get (per-outcome / per-barcode counters, the input of the RCCL all-reduce)
 *
 * The host only moves contiguous bytes: base coding, the sort of the reads by length, 2-bit packing and, on the way back,
 * the per-read records, the rewritten sequences and the labels in the caller's order are all produced on the device.
 * A context is not thread-safe: drive it from one host thread (one context per GPU, like one model copy per pthread).
 */
#ifndef TAGDUST_HIP_H
#define TAGDUST_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TD_OK   0   /* kslOK   */
#define TD_FAIL 1   /* kslFAIL */

#define TD_MAX_SEGMENTS 64    /* MAX_NUM_SUB_MODELS, src/barcode_hmm.h:101 */
#define TD_MAX_HMMS     127   /* labels are signed char (src/io.h:80); the reference itself stops at 100 (barcode_hmm.c:4186) */
#define TD_LOGSUM_SIZE  16000 /* src/misc.h:45 */

/* run modes, src/barcode_hmm.h:128-132 */
#define TD_MODE_GET_LABEL 1   /* label + Q + extraction (+ DUST)           do_label_thread            */
#define TD_MODE_GET_PROB  4   /* Q only (threshold calibration)            do_probability_estimation  */
#define TD_MODE_ARCH_COMP 5   /* backward() only: b_score per read         do_arch_comparison         */

/* extraction outcomes, src/io.h:40-46 */
#define TD_EXTRACT_SUCCESS                    0
#define TD_EXTRACT_FAIL_ARCHITECTURE_MISMATCH 1
#define TD_EXTRACT_FAIL_READ_TOO_SHORT        2
#define TD_EXTRACT_FAIL_BAR_FINGER_NOT_FOUND  3
#define TD_EXTRACT_FAIL_MATCHES_ARTIFACTS     5
#define TD_EXTRACT_FAIL_LOW_COMPLEXITY        6

/* transition indices of td_model_desc.trans, src/barcode_hmm.h:87-96 */
enum { TD_MM = 0, TD_MI = 1, TD_MD = 2, TD_II = 3, TD_IM = 4, TD_DD = 5, TD_DM = 6, TD_MSKIP = 7, TD_ISKIP = 8 };

/* counters returned by td_counts_get: slot = outcome code 0..7, then 256 per-barcode success bins
 * (index = barcode & 0xFF like the writer's file index, src/io.c:930) */
#define TD_NUM_OUTCOME_SLOTS 8
#define TD_NUM_BARCODE_BINS  256
#define TD_NUM_COUNTERS (TD_NUM_OUTCOME_SLOTS + TD_NUM_BARCODE_BINS)

typedef struct td_ctx td_ctx;

/* Flattened read-architecture HMM: exactly the tables struct model_bag holds after init_model_bag()
 * (src/barcode_hmm.h:187-272, barcode_hmm.c:5760-6011).  All floats are natural-log probabilities, log(0) = -inf.
 * Column index of (segment j, HMM f, column g) = sum_{j'<j} n_hmm[j']*n_col[j'] + f*n_col[j] + g. */
typedef struct td_model_desc {
	int32_t S;                  /* mb->num_models                                      */
	int32_t H;                  /* mb->total_hmm_num        (<= TD_MAX_HMMS)           */
	int32_t C;                  /* total number of columns                             */
	int32_t avg_len;            /* mb->average_raw_length                              */
	float   bg[5];              /* model[0]->background_nuc_frequency                  */
	const int32_t* n_hmm;       /* [S] model[j]->num_hmms                              */
	const int32_t* n_col;       /* [S] model[j]->hmms[0]->num_columns                  */
	const float*   skip;        /* [S] model[j]->skip                                  */
	const int8_t*  seg_type;    /* [S] read_structure->type[j]: 'B','R','P','F','S','O','G' */
	const int32_t* finger_len;  /* [S] strlen(sequence_matrix[j][0]) for 'F' segments (extract_reads :3196-3200), else 0 */
	const float*   trans;       /* [C][9] hmm_column->transition                       */
	const float*   eM;          /* [C][5] hmm_column->m_emit                           */
	const float*   eI;          /* [C][5] hmm_column->i_emit                           */
	const float*   sM;          /* [C]    model->silent_to_M[f][g]                     */
	const float*   sI;          /* [C]    model->silent_to_I[f][g]                     */
	const int32_t* label;       /* [H]    mb->label                                    */
	const float*   A;           /* [H][H] mb->transition_matrix (0/1)                  */
} td_model_desc;

/* Per-read outputs (what run_pHMM leaves in struct read_info, src/io.h:76-91) */
typedef struct td_read_result {
	float   f_score;            /* mb->f_score                                          */
	float   b_score;            /* mb->b_score                                          */
	float   r_score;            /* mb->r_score                                          */
	float   bar_prob;           /* ri->bar_prob before it is reset to 100 (:2343)       */
	float   mapq;               /* ri->mapq  (Q)                                        */
	int32_t read_type;          /* ri->read_type                                        */
	int32_t barcode;            /* ri->barcode     ((segment << 16) | hmm), -1 if none  */
	int32_t fingerprint;        /* ri->fingerprint ((key << 8) | len),      -1 if none  */
} td_read_result;

/* ---- context (one per GPU; replaces run_pHMM's per-thread model copies, barcode_hmm.c:1911-1922) ---- */
int  td_ctx_create(int device, td_ctx** out);
void td_ctx_destroy(td_ctx* ctx);
/* last error text of this context (or of the failed td_ctx_create when ctx == NULL) */
const char* td_last_error(const td_ctx* ctx);

/* init_logsum() (src/misc.c:57-63) happens inside td_ctx_create; this returns the 16000-entry host copy */
const float* td_logsum_table(void);

/* ---- model + run parameters ---- */
/* Uploads the tables and (option "specialize", default 1) compiles the decode kernel specialised for this
 * model with hiprtc -- a few seconds, once per architecture.  A compile failure is TD_FAIL, not a fallback. */
int td_model_upload(td_ctx* ctx, const td_model_desc* model);
/* Options:  "specialize" (set before td_model_upload) 1 = model-specialised kernel (default; env TD_SPECIALIZE),
 * 0 = the generic ahead-of-time kernel that reads the model from HBM;  "pipeline_depth" 1..4 (default 3) = batches
 * td_submit may hold in flight;  "host_threads" 1..16 (default: env TD_HOST_THREADS, else the machine's threads, at most 16) =
 * host threads of this context for its copies between pageable caller memory and pinned staging (started once, reused);  "overlap_decode" 1 (default; env TD_OVERLAP) = consecutive td_submit batches run their
 * decode kernels on two streams with a workspace each, so that one batch's kernel starts while the last one's slowest waves
 * finish (falls back to one stream when device memory cannot hold two workspaces);  "poison_workspace" 1 = fill the HBM workspace with 0xFF bytes before every decode launch
 * (tests: a kernel that reads workspace bytes it has not written in this launch then computes on NaNs);
 * "stable_input" 1 = the caller promises to leave a PAGE-LOCKED `bases` buffer handed to td_submit untouched until td_wait(ticket)
 * has returned (default 0: it may be refilled as soon as td_submit returns).  With the promise td_submit does not wait for the upload
 * and the rewritten sequences come back as keep bits that the host applies to the caller's own buffer (the compact egress) instead
 * of as full device-side copies: the page-locked path then does strictly less host work and moves fewer bytes than the pageable one;
 * "compact_egress" 1 (default; env TD_COMPACT_EGRESS) = rewritten sequences travel as keep bits, labels as runs;
 * "length_classes_enabled" 1 (default; env TD_NO_LENGTH_CLASSES turns it off), "rle_cap", "debug_wait", "spec_lsum_limit" = test and
 * diagnosis knobs.  Every environment variable is read once, when the context is created. */
int td_set_option(td_ctx* ctx, const char* name, int32_t value);
/* Read a setting back: "specialize", "pipeline_depth", "overlap_decode", or "spec_lsum_clamped" (1 when the loaded specialised kernel uses the clamped
 * logsum: the clamp-free form is only selected while model parameters x read length bound every score difference).
 * Which fast paths a model / batch actually got: "prune_active" (1 when the loaded specialised kernel prunes the leading /
 * trailing segments by position and the bound tables for the last batch's read lengths are live; 0 for the generic kernel,
 * for models without a read segment behind a bounded prefix, for reads beyond 8192 bases, before the first batch) and
 * "overlap_active" (1 when pipelined batches really alternate between two compute streams and workspaces; 0 when the option
 * is off, the pipeline is one deep, the generic kernel runs, or HBM could not hold the second workspace);
 * "artifacts_active" (1 while a -ref artifact filter is set, td_set_artifacts); "length_classes" (the number of wave slots
 * that were laid out for the last batch's longest read while the others kept the geometry of the reads at the 99 % mark --
 * a batch with a few very long reads among many short ones; 0: one geometry);
 * "hw_queues" (hardware queues of the HIP runtime as far as the library can tell: the user's GPU_MAX_HW_QUEUES, else the 8 the
 * library asks for when it is loaded BEFORE the process's first HIP call, else the runtime's default 4) and "hw_queues_late" (1: the
 * runtime was already initialised when the library was loaded, so its request had no effect -- set GPU_MAX_HW_QUEUES=8 in the
 * environment instead; the pipelined calls keep six streams busy). */
int td_get_option(td_ctx* ctx, const char* name, int32_t* value);
/* The HIP source td_model_upload would compile for this model (no GPU needed).  Returns its length; copies at
 * most cap-1 bytes + NUL into buf when buf != NULL. */
int64_t td_spec_source(const td_model_desc* model, char* buf, int64_t cap);
/* Position pruning of the specialised kernel: *n_seg = the number of leading segments whose forward
 * sweep may stop early (0: none), *sfx_first = the first trailing segment whose backward sweep may stop early (S: none), and
 * -- when tab != NULL -- the eight bound tables of lcap + 8 floats each the kernel decides with (forward bound per position,
 * backward bound per bases to go, the two check thresholds of the adjoining read segment; once for each end); *z = the
 * zero-posterior margin.  No GPU needed; for inspection and tests. */
int td_spec_prune_info(const td_model_desc* model, int32_t lcap, float* tab, float* z, int32_t* n_seg, int32_t* sfx_first);
/* The impulse-response tables the restarted sweeps of the specialised kernel start from: tab[8][lcap + 8]
 * = leading segments 0..3, then the first four trailing segments; *restart = 1 when the kernel compiled for this model restarts
 * its far sweeps (big leading segments; TD_SPEC_RESTART=0 / 1 forces it).  No GPU needed; for inspection and tests. */
int td_spec_restart_info(const td_model_desc* model, int32_t lcap, float* tab, int32_t* restart);
/* -ref artifact filter, match_to_reference() src/barcode_hmm.c:2478-2583 (runs between extraction and DUST in
 * TD_MODE_GET_LABEL): string / s_index[n_seq+1] are struct fasta's fields as read_fasta() leaves them (io.c:1912-2001:
 * per sequence one 'X' byte followed by the base codes); filter_error = param->filter_error (-fe);
 * n_threads = param->num_threads -- the reference pairs reads in fours from the start of each thread's range and
 * scores the up-to-3 left-over reads of a range with a different routine, which is reproduced.  A matching read gets
 * read_type = (1-based sequence index << 8) | 5.  n_seq = 0 switches the filter off. */
int td_set_artifacts(td_ctx* ctx, const uint8_t* string, const int32_t* s_index, int32_t n_seq,
                     int32_t filter_error, int32_t n_threads);
/* -start / -end (param->matchstart, param->matchend; -1, -1 = none): do_probability_estimation / do_label_thread then
 * decode seq + matchstart for matchend - matchstart bases (src/barcode_hmm.c:2195-2210, :2290-2313).  With a window set,
 * td_run works on it in every mode and reproduces what the reference does with the window's labels afterwards:
 * extract_reads walks them (:3189-3193, fingerprint bases from seq[j + matchstart]); make_extracted_read (:3325-3356)
 * puts label j+1 on position j of the WHOLE read and meets ri->labels' initial zeros beyond the window (io.c:1755-1764),
 * i.e. HMM 0 of segment 0; the artifact filter and DUST see that rewritten whole read; labels beyond the window are 0.
 * The reference reads past the end of a read shorter than matchend (undefined); here such a read is decoded on what it
 * has inside the window. */
int td_set_window(td_ctx* ctx, int32_t matchstart, int32_t matchend);
/* The batches this context gets from now on are reads [first_read, first_read + n) of a batch of total_reads reads that is
 * shared out over several contexts (tagdust_multi.h): the artifact filter's thread ranges are then taken over the whole
 * batch, so that every read is scored by the routine the reference would use for it.  total_reads = 0: a batch is whole. */
int td_set_batch_window(td_ctx* ctx, int64_t first_read, int64_t total_reads);
/* param->confidence_threshold in effect, param->minlen, param->dust (0 = off) */
int td_set_params(td_ctx* ctx, float threshold, int32_t minlen, int32_t dust);

/* ---- batches ---- */
/* Stage a batch of reads: codes = base codes 0..4 (A,C,G,T,other: src/nuc_code.c:46-74) of all reads
 * concatenated, offs[n_reads+1] the read boundaries (like ri[i]->seq / ri[i]->len; read i is codes[offs[i] .. offs[i+1])).
 * One host-to-device copy of the bytes as they are; sorting by length and packing to 2 bit + N mask happen on the device.
 * Replaces whatever batch was resident; returns when the caller's buffers may be reused. */
int td_batch_upload(td_ctx* ctx, const uint8_t* codes, const int64_t* offs, int64_t n_reads);
/* Same from ASCII FASTQ sequence lines (applies the nuc_code mapping). */
int td_batch_upload_ascii(td_ctx* ctx, const char* bases, const int64_t* offs, int64_t n_reads);
/* Run the hot path over the resident batch on the context's stream (asynchronous; td_sync / td_batch_download
 * wait).  mode = TD_MODE_GET_LABEL, TD_MODE_GET_PROB or TD_MODE_ARCH_COMP (only td_read_result.b_score is then
 * meaningful).  TD_MODE_GET_LABEL adds this batch's outcomes to the counters. */
int td_run(td_ctx* ctx, int mode);
int td_sync(td_ctx* ctx);
/* Copy results of the resident batch back.  Any pointer may be NULL.
 *   res      [n_reads]
 *   labels   read i at offs[i]+i, len_i+1 bytes (ri->labels, barcode_hmm.c:4503-4514)
 *   seq_out  read i at offs[i], len_i bytes: the sequence as extract_reads() rewrites it
 *            (codes 0..4, non-read positions 65; barcode_hmm.c:3325-3356); untouched codes when not extracted */
int td_batch_download(td_ctx* ctx, td_read_result* res, int8_t* labels, uint8_t* seq_out);

/* ---- pipelined batches: one run_pHMM call per batch (barcode_hmm.c:322), several batches in flight ---- */
/* Hand over a batch and name where its results go; returns once the reads have left the caller's buffers (they may be
 * reused) with the upload, the decode kernel (mode as td_run; parameters, model, window and artifact filter as set at
 * this moment) and the device-side reordering of its results queued; td_wait issues the download when they are ready.  bases: base codes 0..4, or FASTQ sequence text when is_ascii != 0.
 * res / labels / seq_out as in td_batch_download (any may be NULL); they are complete after td_wait(ticket).  offs[0]
 * need not be 0: read i is bases[offs[i] .. offs[i+1]) and output positions count from offs[0] (seq_out + offs[i] - offs[0],
 * labels + offs[i] - offs[0] + i), so a contiguous range of a larger batch can be handed over with its own offsets.
 * At most "pipeline_depth" tickets may be outstanding (TD_FAIL beyond that).  Page-locked buffers (td_host_alloc, or
 * registered with hipHostRegister) are read and written by the DMA engines directly -- td_submit then waits for the upload
 * of `bases` before it returns, so the contract above holds for them too; any other host memory goes through
 * the library's own pinned staging with one extra host copy each way (TD_HOST_THREADS host threads, default all, <= 16).
 * The result buffers belong to the library until td_wait(ticket) returns. */
int td_submit(td_ctx* ctx, const void* bases, int32_t is_ascii, const int64_t* offs, int64_t n_reads, int mode,
              td_read_result* res, int8_t* labels, uint8_t* seq_out, int64_t* ticket);
/* Block until the batch of this ticket is complete and its results are in the buffers named at td_submit. */
int td_wait(td_ctx* ctx, int64_t ticket);
/* Page-locked host memory for batch inputs / outputs (NULL on failure); portable: every device of the process may DMA
 * from / to it (td_multi_decode hands ranges of one caller array to several devices). */
void* td_host_alloc(size_t bytes);
void  td_host_free(void* p);

/* ---- architecture comparison (TD_MODE_ARCH_COMP for many models at once) ----
 * test_architectures() hands run_pHMM every candidate model bag and the first batch of reads (src/test_architectures.c:
 * 182-184); each thread then runs backward() alone for every candidate over its reads (do_arch_comparison,
 * src/barcode_hmm.c:2111-2148).  Here: the reads are staged once, every candidate's tables go to HBM, and ONE launch of
 * the generic kernel (no per-candidate compile) scores all of them -- the launch's second grid dimension is the
 * candidate.  b_scores[k * n_reads + i] = backward score of read i under models[k] (mb->b_score), in the caller's
 * order.  The context's own model, parameters and counters are untouched; its resident batch is replaced.  Whole reads are
 * scored: a window set with td_set_window is not applied here. */
int td_arch_scores(td_ctx* ctx, const td_model_desc* const* models, int32_t n_models, const uint8_t* codes,
                   const int64_t* offs, int64_t n_reads, float* b_scores /* [n_models][n_reads] */);

/* ---- counters (the reference's serial outcome counting, barcode_hmm.c:354-384, done on device) ---- */
int td_counts_reset(td_ctx* ctx);
int td_counts_get(td_ctx* ctx, int64_t* counts /* [TD_NUM_COUNTERS] */);
/* The diagnostic words of the development knobs (TD_SPEC_PROFILE, TD_SPEC_PRUNE_STATS, TD_SPEC_ENDSTATS): a tail of their
 * own behind the counters above, so that no knob can disturb what td_counts_get / the all-reduce report.  Reset with the
 * counters.  diag[k] = historical slot 192 + k of DESIGN.md's knob table. */
#define TD_NUM_DIAG_COUNTERS 64
int td_diag_get(td_ctx* ctx, int64_t* diag /* [TD_NUM_DIAG_COUNTERS] */);
/* device pointer to the int64 counters, for an in-place RCCL all-reduce over xGMI */
void* td_counts_device_ptr(td_ctx* ctx);

/* ---- measurement hooks (bench.py) ---- */
/* Milliseconds the decode kernel of the last td_run -- or of the batch last td_wait'ed for -- took, from HIP events
 * recorded around it on the stream it was launched on.  With "overlap_decode" the span of a pipelined batch's kernel includes
 * the time it waited for compute units while the previous batch's kernel drained; td_run gives the launch's own duration. */
int td_last_kernel_ms(td_ctx* ctx, float* ms);
/* A device-side timeline of the decode launches, so that a caller can show how pipelined launches follow one another without
 * a tracer: td_timeline_origin records the origin (an event on the context's first compute stream, waited for);
 * td_last_kernel_times then gives, for the batch last td_run / td_wait'ed for, the milliseconds from that origin to the HIP
 * events recorded on the batch's compute stream right before and right after its decode kernel, and which of the two
 * compute streams / workspaces it ran on (0 / 1).  start is the moment the stream reaches the launch (the previous kernel on
 * that stream and the batch's pack kernel are done); with "overlap_decode" the kernel's first waves may still wait for the
 * other stream's kernel to free compute units. */
int td_timeline_origin(td_ctx* ctx);
int td_last_kernel_times(td_ctx* ctx, float* start_ms, float* stop_ms, int32_t* stream_index);
/* Number of reads resident, HBM workspace bytes, wave slots in use. */
int td_batch_info(td_ctx* ctx, int64_t* n_reads, int64_t* workspace_bytes, int32_t* wave_slots);


#ifdef __cplusplus
}
#endif
#endif /* TAGDUST_HIP_H */
