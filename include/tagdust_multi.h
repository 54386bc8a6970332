/*
 * tagdust_multi.h -- one process, several MI355X: the static shard of a batch over the GPUs of a node and the one
 * exchange of the path (part of libtagdust_hip.so).
 *
 * The reference splits a batch over T pthreads in contiguous ranges (run_pHMM, src/barcode_hmm.c:1911-1922:
 * interval = numseq / T, thread t takes [t*interval, (t+1)*interval), the last one also the remainder) and joins them
 * before the controller combines the files of a paired / 3-read run per record index (:329-351) and print_all() writes
 * the records in input order (src/io.c:757-1016).  Here the ranges go to devices instead of threads: reads are
 * independent given the model, so there is no data-path collective -- every device decodes its range with its own context
 * (model, threshold and artifact sequences replicated) and writes its results straight into the caller's arrays at the
 * range's offset, which keeps input order.  Every input file of a multi-file run has the same number of records, so the
 * same n gives the same ranges in every file (td_shard_bounds): record i of file 1 and record i of file 2 land on the
 * same device index and at the same output index.  The only exchange is the sum of the 8 outcome + 256 per-barcode
 * counters (src/barcode_hmm.c:354-384 counts serially): one ncclAllReduce (RCCL over xGMI) on the contexts' device
 * counters when more than one device takes part, a plain read-back for one device.
 */
#ifndef TAGDUST_MULTI_H
#define TAGDUST_MULTI_H

#include <stdint.h>
#include "tagdust_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- host-side pieces (no GPU needed) ---- */
/* run_pHMM's contiguous split (barcode_hmm.c:1911-1922) of n reads over `world` parts: part `rank` is [*lo, *hi). */
void td_shard_bounds(int64_t n_reads, int32_t world, int32_t rank, int64_t* lo, int64_t* hi);
/* The device counters, restated on the host from per-read results: slot = read_type & 7 for every read with at least
 * one base (an artifact hit is (sequence << 8) | 5), then per-barcode bins (barcode & 0xFF) of the extracted reads.
 * Adds to counts[TD_NUM_COUNTERS].  lens may be NULL (all reads counted). */
void td_count_outcomes(const td_read_result* res, const int32_t* lens, int64_t n_reads, int64_t* counts);

/* Diagnostic (tools/host_scale.py): the HOST halves of td_submit / td_wait for one batch of n_reads reads of read_len bases, run
 * `iters` times back to back on n_threads threads WITHOUT a device -- the staging copy into page-locked memory, the records copy,
 * the rebuilding of rewritten sequences from keep bits, the expansion of labels from runs: the very routines the library runs
 * around its device calls.  mode 0: pageable caller buffers; 1: page-locked caller buffers under "stable_input".  out[0] = seconds
 * per batch, out[1] = host bytes read + written per batch, out[2] = batches timed.  N processes of this side by side show what N
 * ranks' host halves cost one host -- the ceiling the host sets for the 8-GPU run. */
int td_host_halves_bench(int64_t n_reads, int32_t read_len, int32_t n_threads, int32_t iters, int32_t mode, double* out /* [3] */);

/* Bind the calling host thread to the CPUs of the NUMA node next to `device` (as far as the process may use them); returns
 * the node, or -1 when it is unknown / none of its CPUs is available (the thread then stays where it is).  td_multi's
 * per-device threads call it; a one-process-per-GPU launcher calls it once per rank before it allocates host buffers. */
int32_t td_bind_host_to_device(int32_t device);

/* ---- several devices driven from one process ---- */
typedef struct td_multi td_multi;
/* devices[n_devices] = HIP device indices (NULL: 0 .. n_devices-1).  One context and one host thread per device (started
 * here, reused by every call, bound to the device's NUMA node unless TD_MULTI_BIND=0); the machine's host threads are
 * shared out over the contexts' copy pools.  With n_devices > 1 distinct devices an RCCL communicator is created
 * (librccl.so is loaded then, not before); TD_MULTI_FORCE_RCCL=1 creates one for a single device too, so that the
 * collective path can be exercised on a one-GPU box. */
int  td_multi_create(const int32_t* devices, int32_t n_devices, td_multi** out);
void td_multi_destroy(td_multi* m);
const char* td_multi_last_error(const td_multi* m);   /* m == NULL: the failed td_multi_create */
int32_t td_multi_size(const td_multi* m);
td_ctx* td_multi_ctx(td_multi* m, int32_t k);          /* the k-th device's context (options, per-device inspection) */
/* replicate model / parameters / artifact sequences on every device (uploads and kernel compiles run concurrently) */
int  td_multi_model_upload(td_multi* m, const td_model_desc* model);
int  td_multi_set_params(td_multi* m, float threshold, int32_t minlen, int32_t dust);
int  td_multi_set_window(td_multi* m, int32_t matchstart, int32_t matchend);   /* -start / -end on every device (td_set_window) */
int  td_multi_set_artifacts(td_multi* m, const uint8_t* string, const int32_t* s_index, int32_t n_seq,
                            int32_t filter_error, int32_t n_threads);
/* One run_pHMM call over all devices: reads [0, n) are split with td_shard_bounds, every device runs td_submit / td_wait
 * on its range, results arrive in input order (res / labels / seq_out as in td_batch_download; any may be NULL).
 * Per-read results are identical to a single context's: the artifact filter's thread ranges are taken over the whole
 * batch, not over a device's share. */
int  td_multi_decode(td_multi* m, const void* bases, int32_t is_ascii, const int64_t* offs, int64_t n_reads, int mode,
                     td_read_result* res, int8_t* labels, uint8_t* seq_out);
/* Counters summed over the devices (all-reduced on the devices with RCCL when the communicator exists). */
int  td_multi_counts(td_multi* m, int64_t* counts /* [TD_NUM_COUNTERS] */);
int  td_multi_counts_reset(td_multi* m);
/* 1 when td_multi_counts goes through ncclAllReduce, 0 when it sums on the host (one device without TD_MULTI_FORCE_RCCL,
 * or a device listed twice) */
int32_t td_multi_uses_rccl(const td_multi* m);

#ifdef __cplusplus
}
#endif
#endif
