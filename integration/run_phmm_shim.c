/*
 * run_phmm_shim.c -- the reference-side binding of libtagdust_hip.so: a drop-in definition of
 *
 *     int run_pHMM(struct arch_bag* ab, struct model_bag* mb, struct read_info** ri, struct parameters* param,
 *                  struct fasta* reference_fasta, int numseq, int mode);        (src/barcode_hmm.h:342)
 *
 * that a TagDust2 maintainer links instead of the pthread fan-out in src/barcode_hmm.c:1895-2029.  It is own
 * code written against the reference's public structs (barcode_hmm.h, io.h, interface.h); it flattens the live
 * struct model_bag into a td_model_desc, hands the batch to the GPU through the C-ABI (include/tagdust_hip.h) and
 * writes the results back into struct read_info exactly where do_label_thread / do_probability_estimation leave
 * them (mapq, labels, read_type, barcode, fingerprint, seq/qual rewritten in place, bar_prob = 100).
 *
 * Every mode the live reference calls (MODE_GET_LABEL, MODE_GET_PROB, MODE_ARCH_COMP), with and without -start/-end windows,
 * runs on the GPU.  What is left for the reference's own CPU implementation -- which the build recipe keeps available as
 * ref_run_pHMM() (oracle/Makefile) -- are the training modes (dead code in v2.33, unreachable from its CLI) and whatever
 * mode TAGDUST_HIP_DELEGATE=<mode>[,<mode>] names (a debugging aid: e.g. "4" runs the threshold calibration on the CPU and
 * the labelling on the GPU inside one binary).  Every hand-over is counted, and at exit the shim reports
 *     tagdust_hip: batches gpu=<n> delegated=<m>
 * on stderr when TAGDUST_HIP_REPORT=1 or TAGDUST_HIP_STRICT=1 is set, or when a batch was handed over (otherwise the binary's
 * stderr is the reference's own).  TAGDUST_HIP_STRICT=1 turns a hand-over into an error (run_pHMM returns kslFAIL with a message), so that a
 * test -- or a user -- can be sure that every result came from the GPU.
 *
 * Inputs on which the reference itself has no defined behaviour are not handed to it: a -start/-end window with
 * start < 0 or end <= start (the reference decodes a negative length and crashes) is refused with a message, and a read
 * shorter than -end (the reference reads past its end and corrupts its heap: "munmap_chunk(): invalid pointer") is decoded
 * on the bases it has inside the window, like td_set_window documents.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <math.h>

#include "kslib.h"
#include "interface.h"
#include "io.h"
#include "barcode_hmm.h"
#include "misc.h"

#include "tagdust_hip.h"
#include "tagdust_multi.h"

int ref_run_pHMM(struct arch_bag* ab, struct model_bag* mb, struct read_info** ri, struct parameters* param,
                 struct fasta* reference_fasta, int numseq, int mode);

static td_ctx* g_ctx = NULL;
static td_multi* g_multi = NULL;   /* TAGDUST_HIP_DEVICES=0,1,...: the batches are shared out over these devices (tagdust_multi.h) */
static uint64_t g_model_key = 0;
static long g_batches_gpu = 0, g_batches_delegated = 0;
static int g_report_registered = 0;

/* The drop-in binary's stderr stays the reference's own unless somebody asked (TAGDUST_HIP_REPORT=1, TAGDUST_HIP_STRICT=1) or a
 * batch was in fact handed to the CPU path -- which a user should hear about. */
static void report_at_exit(void)
{
	const char* rep = getenv("TAGDUST_HIP_REPORT");
	const char* strict = getenv("TAGDUST_HIP_STRICT");
	if ((rep && atoi(rep) != 0) || (strict && atoi(strict) != 0) || g_batches_delegated > 0)
		fprintf(stderr, "tagdust_hip: batches gpu=%ld delegated=%ld\n", g_batches_gpu, g_batches_delegated);
}

static void ensure_report(void)
{
	if (!g_report_registered) { g_report_registered = 1; atexit(report_at_exit); }
}

/* a batch the GPU path does not take: the reference's own code -- or, with TAGDUST_HIP_STRICT=1, an error */
static int delegate(const char* why, struct arch_bag* ab, struct model_bag* mb, struct read_info** ri, struct parameters* param,
                    struct fasta* reference_fasta, int numseq, int mode)
{
	const char* strict = getenv("TAGDUST_HIP_STRICT");
	ensure_report();
	if (strict && atoi(strict) != 0) {
		fprintf(stderr, "tagdust_hip: TAGDUST_HIP_STRICT: refusing to hand a batch of %d reads (mode %d) to the CPU path: %s\n", numseq, mode, why);
		return kslFAIL;
	}
	g_batches_delegated++;
	return ref_run_pHMM(ab, mb, ri, param, reference_fasta, numseq, mode);
}

/* one context per device listed in TAGDUST_HIP_DEVICES (default: "0") behind a td_multi; context 0 of it also serves the
 * calls that run on one device only: architecture comparison, windowed scores */
static int ensure_context(void)
{
	if (g_ctx) return TD_OK;
	const char* e = getenv("TAGDUST_HIP_DEVICES");
	/* Default: GPU 0 behind a td_multi as well -- td_multi_decode puts a batch through the pipelined calls in pieces, so that
	 * within the one synchronous run_pHMM call the copies of one piece run beside the decode kernel of another (a 2^20-read
	 * batch with labels: 32 ms instead of 45).  TAGDUST_HIP_SYNC=1: one context, td_batch_upload / td_run / td_batch_download. */
	const char* sync = getenv("TAGDUST_HIP_SYNC");
	if ((!e || !*e) && !(sync && atoi(sync) != 0)) e = "0";
	if (e && *e) {
		int32_t dev[64];
		int n = 0, k;
		while (*e) {
			char* end = NULL;
			const long v = strtol(e, &end, 10);
			if (end == e || v < 0 || v > 1023 || n >= 64 || (*end && *end != ',' && *end != ' ')) {
				fprintf(stderr, "tagdust_hip: TAGDUST_HIP_DEVICES: cannot read a device index at \"%s\" (expected e.g. 0,1,2; at most 64 entries)\n", e);
				return TD_FAIL;
			}
			/* (TAGDUST_HIP_ALLOW_DUPLICATE_DEVICES=1: a testing aid -- several contexts on one GPU exercise the split and the merge) */
			for (k = 0; k < n; k++) if (dev[k] == (int32_t)v && !getenv("TAGDUST_HIP_ALLOW_DUPLICATE_DEVICES")) {
				fprintf(stderr, "tagdust_hip: TAGDUST_HIP_DEVICES lists device %ld twice\n", v);
				return TD_FAIL;
			}
			dev[n++] = (int32_t)v;
			e = end;
			while (*e == ',' || *e == ' ') e++;
		}
		if (n == 0) { fprintf(stderr, "tagdust_hip: TAGDUST_HIP_DEVICES names no device\n"); return TD_FAIL; }
		if (td_multi_create(dev, n, &g_multi) != TD_OK) { fprintf(stderr, "tagdust_hip: %s\n", td_multi_last_error(NULL)); return TD_FAIL; }
		g_ctx = td_multi_ctx(g_multi, 0);
		return TD_OK;
	}
	if (td_ctx_create(0, &g_ctx) != TD_OK) { fprintf(stderr, "tagdust_hip: %s\n", td_last_error(NULL)); return TD_FAIL; }
	return TD_OK;
}

/* FNV-1a over everything that defines the model, so a rebuilt model_bag (calibration, "long sequence" realloc)
 * is re-uploaded only when its tables changed */
static uint64_t fnv(uint64_t h, const void* p, size_t n)
{
	const unsigned char* b = (const unsigned char*)p;
	for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ULL; }
	return h;
}

/* struct model_bag -> td_model_desc; the tables are malloc'ed, flat_free() releases them */
typedef struct flat_model { td_model_desc d; void* mem[11]; } flat_model;

static void flat_free(flat_model* fm)
{
	int k;
	for (k = 0; k < 11; k++) free(fm->mem[k]);
	memset(fm, 0, sizeof *fm);
}

static void flatten_model(struct model_bag* mb, struct parameters* param, int with_types, flat_model* fm)
{
	const int S = mb->num_models, H = mb->total_hmm_num;
	int C = 0, j, f, g, k;
	for (j = 0; j < S; j++) C += mb->model[j]->num_hmms * mb->model[j]->hmms[0]->num_columns;

	int32_t* n_hmm = malloc(sizeof(int32_t) * S);
	int32_t* n_col = malloc(sizeof(int32_t) * S);
	int32_t* finger = malloc(sizeof(int32_t) * S);
	float* skip = malloc(sizeof(float) * S);
	int8_t* type = malloc(S);
	float* trans = malloc(sizeof(float) * C * 9);
	float* eM = malloc(sizeof(float) * C * 5);
	float* eI = malloc(sizeof(float) * C * 5);
	float* sM = malloc(sizeof(float) * C);
	float* sI = malloc(sizeof(float) * C);
	float* A = malloc(sizeof(float) * H * H);
	int c = 0;
	for (j = 0; j < S; j++) {
		struct model* m = mb->model[j];
		n_hmm[j] = m->num_hmms;
		n_col[j] = m->hmms[0]->num_columns;
		skip[j] = m->skip;
		/* segment types only matter for extraction; the candidate models of an architecture comparison are scored
		 * by backward() alone and param->read_structure does not describe them */
		type[j] = with_types ? param->read_structure->type[j] : 'R';
		finger[j] = (type[j] == 'F') ? (int32_t)strlen(param->read_structure->sequence_matrix[j][0]) : 0;
		for (f = 0; f < m->num_hmms; f++) {
			for (g = 0; g < m->hmms[f]->num_columns; g++, c++) {
				struct hmm_column* col = m->hmms[f]->hmm_column[g];
				for (k = 0; k < 9; k++) trans[c * 9 + k] = col->transition[k];
				for (k = 0; k < 5; k++) { eM[c * 5 + k] = col->m_emit[k]; eI[c * 5 + k] = col->i_emit[k]; }
				sM[c] = m->silent_to_M[f][g];
				sI[c] = m->silent_to_I[f][g];
			}
		}
	}
	for (j = 0; j < H; j++)
		for (k = 0; k < H; k++) A[j * H + k] = mb->transition_matrix[j][k];

	td_model_desc* d = &fm->d;
	memset(fm, 0, sizeof *fm);
	d->S = S; d->H = H; d->C = C; d->avg_len = mb->average_raw_length;
	for (k = 0; k < 5; k++) d->bg[k] = mb->model[0]->background_nuc_frequency[k];
	d->n_hmm = n_hmm; d->n_col = n_col; d->skip = skip; d->seg_type = type; d->finger_len = finger;
	d->trans = trans; d->eM = eM; d->eI = eI; d->sM = sM; d->sI = sI; d->label = mb->label; d->A = A;
	void* mem[11] = { n_hmm, n_col, finger, skip, type, trans, eM, eI, sM, sI, A };
	memcpy(fm->mem, mem, sizeof mem);
}

static int upload_model(struct model_bag* mb, struct parameters* param, int with_types)
{
	flat_model fm;
	flatten_model(mb, param, with_types, &fm);
	const td_model_desc* d = &fm.d;
	const int S = d->S, H = d->H, C = d->C;
	uint64_t key = 1469598103934665603ULL;
	key = fnv(key, &d->S, 16); key = fnv(key, d->bg, 20);
	key = fnv(key, d->n_hmm, 4 * S); key = fnv(key, d->n_col, 4 * S); key = fnv(key, d->skip, 4 * S); key = fnv(key, d->seg_type, S);
	key = fnv(key, d->trans, 36 * C); key = fnv(key, d->eM, 20 * C); key = fnv(key, d->eI, 20 * C);
	key = fnv(key, d->sM, 4 * C); key = fnv(key, d->sI, 4 * C); key = fnv(key, d->label, 4 * H); key = fnv(key, d->A, 4 * H * H);
	int rc = TD_OK;
	if (key != g_model_key) {
		rc = g_multi ? td_multi_model_upload(g_multi, d) : td_model_upload(g_ctx, d);
		if (rc == TD_OK) g_model_key = key;
		else if (g_multi) fprintf(stderr, "tagdust_hip: %s\n", td_multi_last_error(g_multi));
	}
	flat_free(&fm);
	return rc;
}

/* MODE_ARCH_COMP (do_arch_comparison, barcode_hmm.c:2111-2148 + the merge at :1995-2016): every candidate model
 * scores every read with backward(); arch_posterior[j] is the float sum of b_score, accumulated per thread range
 * in read order and then over threads -- the summation order (hence the float result) depends on -t, so it is
 * reproduced here on the host from the per-read b_scores the GPU returns.  All candidates are scored by ONE launch
 * (td_arch_scores: the reads are staged once, the candidates' tables sit in HBM, generic kernel, no per-model compile). */
static int arch_comparison(struct arch_bag* ab, struct read_info** ri, struct parameters* param, int numseq)
{
	int i, j, t, rc = kslOK;
	const int T = param->num_threads > 0 ? param->num_threads : 1;
	const int interval = (int)(numseq / T);
	const int NA = ab->num_arch;
	int64_t* offs = malloc(sizeof(int64_t) * ((size_t)numseq + 1));
	offs[0] = 0;
	for (i = 0; i < numseq; i++) offs[i + 1] = offs[i] + ri[i]->len;
	uint8_t* codes = malloc((size_t)offs[numseq] + 1);
	for (i = 0; i < numseq; i++) memcpy(codes + offs[i], ri[i]->seq, (size_t)ri[i]->len);
	float* b = malloc(sizeof(float) * (size_t)numseq * (size_t)NA);
	flat_model* fm = calloc((size_t)NA, sizeof(flat_model));
	const td_model_desc** descs = malloc(sizeof(td_model_desc*) * (size_t)NA);
	for (j = 0; j < NA; j++) { flatten_model(ab->archs[j], param, 0, &fm[j]); descs[j] = &fm[j].d; }

	if (td_set_window(g_ctx, -1, -1) != TD_OK || td_arch_scores(g_ctx, descs, NA, codes, offs, numseq, b) != TD_OK) rc = kslFAIL;
	for (j = 0; j < NA && rc == kslOK; j++) {
		const float* bj = b + (size_t)j * (size_t)numseq;
		for (t = 0; t < T; t++) {
			const int start = t * interval, end = (t == T - 1) ? numseq : (t + 1) * interval;
			float partial = prob2scaledprob(1.0);            /* thread_data[t].ab->arch_posterior[i], :1938 */
			for (i = start; i < end; i++) partial += bj[i];
			ab->arch_posterior[j] += partial;                /* :2003 */
		}
	}
	for (j = 0; j < NA; j++) flat_free(&fm[j]);
	free(fm); free(descs);
	if (rc == kslOK) {                                       /* :2009-2016 */
		float sum = ab->arch_posterior[0];
		for (i = 1; i < ab->num_arch; i++) sum = logsum(sum, ab->arch_posterior[i]);
		for (i = 0; i < ab->num_arch; i++) ab->arch_posterior[i] = ab->arch_posterior[i] - sum;
	} else {
		fprintf(stderr, "tagdust_hip: %s\n", td_last_error(g_ctx));
	}
	free(offs); free(codes); free(b);
	return rc;
}

int run_pHMM(struct arch_bag* ab, struct model_bag* mb, struct read_info** ri, struct parameters* param,
             struct fasta* reference_fasta, int numseq, int mode)
{
	int i, k, status = kslOK;

	const int windowed = param->matchstart != -1 || param->matchend != -1;
	/* not on the GPU path: the dead training modes only.  (-start/-end windows run on the device in every mode; a read that does
	   not reach matchend -- which the reference decodes past its terminator, corrupting its heap -- is decoded on the bases it
	   has inside the window, and an invalid window is refused below: neither is handed to the reference's code.) */
	if (mode != MODE_GET_LABEL && mode != MODE_GET_PROB && mode != MODE_ARCH_COMP)
		return delegate("a training mode (dead code in v2.33)", ab, mb, ri, param, reference_fasta, numseq, mode);
	if (mode == MODE_ARCH_COMP && !ab) return delegate("architecture comparison without candidates", ab, mb, ri, param, reference_fasta, numseq, mode);
	{
		const char* d = getenv("TAGDUST_HIP_DELEGATE");      /* "4" / "1,4": these modes go to the reference's code (A/B aid) */
		while (d && *d) {
			char* end = NULL;
			const long v = strtol(d, &end, 10);
			if (end == d) break;
			if (v == mode) return delegate("TAGDUST_HIP_DELEGATE names this mode", ab, mb, ri, param, reference_fasta, numseq, mode);
			d = end;
			while (*d == ',' || *d == ' ') d++;
		}
	}
	/* (do_arch_comparison scores whole reads whatever -start/-end say, barcode_hmm.c:2111-2148: so does arch_comparison below) */
	if (windowed && mode != MODE_ARCH_COMP && (param->matchstart < 0 || param->matchend <= param->matchstart)) {
		fprintf(stderr, "tagdust_hip: -start %d -end %d is not a window (need 1 <= start <= end; both must be given)\n", param->matchstart + 1, param->matchend);
		return kslFAIL;
	}
	if (numseq <= 0) return kslOK;

	ensure_report();
	if (ensure_context() != TD_OK) return kslFAIL;
	g_batches_gpu++;
	if (mode == MODE_ARCH_COMP) return arch_comparison(ab, ri, param, numseq);
	if (td_set_option(g_ctx, "specialize", 1) != TD_OK) goto ERROR;
	if (upload_model(mb, param, 1) != TD_OK) goto ERROR;
	if ((g_multi ? td_multi_set_params(g_multi, param->confidence_threshold, param->minlen, param->dust)
	             : td_set_params(g_ctx, param->confidence_threshold, param->minlen, param->dust)) != TD_OK) goto ERROR;
	if ((g_multi ? td_multi_set_window(g_multi, windowed ? param->matchstart : -1, windowed ? param->matchend : -1)
	             : td_set_window(g_ctx, windowed ? param->matchstart : -1, windowed ? param->matchend : -1)) != TD_OK) goto ERROR;
	/* -ref: match_to_reference (barcode_hmm.c:2349-2351) moves to the device; it only runs in label mode */
	if (mode == MODE_GET_LABEL && reference_fasta && param->reference_fasta) {
		if ((g_multi ? td_multi_set_artifacts(g_multi, reference_fasta->string, reference_fasta->s_index, reference_fasta->numseq,
		                                      param->filter_error, param->num_threads)
		             : td_set_artifacts(g_ctx, reference_fasta->string, reference_fasta->s_index, reference_fasta->numseq,
		                                param->filter_error, param->num_threads)) != TD_OK) goto ERROR;
	} else if ((g_multi ? td_multi_set_artifacts(g_multi, NULL, NULL, 0, 0, 1) : td_set_artifacts(g_ctx, NULL, NULL, 0, 0, 1)) != TD_OK) goto ERROR;

	int64_t* offs = malloc(sizeof(int64_t) * ((size_t)numseq + 1));
	offs[0] = 0;
	for (i = 0; i < numseq; i++) offs[i + 1] = offs[i] + ri[i]->len;
	uint8_t* codes = malloc((size_t)offs[numseq] + 1);
	for (i = 0; i < numseq; i++) memcpy(codes + offs[i], ri[i]->seq, (size_t)ri[i]->len);   /* already 0..4 (io.c:1759) */

	td_read_result* res = malloc(sizeof(td_read_result) * (size_t)numseq);
	int8_t* labels = malloc((size_t)offs[numseq] + (size_t)numseq);
	uint8_t* seq_out = malloc((size_t)offs[numseq] + 1);
	const int tdmode = mode == MODE_GET_LABEL ? TD_MODE_GET_LABEL : TD_MODE_GET_PROB;
	int8_t* want_labels = mode == MODE_GET_LABEL ? labels : NULL;
	uint8_t* want_seq = mode == MODE_GET_LABEL ? seq_out : NULL;
	int failed;
	if (g_multi)   /* the batch split over the devices like run_pHMM splits it over threads, results in input order */
		failed = td_multi_decode(g_multi, codes, 0, offs, numseq, tdmode, res, want_labels, want_seq) != TD_OK;
	else
		failed = td_batch_upload(g_ctx, codes, offs, numseq) != TD_OK || td_run(g_ctx, tdmode) != TD_OK ||
		         td_batch_download(g_ctx, res, want_labels, want_seq) != TD_OK;
	if (failed) {
		if (g_multi) fprintf(stderr, "tagdust_hip: %s\n", td_multi_last_error(g_multi));
		status = kslFAIL;
	} else {
		for (i = 0; i < numseq; i++) {
			ri[i]->mapq = res[i].mapq;
			if (mode == MODE_GET_LABEL) {
				const int len = ri[i]->len;
				memcpy(ri[i]->labels, labels + offs[i] + i, (size_t)len + 1);
				ri[i]->bar_prob = 100;                       /* barcode_hmm.c:2343 */
				ri[i]->read_type = res[i].read_type;
				if (res[i].barcode != -1) ri[i]->barcode = res[i].barcode;
				if (res[i].fingerprint != -1) ri[i]->fingerprint = res[i].fingerprint;
				for (k = 0; k < len; k++) {                 /* make_extracted_read, barcode_hmm.c:3343-3350 */
					if (seq_out[offs[i] + k] == 65) { ri[i]->seq[k] = 65; ri[i]->qual[k] = 65; }
				}
				ri[i]->qual[len] = 0;                        /* barcode_hmm.c:3308 */
			} else {
				ri[i]->bar_prob = res[i].bar_prob;
			}
		}
	}
	free(offs); free(codes); free(res); free(labels); free(seq_out);
	if (status != kslOK) goto ERROR;
	return kslOK;
ERROR:
	fprintf(stderr, "tagdust_hip: %s\n", g_ctx ? td_last_error(g_ctx) : "no context");
	return kslFAIL;
}
