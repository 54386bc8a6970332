/*
 * ref_dump.c -- TEST INFRASTRUCTURE ONLY (fixture generator), never shipped.
 *
 * Own code that links against the *unmodified* TagDust2 reference sources where
 * they lie under /root/reference/src (see oracle/Makefile; output only into
 * oracle/_ref/, which is git-ignored).  It drives the reference's own functions
 * for the per-read HMM decoding path and dumps everything a parity test needs:
 *
 *   - the model tables produced by init_model_bag()        (barcode_hmm.c:5760)
 *   - per read: b_score / f_score / r_score / bar_prob / labels straight after
 *     backward() + forward_max_posterior_decoding()         (barcode_hmm.c:3439, 4128)
 *   - per read after run_pHMM(MODE_GET_LABEL)               (barcode_hmm.c:1895,
 *     do_label_thread :2269): mapq (Q), read_type, barcode, fingerprint and the
 *     rewritten sequence (extract_reads :3172, dust_sequences :2407)
 *
 * The controller prologue mirrors hmm_controller_multiple() (barcode_hmm.c:163-206)
 * for one input file: init_logsum, get_sequence_stats, optional
 * estimateQthreshold, init_model_bag.
 *
 * usage: ref_dump <out.bin> [-maxreads N] <tagdust command line without argv[0]>
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#include "kslib.h"
#include "tagdust2.h"
#include "interface.h"
#include "nuc_code.h"
#include "misc.h"
#include "io.h"
#include "barcode_hmm.h"

int estimateQthreshold(struct parameters* param, struct sequence_stats_info* ssi);

static FILE* out;
static void w_i32(int32_t v){ fwrite(&v, 4, 1, out); }
static void w_f32(float v){ fwrite(&v, 4, 1, out); }
static void w_f64(double v){ fwrite(&v, 8, 1, out); }
static void w_bytes(const void* p, size_t n){ fwrite(p, 1, n, out); }

int main(int argc, char* argv[])
{
	struct parameters* param = NULL;
	struct sequence_stats_info* ssi = NULL;
	struct model_bag* mb = NULL;
	struct read_info** ri = NULL;
	FILE* file = NULL;
	int i, j, f, g, numseq = 0;
	int maxreads = 1000001;
	int q_given;
	float threshold;

	if(argc < 3){
		fprintf(stderr, "usage: ref_dump <out.bin> [-maxreads N] <tagdust args>\n");
		return 2;
	}
	const char* outname = argv[1];
	int shift = 2;
	if(argc > 4 && !strcmp(argv[2], "-maxreads")){
		maxreads = atoi(argv[3]);
		shift = 4;
	}
	int nargc = argc - shift + 1;
	char** nargv = malloc(sizeof(char*) * (nargc + 1));
	nargv[0] = "tagdust";
	for(i = 1; i < nargc; i++) nargv[i] = argv[shift + i - 1];
	nargv[nargc] = NULL;

	init_nuc_code();
	param = interface(param, nargc, nargv);
	if(!param){ fprintf(stderr, "interface() returned NULL\n"); return 1; }
	if(QC_read_structure(param) != kslOK){ fprintf(stderr, "QC_read_structure failed\n"); return 1; }
	if(param->infiles != 1){ fprintf(stderr, "ref_dump handles exactly one input file\n"); return 1; }

	init_logsum();
	param->num_query = maxreads;
	ri = malloc_read_info(ri, param->num_query);
	ssi = get_sequence_stats(param, ri, 0);

	/* quirk Q1 (barcode_hmm.c:102,190-200,314): -Q skips calibration and the run
	   then uses threshold 0.0 */
	q_given = param->confidence_threshold != 0.0f;
	if(!q_given){
		if(estimateQthreshold(param, ssi) != kslOK){ fprintf(stderr, "estimateQthreshold failed\n"); return 1; }
		threshold = param->confidence_threshold;
	}else{
		threshold = 0.0f;
	}
	mb = init_model_bag(param, ssi);

	out = fopen(outname, "wb");
	if(!out){ perror(outname); return 1; }
	w_bytes("TDRF", 4);
	w_i32(2);                        /* format version */

	/* ---- run parameters ---- */
	struct read_structure* rs = param->read_structure;
	w_f32(param->sequencer_error_rate);
	w_f32(param->indel_frequency);
	w_f32(threshold);
	w_i32(q_given);
	w_i32(param->minlen);
	w_i32(param->dust);
	w_i32(param->matchstart);
	w_i32(param->matchend);

	/* ---- sequence statistics (io.c:52-300) ---- */
	for(i = 0; i < 5; i++) w_f64(ssi->background[i]);
	w_f64(ssi->expected_5_len); w_f64(ssi->expected_3_len);
	w_f64(ssi->mean_5_len); w_f64(ssi->stdev_5_len);
	w_f64(ssi->mean_3_len); w_f64(ssi->stdev_3_len);
	w_f64(ssi->average_length);
	w_i32(ssi->max_seq_len);

	/* ---- read structure as parsed by interface.c ---- */
	w_i32(rs->num_segments);
	for(j = 0; j < rs->num_segments; j++){
		w_i32((int)rs->type[j]);
		w_i32(rs->numseq_in_segment[j]);
		int sl = (int)strlen(rs->sequence_matrix[j][0]);
		w_i32(sl);
		for(f = 0; f < rs->numseq_in_segment[j]; f++) w_bytes(rs->sequence_matrix[j][f], sl);
	}

	/* ---- model tables ---- */
	int S = mb->num_models, H = mb->total_hmm_num, C = 0;
	for(j = 0; j < S; j++) C += mb->model[j]->num_hmms * mb->model[j]->hmms[0]->num_columns;
	w_i32(S); w_i32(H); w_i32(C);
	w_i32(mb->average_raw_length);
	for(i = 0; i < 5; i++) w_f32(mb->model[0]->background_nuc_frequency[i]);
	for(j = 0; j < S; j++){
		w_i32(mb->model[j]->num_hmms);
		w_i32(mb->model[j]->hmms[0]->num_columns);
		w_f32(mb->model[j]->skip);
	}
	for(j = 0; j < S; j++){
		for(f = 0; f < mb->model[j]->num_hmms; f++){
			for(g = 0; g < mb->model[j]->hmms[f]->num_columns; g++){
				struct hmm_column* col = mb->model[j]->hmms[f]->hmm_column[g];
				for(i = 0; i < 9; i++) w_f32(col->transition[i]);
				for(i = 0; i < 5; i++) w_f32(col->m_emit[i]);
				for(i = 0; i < 5; i++) w_f32(col->i_emit[i]);
				w_f32(mb->model[j]->silent_to_M[f][g]);
				w_f32(mb->model[j]->silent_to_I[f][g]);
			}
		}
	}
	for(i = 0; i < H; i++) w_i32(mb->label[i]);
	for(i = 0; i < H; i++) for(j = 0; j < H; j++) w_f32(mb->transition_matrix[i][j]);

	/* ---- reads ---- */
	file = io_handler(file, 0, param);
	if(read_fasta_fastq(ri, param, file, &numseq) != kslOK){ fprintf(stderr, "read failed\n"); return 1; }
	pclose(file);
	w_i32(numseq);

	char** seq_before = malloc(sizeof(char*) * numseq);
	float* bs = malloc(sizeof(float) * numseq);
	float* fs = malloc(sizeof(float) * numseq);
	float* rsx = malloc(sizeof(float) * numseq);
	double* bp = malloc(sizeof(double) * numseq);
	char** lab = malloc(sizeof(char*) * numseq);

	for(i = 0; i < numseq; i++){
		int len = ri[i]->len;
		seq_before[i] = malloc(len + 1);
		memcpy(seq_before[i], ri[i]->seq, len + 1);
		if(param->matchstart != -1 || param->matchend != -1){
			/* -start / -end: do_label_thread decodes seq + matchstart for matchend - matchstart bases (barcode_hmm.c:2290-2296);
			   only reads that reach matchend are defined input (the reference reads past a shorter read's terminator) */
			if(len < param->matchend){ fprintf(stderr, "ref_dump: read %d (%d nt) does not reach -end %d\n", i, len, param->matchend); return 1; }
			mb = backward(mb, ri[i]->seq + param->matchstart, param->matchend - param->matchstart);
			mb = forward_max_posterior_decoding(mb, ri[i], ri[i]->seq + param->matchstart, param->matchend - param->matchstart);
		}else{
			mb = backward(mb, ri[i]->seq, len);
			mb = forward_max_posterior_decoding(mb, ri[i], ri[i]->seq, len);
		}
		bs[i] = mb->b_score; fs[i] = mb->f_score; rsx[i] = mb->r_score; bp[i] = ri[i]->bar_prob;
		lab[i] = malloc(len + 1);
		memcpy(lab[i], ri[i]->labels, len + 1);
	}

	/* -ref <fasta>: artifact matching (match_to_reference, barcode_hmm.c:2478-2583) runs inside run_pHMM between
	   extraction and DUST; it pairs reads in fours from the start of each thread's range, so the thread count
	   matters (REF_DUMP_THREADS, default 1).  The fasta is loaded as hmm_controller_multiple does (:209-216). */
	struct fasta* reference_fasta = 0;
	int n_threads = 1;
	if(getenv("REF_DUMP_THREADS")) n_threads = atoi(getenv("REF_DUMP_THREADS"));
	if(param->reference_fasta){
		reference_fasta = get_fasta(reference_fasta, param->reference_fasta);
		reference_fasta->mer_hash = malloc(sizeof(int) * reference_fasta->numseq);
		for(i = 0; i < reference_fasta->numseq; i++) reference_fasta->mer_hash[i] = 0;
	}
	param->num_threads = n_threads;
	param->confidence_threshold = threshold;
	if(run_pHMM(0, mb, ri, param, reference_fasta, numseq, MODE_GET_LABEL) != kslOK){ fprintf(stderr, "run_pHMM failed\n"); return 1; }

	for(i = 0; i < numseq; i++){
		int len = ri[i]->len;
		if(memcmp(lab[i], ri[i]->labels, len + 1)){
			fprintf(stderr, "ref_dump: labels differ between direct call and run_pHMM (read %d)\n", i);
			return 1;
		}
		w_i32(len);
		int nl = (int)strlen(ri[i]->name);
		w_i32(nl);
		w_bytes(ri[i]->name, nl);
		w_bytes(seq_before[i], len);
		w_bytes(ri[i]->qual, len);
		w_f32(bs[i]); w_f32(fs[i]); w_f32(rsx[i]); w_f64(bp[i]);
		w_bytes(lab[i], len + 1);
		w_f32(ri[i]->mapq);
		w_i32(ri[i]->read_type);
		w_i32(ri[i]->barcode);
		w_i32(ri[i]->fingerprint);
		w_bytes(ri[i]->seq, len);
	}
	/* ---- optional trailer: the artifact sequences as read_fasta() left them ---- */
	if(reference_fasta){
		w_bytes("ARTF", 4);
		w_i32(reference_fasta->numseq);
		w_i32(param->filter_error);
		w_i32(n_threads);
		for(i = 0; i <= reference_fasta->numseq; i++) w_i32(reference_fasta->s_index[i]);
		w_bytes((char*)reference_fasta->string, reference_fasta->s_index[reference_fasta->numseq]);
	}
	fclose(out);
	fprintf(stderr, "ref_dump: %d reads, S=%d H=%d C=%d threshold=%f avg_len=%d -> %s\n",
	        numseq, S, H, C, threshold, mb->average_raw_length, outname);
	return 0;
}
