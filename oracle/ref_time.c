/*
 * ref_time.c -- TEST / MEASUREMENT INFRASTRUCTURE ONLY (the CPU baseline of bench.py), never shipped.
 *
 * Own code that links against the *unmodified* TagDust2 reference sources where they lie under /root/reference/src
 * (oracle/Makefile; output only into oracle/_ref/).  It times exactly the reference's label phase -- one run_pHMM(...,
 * MODE_GET_LABEL) call (src/barcode_hmm.c:1895-2029: pthread fan-out, per-thread copy_model_bag, do_label_thread over
 * the thread's range, join) on reads that are already in memory -- with the prologue of hmm_controller_multiple()
 * (:163-206) done before the clock starts: init_logsum, get_sequence_stats, init_model_bag, read_fasta_fastq.
 * No threshold calibration (the threshold is given), no file output, and none of the controller's per-batch side work:
 * in particular not the "long sequence" check (:292-310), which with -Q and reads of one length rebuilds the model for
 * every read of a batch and would charge the reference for time that is not the label phase.
 *
 * usage: ref_time <threshold> <tagdust command line without argv[0]>       (-t N selects the thread count)
 * prints: {"reads": N, "seconds": S, "threads": T, "extracted": E}
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "kslib.h"
#include "tagdust2.h"
#include "interface.h"
#include "nuc_code.h"
#include "misc.h"
#include "io.h"
#include "barcode_hmm.h"

int main(int argc, char* argv[])
{
	struct parameters* param = NULL;
	struct sequence_stats_info* ssi = NULL;
	struct model_bag* mb = NULL;
	struct read_info** ri = NULL;
	FILE* file = NULL;
	int i, numseq = 0, extracted = 0;
	struct timespec t0, t1;

	if(argc < 4){
		fprintf(stderr, "usage: ref_time <threshold> <tagdust args>\n");
		return 2;
	}
	const float threshold = (float)atof(argv[1]);
	int nargc = argc - 1;
	char** nargv = malloc(sizeof(char*) * (nargc + 1));
	nargv[0] = "tagdust";
	for(i = 1; i < nargc; i++) nargv[i] = argv[i + 1];
	nargv[nargc] = NULL;

	init_nuc_code();
	param = interface(param, nargc, nargv);
	if(!param){ fprintf(stderr, "interface() returned NULL\n"); return 1; }
	if(QC_read_structure(param) != kslOK){ fprintf(stderr, "QC_read_structure failed\n"); return 1; }
	if(param->infiles != 1){ fprintf(stderr, "ref_time handles exactly one input file\n"); return 1; }
	init_logsum();
	param->num_query = 1000001;                     /* one batch, barcode_hmm.c:172 */
	ri = malloc_read_info(ri, param->num_query);
	ssi = get_sequence_stats(param, ri, 0);
	mb = init_model_bag(param, ssi);
	file = io_handler(file, 0, param);
	if(read_fasta_fastq(ri, param, file, &numseq) != kslOK){ fprintf(stderr, "read failed\n"); return 1; }
	pclose(file);
	param->confidence_threshold = threshold;

	clock_gettime(CLOCK_MONOTONIC, &t0);
	if(run_pHMM(0, mb, ri, param, 0, numseq, MODE_GET_LABEL) != kslOK){ fprintf(stderr, "run_pHMM failed\n"); return 1; }
	clock_gettime(CLOCK_MONOTONIC, &t1);

	for(i = 0; i < numseq; i++) if(ri[i]->read_type == EXTRACT_SUCCESS) extracted++;
	printf("{\"reads\": %d, \"seconds\": %.6f, \"threads\": %d, \"extracted\": %d}\n", numseq,
	       (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec), param->num_threads, extracted);
	return 0;
}
