#include "tagdust_io.h"
#include "tagdust_hip.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
// stand-ins for the device entry points (the parse-only run never calls them; a "decoding" run here returns zeroed results)
extern "C" {
// TD_STUB_ALLOC_LIMIT=N: the (N+1)-th and later page-locked allocations fail (a memlock limit met part-way)
static int g_allocs = 0;
void* td_host_alloc(size_t b)
{
	const char* e = getenv("TD_STUB_ALLOC_LIMIT");
	if (e && __atomic_add_fetch(&g_allocs, 1, __ATOMIC_RELAXED) > atoi(e)) return nullptr;
	return malloc(b ? b : 1);
}
void td_host_free(void* p) { free(p); }
int td_get_option(td_ctx*, const char* name, int32_t* v) { *v = !strcmp(name, "pipeline_depth") ? 3 : 0; return 0; }
const char* td_last_error(const td_ctx*) { return "stub"; }
static int64_t g_t = 0;
int td_submit(td_ctx*, const void*, int32_t, const int64_t* offs, int64_t n, int, td_read_result* res, int8_t*, uint8_t* seq_out, int64_t* ticket)
{ memset(res, 0, sizeof(td_read_result) * (size_t)n); for (int64_t i = 0; i < n; i++) { res[i].barcode = (int32_t)(i % 3); res[i].fingerprint = -1; res[i].mapq = 12.345f; } memset(seq_out, 1, (size_t)(offs[n] - offs[0])); *ticket = ++g_t; return 0; }
int td_wait(td_ctx*, int64_t) { return 0; }
}
int main(int argc, char** argv)
{
	td_stream_stats st;
	td_stream_opts o = { atoi(argv[2]), atoi(argv[3]), atol(argv[4]) };
	static char b0[] = "ACGT", b1[] = "TTGA", b2[] = "GGCC", bn[] = "NNNN", rn[] = "N";
	static char* bs[] = { b0, b1, b2, bn };
	static char* rs[] = { rn };
	static td_arch arch;
	arch.n_segments = 2; arch.type[0] = 'B'; arch.type[1] = 'R'; arch.n_seq[0] = 4; arch.n_seq[1] = 1; arch.seq_len[0] = 4; arch.seq_len[1] = 1;
	arch.seqs[0] = bs; arch.seqs[1] = rs;
	td_arch* a = &arch;
	if (argc > 5 && !strcmp(argv[5], "multi")) {
		// two input files in lock-step (the same file twice): file 0 "decoded" by the stub with the barcode architecture, file 1 a
		// plain read (R:N: not decoded, run_rna_dust on the host), two "devices"
		static td_arch plain;
		plain.n_segments = 1; plain.type[0] = 'R'; plain.n_seq[0] = 1; plain.seq_len[0] = 1; plain.seqs[0] = rs;
		td_ctx* ctxs[2] = { (td_ctx*)0x1, (td_ctx*)0x1 };
		td_stream_file fl[2] = { { argv[1], a, ctxs }, { argv[1], &plain, nullptr } };
		int64_t counts[TD_NUM_COUNTERS];
		for (int rep = 0; rep < 2; rep++) {
			int rc = td_stream_run_multi(fl, 2, 2, "/tmp/td_tsan/outm", 100, &o, &st, counts);
			if (rc) { printf("error: %s\n", td_io_last_error()); return 1; }
			printf("rc %d reads %lld batches %lld bytes_out %lld counts %lld %lld %lld\n", rc, (long long)st.n_reads, (long long)st.n_batches, (long long)st.bytes_out,
			       (long long)counts[0], (long long)counts[6], (long long)counts[8] + counts[9] + counts[10]);
		}
		td_stream_release();
		return 0;
	}
	for (int rep = 0; rep < 2; rep++) {
		int rc = td_stream_run(argc > 5 ? (td_ctx*)0x1 : nullptr, argv[1], a, "/tmp/td_tsan/out", &o, &st);
		printf("rc %d reads %lld batches %lld bytes_out %lld fnv %llx%s%s\n", rc, (long long)st.n_reads, (long long)st.n_batches, (long long)st.bytes_out, (unsigned long long)st.codes_fnv,
		       rc ? "  error: " : "", rc ? td_io_last_error() : "");
	}
	td_stream_release();
	return 0;
}
