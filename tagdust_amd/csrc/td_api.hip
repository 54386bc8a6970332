// td_api.hip -- C-ABI host layer of libtagdust_hip.so (declared in include/tagdust_hip.h).
//
// Replaces the reference's run_pHMM() fan-out (src/barcode_hmm.c:1895-2029): instead of T pthreads with
// private model copies, one context per GPU holds the flattened model in HBM and launches the decode
// kernel (td_kernels.hip) over a resident batch.  No torch, no CPU fallback: every entry point fails
// with TD_FAIL (and a message in td_last_error) when HIP does.
#include <hip/hip_runtime.h>
#include <math.h>
#include <cmath>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <thread>
#include <vector>

#include "../../include/tagdust_hip.h"
#include "td_device.h"
#include "td_jit.h"

extern "C" __attribute__((visibility("hidden"))) hipError_t td_launch_decode(const TdKernelArgs* ka, hipStream_t stream);   // library-internal
extern "C" __attribute__((visibility("hidden"))) int td_kernel_block_threads(void);

static float g_logsum[TD_LOGSUM_SIZE];
static bool g_logsum_ready = false;
static std::string g_create_error;

static void init_logsum_host()
{
	// init_logsum(), src/misc.c:57-63: (float) log(1. + exp((double) -i / SCALE)), SCALE = 1000.0f
	if (g_logsum_ready) return;
	for (int i = 0; i < TD_LOGSUM_SIZE; i++) g_logsum[i] = (float)log(1.0 + exp((double)-i / (double)1000.0f));
	g_logsum_ready = true;
}

// prob2scaledprob(), src/misc.c:85-92
static float p2sp(float p) { return p == 0.0f ? -INFINITY : (float)log((double)p); }

struct td_ctx {
	int device = 0;
	hipStream_t stream = nullptr;
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	std::string err;
	int n_cu = 0;
	size_t hbm_total = 0;

	// model
	bool have_model = false;
	TdModelHeader hdr{};
	std::vector<int32_t> label;
	TdModelHeader* d_hdr = nullptr;
	TdCol* d_cols = nullptr;
	uint32_t* d_hinfo = nullptr;
	int32_t* d_pred_off = nullptr;
	int32_t* d_pred_idx = nullptr;
	float* d_logsum = nullptr;
	unsigned long long* d_counters = nullptr;

	// model-specialised kernel (td_spec_kernel.inc through hiprtc)
	int specialize = 1;
	bool spec_ready = false;
	hipModule_t spec_mod = nullptr;
	hipFunction_t spec_fn = nullptr;
	std::vector<int32_t> m_n_hmm, m_n_col;
	std::vector<float> m_trans;
	// deep copy of the uploaded description (a batch with very long reads recompiles the kernel with the clamped logsum)
	td_model_desc m_desc{};
	std::vector<float> m_skip, m_eM, m_eI, m_sM, m_sI, m_A;
	std::vector<int8_t> m_seg_type;
	std::vector<int32_t> m_finger_len;
	bool spec_oob = false;      // the loaded kernel uses the clamp-free logsum
	float m_maxabs = 0.0f;      // largest |finite parameter|
	TdSpecLayout slay{};
	int spec_block = 256, spec_waves_per_cu = 8;

	// params
	float threshold = 0.0f;
	int32_t minlen = 16, dust = 100;

	// batch
	int64_t n_reads = 0;
	int32_t n_tiles = 0, lmax = 0, nw2 = 0, nw1 = 0;
	std::vector<int64_t> offs;
	std::vector<uint8_t> codes_host; // kept for td_batch_download(seq_out)
	std::vector<int64_t> pos_of;     // pos_of[i] = position of read i in the length-sorted device order
	uint32_t* d_packed = nullptr; size_t cap_packed = 0;
	int32_t* d_lens = nullptr;    size_t cap_lens = 0;
	// pinned host staging (H2D of the packed batch, D2H of the outputs): pageable copies run at a fraction of the link
	// rate and a fresh std::vector per batch is zero-filled first
	uint32_t* h_packed = nullptr; size_t cap_h_packed = 0;
	int32_t*  h_lens = nullptr;   size_t cap_h_lens = 0;
	uint8_t*  h_out = nullptr;    size_t cap_h_out = 0;
	// -ref artifact filter
	uint8_t* d_art_text = nullptr; int32_t* d_art_index = nullptr;
	uint8_t* d_art_left = nullptr; size_t cap_art_left = 0;
	int32_t art_n = 0, art_fe = 0, art_threads = 1;
	uint8_t* d_out = nullptr;     size_t cap_out = 0;   // all per-read outputs in one allocation
	uint8_t* d_ws = nullptr;      size_t cap_ws = 0;
	TdWsLayout lay{};
	int32_t n_slots = 0;
	bool ran = false;
	float last_ms = -1.0f;
};

// run fn(lo, hi) over [0, n) on up to 16 host threads (TD_HOST_THREADS overrides)
template <typename F>
static void parallel_ranges(int64_t n, F fn)
{
	int nt = (int)std::thread::hardware_concurrency();
	if (const char* e = getenv("TD_HOST_THREADS")) nt = atoi(e);
	if (nt > 16) nt = 16;
	if (nt < 1 || n < 65536) nt = 1;
	if (nt == 1) { fn((int64_t)0, n); return; }
	std::vector<std::thread> th;
	const int64_t per = (n + nt - 1) / nt;
	for (int t = 0; t < nt; t++) {
		const int64_t lo = t * per, hi = lo + per < n ? lo + per : n;
		if (lo < hi) th.emplace_back(fn, lo, hi);
	}
	for (auto& t : th) t.join();
}

static int fail(td_ctx* c, const char* fmt, ...)
{
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	if (c) c->err = buf; else g_create_error = buf;
	return TD_FAIL;
}

#define HIPCHK(c, call)                                                                       \
	do {                                                                                      \
		hipError_t e_ = (call);                                                               \
		if (e_ != hipSuccess) return fail((c), "%s failed: %s", #call, hipGetErrorString(e_)); \
	} while (0)

template <typename T>
static int ensure(td_ctx* c, T** p, size_t* cap, size_t bytes)
{
	if (*cap >= bytes && *p) return TD_OK;
	if (*p) { HIPCHK(c, hipFree(*p)); *p = nullptr; *cap = 0; }
	if (bytes == 0) bytes = 256;
	HIPCHK(c, hipMalloc((void**)p, bytes));
	*cap = bytes;
	return TD_OK;
}

template <typename T>
static int ensure_pinned(td_ctx* c, T** p, size_t* cap, size_t bytes)
{
	if (*cap >= bytes && *p) return TD_OK;
	if (*p) { HIPCHK(c, hipHostFree(*p)); *p = nullptr; *cap = 0; }
	if (bytes == 0) bytes = 256;
	bytes += bytes / 4;   // head room: batches of a run differ a little in size
	HIPCHK(c, hipHostMalloc((void**)p, bytes, hipHostMallocDefault));
	*cap = bytes;
	return TD_OK;
}

extern "C" const float* td_logsum_table(void)
{
	init_logsum_host();
	return g_logsum;
}

extern "C" const char* td_last_error(const td_ctx* ctx)
{
	return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

extern "C" int td_ctx_create(int device, td_ctx** out)
{
	if (!out) return fail(nullptr, "td_ctx_create: out is NULL");
	*out = nullptr;
	int ndev = 0;
	hipError_t e = hipGetDeviceCount(&ndev);
	if (e != hipSuccess || ndev <= 0)
		return fail(nullptr, "td_ctx_create: no HIP device available (%s) -- this library has no CPU path",
		            e == hipSuccess ? "device count 0" : hipGetErrorString(e));
	if (device < 0 || device >= ndev) return fail(nullptr, "td_ctx_create: device %d out of range (0..%d)", device, ndev - 1);
	td_ctx* c = new td_ctx();
	c->device = device;
	hipDeviceProp_t prop;
	if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) {
		delete c;
		return fail(nullptr, "td_ctx_create: cannot select device %d", device);
	}
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
		std::string arch = prop.gcnArchName;
		delete c;
		return fail(nullptr, "td_ctx_create: device %d is %s; this library is built for gfx950 (MI355X) only", device, arch.c_str());
	}
	c->n_cu = prop.multiProcessorCount;
	if (const char* e = getenv("TD_SPECIALIZE")) c->specialize = atoi(e) != 0;
	c->hbm_total = prop.totalGlobalMem;
	init_logsum_host();
	bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess &&
	          hipEventCreate(&c->ev0) == hipSuccess && hipEventCreate(&c->ev1) == hipSuccess &&
	          hipMalloc((void**)&c->d_logsum, sizeof(float) * TD_LOGSUM_LIVE) == hipSuccess &&
	          hipMalloc((void**)&c->d_counters, sizeof(unsigned long long) * TD_NUM_COUNTERS) == hipSuccess &&
	          hipMemcpy(c->d_logsum, g_logsum, sizeof(float) * TD_LOGSUM_LIVE, hipMemcpyHostToDevice) == hipSuccess &&
	          hipMemset(c->d_counters, 0, sizeof(unsigned long long) * TD_NUM_COUNTERS) == hipSuccess;
	if (!ok) {
		td_ctx_destroy(c);
		return fail(nullptr, "td_ctx_create: HIP resource allocation failed: %s", hipGetErrorString(hipGetLastError()));
	}
	*out = c;
	return TD_OK;
}

extern "C" void td_ctx_destroy(td_ctx* c)
{
	if (!c) return;
	(void)hipSetDevice(c->device);
	if (c->stream) (void)hipStreamSynchronize(c->stream);
	void* bufs[] = { c->d_hdr, c->d_cols, c->d_hinfo, c->d_pred_off, c->d_pred_idx, c->d_logsum, c->d_counters,
	                 c->d_packed, c->d_lens, c->d_out, c->d_ws, c->d_art_text, c->d_art_index, c->d_art_left };
	for (void* p : bufs) if (p) (void)hipFree(p);
	void* pinned[] = { c->h_packed, c->h_lens, c->h_out };
	for (void* p : pinned) if (p) (void)hipHostFree(p);
	if (c->spec_mod) (void)hipModuleUnload(c->spec_mod);
	if (c->ev0) (void)hipEventDestroy(c->ev0);
	if (c->ev1) (void)hipEventDestroy(c->ev1);
	if (c->stream) (void)hipStreamDestroy(c->stream);
	delete c;
}

// ---------------------------------------------------------------------------------------------------------
// model
// ---------------------------------------------------------------------------------------------------------
// Compile (or fetch from the cache) and load the model-specialised kernel.  lsum_oob selects the clamp-free logsum.
static int load_spec_kernel(td_ctx* c, int lsum_oob)
{
	if (c->spec_mod) { HIPCHK(c, hipModuleUnload(c->spec_mod)); c->spec_mod = nullptr; }
	c->spec_fn = nullptr; c->spec_ready = false;
	std::vector<char> code;
	std::string log;
	if (td_spec_compile(&c->m_desc, code, log, lsum_oob) != TD_OK) return fail(c, "td_model_upload: specialised kernel did not compile: %.400s", log.c_str());
	HIPCHK(c, hipModuleLoadData(&c->spec_mod, code.data()));
	HIPCHK(c, hipModuleGetFunction(&c->spec_fn, c->spec_mod, "td_spec_kernel"));
	c->spec_ready = true;
	c->spec_oob = lsum_oob != 0;
	return TD_OK;
}

// The clamp-free logsum of the specialised kernel turns |a - b| * 1000 into an LDS byte address that wraps at
// |a - b| = 2^30 / 1000.  Every finite DP value is a sum of at most 2 parameters per position, and the posterior terms
// add two such values, so 4 * max|parameter| * (L + 2) bounds every finite difference; 6 * keeps a margin.
static bool spec_lsum_range_ok(const td_ctx* c, int lmax)
{
	double limit = 1.0e6;
	if (const char* e = getenv("TD_SPEC_LSUM_LIMIT")) limit = atof(e);   // tests: force the switch to the clamped form
	return 6.0 * (double)c->m_maxabs * ((double)lmax + 2.0) < limit;
}

extern "C" int td_model_upload(td_ctx* c, const td_model_desc* m)
{
	if (!c || !m) return fail(c, "td_model_upload: NULL argument");
	HIPCHK(c, hipSetDevice(c->device));
	if (m->S < 1 || m->S > TD_MAX_SEGMENTS) return fail(c, "td_model_upload: %d segments (1..%d supported)", m->S, TD_MAX_SEGMENTS);
	if (m->H < 1 || m->H > TD_MAX_HMMS) return fail(c, "td_model_upload: %d HMMs (1..%d supported)", m->H, TD_MAX_HMMS);
	if (!m->n_hmm || !m->n_col || !m->skip || !m->seg_type || !m->finger_len || !m->trans || !m->eM || !m->eI ||
	    !m->sM || !m->sI || !m->label || !m->A)
		return fail(c, "td_model_upload: NULL table pointer");
	TdModelHeader h{};
	h.S = m->S; h.H = m->H; h.C = m->C; h.avg_len = m->avg_len;
	for (int i = 0; i < 5; i++) h.bg[i] = m->bg[i];
	// random model constants, barcode_hmm.c:4520,4523 (float/double mix exactly as written there)
	h.r_stay = p2sp((float)(1.0 - (1.0 / (double)(float)m->avg_len)));
	h.r_exit = p2sp((float)(1.0 / (double)(float)m->avg_len));
	int co = 0, ho = 0, req = 0, maxc = 0;
	for (int j = 0; j < m->S; j++) {
		if (m->n_hmm[j] < 1 || m->n_col[j] < 1) return fail(c, "td_model_upload: segment %d has %d HMMs x %d columns", j, m->n_hmm[j], m->n_col[j]);
		TdSeg& s = h.seg[j];
		s.n_hmm = m->n_hmm[j]; s.n_col = m->n_col[j]; s.col_off = co; s.hmm_off = ho;
		s.skip = m->skip[j]; s.skip_live = !(m->skip[j] == -INFINITY); s.type = m->seg_type[j];
		co += s.n_hmm * s.n_col; ho += s.n_hmm;
		if (m->seg_type[j] == 'F') req += m->finger_len[j];
		if (s.n_col > maxc) maxc = s.n_col;
	}
	if (co != m->C || ho != m->H) return fail(c, "td_model_upload: inconsistent sizes (columns %d vs C %d, HMMs %d vs H %d)", co, m->C, ho, m->H);
	h.required_finger_len = req;
	h.max_ncol = maxc;

	std::vector<TdCol> cols(m->C);
	for (int k = 0; k < m->C; k++) {
		TdCol& q = cols[k];
		memset(&q, 0, sizeof q);
		uint32_t fl = 0;
		for (int t = 0; t < 9; t++) { q.t[t] = m->trans[k * 9 + t]; if (!(q.t[t] == -INFINITY)) fl |= 1u << t; }
		q.sM = m->sM[k]; if (!(q.sM == -INFINITY)) fl |= TDF_SM;
		q.sI = m->sI[k]; if (!(q.sI == -INFINITY)) fl |= TDF_SI;
		for (int e = 0; e < 5; e++) { q.eM[e] = m->eM[k * 5 + e]; q.eI[e] = m->eI[k * 5 + e]; }
		q.flags = fl;
	}
	// per-HMM info for extract_reads (barcode_hmm.c:3205-3226): type | segment | hmm | decoy
	std::vector<uint32_t> hinfo(m->H);
	for (int hh = 0; hh < m->H; hh++) {
		const int seg = m->label[hh] & 0xFFFF, f = (m->label[hh] >> 16) & 0x7FFF;
		if (seg >= m->S) return fail(c, "td_model_upload: label[%d] names segment %d", hh, seg);
		uint32_t v = (uint32_t)(uint8_t)m->seg_type[seg] | ((uint32_t)seg << 8) | ((uint32_t)f << 16);
		if (m->seg_type[seg] == 'B' && f == m->n_hmm[seg] - 1) v |= 0x80000000u; // all-N decoy, interface.c:521-527
		hinfo[hh] = v;
	}
	// label transition matrix -> predecessor lists (only u < v; staying in v is handled in the kernel)
	std::vector<int32_t> poff(m->H + 1, 0), pidx;
	for (int v = 0; v < m->H; v++) {
		poff[v] = (int32_t)pidx.size();
		for (int u = 0; u < v; u++) {
			const float a = m->A[u * m->H + v];
			if (a != 0.0f && a != 1.0f) return fail(c, "td_model_upload: transition matrix entry [%d][%d] = %g is not 0/1", u, v, a);
			if (a == 1.0f) pidx.push_back(u);
		}
		if (m->A[v * m->H + v] != 1.0f) return fail(c, "td_model_upload: transition matrix diagonal [%d] must be 1", v);
	}
	poff[m->H] = (int32_t)pidx.size();
	if (pidx.empty()) pidx.push_back(0);

	void* old[] = { c->d_hdr, c->d_cols, c->d_hinfo, c->d_pred_off, c->d_pred_idx };
	for (void* p : old) if (p) HIPCHK(c, hipFree(p));
	c->d_hdr = nullptr; c->d_cols = nullptr; c->d_hinfo = nullptr; c->d_pred_off = nullptr; c->d_pred_idx = nullptr;
	c->have_model = false;
	HIPCHK(c, hipMalloc((void**)&c->d_hdr, sizeof h));
	HIPCHK(c, hipMalloc((void**)&c->d_cols, sizeof(TdCol) * cols.size()));
	HIPCHK(c, hipMalloc((void**)&c->d_hinfo, sizeof(uint32_t) * hinfo.size()));
	HIPCHK(c, hipMalloc((void**)&c->d_pred_off, sizeof(int32_t) * poff.size()));
	HIPCHK(c, hipMalloc((void**)&c->d_pred_idx, sizeof(int32_t) * pidx.size()));
	HIPCHK(c, hipMemcpy(c->d_hdr, &h, sizeof h, hipMemcpyHostToDevice));
	HIPCHK(c, hipMemcpy(c->d_cols, cols.data(), sizeof(TdCol) * cols.size(), hipMemcpyHostToDevice));
	HIPCHK(c, hipMemcpy(c->d_hinfo, hinfo.data(), sizeof(uint32_t) * hinfo.size(), hipMemcpyHostToDevice));
	HIPCHK(c, hipMemcpy(c->d_pred_off, poff.data(), sizeof(int32_t) * poff.size(), hipMemcpyHostToDevice));
	HIPCHK(c, hipMemcpy(c->d_pred_idx, pidx.data(), sizeof(int32_t) * pidx.size(), hipMemcpyHostToDevice));
	c->hdr = h;
	c->label.assign(m->label, m->label + m->H);
	c->m_n_hmm.assign(m->n_hmm, m->n_hmm + m->S);
	c->m_n_col.assign(m->n_col, m->n_col + m->S);
	c->m_trans.assign(m->trans, m->trans + (size_t)m->C * 9);
	c->have_model = true;
	c->ran = false;
	c->n_reads = 0; c->n_tiles = 0;

	// model-specialised kernel: compile now (seconds); a failure is an error, never a silent fallback
	if (c->spec_mod) { HIPCHK(c, hipModuleUnload(c->spec_mod)); c->spec_mod = nullptr; }
	c->spec_fn = nullptr; c->spec_ready = false;
	if (c->specialize) {
		// keep what a later recompile needs
		c->m_skip.assign(m->skip, m->skip + m->S);
		c->m_seg_type.assign(m->seg_type, m->seg_type + m->S);
		c->m_finger_len.assign(m->finger_len, m->finger_len + m->S);
		c->m_eM.assign(m->eM, m->eM + (size_t)m->C * 5); c->m_eI.assign(m->eI, m->eI + (size_t)m->C * 5);
		c->m_sM.assign(m->sM, m->sM + m->C); c->m_sI.assign(m->sI, m->sI + m->C);
		c->m_A.assign(m->A, m->A + (size_t)m->H * m->H);
		c->m_desc = *m;
		c->m_desc.n_hmm = c->m_n_hmm.data(); c->m_desc.n_col = c->m_n_col.data(); c->m_desc.skip = c->m_skip.data();
		c->m_desc.seg_type = c->m_seg_type.data(); c->m_desc.finger_len = c->m_finger_len.data();
		c->m_desc.trans = c->m_trans.data(); c->m_desc.eM = c->m_eM.data(); c->m_desc.eI = c->m_eI.data();
		c->m_desc.sM = c->m_sM.data(); c->m_desc.sI = c->m_sI.data(); c->m_desc.label = c->label.data(); c->m_desc.A = c->m_A.data();
		float mx = 0.0f;
		auto scan = [&](const float* v, size_t n) { for (size_t i = 0; i < n; i++) if (std::isfinite(v[i]) && fabsf(v[i]) > mx) mx = fabsf(v[i]); };
		scan(m->trans, (size_t)m->C * 9); scan(m->eM, (size_t)m->C * 5); scan(m->eI, (size_t)m->C * 5);
		scan(m->sM, m->C); scan(m->sI, m->C); scan(m->skip, m->S); scan(m->bg, 5);
		c->m_maxabs = mx;
		if (load_spec_kernel(c, td_spec_lsum_oob()) != TD_OK) return TD_FAIL;
		c->spec_block = td_spec_block_threads();
		// resident waves per CU: two LDS tables fit a CU; a 1024-thread workgroup fills it alone
		{
			const int wpb = c->spec_block / TD_WAVE;
			int blocks_per_cu = 32 / wpb;             // 32 waves per CU
			if (blocks_per_cu > 2) blocks_per_cu = 2; // two logsum tables (<= 66.5 KB each) per 160 KB of LDS
			if (blocks_per_cu < 1) blocks_per_cu = 1;
			c->spec_waves_per_cu = blocks_per_cu * wpb;
		}
		const int by_regs = 4 * td_spec_min_waves();
		if (c->spec_waves_per_cu > by_regs) c->spec_waves_per_cu = by_regs;
	}
	return TD_OK;
}

extern "C" int td_set_option(td_ctx* c, const char* name, int32_t value)
{
	if (!c || !name) return TD_FAIL;
	if (!strcmp(name, "specialize")) {
		c->specialize = value != 0; // takes effect at the next td_model_upload
		return TD_OK;
	}
	return fail(c, "td_set_option: unknown option %s", name);
}

extern "C" int td_get_option(td_ctx* c, const char* name, int32_t* value)
{
	if (!c || !name || !value) return TD_FAIL;
	if (!strcmp(name, "specialize")) { *value = c->specialize; return TD_OK; }
	if (!strcmp(name, "spec_lsum_clamped")) { *value = c->spec_ready && !c->spec_oob; return TD_OK; }
	return fail(c, "td_get_option: unknown option %s", name);
}

extern "C" int td_set_artifacts(td_ctx* c, const uint8_t* string, const int32_t* s_index, int32_t n_seq,
                                int32_t filter_error, int32_t n_threads)
{
	if (!c) return TD_FAIL;
	HIPCHK(c, hipSetDevice(c->device));
	if (c->d_art_text) { HIPCHK(c, hipFree(c->d_art_text)); c->d_art_text = nullptr; }
	if (c->d_art_index) { HIPCHK(c, hipFree(c->d_art_index)); c->d_art_index = nullptr; }
	c->art_n = 0;
	if (n_seq <= 0) return TD_OK;
	if (!string || !s_index) return fail(c, "td_set_artifacts: null argument");
	if (n_threads < 1) return fail(c, "td_set_artifacts: n_threads = %d", n_threads);
	for (int32_t j = 0; j < n_seq; j++)
		if (s_index[j + 1] < s_index[j] || s_index[j] < 0) return fail(c, "td_set_artifacts: s_index is not ascending at %d", j);
	const size_t bytes = (size_t)s_index[n_seq];
	HIPCHK(c, hipMalloc((void**)&c->d_art_text, bytes ? bytes : 1));
	HIPCHK(c, hipMalloc((void**)&c->d_art_index, sizeof(int32_t) * ((size_t)n_seq + 1)));
	if (bytes) HIPCHK(c, hipMemcpy(c->d_art_text, string, bytes, hipMemcpyHostToDevice));
	HIPCHK(c, hipMemcpy(c->d_art_index, s_index, sizeof(int32_t) * ((size_t)n_seq + 1), hipMemcpyHostToDevice));
	c->art_n = n_seq; c->art_fe = filter_error; c->art_threads = n_threads;
	return TD_OK;
}

extern "C" int td_set_params(td_ctx* c, float threshold, int32_t minlen, int32_t dust)
{
	if (!c) return TD_FAIL;
	c->threshold = threshold; c->minlen = minlen; c->dust = dust;
	return TD_OK;
}

// ---------------------------------------------------------------------------------------------------------
// batches
// ---------------------------------------------------------------------------------------------------------
static inline int64_t align256(int64_t v) { return (v + 255) & ~(int64_t)255; }

static void make_layout(TdWsLayout& L, int S, int H, int C, int lmax, int max_ncol)
{
	int64_t o = 0;
	L.codes = o; o = align256(o + (int64_t)(lmax + 2) * TD_WAVE);
	L.sb = o;    o = align256(o + (int64_t)(S + 1) * (lmax + 2) * TD_WAVE * 4);
	L.sf = o;    o = align256(o + (int64_t)(S + 1) * (lmax + 2) * TD_WAVE * 4);
	L.bw = o;    o = align256(o + (int64_t)C * lmax * TD_WAVE * 8);
	L.fwrow = o; o = align256(o + (int64_t)max_ncol * TD_WAVE * 8);
	L.dp = o;    o = align256(o + (int64_t)lmax * H * TD_WAVE * 4);
	L.path = o;  o = align256(o + (int64_t)lmax * H * TD_WAVE);
	L.acc = o;   o = align256(o + (int64_t)H * TD_WAVE * 4);
	L.total = o; o = align256(o + (int64_t)H * TD_WAVE * 4);
	L.dust = o;  o = align256(o + (int64_t)64 * TD_WAVE);
	L.slot_bytes = o;
}

// output block: SoA arrays over n_tiles*64 reads, then keep words, then labels
struct OutLayout { int64_t f, b, r, bar, q, type, barcode, finger, keep, labels, total; };
static OutLayout out_layout(int64_t n_tiles, int lmax, int nw1)
{
	OutLayout o; int64_t p = 0; const int64_t n = n_tiles * TD_WAVE;
	o.f = p; p = align256(p + n * 4); o.b = p; p = align256(p + n * 4); o.r = p; p = align256(p + n * 4);
	o.bar = p; p = align256(p + n * 4); o.q = p; p = align256(p + n * 4); o.type = p; p = align256(p + n * 4);
	o.barcode = p; p = align256(p + n * 4); o.finger = p; p = align256(p + n * 4);
	o.keep = p; p = align256(p + n_tiles * nw1 * TD_WAVE * 4);
	o.labels = p; p = align256(p + n_tiles * (int64_t)(lmax + 1) * TD_WAVE);
	o.total = p;
	return o;
}

static int upload_common(td_ctx* c, const uint8_t* codes, const char* ascii, const int64_t* offs, int64_t n)
{
	if (!c) return TD_FAIL;
	if (!c->have_model) return fail(c, "td_batch_upload: no model uploaded");
	if ((!codes && !ascii) || !offs || n < 0) return fail(c, "td_batch_upload: bad arguments");
	HIPCHK(c, hipSetDevice(c->device));
	int lmax = 1;
	for (int64_t i = 0; i < n; i++) {
		const int64_t l = offs[i + 1] - offs[i];
		if (l < 0 || l > 100000) return fail(c, "td_batch_upload: read %lld has length %lld", (long long)i, (long long)l);
		if (l > lmax) lmax = (int)l;
	}
	const int64_t n_tiles = (n + TD_WAVE - 1) / TD_WAVE;
	const int nw2 = (lmax + 15) / 16, nw1 = (lmax + 31) / 32;
	const int64_t tile_words = (int64_t)(nw2 + nw1) * TD_WAVE;

	// pack: 2 bits per base + 1 bit "is N" per base, lane-interleaved per tile so a wave reads 256 contiguous bytes per word
	const size_t n_packed = (size_t)(n_tiles * tile_words), n_lens = (size_t)(n_tiles * TD_WAVE);
	if (ensure_pinned(c, &c->h_packed, &c->cap_h_packed, n_packed * 4) != TD_OK) return TD_FAIL;
	if (ensure_pinned(c, &c->h_lens, &c->cap_h_lens, n_lens * 4) != TD_OK) return TD_FAIL;
	uint32_t* const packed = c->h_packed;
	int32_t* const lens = c->h_lens;
	parallel_ranges((int64_t)n_packed, [&](int64_t lo, int64_t hi) { memset(packed + lo, 0, (size_t)(hi - lo) * 4); });
	memset(lens, 0, n_lens * 4);
	c->codes_host.resize((size_t)offs[n]);
	// init_nuc_code(), src/nuc_code.c:46-74: ACGTU (either case) -> 0..3(3), everything else 4 (thread-safe static init)
	struct AscTable {
		uint8_t t[256];
		AscTable()
		{
			for (int k = 0; k < 256; k++) t[k] = 4;
			t['A'] = t['a'] = 0; t['C'] = t['c'] = 1; t['G'] = t['g'] = 2; t['T'] = t['t'] = 3; t['U'] = t['u'] = 3;
		}
	};
	static const AscTable asc_table;
	const uint8_t* asc2code = asc_table.t;
	// Reads are laid out on the device sorted by length (stable counting sort), so that the 64 reads of a tile have
	// (nearly) the same length and no lane idles through another read's extra positions; results are un-permuted on
	// download.  Per-read results do not depend on the order (every read is decoded independently).
	c->pos_of.assign((size_t)n, 0);
	{
		std::vector<int64_t> start((size_t)lmax + 2, 0);
		for (int64_t i = 0; i < n; i++) start[(size_t)(offs[i + 1] - offs[i]) + 1]++;
		for (int l = 0; l <= lmax; l++) start[(size_t)l + 1] += start[(size_t)l];
		for (int64_t i = 0; i < n; i++) c->pos_of[(size_t)i] = start[(size_t)(offs[i + 1] - offs[i])]++;
	}
	// every read owns its (tile, lane) words, so reads can be packed by several host threads without synchronisation
	auto pack_range = [&](int64_t lo, int64_t hi) {
		for (int64_t i = lo; i < hi; i++) {
			const int64_t k = c->pos_of[(size_t)i];
			const int64_t tile = k / TD_WAVE; const int lane = (int)(k % TD_WAVE);
			const int l = (int)(offs[i + 1] - offs[i]);
			lens[(size_t)k] = l;
			uint32_t* pk = packed + tile * tile_words;
			uint32_t w2 = 0, w1 = 0;   // a word is built in a register and stored once
			for (int kk = 0; kk < l; kk++) {
				uint8_t cd = codes ? codes[offs[i] + kk] : asc2code[(uint8_t)ascii[offs[i] + kk]];
				if (cd > 4) cd = 4;
				c->codes_host[(size_t)(offs[i] + kk)] = cd;
				if (cd == 4) w1 |= 1u << (kk & 31);
				else w2 |= (uint32_t)cd << (2 * (kk & 15));
				if ((kk & 15) == 15 || kk == l - 1) { pk[(kk >> 4) * TD_WAVE + lane] = w2; w2 = 0; }
				if ((kk & 31) == 31 || kk == l - 1) { pk[(nw2 + (kk >> 5)) * TD_WAVE + lane] = w1; w1 = 0; }
			}
		}
	};
	parallel_ranges(n, pack_range);
	if (ensure(c, &c->d_packed, &c->cap_packed, n_packed * 4) != TD_OK) return TD_FAIL;
	if (ensure(c, &c->d_lens, &c->cap_lens, n_lens * 4) != TD_OK) return TD_FAIL;
	if (n_packed) HIPCHK(c, hipMemcpyAsync(c->d_packed, packed, n_packed * 4, hipMemcpyHostToDevice, c->stream));
	if (n_lens) HIPCHK(c, hipMemcpyAsync(c->d_lens, lens, n_lens * 4, hipMemcpyHostToDevice, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));

	c->n_reads = n; c->n_tiles = (int32_t)n_tiles; c->lmax = lmax; c->nw2 = nw2; c->nw1 = nw1;
	c->offs.assign(offs, offs + n + 1);
	c->ran = false;

	// outputs + workspace
	const OutLayout ol = out_layout(n_tiles, lmax, nw1);
	if (ensure(c, &c->d_out, &c->cap_out, (size_t)ol.total) != TD_OK) return TD_FAIL;
	make_layout(c->lay, c->hdr.S, c->hdr.H, c->hdr.C, lmax, c->hdr.max_ncol);
	int64_t slot_bytes = c->lay.slot_bytes;
	if (c->spec_ready && c->spec_oob && !spec_lsum_range_ok(c, lmax)) {
		if (load_spec_kernel(c, 0) != TD_OK) return TD_FAIL;   // reads this long need the clamped logsum (seconds, once)
	}
	if (c->spec_ready) {
		td_model_desc md{};
		md.S = c->hdr.S; md.H = c->hdr.H; md.C = c->hdr.C;
		md.n_hmm = c->m_n_hmm.data(); md.n_col = c->m_n_col.data(); md.trans = c->m_trans.data();
		td_spec_layout(c->slay, &md, lmax);
		slot_bytes = c->slay.slot_bytes;
	}
	// wave slots: enough to fill the chip (2 workgroups of 4 waves per CU share the LDS), bounded by HBM
	const int wpb = (c->spec_ready ? c->spec_block : td_kernel_block_threads()) / TD_WAVE;
	int64_t want = (int64_t)c->n_cu * (c->spec_ready ? c->spec_waves_per_cu : 2 * wpb);
	if (const char* e = getenv("TD_WAVE_SLOTS")) { const long v = atol(e); if (v > 0) want = v; }
	size_t free_b = 0, total_b = 0;
	HIPCHK(c, hipMemGetInfo(&free_b, &total_b));
	const int64_t budget = (int64_t)((double)(free_b + c->cap_ws) * 0.85);
	int64_t slots = want;
	if (slots * slot_bytes > budget) slots = budget / slot_bytes;
	if (slots > n_tiles) slots = n_tiles;
	if (slots < 1) {
		if (n_tiles == 0) slots = 1;
		else return fail(c, "td_batch_upload: workspace of %lld bytes per wave does not fit in HBM", (long long)slot_bytes);
	}
	slots = (slots + wpb - 1) / wpb * wpb; // whole workgroups
	if (ensure(c, &c->d_ws, &c->cap_ws, (size_t)(slots * slot_bytes)) != TD_OK) return TD_FAIL;
	c->n_slots = (int32_t)slots;
	return TD_OK;
}

extern "C" int td_batch_upload(td_ctx* c, const uint8_t* codes, const int64_t* offs, int64_t n)
{
	return upload_common(c, codes, nullptr, offs, n);
}

extern "C" int td_batch_upload_ascii(td_ctx* c, const char* bases, const int64_t* offs, int64_t n)
{
	return upload_common(c, nullptr, bases, offs, n);
}

extern "C" int td_run(td_ctx* c, int mode)
{
	if (!c) return TD_FAIL;
	if (!c->have_model) return fail(c, "td_run: no model uploaded");
	if (mode != TD_MODE_GET_LABEL && mode != TD_MODE_GET_PROB && mode != TD_MODE_ARCH_COMP) return fail(c, "td_run: unsupported mode %d", mode);
	HIPCHK(c, hipSetDevice(c->device));
	if (c->n_tiles == 0) { c->ran = true; c->last_ms = 0.0f; return TD_OK; }
	const OutLayout ol = out_layout(c->n_tiles, c->lmax, c->nw1);
	TdKernelArgs ka{};
	ka.hdr = c->d_hdr; ka.cols = c->d_cols; ka.hinfo = c->d_hinfo;
	ka.pred_off = c->d_pred_off; ka.pred_idx = c->d_pred_idx; ka.logsum = c->d_logsum;
	ka.packed = c->d_packed; ka.lens = c->d_lens;
	ka.n_tiles = c->n_tiles; ka.n_slots = c->n_slots; ka.lmax = c->lmax; ka.nw2 = c->nw2; ka.nw1 = c->nw1;
	ka.mode = mode; ka.threshold = c->threshold; ka.minlen = c->minlen; ka.dust = c->dust; ka.want_labels = 1;
	ka.out_f = (float*)(c->d_out + ol.f); ka.out_b = (float*)(c->d_out + ol.b); ka.out_r = (float*)(c->d_out + ol.r);
	ka.out_bar = (float*)(c->d_out + ol.bar); ka.out_q = (float*)(c->d_out + ol.q);
	ka.out_type = (int32_t*)(c->d_out + ol.type); ka.out_barcode = (int32_t*)(c->d_out + ol.barcode);
	ka.out_finger = (int32_t*)(c->d_out + ol.finger);
	ka.out_keep = (uint32_t*)(c->d_out + ol.keep); ka.out_labels = (int8_t*)(c->d_out + ol.labels);
	ka.counters = c->d_counters;
	ka.ws = c->d_ws; ka.lay = c->lay;
	if (c->art_n > 0 && mode == TD_MODE_GET_LABEL) {
		// match_to_reference takes the reads of each thread range [t*interval, ...) in fours and gives the
		// (range length mod 4) left-over reads to another routine (barcode_hmm.c:2495-2575, ranges :1911-1922)
		const int64_t n = c->n_reads, T = c->art_threads, interval = n / T;
		std::vector<uint8_t> left((size_t)c->n_tiles * TD_WAVE, 0);
		for (int64_t t = 0; t < T; t++) {
			const int64_t start = t * interval, end = (t == T - 1) ? n : (t + 1) * interval;
			for (int64_t i = start + (end - start) / 4 * 4; i < end; i++) left[(size_t)c->pos_of[(size_t)i]] = 1;
		}
		if (ensure(c, &c->d_art_left, &c->cap_art_left, left.size()) != TD_OK) return TD_FAIL;
		HIPCHK(c, hipMemcpyAsync(c->d_art_left, left.data(), left.size(), hipMemcpyHostToDevice, c->stream));
		HIPCHK(c, hipStreamSynchronize(c->stream));   // `left` goes out of scope
		ka.art_text = c->d_art_text; ka.art_index = c->d_art_index; ka.art_left = c->d_art_left;
		ka.art_n = c->art_n; ka.art_fe = c->art_fe;
	}
	HIPCHK(c, hipEventRecord(c->ev0, c->stream));
	if (c->spec_ready) {
		TdSpecArgs sa{};
		sa.logsum = ka.logsum; sa.packed = ka.packed; sa.lens = ka.lens;
		sa.n_tiles = ka.n_tiles; sa.n_slots = ka.n_slots; sa.lmax = ka.lmax; sa.nw2 = ka.nw2; sa.nw1 = ka.nw1;
		sa.mode = ka.mode; sa.threshold = ka.threshold; sa.minlen = ka.minlen; sa.dust = ka.dust;
		sa.out_f = ka.out_f; sa.out_b = ka.out_b; sa.out_r = ka.out_r; sa.out_bar = ka.out_bar; sa.out_q = ka.out_q;
		sa.out_type = ka.out_type; sa.out_barcode = ka.out_barcode; sa.out_finger = ka.out_finger;
		sa.out_keep = ka.out_keep; sa.out_labels = ka.out_labels; sa.counters = ka.counters;
		sa.art_text = ka.art_text; sa.art_index = ka.art_index; sa.art_left = ka.art_left; sa.art_n = ka.art_n; sa.art_fe = ka.art_fe;
		sa.ws = ka.ws; sa.lay = c->slay;
		size_t sz = sizeof sa;
		void* cfg[] = { HIP_LAUNCH_PARAM_BUFFER_POINTER, &sa, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END };
		const int wpb = c->spec_block / TD_WAVE;
		const unsigned blocks = (unsigned)((c->n_slots + wpb - 1) / wpb);
		HIPCHK(c, hipModuleLaunchKernel(c->spec_fn, blocks, 1, 1, (unsigned)c->spec_block, 1, 1, 0, c->stream, nullptr, cfg));
	} else {
		HIPCHK(c, td_launch_decode(&ka, c->stream));
	}
	HIPCHK(c, hipEventRecord(c->ev1, c->stream));
	c->ran = true;
	c->last_ms = -1.0f;
	return TD_OK;
}

extern "C" int td_sync(td_ctx* c)
{
	if (!c) return TD_FAIL;
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	return TD_OK;
}

extern "C" int td_last_kernel_ms(td_ctx* c, float* ms)
{
	if (!c || !ms) return TD_FAIL;
	if (!c->ran) return fail(c, "td_last_kernel_ms: nothing has run");
	HIPCHK(c, hipSetDevice(c->device));
	if (c->last_ms < 0.0f && c->n_tiles > 0) {
		HIPCHK(c, hipEventSynchronize(c->ev1));
		HIPCHK(c, hipEventElapsedTime(&c->last_ms, c->ev0, c->ev1));
	}
	*ms = c->last_ms;
	return TD_OK;
}

extern "C" int td_batch_info(td_ctx* c, int64_t* n_reads, int64_t* workspace_bytes, int32_t* wave_slots)
{
	if (!c) return TD_FAIL;
	if (n_reads) *n_reads = c->n_reads;
	if (workspace_bytes) *workspace_bytes = (int64_t)c->n_slots * (c->spec_ready ? c->slay.slot_bytes : c->lay.slot_bytes);
	if (wave_slots) *wave_slots = c->n_slots;
	return TD_OK;
}

extern "C" int td_batch_download(td_ctx* c, td_read_result* res, int8_t* labels, uint8_t* seq_out)
{
	if (!c) return TD_FAIL;
	if (!c->ran) return fail(c, "td_batch_download: td_run has not been called on this batch");
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	const int64_t n = c->n_reads;
	if (n == 0) return TD_OK;
	const OutLayout ol = out_layout(c->n_tiles, c->lmax, c->nw1);
	// one pinned staging buffer with the device layout; only the regions asked for cross the link
	if (ensure_pinned(c, &c->h_out, &c->cap_h_out, (size_t)ol.total) != TD_OK) return TD_FAIL;
	const size_t keep_bytes = (size_t)c->n_tiles * c->nw1 * TD_WAVE * 4;
	const size_t label_bytes = (size_t)c->n_tiles * (c->lmax + 1) * TD_WAVE;
	if (res) HIPCHK(c, hipMemcpyAsync(c->h_out, c->d_out, (size_t)ol.keep, hipMemcpyDeviceToHost, c->stream));
	if (seq_out) HIPCHK(c, hipMemcpyAsync(c->h_out + ol.keep, c->d_out + ol.keep, keep_bytes, hipMemcpyDeviceToHost, c->stream));
	if (labels) HIPCHK(c, hipMemcpyAsync(c->h_out + ol.labels, c->d_out + ol.labels, label_bytes, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	const uint8_t* h = c->h_out;
	const float* f = (const float*)(h + ol.f); const float* b = (const float*)(h + ol.b);
	const float* r = (const float*)(h + ol.r); const float* bar = (const float*)(h + ol.bar);
	const float* q = (const float*)(h + ol.q); const int32_t* ty = (const int32_t*)(h + ol.type);
	const int32_t* bc = (const int32_t*)(h + ol.barcode); const int32_t* fg = (const int32_t*)(h + ol.finger);
	const int8_t* hl = (const int8_t*)(h + ol.labels);
	const uint32_t* hk = (const uint32_t*)(h + ol.keep);
	// un-permute (device order is length-sorted) and un-interleave ([..][64 lanes] -> per read), all three in one pass
	parallel_ranges(n, [&](int64_t lo, int64_t hi) {
		for (int64_t i = lo; i < hi; i++) {
			const int64_t k = c->pos_of[(size_t)i];
			const int64_t tile = k / TD_WAVE; const int lane = (int)(k % TD_WAVE);
			const int l = (int)(c->offs[i + 1] - c->offs[i]);
			if (res) {
				res[i].f_score = f[k]; res[i].b_score = b[k]; res[i].r_score = r[k]; res[i].bar_prob = bar[k];
				res[i].mapq = q[k]; res[i].read_type = ty[k]; res[i].barcode = bc[k]; res[i].fingerprint = fg[k];
			}
			if (labels) {
				const int8_t* src = hl + tile * (int64_t)(c->lmax + 1) * TD_WAVE + lane;
				int8_t* dst = labels + c->offs[i] + i;
				for (int kk = 0; kk <= l; kk++) dst[kk] = src[kk * TD_WAVE];
			}
			if (seq_out) {
				const uint32_t* kw = hk + tile * (int64_t)c->nw1 * TD_WAVE + lane;
				const uint8_t* cd = c->codes_host.data() + c->offs[i];
				uint8_t* dst = seq_out + c->offs[i];
				for (int k0 = 0; k0 < l; k0 += 32) {
					const uint32_t w = kw[(k0 >> 5) * TD_WAVE];
					const int e = l - k0 < 32 ? l - k0 : 32;
					if (w == 0xFFFFFFFFu) memcpy(dst + k0, cd + k0, (size_t)e);      // whole word kept (unextracted reads)
					else for (int kk = 0; kk < e; kk++) dst[k0 + kk] = ((w >> kk) & 1u) ? cd[k0 + kk] : 65; // spacer byte, barcode_hmm.c:3348
				}
			}
		}
	});
	return TD_OK;
}

// ---------------------------------------------------------------------------------------------------------
// counters
// ---------------------------------------------------------------------------------------------------------
extern "C" int td_counts_reset(td_ctx* c)
{
	if (!c) return TD_FAIL;
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipMemsetAsync(c->d_counters, 0, sizeof(unsigned long long) * TD_NUM_COUNTERS, c->stream));
	return TD_OK;
}

extern "C" int td_counts_get(td_ctx* c, int64_t* counts)
{
	if (!c || !counts) return TD_FAIL;
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	HIPCHK(c, hipMemcpy(counts, c->d_counters, sizeof(int64_t) * TD_NUM_COUNTERS, hipMemcpyDeviceToHost));
	return TD_OK;
}

extern "C" void* td_counts_device_ptr(td_ctx* c) { return c ? (void*)c->d_counters : nullptr; }
