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
#include <time.h>
#include <unistd.h>

#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/tagdust_hip.h"
#include "td_device.h"
#include "td_jit.h"
#include "td_stage.h"
#include "td_host_inner.h"

extern "C" __attribute__((visibility("hidden"))) hipError_t td_launch_decode(const TdKernelArgs* ka, hipStream_t stream);   // library-internal
extern "C" __attribute__((visibility("hidden"))) int td_kernel_block_threads(void);
extern "C" __attribute__((visibility("hidden"))) hipError_t td_launch_decode_multi(const TdKernelArgs* d_args, int n_models, int max_slots, hipStream_t stream);

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

#define TD_MAX_PIPELINE 4
// the device counter block: what the ABI reports, then the diagnostic tail of the development knobs (td_diag_get)
#define TD_COUNTER_WORDS (TD_NUM_COUNTERS + TD_NUM_DIAG_COUNTERS)

// One batch on its way through the device (see "batches" below).  What a slot knows about ITS BATCH lives in three groups, each
// value-initialised as a whole by the step that owns it -- so that nothing a branch of that step does not set can survive from
// the batch before (round 3's soak fault was exactly that: a label-run table of the previous, smaller batch):
//   TdStaged  <- slot_stage():       the reads as staged on the device and the workspace geometry chosen for them
//   TdDecoded <- slot_decode():      what the last launch over the staged batch was and left behind
//   TdFetch   <- slot_fetch_begin(): where the results go and in which form they travel
// slot_stage() also resets the two later groups, slot_decode() the last one.  What is left in TdSlot itself belongs to the slot,
// not to a batch: device / pinned buffers with their capacities, events, the ticket.
struct TdRoute {             // where a batch runs (chosen by the caller of slot_stage, per batch)
	hipStream_t cs = nullptr;   // the compute stream (c->stream, or c->stream2 for every other pipelined batch)
	hipStream_t aux = nullptr;  // the stream of its sort / pack kernels: cs itself, or the context's high-priority stream for them
	hipStream_t fin = nullptr;  // ... and of its finish kernel (a stream of its own: it waits for the decode kernel, the next batch's pack must not)
	int wsi = 0;                // ... and the workspace (0 / 1) that goes with cs
	bool pipelined = false;     // a td_submit batch (the synchronous calls use slot 0 with pipelined = false)
};
struct TdStaged : TdRoute {
	int64_t n_reads = 0, n_bases = 0;
	int32_t n_tiles = 0, lmax = 0, nw2 = 0, nw1 = 0;
	int is_ascii = 0;
	bool sorted = false;      // device order differs from the caller's (reads of several lengths)
	bool staged = false;      // inputs are packed on the device: td_run may launch
	bool raw_direct = false;  // the upload read the caller's page-locked buffer itself (no staging copy)
	const uint8_t* raw_host = nullptr;   // the batch's bases on the host, valid until the batch has been waited for: the pinned staging
	                                     // copy, or the caller's own page-locked buffer under the "stable_input" contract; else NULL
	TdStageBatch sb{};
	TdWsLayout lay{};
	TdSpecLayout slay{};
	int32_t n_wave_slots = 0;
	int64_t ws_slot_bytes = 0;
	// length classes (specialised kernel): the n_long longest tiles are longer than lmax_small, the geometry of most wave slots;
	// n_big >= n_long slots keep the geometry of the batch's longest read (slay_big).  n_long = 0: one geometry.
	int32_t n_long = 0, lmax_small = 0, n_big = 0;
	TdSpecLayout slay_big{};
	int64_t ws_bytes = 0;       // workspace bytes this batch's launch uses
};
struct TdDecoded {
	int mode = 0;
	bool ran = false;
	float last_ms = -1.0f;
	int32_t runs_cap = 0;       // entries per read in d_runs (0: the last launch left no label runs)
};
struct TdFetch {
	td_read_result* u_res = nullptr; int8_t* u_labels = nullptr; uint8_t* u_seq = nullptr;   // the caller's output buffers
	bool res_direct = false, lab_direct = false, seq_direct = false;                          // ... are page-locked
	bool copies_deferred = false;   // td_wait issues the device-to-host copies (pipelined calls)
	bool use_keep = false, use_rle = false;   // compact egress: keep bits instead of the rewritten sequence, label runs instead of labels
	int32_t rle_cap = 0;
	bool finished = false;          // the finish kernel is queued: slot_fetch_end has something to collect
};
struct TdSlot : TdStaged, TdDecoded, TdFetch {
	int64_t ticket = 0;       // td_submit: 0 = free
	void reset_staged(const TdRoute& r) { static_cast<TdStaged&>(*this) = TdStaged(); static_cast<TdRoute&>(*this) = r; reset_decoded(); }
	void reset_decoded() { static_cast<TdDecoded&>(*this) = TdDecoded(); reset_fetch(); }
	void reset_fetch() { static_cast<TdFetch&>(*this) = TdFetch(); }
	// device
	uint8_t* d_raw = nullptr;      size_t cap_raw = 0;
	int64_t* d_offs = nullptr;     size_t cap_offs = 0;
	int32_t* d_read_at = nullptr;  size_t cap_read_at = 0;
	uint32_t* d_keys = nullptr;    size_t cap_keys = 0;
	int32_t* d_vals = nullptr;     size_t cap_vals = 0;
	uint8_t* d_sort_tmp = nullptr; size_t cap_sort_tmp = 0;
	uint32_t* d_packed = nullptr;  size_t cap_packed = 0;
	int32_t* d_lens = nullptr;     size_t cap_lens = 0;
	uint8_t* d_art_left = nullptr; size_t cap_art_left = 0;
	uint8_t* d_out = nullptr;      size_t cap_out = 0;    // decode-kernel outputs, device order
	uint8_t* d_res = nullptr;      size_t cap_res = 0;    // results in the caller's order
	uint8_t* d_seq = nullptr;      size_t cap_seq = 0;
	int8_t*  d_lab = nullptr;      size_t cap_lab = 0;
	// compact egress: what the host rebuilds the rewritten sequences and the labels from (slot_fetch_begin)
	uint32_t* d_keepo = nullptr;   size_t cap_keepo = 0;
	uint32_t* d_rle = nullptr;     size_t cap_rle = 0;     // (+ one word behind the runs: the overflow flag)
	uint32_t* d_runs = nullptr;    size_t cap_runs = 0;    // label runs in device order, left by the specialised kernel (+ the overflow flag)
	uint32_t* h_keepo = nullptr;   size_t cap_h_keepo = 0;
	uint32_t* h_rle = nullptr;     size_t cap_h_rle = 0;
	// pinned host staging for pageable caller memory
	uint8_t* h_raw = nullptr;      size_t cap_h_raw = 0;
	int64_t* h_offs = nullptr;     size_t cap_h_offs = 0;
	uint8_t* h_res = nullptr;      size_t cap_h_res = 0;
	uint8_t* h_seq = nullptr;      size_t cap_h_seq = 0;
	int8_t*  h_lab = nullptr;      size_t cap_h_lab = 0;
	hipEvent_t ev_up = nullptr, ev_k0 = nullptr, ev_k1 = nullptr, ev_done = nullptr, ev_down = nullptr, ev_pack = nullptr;
};

// Host threads a context may use for its copies between pageable caller memory and pinned staging: TD_HOST_THREADS, else the
// machine's threads, at most 16 (option "host_threads"; td_multi_create shares the machine's threads out over its devices).
static int default_host_threads()
{
	int nt = (int)std::thread::hardware_concurrency();
	if (const char* e = getenv("TD_HOST_THREADS")) nt = atoi(e);
	if (nt > 16) nt = 16;
	if (nt < 1) nt = 1;
	return nt;
}

// The copy threads of one context: started once, reused by every batch (the calling thread takes a share of each copy itself).
// The pipelined calls keep six streams busy (two compute streams, upload, download and two for the small kernels around the
// decode).  The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues, four unless told otherwise: two of the
// six then share one, and whatever sits behind a 20 ms decode launch in its queue waits for it.  Eight leaves every stream its
// own queue (measured on one box: 17.6 / 17.6 ms per 2^20-read step against 17.6 / 18.5 with four).  The runtime reads the
// variable when it initialises, at the process's first HIP call; a value the user has set is left alone.
// That works when this library is loaded before the process's first HIP call (the drop-in binary, the Python harness); a host
// that initialised HIP first keeps the runtime's own default, and can see so: td_get_option("hw_queues") is what the runtime
// was configured with when it initialised as far as this library can tell, "hw_queues_late" says the request came too late.
// (Whether the runtime is up is read off the process's open files: initialising it opens /dev/kfd.)
static int g_hwq_user = 0;       // GPU_MAX_HW_QUEUES as the user had set it when the library was loaded (0: not set)
static bool g_hwq_late = false;  // ... the HIP runtime was already initialised then
static bool kfd_is_open()
{
	char path[64], target[64];
	for (int fd = 0; fd < 256; fd++) {
		snprintf(path, sizeof path, "/proc/self/fd/%d", fd);
		const ssize_t n = readlink(path, target, sizeof target - 1);
		if (n <= 0) continue;
		target[n] = 0;
		if (!strcmp(target, "/dev/kfd")) return true;
	}
	return false;
}
__attribute__((constructor)) static void td_want_hw_queues()
{
	if (const char* e = getenv("GPU_MAX_HW_QUEUES")) { g_hwq_user = atoi(e) > 0 ? atoi(e) : 0; if (g_hwq_user) return; }
	g_hwq_late = kfd_is_open();
	if (!g_hwq_late) setenv("GPU_MAX_HW_QUEUES", "8", 0);
}

struct td_ctx {
	int device = 0;
	int host_threads = default_host_threads();
	CopyPool pool;
	hipStream_t stream = nullptr;
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
	bool spec_window = false;   // the loaded kernel has the -start/-end window arithmetic compiled in
	bool spec_oob_unsafe = false; // the clamp-free form failed its self-check once: never again in this context
	float m_maxabs = 0.0f;      // largest |finite parameter|
	int spec_block = 256, spec_waves_per_cu = 8;

	// params
	float threshold = 0.0f;
	int32_t minlen = 16, dust = 100;

	// -ref artifact filter
	uint8_t* d_art_text = nullptr; int32_t* d_art_index = nullptr;
	int32_t art_n = 0, art_fe = 0, art_threads = 1;
	int64_t win_first = 0, win_total = 0;   // td_set_batch_window
	int32_t match_start = 0, match_len = 0;  // td_set_window (-start / -end); match_len = 0: whole reads
	// batches: slot 0 is the resident batch of the synchronous calls; td_submit rotates over pipeline_depth slots
	TdSlot slots[TD_MAX_PIPELINE];
	int pipeline_depth = 3, next_slot = 0, last_slot = 0;
	bool counted = false;   // td_ctx_create finished: this context counts among the live ones (the last one to go frees the stream cache)
	int poison = 0;   // option "poison_workspace": fill the workspace with 0xFF bytes before every decode launch (tests)
	// development / test knobs: read from the environment ONCE, when the context is created (never on the per-batch path), and
	// settable afterwards through td_set_option under the names in brackets
	int compact_egress = 1;    // TD_COMPACT_EGRESS ["compact_egress"]: keep bits + label runs instead of plain copies
	int stable_input = 0;      // ["stable_input"]: the caller leaves a page-locked input buffer alone until td_wait (see tagdust_hip.h)
	int rle_cap_forced = 0;    // TD_RLE_CAP ["rle_cap"]: entries of the label-run table (0: S + 2)
	int length_classes = 1;    // TD_NO_LENGTH_CLASSES ["length_classes_enabled"]
	int debug_wait = 0;        // TD_DEBUG_WAIT ["debug_wait"]
	int debug_alloc = 0;       // TD_DEBUG_ALLOC
	long wave_slots_forced = 0;   // TD_WAVE_SLOTS
	int ws_candidates = 3;     // TD_WS_CANDIDATES
	double lsum_limit = 1.0e6; // TD_SPEC_LSUM_LIMIT (tests: force the switch to the clamped logsum)
	int selfcheck_fail = 0;    // TD_SPEC_SELFCHECK_FAIL (tests: exercise the fallback)
	int64_t ticket_counter = 0;
	hipStream_t s_up = nullptr, s_down = nullptr;   // copy streams of the pipelined calls
	uint8_t* d_ws = nullptr;      size_t cap_ws = 0;  // workspace of the decode kernels on `stream` (they run one after the other)
	// Pipelined batches alternate between two compute streams with a workspace each: a launch ends with its slowest wave
	// (the waves of some XCDs take ~10 % longer for the same tiles), and the next batch's workgroups move in as the first
	// one's retire instead of waiting for the last (option "overlap_decode", TD_OVERLAP; off when HBM cannot hold both).
	hipStream_t stream2 = nullptr;
	// With two decode kernels queued the machine never falls idle, so the small kernels around them (sort / pack of the next
	// batch, finish of the last one) and the download's blit kernels would wait for a whole decode kernel: they run on a
	// high-priority stream and take the compute units the retiring workgroups free before the next decode kernel does.
	hipStream_t s_aux = nullptr, s_fin = nullptr;
	uint8_t* d_ws2 = nullptr;     size_t cap_ws2 = 0;
	int32_t* d_tile_next2 = nullptr;
	int overlap = 1, submit_parity = 0;
	bool half_slots = false;   // two workspaces of the full slot count do not fit: the pipelined launches use half the slots each
	// position pruning tables of the specialised kernel (td_spec_prune_tables), for reads up to prune_lcap bases
	float* d_prune = nullptr;     int prune_lcap = 0, prune_stride = 0;
	bool prune_live = false;      // ... and they are real bounds (not the all-zero tables of reads beyond 8192 bases)
	int32_t* d_tile_next = nullptr;   // tile counter of the specialised kernel's dynamic tile assignment
	hipEvent_t ev_origin = nullptr;   // td_timeline_origin: the common origin of td_last_kernel_times
};

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

// everything queued on the compute streams has finished
static hipError_t sync_compute(td_ctx* c)
{
	hipError_t e = hipStreamSynchronize(c->stream);
	if (e == hipSuccess && c->stream2) e = hipStreamSynchronize(c->stream2);
	if (e == hipSuccess && c->s_aux) e = hipStreamSynchronize(c->s_aux);
	if (e == hipSuccess && c->s_fin) e = hipStreamSynchronize(c->s_fin);
	return e;
}

template <typename T>
static int ensure(td_ctx* c, T** p, size_t* cap, size_t bytes)
{
	if (*cap >= bytes && *p) return TD_OK;
	if (*p) { HIPCHK(c, hipFree(*p)); *p = nullptr; *cap = 0; }
	if (bytes == 0) bytes = 256;
	HIPCHK(c, hipMalloc((void**)p, bytes));
	*cap = bytes;
	if (c && c->debug_alloc && bytes > (1u << 30)) fprintf(stderr, "tagdust_hip: hipMalloc(%zu) = %p\n", bytes, (void*)*p);
	return TD_OK;
}

template <typename T>
static int ensure_pinned(td_ctx* c, T** p, size_t* cap, size_t bytes)
{
	if (*cap >= bytes && *p) return TD_OK;
	if (*p) { HIPCHK(c, hipHostFree(*p)); *p = nullptr; *cap = 0; }
	if (bytes == 0) bytes = 256;
	bytes += bytes / 4;   // head room: batches of a run differ a little in size
	HIPCHK(c, hipHostMalloc((void**)p, bytes, hipHostMallocPortable));   // (several devices of one process may DMA from it)
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

extern "C" void td_stream_release(void);   // td_stream.cpp: the page-locked batch buffers td_stream_run keeps between runs
static std::mutex g_live_mu;
static int g_live_ctx = 0;

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
	if (const char* e = getenv("TD_OVERLAP")) c->overlap = atoi(e) != 0;
	if (const char* e = getenv("TD_COMPACT_EGRESS")) c->compact_egress = atoi(e) != 0;
	if (const char* e = getenv("TD_RLE_CAP")) { const int v = atoi(e); if (v >= 1 && v <= 127) c->rle_cap_forced = v; }
	if (getenv("TD_NO_LENGTH_CLASSES")) c->length_classes = 0;
	if (getenv("TD_DEBUG_WAIT")) c->debug_wait = 1;
	if (getenv("TD_DEBUG_ALLOC")) c->debug_alloc = 1;
	if (const char* e = getenv("TD_WAVE_SLOTS")) { const long v = atol(e); if (v > 0) c->wave_slots_forced = v; }
	if (const char* e = getenv("TD_WS_CANDIDATES")) c->ws_candidates = atoi(e);
	if (const char* e = getenv("TD_SPEC_LSUM_LIMIT")) c->lsum_limit = atof(e);
	if (getenv("TD_SPEC_SELFCHECK_FAIL")) c->selfcheck_fail = 1;
	c->hbm_total = prop.totalGlobalMem;
	init_logsum_host();
	bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess &&
	          hipMalloc((void**)&c->d_logsum, sizeof(float) * TD_LOGSUM_LIVE) == hipSuccess &&
	          hipMalloc((void**)&c->d_counters, sizeof(unsigned long long) * TD_COUNTER_WORDS) == hipSuccess &&
	          hipMemcpy(c->d_logsum, g_logsum, sizeof(float) * TD_LOGSUM_LIVE, hipMemcpyHostToDevice) == hipSuccess &&
	          hipMemset(c->d_counters, 0, sizeof(unsigned long long) * TD_COUNTER_WORDS) == hipSuccess;
	if (!ok) {
		td_ctx_destroy(c);
		return fail(nullptr, "td_ctx_create: HIP resource allocation failed: %s", hipGetErrorString(hipGetLastError()));
	}
	{ std::lock_guard<std::mutex> lk(g_live_mu); g_live_ctx++; c->counted = true; }
	*out = c;
	return TD_OK;
}

static void slot_release(TdSlot& s);
static bool tickets_outstanding(const td_ctx* c);

extern "C" void td_ctx_destroy(td_ctx* c)
{
	if (!c) return;
	(void)hipSetDevice(c->device);
	if (c->stream) (void)hipStreamSynchronize(c->stream);
	if (c->stream2) (void)hipStreamSynchronize(c->stream2);
	if (c->s_aux) (void)hipStreamSynchronize(c->s_aux);
	if (c->s_fin) (void)hipStreamSynchronize(c->s_fin);
	if (c->s_up) (void)hipStreamSynchronize(c->s_up);
	if (c->s_down) (void)hipStreamSynchronize(c->s_down);
	void* bufs[] = { c->d_hdr, c->d_cols, c->d_hinfo, c->d_pred_off, c->d_pred_idx, c->d_logsum, c->d_counters,
	                 c->d_ws, c->d_art_text, c->d_art_index, c->d_prune, c->d_tile_next, c->d_ws2, c->d_tile_next2 };
	for (void* p : bufs) if (p) (void)hipFree(p);
	for (int k = 0; k < TD_MAX_PIPELINE; k++) slot_release(c->slots[k]);
	if (c->ev_origin) (void)hipEventDestroy(c->ev_origin);
	if (c->s_up) (void)hipStreamDestroy(c->s_up);
	if (c->s_down) (void)hipStreamDestroy(c->s_down);
	if (c->spec_mod) (void)hipModuleUnload(c->spec_mod);
	if (c->stream) (void)hipStreamDestroy(c->stream);
	if (c->stream2) (void)hipStreamDestroy(c->stream2);
	if (c->s_aux) (void)hipStreamDestroy(c->s_aux);
	if (c->s_fin) (void)hipStreamDestroy(c->s_fin);
	bool last = false;
	if (c->counted) { std::lock_guard<std::mutex> lk(g_live_mu); last = --g_live_ctx == 0; }
	delete c;
	// td_stream_run keeps up to 1 GiB of page-locked batch buffers for the next run of the process: with the last context gone
	// there is no next run to serve
	if (last) td_stream_release();
}

// ---------------------------------------------------------------------------------------------------------
// model
// ---------------------------------------------------------------------------------------------------------
// Compile (or fetch from the cache) and load the model-specialised kernel.  lsum_oob selects the clamp-free logsum.
static int load_spec_kernel(td_ctx* c, int lsum_oob, int window = -1)
{
	if (window < 0) window = c->match_len > 0;   // a context with a -start/-end window gets the kernel that can apply it
	if (c->spec_mod) { HIPCHK(c, hipModuleUnload(c->spec_mod)); c->spec_mod = nullptr; }
	c->spec_fn = nullptr; c->spec_ready = false;
	std::vector<char> code;
	std::string log;
	if (td_spec_compile(&c->m_desc, code, log, lsum_oob, window) != TD_OK) return fail(c, "td_model_upload: specialised kernel did not compile: %.400s", log.c_str());
	HIPCHK(c, hipModuleLoadData(&c->spec_mod, code.data()));
	HIPCHK(c, hipModuleGetFunction(&c->spec_fn, c->spec_mod, "td_spec_kernel"));
	// lsum() as compiled against the reference's formula on the operand pairs that matter (either or both operands -inf,
	// gaps just below / at / above the 15.7 cut, huge gaps, equal operands).  The clamp-free form depends on hardware and
	// toolchain behaviour nobody documents; if it ever stops holding, the clamped form is loaded instead -- loudly.
	{
		static const float g[] = { 0.0f, 0.0005f, 0.001f, 1.0f, 15.699f, 15.6999f, 15.7f, 15.7001f, 16.639f, 16.64f, 16.7f, 100.0f, 1.0e5f, 1.0e6f };
		std::vector<float> pairs;
		for (float base : { 0.0f, -3.25f, -700.0f }) {
			for (float d : g) { pairs.push_back(base); pairs.push_back(base - d); pairs.push_back(base - d); pairs.push_back(base); }
			pairs.push_back(base); pairs.push_back(-INFINITY); pairs.push_back(-INFINITY); pairs.push_back(base);
		}
		pairs.push_back(-INFINITY); pairs.push_back(-INFINITY);
		const int n_pairs = (int)(pairs.size() / 2);
		hipFunction_t chk = nullptr;
		HIPCHK(c, hipModuleGetFunction(&chk, c->spec_mod, "td_spec_selfcheck"));
		float* d_pairs = nullptr; int* d_bad = nullptr; int bad = -1;
		HIPCHK(c, hipMalloc((void**)&d_pairs, pairs.size() * 4));
		HIPCHK(c, hipMalloc((void**)&d_bad, 4));
		HIPCHK(c, hipMemcpy(d_pairs, pairs.data(), pairs.size() * 4, hipMemcpyHostToDevice));
		HIPCHK(c, hipMemset(d_bad, 0, 4));
		struct { const float* logsum; const float* pairs; int n; int pad; int* bad; } a = { c->d_logsum, d_pairs, n_pairs, 0, d_bad };
		size_t sz = sizeof a;
		void* cfg[] = { HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END };
		const int block = td_spec_block_threads();
		hipError_t e = n_pairs <= block ? hipModuleLaunchKernel(chk, 1, 1, 1, (unsigned)block, 1, 1, 0, c->stream, nullptr, cfg) : hipErrorInvalidValue;
		if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
		if (e == hipSuccess) e = hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost);
		(void)hipFree(d_pairs); (void)hipFree(d_bad);
		if (e != hipSuccess) return fail(c, "td_model_upload: logsum self-check did not run: %s", hipGetErrorString(e));
		if (c->selfcheck_fail && lsum_oob) bad = 1;   // tests: exercise the fallback
		if (bad != 0) {
			if (!lsum_oob) return fail(c, "td_model_upload: the compiled logsum differs from the reference formula on %d of %d operand pairs", bad, n_pairs);
			fprintf(stderr, "tagdust_hip: clamp-free logsum failed its self-check on this device / toolchain (%d of %d pairs); using the clamped form\n", bad, n_pairs);
			c->spec_oob_unsafe = true;
			return load_spec_kernel(c, 0, window);
		}
	}
	c->spec_ready = true;
	c->spec_window = window != 0;
	c->spec_oob = lsum_oob != 0;
	return TD_OK;
}

// The clamp-free logsum of the specialised kernel turns |a - b| * 1000 into an LDS byte address that wraps at
// |a - b| = 2^30 / 1000.  Every finite DP value is a sum of at most 2 parameters per position, and the posterior terms
// add two such values, so 4 * max|parameter| * (L + 2) bounds every finite difference; 6 * keeps a margin.
static bool spec_lsum_range_ok(const td_ctx* c, int lmax)
{
	const double limit = c->lsum_limit;   // (1e6; tests lower it to force the switch to the clamped form)
	return 6.0 * (double)c->m_maxabs * ((double)lmax + 2.0) < limit;
}

// The model as the generic kernel reads it from HBM: header, columns, per-HMM info, label predecessor lists.
struct DevModel {
	TdModelHeader h{};
	TdModelHeader* d_hdr = nullptr;
	TdCol* d_cols = nullptr;
	uint32_t* d_hinfo = nullptr;
	int32_t* d_pred_off = nullptr;
	int32_t* d_pred_idx = nullptr;
};

static void free_dev_model(DevModel& d)
{
	void* p[] = { d.d_hdr, d.d_cols, d.d_hinfo, d.d_pred_off, d.d_pred_idx };
	for (void* q : p) if (q) (void)hipFree(q);
	d = DevModel();
}

// validate a description and put its tables on the device (synchronous copies)
static int build_dev_model(td_ctx* c, const td_model_desc* m, DevModel& out)
{
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

	DevModel d;
	d.h = h;
	bool ok = hipMalloc((void**)&d.d_hdr, sizeof h) == hipSuccess && hipMalloc((void**)&d.d_cols, sizeof(TdCol) * cols.size()) == hipSuccess &&
	          hipMalloc((void**)&d.d_hinfo, sizeof(uint32_t) * hinfo.size()) == hipSuccess &&
	          hipMalloc((void**)&d.d_pred_off, sizeof(int32_t) * poff.size()) == hipSuccess &&
	          hipMalloc((void**)&d.d_pred_idx, sizeof(int32_t) * pidx.size()) == hipSuccess &&
	          hipMemcpy(d.d_hdr, &h, sizeof h, hipMemcpyHostToDevice) == hipSuccess &&
	          hipMemcpy(d.d_cols, cols.data(), sizeof(TdCol) * cols.size(), hipMemcpyHostToDevice) == hipSuccess &&
	          hipMemcpy(d.d_hinfo, hinfo.data(), sizeof(uint32_t) * hinfo.size(), hipMemcpyHostToDevice) == hipSuccess &&
	          hipMemcpy(d.d_pred_off, poff.data(), sizeof(int32_t) * poff.size(), hipMemcpyHostToDevice) == hipSuccess &&
	          hipMemcpy(d.d_pred_idx, pidx.data(), sizeof(int32_t) * pidx.size(), hipMemcpyHostToDevice) == hipSuccess;
	if (!ok) { free_dev_model(d); return fail(c, "td_model_upload: device tables: %s", hipGetErrorString(hipGetLastError())); }
	out = d;
	return TD_OK;
}

extern "C" int td_model_upload(td_ctx* c, const td_model_desc* m)
{
	if (!c || !m) return fail(c, "td_model_upload: NULL argument");
	HIPCHK(c, hipSetDevice(c->device));
	// a rejected upload (tickets in flight, an invalid description) leaves the context as it was
	if (tickets_outstanding(c)) return fail(c, "td_model_upload: td_submit tickets are outstanding (td_wait them first)");
	DevModel dm;
	if (build_dev_model(c, m, dm) != TD_OK) return TD_FAIL;
	const TdModelHeader h = dm.h;
	// from here on the context holds no usable model / batch until every step below has succeeded
	c->have_model = false;
	for (int k = 0; k < TD_MAX_PIPELINE; k++) { c->slots[k].staged = false; c->slots[k].ran = false; }   // batches are staged per model
	HIPCHK(c, sync_compute(c));
	void* old[] = { c->d_hdr, c->d_cols, c->d_hinfo, c->d_pred_off, c->d_pred_idx };
	for (void* p : old) if (p) (void)hipFree(p);
	c->d_hdr = dm.d_hdr; c->d_cols = dm.d_cols; c->d_hinfo = dm.d_hinfo; c->d_pred_off = dm.d_pred_off; c->d_pred_idx = dm.d_pred_idx;
	c->hdr = h;
	c->label.assign(m->label, m->label + m->H);
	c->m_n_hmm.assign(m->n_hmm, m->n_hmm + m->S);
	c->m_n_col.assign(m->n_col, m->n_col + m->S);
	c->m_trans.assign(m->trans, m->trans + (size_t)m->C * 9);

	// model-specialised kernel: compile now (seconds); a failure is an error, never a silent fallback
	if (c->spec_mod) { HIPCHK(c, hipModuleUnload(c->spec_mod)); c->spec_mod = nullptr; }
	c->spec_fn = nullptr; c->spec_ready = false;
	c->prune_lcap = 0; c->prune_live = false;   // the pruning tables belong to the model
	c->half_slots = false;                      // ... and so does the workspace geometry
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
		if (load_spec_kernel(c, c->spec_oob_unsafe ? 0 : td_spec_lsum_oob()) != TD_OK) return TD_FAIL;
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
	c->have_model = true;
	return TD_OK;
}

extern "C" int td_set_option(td_ctx* c, const char* name, int32_t value)
{
	if (!c || !name) return TD_FAIL;
	if (!strcmp(name, "specialize")) {
		c->specialize = value != 0; // takes effect at the next td_model_upload
		return TD_OK;
	}
	if (!strcmp(name, "poison_workspace")) { c->poison = value != 0; return TD_OK; }
	if (!strcmp(name, "compact_egress")) { c->compact_egress = value != 0; return TD_OK; }          // takes effect with the next download / td_submit
	if (!strcmp(name, "stable_input")) {
		if (tickets_outstanding(c)) return fail(c, "td_set_option: stable_input cannot change while tickets are outstanding");
		c->stable_input = value != 0;
		return TD_OK;
	}
	if (!strcmp(name, "length_classes_enabled")) { c->length_classes = value != 0; return TD_OK; }  // ... the next upload
	if (!strcmp(name, "debug_wait")) { c->debug_wait = value != 0; return TD_OK; }
	if (!strcmp(name, "spec_lsum_limit")) { c->lsum_limit = value > 0 ? (double)value : 1.0e6; return TD_OK; }   // tests; next upload
	if (!strcmp(name, "rle_cap")) {
		if (value < 0 || value > 127) return fail(c, "td_set_option: rle_cap must be 0 (default) .. 127");
		c->rle_cap_forced = value;
		return TD_OK;
	}
	if (!strcmp(name, "host_threads")) {
		if (value < 1 || value > 16) return fail(c, "td_set_option: host_threads must be 1..16");
		c->host_threads = value;
		return TD_OK;
	}
	if (!strcmp(name, "overlap_decode")) {
		if (tickets_outstanding(c)) return fail(c, "td_set_option: overlap_decode cannot change while tickets are outstanding");
		c->overlap = value != 0; c->submit_parity = 0;
		return TD_OK;
	}
	if (!strcmp(name, "pipeline_depth")) {
		if (value < 1 || value > TD_MAX_PIPELINE) return fail(c, "td_set_option: pipeline_depth must be 1..%d", TD_MAX_PIPELINE);
		if (tickets_outstanding(c)) return fail(c, "td_set_option: pipeline_depth cannot change while tickets are outstanding");
		c->pipeline_depth = value; c->next_slot = 0;
		return TD_OK;
	}
	return fail(c, "td_set_option: unknown option %s", name);
}

extern "C" int td_get_option(td_ctx* c, const char* name, int32_t* value)
{
	if (!c || !name || !value) return TD_FAIL;
	if (!strcmp(name, "specialize")) { *value = c->specialize; return TD_OK; }
	if (!strcmp(name, "spec_lsum_clamped")) { *value = c->spec_ready && !c->spec_oob; return TD_OK; }
	if (!strcmp(name, "pipeline_depth")) { *value = c->pipeline_depth; return TD_OK; }
	if (!strcmp(name, "host_threads")) { *value = c->host_threads; return TD_OK; }
	if (!strcmp(name, "artifacts_active")) { *value = c->art_n > 0; return TD_OK; }
	if (!strcmp(name, "length_classes")) { *value = c->slots[c->last_slot].n_big; return TD_OK; }   // wave slots of the long geometry in the last batch
	if (!strcmp(name, "overlap_decode")) { *value = c->overlap; return TD_OK; }
	if (!strcmp(name, "compact_egress")) { *value = c->compact_egress; return TD_OK; }
	if (!strcmp(name, "stable_input")) { *value = c->stable_input; return TD_OK; }
	if (!strcmp(name, "length_classes_enabled")) { *value = c->length_classes; return TD_OK; }
	// hardware queues of the HIP runtime as far as the library can tell (see td_want_hw_queues): the user's GPU_MAX_HW_QUEUES, else
	// the 8 the library asked for when it was loaded before the runtime initialised, else the runtime's own default of 4
	if (!strcmp(name, "hw_queues")) { *value = g_hwq_user ? g_hwq_user : (g_hwq_late ? 4 : 8); return TD_OK; }
	if (!strcmp(name, "hw_queues_late")) { *value = g_hwq_late ? 1 : 0; return TD_OK; }
	// which fast paths the model / the last batch actually got (read-only)
	if (!strcmp(name, "prune_active")) {
		// the loaded specialised kernel prunes by position AND the bound tables of the last batch's geometry are live
		*value = c->spec_ready && c->prune_live && (td_spec_prune_segs(&c->m_desc) > 0 || td_spec_prune_sfx(&c->m_desc) < c->m_desc.S);
		return TD_OK;
	}
	if (!strcmp(name, "overlap_active")) {
		// pipelined batches alternate between two compute streams / workspaces (off: option, generic kernel, depth 1, or HBM
		// could not hold the second workspace)
		*value = c->overlap && c->pipeline_depth > 1 && c->spec_ready && c->stream2 != nullptr && c->d_ws2 != nullptr;
		return TD_OK;
	}
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

extern "C" int td_set_window(td_ctx* c, int32_t matchstart, int32_t matchend)
{
	if (!c) return TD_FAIL;
	if (matchstart == -1 && matchend == -1) { c->match_start = 0; c->match_len = 0; return TD_OK; }
	if (matchstart < 0 || matchend <= matchstart) return fail(c, "td_set_window: need 0 <= matchstart < matchend (or -1, -1 for none)");
	c->match_start = matchstart; c->match_len = matchend - matchstart;
	return TD_OK;
}

extern "C" int td_set_batch_window(td_ctx* c, int64_t first_read, int64_t total_reads)
{
	if (!c) return TD_FAIL;
	if (first_read < 0 || total_reads < 0) return fail(c, "td_set_batch_window: negative argument");
	c->win_first = first_read; c->win_total = total_reads;
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
// A batch lives in a slot.  The host hands over the reads as they are (base codes or FASTQ sequence text + offsets) and
// gets per-read records, rewritten sequences and labels back in the same order; everything in between -- base coding,
// the stable sort by length, 2-bit packing and lane interleave, and on the way back un-permuting and de-interleaving --
// runs on the device (td_stage.hip), so a transfer is one copy of contiguous bytes each way.  The synchronous calls
// (td_batch_upload / td_run / td_batch_download) work on slot 0; td_submit / td_wait rotate over `pipeline_depth` slots
// with the copies on their own streams, so that the upload of batch k+1 and the download of batch k-1 overlap the decode
// kernel of batch k.  One HBM workspace serves all slots (decode kernels are serialised on the compute stream).
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

// output block of the decode kernels: eight SoA arrays over n_tiles*64 reads (f, b, r, bar, q, type, barcode, finger: equal
// strides), then keep words, then labels -- all in device order
struct OutLayout { int64_t soa_stride, keep, labels, total; };
static OutLayout out_layout(int64_t n_tiles, int lmax, int nw1)
{
	OutLayout o;
	o.soa_stride = n_tiles * TD_WAVE * 4;      // a multiple of 256
	int64_t p = 8 * o.soa_stride;
	o.keep = p; p = align256(p + n_tiles * nw1 * TD_WAVE * 4);
	o.labels = p; p = align256(p + n_tiles * (int64_t)(lmax + 1) * TD_WAVE);
	o.total = p;
	return o;
}

// is this host pointer page-locked (hipHostMalloc / hipHostRegister)?  Then the DMA engines can use it directly.
static bool is_pinned(const void* p)
{
	if (!p) return false;
	hipPointerAttribute_t a;
	if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
	return a.type == hipMemoryTypeHost;
}

// memcpy on the context's host threads (pageable caller memory <-> pinned staging)
static void parallel_copy(td_ctx* c, void* dst, const void* src, size_t bytes)
{
	c->pool.copy(dst, src, bytes, c->host_threads);
}

static int slot_events(td_ctx* c, TdSlot& s)
{
	if (s.ev_k0) return TD_OK;
	HIPCHK(c, hipEventCreate(&s.ev_k0));
	HIPCHK(c, hipEventCreate(&s.ev_k1));
	HIPCHK(c, hipEventCreateWithFlags(&s.ev_up, hipEventDisableTiming));
	HIPCHK(c, hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming));
	HIPCHK(c, hipEventCreateWithFlags(&s.ev_down, hipEventDisableTiming));
	HIPCHK(c, hipEventCreateWithFlags(&s.ev_pack, hipEventDisableTiming));
	return TD_OK;
}

static void slot_release(TdSlot& s)
{
	void* dev[] = { s.d_raw, s.d_offs, s.d_read_at, s.d_keys, s.d_vals, s.d_sort_tmp, s.d_packed, s.d_lens, s.d_art_left,
	                s.d_out, s.d_res, s.d_seq, s.d_lab, s.d_keepo, s.d_rle, s.d_runs };
	for (void* p : dev) if (p) (void)hipFree(p);
	void* pinned[] = { s.h_raw, s.h_offs, s.h_res, s.h_seq, s.h_lab, s.h_keepo, s.h_rle };
	for (void* p : pinned) if (p) (void)hipHostFree(p);
	hipEvent_t ev[] = { s.ev_up, s.ev_k0, s.ev_k1, s.ev_done, s.ev_down, s.ev_pack };
	for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
	s = TdSlot();
}

// Size the shared workspace for a batch with this geometry.  Growing it (or switching the kernel to the clamped logsum)
// waits for the decode kernels in flight first.
static int ensure_workspace(td_ctx* c, TdSlot& s)
{
	make_layout(s.lay, c->hdr.S, c->hdr.H, c->hdr.C, s.lmax, c->hdr.max_ncol);
	int64_t slot_bytes = s.lay.slot_bytes;
	if (c->spec_ready && c->spec_oob && !spec_lsum_range_ok(c, s.lmax)) {
		HIPCHK(c, sync_compute(c));
		if (load_spec_kernel(c, 0) != TD_OK) return TD_FAIL;   // reads this long need the clamped logsum (seconds, once)
	}
	if (c->spec_ready) {
		td_model_desc md{};
		md.S = c->hdr.S; md.H = c->hdr.H; md.C = c->hdr.C;
		md.n_hmm = c->m_n_hmm.data(); md.n_col = c->m_n_col.data(); md.trans = c->m_trans.data();
		td_spec_layout(s.slay, &md, s.n_long > 0 ? s.lmax_small : s.lmax);   // the geometry of the many
		td_spec_layout(s.slay_big, &md, s.lmax);
		slot_bytes = s.slay.slot_bytes;
		if (s.lmax > c->prune_lcap || !c->d_prune) {
			// bound tables of the position pruning, for reads up to lcap bases (kernels in flight read the old ones)
			HIPCHK(c, sync_compute(c));
			const int lcap = (s.lmax + 2 + 255) / 256 * 256, stride = lcap + 24;   // (the scans request TDS_SCAN_B = 16 entries at a time: spare entries behind lcap)
			std::vector<float> tab;
			const int ps = td_spec_prune_segs(&c->m_desc), sf = td_spec_prune_sfx(&c->m_desc);
			// (the bound recurrences cost columns x positions on the host: for reads beyond 8192 bases the tables stay zero, which
			// the kernel reads as "nothing can be pruned" -- every position violates the zero bound -- and decodes densely)
			c->prune_live = (ps > 0 || sf < c->m_desc.S) && lcap <= 8192;
			if (c->prune_live) td_spec_prune_tables(&c->m_desc, ps, sf, lcap, stride, tab);
			else tab.assign((size_t)TD_PRUNE_TABLES * stride, 0.0f);
			if (c->d_prune) { HIPCHK(c, hipFree(c->d_prune)); c->d_prune = nullptr; }
			HIPCHK(c, hipMalloc((void**)&c->d_prune, tab.size() * sizeof(float)));
			HIPCHK(c, hipMemcpy(c->d_prune, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
			c->prune_lcap = lcap; c->prune_stride = stride;
		}
	}
	// wave slots: enough to fill the chip (2 workgroups of 4 waves per CU share the LDS), bounded by HBM
	const int wpb = (c->spec_ready ? c->spec_block : td_kernel_block_threads()) / TD_WAVE;
	int64_t want = (int64_t)c->n_cu * (c->spec_ready ? c->spec_waves_per_cu : 2 * wpb);
	if (c->wave_slots_forced > 0) want = c->wave_slots_forced;
	int64_t slots = want;
	if (slots > s.n_tiles) slots = s.n_tiles;
	if (slots < 1) slots = 1;
	slots = (slots + wpb - 1) / wpb * wpb; // whole workgroups
	// length classes: the long tiles need a slot each of the big geometry (they are the first tiles the lowest slots take); when
	// they are too many for that to pay -- more than a quarter of the slots -- the batch keeps one geometry
	s.n_big = 0;
	int64_t big_extra = 0;       // bytes the big slots take beyond a small slot each
	if (c->spec_ready && s.n_long > 0) {
		if ((int64_t)s.n_long * 4 <= slots) {
			s.n_big = s.n_long;
			big_extra = (int64_t)s.n_big * (s.slay_big.slot_bytes - s.slay.slot_bytes);
		} else {
			s.n_long = 0; s.lmax_small = s.lmax;
			s.slay = s.slay_big;
			slot_bytes = s.slay.slot_bytes;
		}
	}
	uint8_t*& ws = s.wsi ? c->d_ws2 : c->d_ws;
	size_t& cap_ws = s.wsi ? c->cap_ws2 : c->cap_ws;
	// Pipelined batches overlap their launches on two streams with a workspace each.  When HBM cannot hold two workspaces of the
	// full wave-slot count (config 5: 42 MiB per slot, 172 GB for 4096 slots) each launch gets half the slots instead -- two
	// launches side by side still fill every SIMD, and the machine stays busy while one launch's slowest waves finish.
	if (s.pipelined && c->overlap && c->pipeline_depth > 1 && c->spec_ready && !c->half_slots) {
		size_t free_b = 0, total_b = 0;
		HIPCHK(c, hipMemGetInfo(&free_b, &total_b));
		const double avail = (double)free_b + (double)c->cap_ws + (double)c->cap_ws2;
		if (2.0 * (double)(slots * slot_bytes + big_extra) > 0.8 * avail && (double)(slots * slot_bytes + big_extra) <= 0.8 * avail) c->half_slots = true;
	}
	if (c->half_slots && s.pipelined) {
		int64_t half = (slots / 2 + wpb - 1) / wpb * wpb;
		if (half < wpb) half = wpb;
		if (half >= s.n_big * 4 || s.n_big == 0) slots = half;
	}
	if ((size_t)(slots * slot_bytes + big_extra) > cap_ws) {
		size_t free_b = 0, total_b = 0;
		HIPCHK(c, hipMemGetInfo(&free_b, &total_b));
		if (s.wsi == 1 && (double)(slots * slot_bytes + big_extra) > (c->half_slots ? 0.6 : 0.4) * (double)(free_b + cap_ws)) {
			// HBM cannot hold a second workspace of this size beside the first: this and all later batches run on the first stream
			c->overlap = 0;
			s.wsi = 0; s.cs = c->stream;
			return ensure_workspace(c, s);
		}
		const int64_t budget = (int64_t)((double)(free_b + cap_ws) * 0.85);
		if (slots * slot_bytes + big_extra > budget) {
			if (big_extra > budget / 2) {   // not even the long tiles' slots fit beside a useful number of others: one geometry
				s.n_long = 0; s.lmax_small = s.lmax; s.n_big = 0; big_extra = 0;
				s.slay = s.slay_big;
				slot_bytes = s.slay.slot_bytes;
			}
			slots = (budget - big_extra) / slot_bytes / wpb * wpb;
		}
		if (slots < wpb || slots < s.n_big) return fail(c, "td_batch_upload: workspace of %lld bytes per wave does not fit in HBM", (long long)slot_bytes);
		if ((size_t)(slots * slot_bytes + big_extra) > cap_ws) {
			HIPCHK(c, sync_compute(c));
			const size_t need = (size_t)(slots * slot_bytes + big_extra);
			// Where the driver places a large allocation decides how fast the kernel's spill stream runs over it (up to
			// 9 % on one box).  A large workspace is therefore chosen among a few candidates that
			// exist side by side: a memory-side probe runs over each, the fastest stays, the others are freed.
			int n_cand = c->ws_candidates;
			if (n_cand > 4) n_cand = 4;
			if (need < ((size_t)2 << 30) || (double)need * n_cand > 0.6 * (double)(free_b + cap_ws)) n_cand = 1;
			if (n_cand <= 1) {
				if (ensure(c, &ws, &cap_ws, need) != TD_OK) return TD_FAIL;
			} else {
				if (ws) { HIPCHK(c, hipFree(ws)); ws = nullptr; cap_ws = 0; }
				uint8_t* cand[4] = { nullptr, nullptr, nullptr, nullptr };
				float ms[4] = { 0, 0, 0, 0 };
				int n_ok = 0, best = -1;
				for (int k = 0; k < n_cand; k++) {
					if (hipMalloc((void**)&cand[k], need) != hipSuccess) { (void)hipGetLastError(); cand[k] = nullptr; break; }
					n_ok++;
				}
				for (int k = 0; k < n_ok; k++) {
					if (td_ws_probe(cand[k], slot_bytes, (int)slots, c->stream, &ms[k]) != hipSuccess) ms[k] = 1e30f;
					if (best < 0 || ms[k] < ms[best]) best = k;
				}
				if (best < 0) return fail(c, "td_batch_upload: workspace of %zu bytes could not be allocated", need);
				if (c->debug_alloc) fprintf(stderr, "tagdust_hip: workspace candidates: probe %.2f %.2f %.2f %.2f ms -> #%d\n", ms[0], ms[1], ms[2], ms[3], best);
				for (int k = 0; k < n_ok; k++) if (k != best) (void)hipFree(cand[k]);
				ws = cand[best]; cap_ws = need;
			}
		}
	}
	s.n_wave_slots = (int32_t)slots;
	s.ws_slot_bytes = slot_bytes;
	s.ws_bytes = slots * slot_bytes + big_extra;
	return TD_OK;
}

// Reads -> device, sorted and packed.  Copies go on `up` (the compute stream itself for the synchronous calls); the
// kernels on the compute stream wait for them through ev_up.
static int slot_stage(td_ctx* c, TdSlot& s, const TdRoute& route, const void* bases, int is_ascii, const int64_t* offs, int64_t n, hipStream_t up, bool with_workspace = true)
{
	s.reset_staged(route);   // nothing of the slot's previous batch survives (nor of its launch, nor of its download)
	if (with_workspace && !c->have_model) return fail(c, "td_batch_upload: no model uploaded");
	if (!offs || n < 0 || (!bases && n > 0 && offs[n] > offs[0])) return fail(c, "td_batch_upload: bad arguments");
	if (n > 0x7fffffffLL - TD_WAVE) return fail(c, "td_batch_upload: %lld reads in one batch", (long long)n);
	HIPCHK(c, hipSetDevice(c->device));
	if (slot_events(c, s) != TD_OK) return TD_FAIL;
	const int64_t base = offs[0];   // a sub-range of a larger batch keeps the caller's offsets (td_multi_decode)
	// offsets: one pass that copies them into pinned memory and finds the longest / shortest read
	if (ensure_pinned(c, &s.h_offs, &s.cap_h_offs, (size_t)(n + 1) * 8) != TD_OK) return TD_FAIL;
	int lmax = 1, lmin = 0x7fffffff;
	int64_t bad = -1;
	s.h_offs[0] = 0;
	for (int64_t i = 0; i < n; i++) {
		const int64_t l = offs[i + 1] - offs[i];
		s.h_offs[i + 1] = offs[i + 1] - base;
		if (l < 0 || l > 100000) { bad = i; break; }
		if (l > lmax) lmax = (int)l;
		if (l < lmin) lmin = (int)l;
	}
	if (bad >= 0) return fail(c, "td_batch_upload: read %lld has length %lld", (long long)bad, (long long)(offs[bad + 1] - offs[bad]));
	const int64_t n_bases = n > 0 ? offs[n] - base : 0;
	const int64_t n_tiles = (n + TD_WAVE - 1) / TD_WAVE;
	// Length classes.  The reads are sorted by length on the device, so tile t holds the reads ranked 64 t .. 64 t + 63: from the
	// histogram of the lengths, the longest read of every tile.  When the batch's longest read is at least half as long again as
	// the reads at the 99 % mark, the tiles beyond that mark (n_long of them) get wave slots of their own geometry and the rest
	// keeps the geometry of the many -- one 1000-base read among 150-base reads no longer costs every slot 6.6 times the memory.
	s.n_long = 0; s.lmax_small = lmax;
	if (c->spec_ready && with_workspace && lmin != lmax && n_tiles >= 64 && c->length_classes) {
		std::vector<int64_t> hist((size_t)lmax + 2, 0);
		for (int64_t i = 0; i < n; i++) hist[(size_t)(offs[i + 1] - offs[i])]++;
		// rank of the last read of the tile at the 99 % mark, its length, and the tiles that hold anything longer
		const int64_t t99 = n_tiles - 1 - (n_tiles + 99) / 100;
		const int64_t rank = t99 * TD_WAVE + TD_WAVE - 1;
		int64_t acc = 0;
		int l99 = lmax;
		for (int l = 0; l <= lmax; l++) { acc += hist[(size_t)l]; if (acc > rank) { l99 = l; break; } }
		if (l99 < 1) l99 = 1;
		if ((int64_t)lmax * 2 >= (int64_t)l99 * 3) {
			int64_t upto = 0;   // reads of length <= l99
			for (int l = 0; l <= l99; l++) upto += hist[(size_t)l];
			const int64_t first_long_tile = upto / TD_WAVE;                 // (a tile that mixes both kinds counts as long)
			s.n_long = (int32_t)(n_tiles - first_long_tile);
			s.lmax_small = l99;
		}
	}
	const int nw2 = (lmax + 15) / 16, nw1 = (lmax + 31) / 32;
	const bool sorted = n > 0 && lmin != lmax;

	// device buffers
	const size_t n_packed = (size_t)(n_tiles * (int64_t)(nw2 + nw1) * TD_WAVE), n_lanes = (size_t)(n_tiles * TD_WAVE);
	const OutLayout ol = out_layout(n_tiles, lmax, nw1);
	if (ensure(c, &s.d_raw, &s.cap_raw, (size_t)n_bases) != TD_OK || ensure(c, &s.d_offs, &s.cap_offs, (size_t)(n + 1) * 8) != TD_OK ||
	    ensure(c, &s.d_packed, &s.cap_packed, n_packed * 4) != TD_OK || ensure(c, &s.d_lens, &s.cap_lens, n_lanes * 4) != TD_OK ||
	    ensure(c, &s.d_art_left, &s.cap_art_left, n_lanes) != TD_OK || ensure(c, &s.d_out, &s.cap_out, (size_t)ol.total) != TD_OK)
		return TD_FAIL;
	size_t sort_tmp = 0;
	if (sorted) {
		sort_tmp = td_stage_sort_temp_bytes(n, lmax);
		if (ensure(c, &s.d_read_at, &s.cap_read_at, (size_t)n * 4) != TD_OK || ensure(c, &s.d_keys, &s.cap_keys, (size_t)n * 8) != TD_OK ||
		    ensure(c, &s.d_vals, &s.cap_vals, (size_t)n * 4) != TD_OK || ensure(c, &s.d_sort_tmp, &s.cap_sort_tmp, sort_tmp) != TD_OK)
			return TD_FAIL;
	}
	s.lmax = lmax; s.nw2 = nw2; s.nw1 = nw1; s.n_tiles = (int32_t)n_tiles;
	if (with_workspace && ensure_workspace(c, s) != TD_OK) { s.n_tiles = 0; return TD_FAIL; }
	s.n_tiles = 0;

	// host -> device: page-locked caller memory goes straight to the DMA engine, anything else through pinned staging
	const void* src = (const uint8_t*)bases + base;
	s.raw_direct = n_bases > 0 && is_pinned(src);
	if (n_bases > 0 && !s.raw_direct) {
		if (ensure_pinned(c, &s.h_raw, &s.cap_h_raw, (size_t)n_bases) != TD_OK) return TD_FAIL;
		parallel_copy(c, s.h_raw, src, (size_t)n_bases);
		src = s.h_raw;
		s.raw_host = s.h_raw;
	} else if (n_bases > 0 && c->stable_input && s.pipelined) {
		s.raw_host = (const uint8_t*)src;   // the caller's promise: untouched until td_wait -- the compact egress rebuilds from it
	}
	HIPCHK(c, hipMemcpyAsync(s.d_offs, s.h_offs, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, up));
	if (n_bases > 0) HIPCHK(c, hipMemcpyAsync(s.d_raw, src, (size_t)n_bases, hipMemcpyHostToDevice, up));
	if (up != s.aux) {
		HIPCHK(c, hipEventRecord(s.ev_up, up));
		HIPCHK(c, hipStreamWaitEvent(s.aux, s.ev_up, 0));
	}
	if (sorted)
		HIPCHK(c, td_stage_sort(s.d_offs, n, lmax, s.d_read_at, s.d_keys, s.d_keys + n, s.d_vals, s.d_sort_tmp, sort_tmp, s.aux));
	TdStageBatch& b = s.sb;
	b = TdStageBatch{};
	b.raw = s.d_raw; b.offs = s.d_offs; b.n_reads = n; b.is_ascii = is_ascii;
	b.n_tiles = (int32_t)n_tiles; b.lmax = lmax; b.nw2 = nw2; b.nw1 = nw1;
	b.read_at = sorted ? s.d_read_at : nullptr;
	b.packed = s.d_packed; b.lens = s.d_lens; b.art_left = s.d_art_left;
	b.out_soa = s.d_out; b.soa_stride = ol.soa_stride;
	b.keep = (const uint32_t*)(s.d_out + ol.keep); b.labels = (const int8_t*)(s.d_out + ol.labels);
	HIPCHK(c, td_stage_pack(b, s.aux));
	if (s.aux != s.cs) {   // the decode kernel's stream waits for the packed batch
		HIPCHK(c, hipEventRecord(s.ev_pack, s.aux));
		HIPCHK(c, hipStreamWaitEvent(s.cs, s.ev_pack, 0));
	}

	s.n_reads = n; s.n_bases = n_bases; s.n_tiles = (int32_t)n_tiles; s.is_ascii = is_ascii; s.sorted = sorted;
	s.staged = true;
	return TD_OK;
}

// entries of a read's label-run table: a path visits one label per segment (+ the zeros behind a window, + one to spare)
static int rle_capacity(const td_ctx* c)
{
	int cap = c->hdr.S + 2 < c->hdr.H ? c->hdr.S + 2 : c->hdr.H;
	if (c->rle_cap_forced > 0) cap = c->rle_cap_forced;   // tests: force the overflow route
	return cap;
}

// the decode kernel over a staged slot
static int slot_decode(td_ctx* c, TdSlot& s, int mode, bool want_labels = true)
{
	if (!c->have_model) return fail(c, "td_run: no model uploaded");
	if (mode != TD_MODE_GET_LABEL && mode != TD_MODE_GET_PROB && mode != TD_MODE_ARCH_COMP) return fail(c, "td_run: unsupported mode %d", mode);
	if (!s.staged) return fail(c, "td_run: no batch resident (td_batch_upload failed or was not called)");
	HIPCHK(c, hipSetDevice(c->device));
	s.reset_decoded();   // (runs_cap above all: a launch of the generic kernel, or an empty batch, must not hand the label runs of this
	                     // slot's previous launch to the finish kernel)
	s.mode = mode;
	if (s.n_tiles == 0) { s.ran = true; s.last_ms = 0.0f; return TD_OK; }
	if (c->spec_ready && c->match_len > 0 && !c->spec_window) {   // first batch through a window: the kernel variant that applies it
		HIPCHK(c, sync_compute(c));
		if (load_spec_kernel(c, c->spec_oob ? 1 : 0, 1) != TD_OK) return TD_FAIL;
	}
	const OutLayout ol = out_layout(s.n_tiles, s.lmax, s.nw1);
	TdKernelArgs ka{};
	ka.hdr = c->d_hdr; ka.cols = c->d_cols; ka.hinfo = c->d_hinfo;
	ka.pred_off = c->d_pred_off; ka.pred_idx = c->d_pred_idx; ka.logsum = c->d_logsum;
	ka.packed = s.d_packed; ka.lens = s.d_lens;
	ka.n_tiles = s.n_tiles; ka.n_slots = s.n_wave_slots; ka.lmax = s.lmax; ka.nw2 = s.nw2; ka.nw1 = s.nw1;
	ka.mode = mode; ka.threshold = c->threshold; ka.minlen = c->minlen; ka.dust = c->dust; ka.want_labels = 1;
	ka.win_start = c->match_start; ka.win_len = c->match_len;
	float* soa = (float*)s.d_out;
	const int64_t st = ol.soa_stride / 4;
	ka.out_f = soa; ka.out_b = soa + st; ka.out_r = soa + 2 * st; ka.out_bar = soa + 3 * st; ka.out_q = soa + 4 * st;
	ka.out_type = (int32_t*)(soa + 5 * st); ka.out_barcode = (int32_t*)(soa + 6 * st); ka.out_finger = (int32_t*)(soa + 7 * st);
	ka.out_keep = (uint32_t*)(s.d_out + ol.keep); ka.out_labels = (int8_t*)(s.d_out + ol.labels);
	ka.counters = c->d_counters;
	ka.ws = s.wsi ? c->d_ws2 : c->d_ws; ka.lay = s.lay;
	if (c->art_n > 0 && mode == TD_MODE_GET_LABEL) {
		s.sb.art_threads = c->art_threads;
		s.sb.art_first = c->win_first; s.sb.art_total = c->win_total;
		HIPCHK(c, td_stage_art_left(s.sb, s.cs));
		ka.art_text = c->d_art_text; ka.art_index = c->d_art_index; ka.art_left = s.d_art_left;
		ka.art_n = c->art_n; ka.art_fe = c->art_fe;
	}
	// tests: every byte of the workspace the kernel reads must have been written by this launch -- garbage (NaN floats,
	// all-ones masks) in place of whatever an earlier batch or model left there makes a read-before-write show
	if (c->poison) HIPCHK(c, hipMemsetAsync(ka.ws, 0xFF, (size_t)s.ws_bytes, s.cs));
	if (c->spec_ready) {   // the tile counter of the dynamic tile assignment starts at zero (the first tiles go by slot number)
		int32_t*& tn = s.wsi ? c->d_tile_next2 : c->d_tile_next;
		if (!tn) HIPCHK(c, hipMalloc((void**)&tn, 256));
		HIPCHK(c, hipMemsetAsync(tn, 0, sizeof(int32_t), s.cs));
	}
	HIPCHK(c, hipEventRecord(s.ev_k0, s.cs));
	if (c->spec_ready) {
		TdSpecArgs sa{};
		sa.logsum = ka.logsum; sa.packed = ka.packed; sa.lens = ka.lens;
		sa.n_tiles = ka.n_tiles; sa.n_slots = ka.n_slots; sa.lmax = ka.lmax; sa.nw2 = ka.nw2; sa.nw1 = ka.nw1;
		sa.mode = ka.mode; sa.threshold = ka.threshold; sa.minlen = ka.minlen; sa.dust = ka.dust;
		sa.win_start = ka.win_start; sa.win_len = ka.win_len;
		sa.out_f = ka.out_f; sa.out_b = ka.out_b; sa.out_r = ka.out_r; sa.out_bar = ka.out_bar; sa.out_q = ka.out_q;
		sa.out_type = ka.out_type; sa.out_barcode = ka.out_barcode; sa.out_finger = ka.out_finger;
		sa.out_keep = ka.out_keep; sa.out_labels = ka.out_labels; sa.counters = ka.counters;
		sa.art_text = ka.art_text; sa.art_index = ka.art_index; sa.art_left = ka.art_left; sa.art_n = ka.art_n; sa.art_fe = ka.art_fe;
		sa.ws = ka.ws; sa.lay = s.slay;
		sa.lmax = s.n_big > 0 ? s.lmax_small : s.lmax;
		sa.n_big = s.n_big; sa.lmax_big = s.lmax; sa.lay_big = s.slay_big; sa.out_lmax = s.lmax;
		// the label runs for the compact egress, and the flag that says a read had more of them -- only for a batch whose labels
		// somebody will fetch (td_submit knows; a td_run batch may still be asked for them by td_batch_download)
		if (mode == TD_MODE_GET_LABEL && want_labels && c->compact_egress) {
			const int cap = rle_capacity(c);
			const size_t words = (size_t)s.n_tiles * (size_t)cap * TD_WAVE + 1;
			if (ensure(c, &s.d_runs, &s.cap_runs, words * 4) != TD_OK) return TD_FAIL;
			if (c->poison) HIPCHK(c, hipMemsetAsync(s.d_runs, 0xFF, (words - 1) * 4, s.cs));   // (tests: a run the finish kernel reads must be this launch's)
			HIPCHK(c, hipMemsetAsync(s.d_runs + words - 1, 0, 4, s.cs));
			sa.out_runs = s.d_runs; sa.rle_overflow = (int32_t*)(s.d_runs + words - 1); sa.rle_cap = cap;
			s.runs_cap = cap;
		}
		sa.prune = c->d_prune; sa.prune_stride = c->prune_stride;
		sa.tile_next = s.wsi ? c->d_tile_next2 : c->d_tile_next;
		size_t sz = sizeof sa;
		void* cfg[] = { HIP_LAUNCH_PARAM_BUFFER_POINTER, &sa, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END };
		const int wpb = c->spec_block / TD_WAVE;
		const unsigned blocks = (unsigned)((s.n_wave_slots + wpb - 1) / wpb);
		HIPCHK(c, hipModuleLaunchKernel(c->spec_fn, blocks, 1, 1, (unsigned)c->spec_block, 1, 1, 0, s.cs, nullptr, cfg));
	} else {
		HIPCHK(c, td_launch_decode(&ka, s.cs));
	}
	HIPCHK(c, hipEventRecord(s.ev_k1, s.cs));
	s.ran = true;
	s.last_ms = -1.0f;
	c->last_slot = (int)(&s - c->slots);
	return TD_OK;
}

// Results into the caller's order on the device (finish kernel on the compute stream), then device -> host.  A page-locked
// destination is written by the DMA engine directly; anything else is reached through pinned staging and copied out by
// slot_fetch_end.  The synchronous calls queue the copies behind the finish kernel on the compute stream.  The pipelined
// calls must not: a copy queued behind an event is carried out by a blit *kernel*, which then competes for CUs with the
// next batch's decode kernel -- a persistent kernel that owns every register file -- and finishes when that does (measured:
// 41 ms instead of 3).  So td_wait waits for the finish kernel on the host and issues the copies then, with nothing
// pending in front of them: they go to the SDMA engines and run beside the decode kernel.
static int slot_issue_copies(td_ctx* c, TdSlot& s, hipStream_t down)
{
	const int64_t n = s.n_reads;
	const size_t res_bytes = (size_t)n * sizeof(td_read_result), seq_bytes = (size_t)s.n_bases, lab_bytes = (size_t)(s.n_bases + n);
	if (s.u_res) HIPCHK(c, hipMemcpyAsync(s.res_direct ? (void*)s.u_res : (void*)s.h_res, s.d_res, res_bytes, hipMemcpyDeviceToHost, down));
	if (s.u_seq && seq_bytes) {
		if (s.use_keep) HIPCHK(c, hipMemcpyAsync(s.h_keepo, s.d_keepo, (size_t)n * (size_t)s.nw1 * 4, hipMemcpyDeviceToHost, down));
		else HIPCHK(c, hipMemcpyAsync(s.seq_direct ? (void*)s.u_seq : (void*)s.h_seq, s.d_seq, seq_bytes, hipMemcpyDeviceToHost, down));
	}
	if (s.u_labels) {
		if (s.use_rle) {
			// (the overflow flag travels behind the runs in the same copy -- the finish kernel folded the decode kernel's own flag
			// into it: a copy of a few bytes is carried out by a blit kernel, which gets no CU while a decode launch holds every
			// register file and made this wait last as long as that launch: tools/ubench/copy_kinds.cpp)
			HIPCHK(c, hipMemcpyAsync(s.h_rle, s.d_rle, ((size_t)n * (size_t)s.rle_cap + 1) * 4, hipMemcpyDeviceToHost, down));
		}
		else HIPCHK(c, hipMemcpyAsync(s.lab_direct ? (void*)s.u_labels : (void*)s.h_lab, s.d_lab, lab_bytes, hipMemcpyDeviceToHost, down));
	}
	HIPCHK(c, hipEventRecord(s.ev_down, down));
	return TD_OK;
}

static int slot_fetch_begin(td_ctx* c, TdSlot& s, td_read_result* res, int8_t* labels, uint8_t* seq_out, bool deferred)
{
	if (!s.ran) return fail(c, "td_batch_download: td_run has not been called on this batch");
	s.reset_fetch();
	s.u_res = res; s.u_labels = labels; s.u_seq = seq_out;
	s.copies_deferred = deferred;
	const int64_t n = s.n_reads;
	if (n == 0) return TD_OK;
	const size_t res_bytes = (size_t)n * sizeof(td_read_result), seq_bytes = (size_t)s.n_bases, lab_bytes = (size_t)(s.n_bases + n);
	// Compact egress.  The rewritten sequence is the input with some positions turned into the spacer byte: the keep bits (one
	// per base) come back instead and the host rebuilds it from its staging copy of the input -- unless the caller's page-locked
	// buffer was the DMA source, which the caller may have refilled since.  ri->labels is a handful of runs per read (the path
	// moves through the segments in order): (length, label) pairs come back and the host expands them.  A fifth of the bytes
	// over PCIe, and that much less work for the download's blit kernels, which compete with the decode kernel for CUs.
	const bool compact = c->compact_egress != 0;
	s.use_keep = compact && seq_out && seq_bytes && s.raw_host != nullptr;
	s.use_rle = compact && labels;
	s.rle_cap = s.runs_cap > 0 ? s.runs_cap : rle_capacity(c);
	if (res && ensure(c, &s.d_res, &s.cap_res, res_bytes) != TD_OK) return TD_FAIL;
	if (seq_out && !s.use_keep && ensure(c, &s.d_seq, &s.cap_seq, seq_bytes) != TD_OK) return TD_FAIL;
	if (labels && !s.use_rle && ensure(c, &s.d_lab, &s.cap_lab, lab_bytes) != TD_OK) return TD_FAIL;
	if (res && !(s.res_direct = is_pinned(res)) && ensure_pinned(c, &s.h_res, &s.cap_h_res, res_bytes) != TD_OK) return TD_FAIL;
	if (seq_out && !s.use_keep && !(s.seq_direct = is_pinned(seq_out)) && ensure_pinned(c, &s.h_seq, &s.cap_h_seq, seq_bytes) != TD_OK) return TD_FAIL;
	if (labels && !s.use_rle && !(s.lab_direct = is_pinned(labels)) && ensure_pinned(c, &s.h_lab, &s.cap_h_lab, lab_bytes) != TD_OK) return TD_FAIL;
	const size_t keepo_bytes = (size_t)n * (size_t)s.nw1 * 4, rle_bytes = ((size_t)n * (size_t)s.rle_cap + 2) * 4;
	if (s.use_keep && (ensure(c, &s.d_keepo, &s.cap_keepo, keepo_bytes) != TD_OK || ensure_pinned(c, &s.h_keepo, &s.cap_h_keepo, keepo_bytes) != TD_OK)) return TD_FAIL;
	if (s.use_rle) {
		if (ensure(c, &s.d_rle, &s.cap_rle, rle_bytes) != TD_OK || ensure_pinned(c, &s.h_rle, &s.cap_h_rle, rle_bytes) != TD_OK) return TD_FAIL;
		if (c->poison) HIPCHK(c, hipMemsetAsync(s.d_rle, 0xFF, (size_t)n * (size_t)s.rle_cap * 4, s.fin));   // (tests: every entry the host expands must be this batch's)
		HIPCHK(c, hipMemsetAsync(s.d_rle + (size_t)n * (size_t)s.rle_cap, 0, 4, s.fin));   // the overflow flag
	}
	if (s.use_keep && c->poison) HIPCHK(c, hipMemsetAsync(s.d_keepo, 0xFF, keepo_bytes, s.fin));
	s.sb.res = res ? s.d_res : nullptr;
	s.sb.seq_out = (seq_out && !s.use_keep) ? s.d_seq : nullptr;
	s.sb.labels_out = (labels && !s.use_rle) ? s.d_lab : nullptr;
	s.sb.keep_out = s.use_keep ? s.d_keepo : nullptr;
	s.sb.rle_out = s.use_rle ? s.d_rle : nullptr;
	s.sb.rle_cap = s.rle_cap;
	s.sb.runs = (s.use_rle && s.runs_cap > 0) ? s.d_runs : nullptr;
	s.sb.rle_overflow = s.use_rle ? (int32_t*)(s.d_rle + (size_t)n * (size_t)s.rle_cap) : nullptr;
	s.sb.runs_overflow = (s.use_rle && s.runs_cap > 0) ? (const int32_t*)(s.d_runs + (size_t)s.n_tiles * (size_t)s.runs_cap * TD_WAVE) : nullptr;
	if (s.fin != s.cs) HIPCHK(c, hipStreamWaitEvent(s.fin, s.ev_k1, 0));
	HIPCHK(c, td_stage_finish(s.sb, s.fin));
	if (deferred) HIPCHK(c, hipEventRecord(s.ev_done, s.fin));
	else if (slot_issue_copies(c, s, s.fin) != TD_OK) return TD_FAIL;
	s.finished = true;
	return TD_OK;
}

static double wall_ms()
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
}

static int slot_fetch_end(td_ctx* c, TdSlot& s)
{
	if (s.n_reads == 0 || !s.finished) return TD_OK;
	const bool dbg = c->debug_wait != 0;
	const double t0 = dbg ? wall_ms() : 0.0;
	double t1 = t0;
	if (s.copies_deferred) {
		HIPCHK(c, hipEventSynchronize(s.ev_done));
		t1 = dbg ? wall_ms() : 0.0;
		if (slot_issue_copies(c, s, c->s_down) != TD_OK) return TD_FAIL;
	}
	HIPCHK(c, hipEventSynchronize(s.ev_down));
	const double t2 = dbg ? wall_ms() : 0.0;
	struct Rep { bool on; double t0, t1, t2; ~Rep() { if (on) fprintf(stderr, "td_wait: device %.2f ms, download %.2f ms, host %.2f ms\n", t1 - t0, t2 - t1, wall_ms() - t2); } } rep_{ dbg, t0, t1, t2 };
	const int64_t n = s.n_reads;
	if (s.use_rle && s.u_labels && s.h_rle[(size_t)n * (size_t)s.rle_cap] != 0) {
		// a read with more label runs than the table holds (a model whose labels are not one per segment): the labels as they are
		const size_t lab_bytes = (size_t)(s.n_bases + n);
		if (ensure(c, &s.d_lab, &s.cap_lab, lab_bytes) != TD_OK) return TD_FAIL;
		if (!(s.lab_direct = is_pinned(s.u_labels)) && ensure_pinned(c, &s.h_lab, &s.cap_h_lab, lab_bytes) != TD_OK) return TD_FAIL;
		TdStageBatch b2 = s.sb;
		b2.res = nullptr; b2.seq_out = nullptr; b2.keep_out = nullptr; b2.rle_out = nullptr; b2.labels_out = s.d_lab;
		HIPCHK(c, td_stage_finish(b2, s.fin));
		HIPCHK(c, hipStreamSynchronize(s.fin));
		HIPCHK(c, hipMemcpy(s.lab_direct ? (void*)s.u_labels : (void*)s.h_lab, s.d_lab, lab_bytes, hipMemcpyDeviceToHost));
		s.use_rle = false;
	}
	if (s.u_res && !s.res_direct) parallel_copy(c, s.u_res, s.h_res, (size_t)n * sizeof(td_read_result));
	if (s.u_seq && s.n_bases) {
		if (s.use_keep) {
			td_host_rebuild_sequences(c->pool, c->host_threads, n, s.raw_host, s.h_offs, s.h_keepo, s.nw1, s.is_ascii, s.u_seq);
		} else if (!s.seq_direct) parallel_copy(c, s.u_seq, s.h_seq, (size_t)s.n_bases);
	}
	if (s.u_labels) {
		if (s.use_rle) {
			td_host_expand_labels(c->pool, c->host_threads, n, s.h_offs, s.h_rle, s.rle_cap, s.u_labels);
		} else if (!s.lab_direct) parallel_copy(c, s.u_labels, s.h_lab, (size_t)(s.n_bases + n));
	}
	return TD_OK;
}

static bool tickets_outstanding(const td_ctx* c)
{
	for (int k = 0; k < TD_MAX_PIPELINE; k++) if (c->slots[k].ticket) return true;
	return false;
}

static int upload_common(td_ctx* c, const void* bases, int is_ascii, const int64_t* offs, int64_t n)
{
	if (!c) return TD_FAIL;
	if (tickets_outstanding(c)) return fail(c, "td_batch_upload: td_submit tickets are outstanding (td_wait them first)");
	TdSlot& s = c->slots[0];
	TdRoute rt;
	rt.cs = c->stream; rt.wsi = 0; rt.aux = c->stream; rt.fin = c->stream; rt.pipelined = false;
	if (slot_stage(c, s, rt, bases, is_ascii, offs, n, c->stream) != TD_OK) return TD_FAIL;
	HIPCHK(c, hipStreamSynchronize(c->stream));   // the caller may reuse its buffers
	c->last_slot = 0;
	return TD_OK;
}

extern "C" int td_batch_upload(td_ctx* c, const uint8_t* codes, const int64_t* offs, int64_t n)
{
	return upload_common(c, codes, 0, offs, n);
}

extern "C" int td_batch_upload_ascii(td_ctx* c, const char* bases, const int64_t* offs, int64_t n)
{
	return upload_common(c, bases, 1, offs, n);
}

extern "C" int td_run(td_ctx* c, int mode)
{
	if (!c) return TD_FAIL;
	if (!c->have_model) return fail(c, "td_run: no model uploaded");
	return slot_decode(c, c->slots[0], mode);
}

extern "C" int td_sync(td_ctx* c)
{
	if (!c) return TD_FAIL;
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, sync_compute(c));
	return TD_OK;
}

extern "C" int td_batch_download(td_ctx* c, td_read_result* res, int8_t* labels, uint8_t* seq_out)
{
	if (!c) return TD_FAIL;
	TdSlot& s = c->slots[0];
	HIPCHK(c, hipSetDevice(c->device));
	if (slot_fetch_begin(c, s, res, labels, seq_out, false) != TD_OK) return TD_FAIL;
	return slot_fetch_end(c, s);
}

// ---- pipelined batches ----
extern "C" int td_submit(td_ctx* c, const void* bases, int32_t is_ascii, const int64_t* offs, int64_t n_reads, int mode,
                         td_read_result* res, int8_t* labels, uint8_t* seq_out, int64_t* ticket)
{
	if (!c || !ticket) return fail(c, "td_submit: NULL argument");
	*ticket = 0;
	HIPCHK(c, hipSetDevice(c->device));
	if (!c->s_up) {
		int lo = 0, hi = 0;   // (numerically lower = higher priority)
		HIPCHK(c, hipDeviceGetStreamPriorityRange(&lo, &hi));
		HIPCHK(c, hipStreamCreateWithFlags(&c->s_up, hipStreamNonBlocking));
		HIPCHK(c, hipStreamCreateWithPriority(&c->s_down, hipStreamNonBlocking, hi));
		HIPCHK(c, hipStreamCreateWithPriority(&c->s_aux, hipStreamNonBlocking, hi));
		HIPCHK(c, hipStreamCreateWithPriority(&c->s_fin, hipStreamNonBlocking, hi));
	}
	int k = -1;
	for (int j = 0; j < c->pipeline_depth; j++) {   // the slot after the one used last, if free
		const int cand = (c->next_slot + j) % c->pipeline_depth;
		if (!c->slots[cand].ticket) { k = cand; break; }
	}
	if (k < 0) return fail(c, "td_submit: all %d pipeline slots hold batches that have not been waited for", c->pipeline_depth);
	TdSlot& s = c->slots[k];
	TdRoute rt;
	rt.cs = c->stream; rt.wsi = 0; rt.aux = c->stream; rt.fin = c->stream; rt.pipelined = true;
	if (c->overlap && c->pipeline_depth > 1 && c->spec_ready) {   // every other batch on the second stream / workspace
		if (!c->stream2) HIPCHK(c, hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
		if (c->submit_parity) { rt.cs = c->stream2; rt.wsi = 1; }
		c->submit_parity ^= 1;
		rt.aux = c->s_aux; rt.fin = c->s_fin;
	}
	if (slot_stage(c, s, rt, bases, is_ascii != 0, offs, n_reads, c->s_up) != TD_OK) return TD_FAIL;
	if (slot_decode(c, s, mode, labels != nullptr) != TD_OK) return TD_FAIL;
	if (slot_fetch_begin(c, s, res, labels, seq_out, true) != TD_OK) return TD_FAIL;
	// "returns once the reads have left the caller's buffers": a page-locked source is read by the DMA engine itself, so wait
	// for that copy (a few ms at PCIe rate, beside the previous batch's kernel; everything of this batch is queued already)
	if (s.raw_direct && !c->stable_input) HIPCHK(c, hipEventSynchronize(s.ev_up));
	s.ticket = ++c->ticket_counter;
	c->next_slot = (k + 1) % c->pipeline_depth;
	*ticket = s.ticket;
	return TD_OK;
}

extern "C" int td_wait(td_ctx* c, int64_t ticket)
{
	if (!c) return TD_FAIL;
	HIPCHK(c, hipSetDevice(c->device));
	for (int k = 0; k < TD_MAX_PIPELINE; k++) {
		TdSlot& s = c->slots[k];
		if (ticket != 0 && s.ticket == ticket) {
			const int rc = slot_fetch_end(c, s);
			s.ticket = 0;
			c->last_slot = k;   // td_last_kernel_ms / td_batch_info now describe this batch
			return rc;
		}
	}
	return fail(c, "td_wait: no batch with ticket %lld is in flight", (long long)ticket);
}

// ---- architecture comparison: every candidate model over one batch, backward() only, one launch ----
extern "C" int td_arch_scores(td_ctx* c, const td_model_desc* const* models, int32_t n_models, const uint8_t* codes,
                              const int64_t* offs, int64_t n_reads, float* b_scores)
{
	if (!c || !models || n_models < 1 || !offs || !b_scores || n_reads < 0) return fail(c, "td_arch_scores: bad arguments");
	if (tickets_outstanding(c)) return fail(c, "td_arch_scores: td_submit tickets are outstanding (td_wait them first)");
	HIPCHK(c, hipSetDevice(c->device));
	TdSlot& s = c->slots[0];
	// the reads are staged (sorted by length, packed) once; no model of the context is involved
	TdRoute rt;
	rt.cs = c->stream; rt.wsi = 0; rt.aux = c->stream; rt.fin = c->stream; rt.pipelined = false;
	if (slot_stage(c, s, rt, codes, 0, offs, n_reads, c->stream, false) != TD_OK) return TD_FAIL;
	s.staged = false;                      // not a batch td_run could use: it has no workspace geometry
	if (s.n_tiles == 0) return TD_OK;
	const int wpb = td_kernel_block_threads() / TD_WAVE;
	const int64_t n_lanes = (int64_t)s.n_tiles * TD_WAVE;
	const int64_t wave_budget = (int64_t)c->n_cu * 2 * wpb;
	std::vector<DevModel> dm((size_t)n_models);
	int rc = TD_OK;
	float* d_b = nullptr;
	TdKernelArgs* d_args = nullptr;
	std::vector<float> h_b;
	std::vector<int32_t> h_read_at;
	auto cleanup = [&]() {
		for (auto& d : dm) free_dev_model(d);
		if (d_b) (void)hipFree(d_b);
		if (d_args) (void)hipFree(d_args);
	};
	if (hipMalloc((void**)&d_b, sizeof(float) * (size_t)(n_lanes * n_models)) != hipSuccess ||
	    hipMalloc((void**)&d_args, sizeof(TdKernelArgs) * (size_t)n_models) != hipSuccess) { cleanup(); return fail(c, "td_arch_scores: out of device memory"); }
	size_t free_b = 0, total_b = 0;
	if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { cleanup(); return fail(c, "td_arch_scores: hipMemGetInfo failed"); }
	const int64_t ws_budget = (int64_t)((double)(free_b + c->cap_ws) * 0.85);
	// candidates go in chunks whose workspaces fit side by side; within a chunk the chip's wave slots are shared out evenly
	std::vector<TdKernelArgs> args((size_t)n_models);
	int k0 = 0;
	while (k0 < n_models && rc == TD_OK) {
		int k1 = k0;
		int64_t ws_total = 0;
		std::vector<int64_t> ws_off;
		int max_slots = 0;
		// first guess: all remaining candidates in one launch
		int chunk = n_models - k0;
		for (;;) {
			int64_t per = wave_budget / chunk / wpb * wpb;
			if (per < wpb) per = wpb;
			int64_t cap = ((int64_t)s.n_tiles + wpb - 1) / wpb * wpb;
			if (per > cap) per = cap;
			ws_total = 0; ws_off.clear(); max_slots = (int)per;
			bool fits = true;
			for (int k = k0; k < k0 + chunk; k++) {
				if (!dm[(size_t)k].d_hdr && build_dev_model(c, models[k], dm[(size_t)k]) != TD_OK) { cleanup(); return TD_FAIL; }
				TdWsLayout lay;
				make_layout(lay, dm[(size_t)k].h.S, dm[(size_t)k].h.H, dm[(size_t)k].h.C, s.lmax, dm[(size_t)k].h.max_ncol);
				ws_off.push_back(ws_total);
				ws_total += align256(per * lay.slot_bytes);
				if (ws_total > ws_budget) { fits = false; break; }
			}
			if (fits) { k1 = k0 + chunk; break; }
			if (chunk == 1) { cleanup(); return fail(c, "td_arch_scores: the workspace of candidate %d does not fit in HBM", k0); }
			chunk = (chunk + 1) / 2;
		}
		HIPCHK(c, sync_compute(c));
		if (ensure(c, &c->d_ws, &c->cap_ws, (size_t)ws_total) != TD_OK) { cleanup(); return TD_FAIL; }
		for (int k = k0; k < k1; k++) {
			const DevModel& d = dm[(size_t)k];
			TdKernelArgs ka{};
			ka.hdr = d.d_hdr; ka.cols = d.d_cols; ka.hinfo = d.d_hinfo; ka.pred_off = d.d_pred_off; ka.pred_idx = d.d_pred_idx;
			ka.logsum = c->d_logsum; ka.packed = s.d_packed; ka.lens = s.d_lens;
			ka.n_tiles = s.n_tiles; ka.n_slots = max_slots; ka.lmax = s.lmax; ka.nw2 = s.nw2; ka.nw1 = s.nw1;
			ka.mode = TD_MODE_ARCH_COMP; ka.threshold = 0.0f; ka.minlen = c->minlen; ka.dust = 0; ka.want_labels = 0;
			ka.out_b = d_b + (int64_t)k * n_lanes;       // backward-only mode writes nothing else
			ka.counters = c->d_counters;
			ka.ws = c->d_ws + ws_off[(size_t)(k - k0)];
			make_layout(ka.lay, d.h.S, d.h.H, d.h.C, s.lmax, d.h.max_ncol);
			args[(size_t)k] = ka;
		}
		if (hipMemcpyAsync(d_args + k0, args.data() + k0, sizeof(TdKernelArgs) * (size_t)(k1 - k0), hipMemcpyHostToDevice, c->stream) != hipSuccess ||
		    td_launch_decode_multi(d_args + k0, k1 - k0, max_slots, c->stream) != hipSuccess ||
		    hipStreamSynchronize(c->stream) != hipSuccess) {   // args.data() must outlive the copy
			rc = fail(c, "td_arch_scores: launch failed: %s", hipGetErrorString(hipGetLastError()));
		}
		k0 = k1;
	}
	if (rc == TD_OK) {
		h_b.resize((size_t)(n_lanes * n_models));
		if (hipMemcpy(h_b.data(), d_b, sizeof(float) * h_b.size(), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(c, "td_arch_scores: download failed");
		if (rc == TD_OK && s.sorted) {
			h_read_at.resize((size_t)n_reads);
			if (hipMemcpy(h_read_at.data(), s.d_read_at, sizeof(int32_t) * (size_t)n_reads, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(c, "td_arch_scores: download failed");
		}
	}
	if (rc == TD_OK) {
		for (int k = 0; k < n_models; k++) {
			const float* src = h_b.data() + (int64_t)k * n_lanes;
			float* dst = b_scores + (int64_t)k * n_reads;
			if (s.sorted) for (int64_t p = 0; p < n_reads; p++) dst[h_read_at[(size_t)p]] = src[p];
			else memcpy(dst, src, sizeof(float) * (size_t)n_reads);
		}
	}
	cleanup();
	return rc;
}

extern "C" void* td_host_alloc(size_t bytes)
{
	void* p = nullptr;
	if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
	return p;
}

extern "C" void td_host_free(void* p)
{
	if (p) (void)hipHostFree(p);
}

extern "C" int td_last_kernel_ms(td_ctx* c, float* ms)
{
	if (!c || !ms) return TD_FAIL;
	TdSlot& s = c->slots[c->last_slot];
	if (!s.ran) return fail(c, "td_last_kernel_ms: nothing has run");
	HIPCHK(c, hipSetDevice(c->device));
	if (s.last_ms < 0.0f && s.n_tiles > 0) {
		HIPCHK(c, hipEventSynchronize(s.ev_k1));
		HIPCHK(c, hipEventElapsedTime(&s.last_ms, s.ev_k0, s.ev_k1));
	}
	*ms = s.last_ms;
	return TD_OK;
}

extern "C" int td_timeline_origin(td_ctx* c)
{
	if (!c) return TD_FAIL;
	HIPCHK(c, hipSetDevice(c->device));
	if (!c->ev_origin) HIPCHK(c, hipEventCreate(&c->ev_origin));
	HIPCHK(c, hipEventRecord(c->ev_origin, c->stream));
	HIPCHK(c, hipEventSynchronize(c->ev_origin));
	return TD_OK;
}

extern "C" int td_last_kernel_times(td_ctx* c, float* start_ms, float* stop_ms, int32_t* stream_index)
{
	if (!c || !start_ms || !stop_ms) return TD_FAIL;
	TdSlot& s = c->slots[c->last_slot];
	if (!s.ran) return fail(c, "td_last_kernel_times: nothing has run");
	if (!c->ev_origin) return fail(c, "td_last_kernel_times: td_timeline_origin has not been called");
	HIPCHK(c, hipSetDevice(c->device));
	*start_ms = *stop_ms = 0.0f;
	if (stream_index) *stream_index = s.wsi;
	if (s.n_tiles == 0) return TD_OK;
	HIPCHK(c, hipEventSynchronize(s.ev_k1));
	HIPCHK(c, hipEventElapsedTime(start_ms, c->ev_origin, s.ev_k0));
	HIPCHK(c, hipEventElapsedTime(stop_ms, c->ev_origin, s.ev_k1));
	return TD_OK;
}

extern "C" int td_batch_info(td_ctx* c, int64_t* n_reads, int64_t* workspace_bytes, int32_t* wave_slots)
{
	if (!c) return TD_FAIL;
	const TdSlot& s = c->slots[c->last_slot];
	if (n_reads) *n_reads = s.n_reads;
	if (workspace_bytes) *workspace_bytes = s.ws_bytes ? s.ws_bytes : (int64_t)s.n_wave_slots * s.ws_slot_bytes;
	if (wave_slots) *wave_slots = s.n_wave_slots;
	return TD_OK;
}

// ---------------------------------------------------------------------------------------------------------
// counters
// ---------------------------------------------------------------------------------------------------------
extern "C" int td_counts_reset(td_ctx* c)
{
	if (!c) return TD_FAIL;
	HIPCHK(c, hipSetDevice(c->device));
	if (c->stream2) HIPCHK(c, hipStreamSynchronize(c->stream2));   // (kernels of pipelined batches may still be counting)
	HIPCHK(c, hipMemsetAsync(c->d_counters, 0, sizeof(unsigned long long) * TD_COUNTER_WORDS, c->stream));
	if (c->stream2) HIPCHK(c, hipStreamSynchronize(c->stream));
	return TD_OK;
}

extern "C" int td_counts_get(td_ctx* c, int64_t* counts)
{
	if (!c || !counts) return TD_FAIL;
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, sync_compute(c));
	HIPCHK(c, hipMemcpy(counts, c->d_counters, sizeof(int64_t) * TD_NUM_COUNTERS, hipMemcpyDeviceToHost));
	return TD_OK;
}

extern "C" int td_diag_get(td_ctx* c, int64_t* diag)
{
	if (!c || !diag) return TD_FAIL;
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, sync_compute(c));
	HIPCHK(c, hipMemcpy(diag, c->d_counters + TD_NUM_COUNTERS, sizeof(int64_t) * TD_NUM_DIAG_COUNTERS, hipMemcpyDeviceToHost));
	return TD_OK;
}

extern "C" void* td_counts_device_ptr(td_ctx* c) { return c ? (void*)c->d_counters : nullptr; }
