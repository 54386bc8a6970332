// td_multi.cpp -- several MI355X driven from one process (include/tagdust_multi.h): the static shard of run_pHMM's
// thread split (src/barcode_hmm.c:1911-1922) over devices, results in input order, counters all-reduced with RCCL.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <sched.h>
#include <stdlib.h>

#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/tagdust_multi.h"

extern "C" void td_shard_bounds(int64_t n, int32_t world, int32_t rank, int64_t* lo, int64_t* hi)
{
	if (world < 1) world = 1;
	if (rank < 0) rank = 0;
	if (rank > world - 1) rank = world - 1;
	const int64_t interval = n / world;              // barcode_hmm.c:1911
	if (lo) *lo = rank * interval;
	if (hi) *hi = (rank == world - 1) ? n : (rank + 1) * interval;   // the last part takes the remainder, :1918-1922
}

extern "C" void td_count_outcomes(const td_read_result* res, const int32_t* lens, int64_t n, int64_t* counts)
{
	if (!res || !counts) return;
	for (int64_t i = 0; i < n; i++) {
		if (lens && lens[i] < 1) continue;
		counts[res[i].read_type & (TD_NUM_OUTCOME_SLOTS - 1)]++;
		if (res[i].read_type == TD_EXTRACT_SUCCESS && res[i].barcode >= 0) counts[TD_NUM_OUTCOME_SLOTS + (res[i].barcode & 0xFF)]++;
	}
}

// RCCL is loaded when a communicator is first needed, so a single-GPU user never maps it
struct RcclApi {
	void* lib = nullptr;
	ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
	const char* (*GetErrorString)(ncclResult_t) = nullptr;
	bool load(std::string& err)
	{
		if (lib) return true;
		lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
		if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
		if (!lib) { err = std::string("cannot load librccl.so: ") + dlerror(); return false; }
		CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
		CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
		AllReduce = (decltype(AllReduce))dlsym(lib, "ncclAllReduce");
		GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
		GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
		GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
		if (!CommInitAll || !CommDestroy || !AllReduce || !GroupStart || !GroupEnd || !GetErrorString) { err = "librccl.so lacks an expected symbol"; return false; }
		return true;
	}
};
static RcclApi g_rccl;
static std::string g_multi_create_error;

// Bind the calling thread to the CPUs of the NUMA node the device's PCIe function hangs off (the ones the process may use
// anyway): the staging copies of that device's batches and the pages of its pinned buffers then stay on the node next to it.
extern "C" int32_t td_bind_host_to_device(int32_t device)
{
	char bus[64] = { 0 };
	if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) { (void)hipGetLastError(); return -1; }
	for (char* p = bus; *p; p++) if (*p >= 'A' && *p <= 'Z') *p = (char)(*p - 'A' + 'a');   // sysfs names are lower case
	char path[160];
	snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
	FILE* f = fopen(path, "r");
	int node = -1;
	if (f) { if (fscanf(f, "%d", &node) != 1) node = -1; fclose(f); }
	if (node < 0) return -1;
	snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
	f = fopen(path, "r");
	if (!f) return -1;
	char list[4096] = { 0 };
	const size_t got = fread(list, 1, sizeof list - 1, f);
	fclose(f);
	list[got] = 0;
	cpu_set_t allowed, want;
	CPU_ZERO(&want);
	if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return -1;
	int n_set = 0;
	for (char* p = list; *p;) {   // "0-15,128-143"
		char* e = nullptr;
		const long a = strtol(p, &e, 10);
		if (e == p) break;
		long b = a;
		p = e;
		if (*p == '-') { b = strtol(p + 1, &e, 10); if (e == p + 1) break; p = e; }
		for (long c = a; c <= b && c < CPU_SETSIZE; c++) if (c >= 0 && CPU_ISSET((int)c, &allowed)) { CPU_SET((int)c, &want); n_set++; }
		while (*p == ',' || *p == ' ' || *p == '\n') p++;
	}
	if (n_set == 0) return -1;   // none of the node's CPUs is ours: leave the thread where it is
	if (sched_setaffinity(0, sizeof want, &want) != 0) return -1;
	return node;
}

// One host thread per device, started with the td_multi and reused by every call (model uploads, decodes): jobs are handed
// over one at a time and waited for.
struct DeviceWorker {
	std::thread th;
	std::mutex mu;
	std::condition_variable cv;
	std::function<int()> job;
	bool has_job = false, done = false, stop = false;
	int rc = TD_OK;
	int32_t numa_node = -1;

	void start(int32_t device, bool bind)
	{
		th = std::thread([this, device, bind] {
			(void)hipSetDevice(device);
			if (bind) numa_node = td_bind_host_to_device(device);
			for (;;) {
				std::function<int()> j;
				{
					std::unique_lock<std::mutex> lk(mu);
					cv.wait(lk, [this] { return stop || has_job; });
					if (!has_job) return;
					j = job; has_job = false;
				}
				const int r = j();
				{ std::lock_guard<std::mutex> lk(mu); rc = r; done = true; }
				cv.notify_all();
			}
		});
	}
	void post(std::function<int()> j)
	{
		{ std::lock_guard<std::mutex> lk(mu); job = std::move(j); has_job = true; done = false; }
		cv.notify_all();
	}
	int wait()
	{
		std::unique_lock<std::mutex> lk(mu);
		cv.wait(lk, [this] { return done; });
		return rc;
	}
	~DeviceWorker()
	{
		if (!th.joinable()) return;
		{ std::lock_guard<std::mutex> lk(mu); stop = true; }
		cv.notify_all();
		th.join();
	}
};

struct td_multi {
	std::vector<int32_t> devices;
	std::vector<td_ctx*> ctx;
	std::vector<DeviceWorker*> worker;       // one per device when there are several
	std::vector<ncclComm_t> comm;            // empty: host sum
	std::vector<hipStream_t> red_stream;     // per device, for the all-reduce
	std::vector<int64_t*> d_sum;             // per device: all-reduced counters
	std::string err;
};

static int mfail(td_multi* m, const char* fmt, ...)
{
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	if (m) m->err = buf; else g_multi_create_error = buf;
	return TD_FAIL;
}

// run fn(k) for every device on its own host thread; returns the first failing device or -1
template <typename F>
static int for_each_device(td_multi* m, F fn)
{
	const int n = (int)m->ctx.size();
	std::vector<int> rc((size_t)n, TD_OK);
	if (m->worker.empty()) { for (int k = 0; k < n; k++) rc[(size_t)k] = fn(k); }
	else {
		for (int k = 0; k < n; k++) m->worker[(size_t)k]->post([&fn, k] { return fn(k); });
		for (int k = 0; k < n; k++) rc[(size_t)k] = m->worker[(size_t)k]->wait();
	}
	for (int k = 0; k < n; k++) if (rc[(size_t)k] != TD_OK) return k;
	return -1;
}

extern "C" const char* td_multi_last_error(const td_multi* m) { return m ? m->err.c_str() : g_multi_create_error.c_str(); }
extern "C" int32_t td_multi_size(const td_multi* m) { return m ? (int32_t)m->ctx.size() : 0; }
extern "C" td_ctx* td_multi_ctx(td_multi* m, int32_t k) { return (m && k >= 0 && k < (int32_t)m->ctx.size()) ? m->ctx[(size_t)k] : nullptr; }
extern "C" int32_t td_multi_uses_rccl(const td_multi* m) { return m && !m->comm.empty(); }

extern "C" void td_multi_destroy(td_multi* m)
{
	if (!m) return;
	for (DeviceWorker* w : m->worker) delete w;   // (joins)
	m->worker.clear();
	for (size_t k = 0; k < m->comm.size(); k++) if (m->comm[k]) (void)g_rccl.CommDestroy(m->comm[k]);
	for (size_t k = 0; k < m->ctx.size(); k++) {
		(void)hipSetDevice(m->devices[k]);
		if (k < m->red_stream.size() && m->red_stream[k]) (void)hipStreamDestroy(m->red_stream[k]);
		if (k < m->d_sum.size() && m->d_sum[k]) (void)hipFree(m->d_sum[k]);
		td_ctx_destroy(m->ctx[k]);
	}
	delete m;
}

extern "C" int td_multi_create(const int32_t* devices, int32_t n_devices, td_multi** out)
{
	if (!out || n_devices < 1 || n_devices > 64) return mfail(nullptr, "td_multi_create: bad arguments");
	*out = nullptr;
	td_multi* m = new td_multi();
	bool distinct = true;
	for (int k = 0; k < n_devices; k++) {
		const int32_t d = devices ? devices[k] : k;
		for (int32_t e : m->devices) if (e == d) distinct = false;
		m->devices.push_back(d);
	}
	for (int k = 0; k < n_devices; k++) {
		td_ctx* c = nullptr;
		if (td_ctx_create(m->devices[(size_t)k], &c) != TD_OK) {
			const std::string why = td_last_error(nullptr);
			td_multi_destroy(m);
			return mfail(nullptr, "td_multi_create: device %d: %s", devices ? devices[k] : k, why.c_str());
		}
		m->ctx.push_back(c);
	}
	if (n_devices > 1) {
		// the machine's host threads shared out over the devices' copy pools (TD_HOST_THREADS, when set, is per device)
		if (!getenv("TD_HOST_THREADS")) {
			int per = (int)std::thread::hardware_concurrency() / n_devices;
			if (per > 16) per = 16;
			if (per < 1) per = 1;
			for (td_ctx* c : m->ctx) (void)td_set_option(c, "host_threads", per);
		}
		// worker threads bound to their device's NUMA node (TD_MULTI_BIND=0: leave them unbound)
		const char* b = getenv("TD_MULTI_BIND");
		const bool bind = distinct && !(b && atoi(b) == 0);
		for (int k = 0; k < n_devices; k++) { DeviceWorker* w = new DeviceWorker(); m->worker.push_back(w); w->start(m->devices[(size_t)k], bind); }
	}
	// RCCL communicator: several distinct devices -- or one, when TD_MULTI_FORCE_RCCL=1 asks for the collective path on a
	// one-GPU box (a 1-rank communicator: same calls, same library, no peer)
	const char* force = getenv("TD_MULTI_FORCE_RCCL");
	if (distinct && (n_devices > 1 || (force && atoi(force) != 0))) {
		std::string why;
		if (!g_rccl.load(why)) { td_multi_destroy(m); return mfail(nullptr, "td_multi_create: %s", why.c_str()); }
		m->comm.assign((size_t)n_devices, nullptr);
		std::vector<int> devs(m->devices.begin(), m->devices.end());
		const ncclResult_t r = g_rccl.CommInitAll(m->comm.data(), n_devices, devs.data());
		if (r != ncclSuccess) {
			m->comm.clear();
			td_multi_destroy(m);
			return mfail(nullptr, "td_multi_create: ncclCommInitAll: %s", g_rccl.GetErrorString(r));
		}
		m->red_stream.assign((size_t)n_devices, nullptr);
		m->d_sum.assign((size_t)n_devices, nullptr);
		for (int k = 0; k < n_devices; k++) {
			if (hipSetDevice(m->devices[(size_t)k]) != hipSuccess || hipStreamCreateWithFlags(&m->red_stream[(size_t)k], hipStreamNonBlocking) != hipSuccess ||
			    hipMalloc((void**)&m->d_sum[(size_t)k], sizeof(int64_t) * TD_NUM_COUNTERS) != hipSuccess) {
				td_multi_destroy(m);
				return mfail(nullptr, "td_multi_create: HIP resources for the reduce on device %d", m->devices[(size_t)k]);
			}
		}
	}
	*out = m;
	return TD_OK;
}

extern "C" int td_multi_model_upload(td_multi* m, const td_model_desc* model)
{
	if (!m || !model) return TD_FAIL;
	// the first device compiles the specialised kernel (seconds); the others find it in the code cache
	if (td_model_upload(m->ctx[0], model) != TD_OK) return mfail(m, "device %d: %s", m->devices[0], td_last_error(m->ctx[0]));
	const int bad = for_each_device(m, [&](int k) { return k == 0 ? TD_OK : td_model_upload(m->ctx[(size_t)k], model); });
	if (bad >= 0) return mfail(m, "device %d: %s", m->devices[(size_t)bad], td_last_error(m->ctx[(size_t)bad]));
	return TD_OK;
}

extern "C" int td_multi_set_params(td_multi* m, float threshold, int32_t minlen, int32_t dust)
{
	if (!m) return TD_FAIL;
	for (td_ctx* c : m->ctx) if (td_set_params(c, threshold, minlen, dust) != TD_OK) return mfail(m, "%s", td_last_error(c));
	return TD_OK;
}

extern "C" int td_multi_set_window(td_multi* m, int32_t matchstart, int32_t matchend)
{
	if (!m) return TD_FAIL;
	for (td_ctx* c : m->ctx) if (td_set_window(c, matchstart, matchend) != TD_OK) return mfail(m, "%s", td_last_error(c));
	return TD_OK;
}

extern "C" int td_multi_set_artifacts(td_multi* m, const uint8_t* string, const int32_t* s_index, int32_t n_seq, int32_t filter_error, int32_t n_threads)
{
	if (!m) return TD_FAIL;
	for (td_ctx* c : m->ctx) if (td_set_artifacts(c, string, s_index, n_seq, filter_error, n_threads) != TD_OK) return mfail(m, "%s", td_last_error(c));
	return TD_OK;
}

extern "C" int td_multi_decode(td_multi* m, const void* bases, int32_t is_ascii, const int64_t* offs, int64_t n, int mode,
                               td_read_result* res, int8_t* labels, uint8_t* seq_out)
{
	if (!m || !offs || n < 0) return mfail(m, "td_multi_decode: bad arguments");
	const int world = (int)m->ctx.size();
	const int bad = for_each_device(m, [&](int k) {
		int64_t lo = 0, hi = 0;
		td_shard_bounds(n, world, k, &lo, &hi);
		td_ctx* c = m->ctx[(size_t)k];
		// A device's range goes through the pipelined calls in pieces (whole tiles of 64 reads), as many in flight as the
		// context's pipeline is deep: the upload of one piece and the download of another run beside the decode kernel of a
		// third, and the decode kernels overlap at their ends -- one run_pHMM call per batch stays one call for the caller.
		// Per-read results do not depend on which reads share a tile or a launch; the artifact filter's thread ranges are those
		// of the whole batch (td_set_batch_window per piece).
		// (pieces of 2^17 reads = 2048 tiles: two of them, on the two streams, fill the 4096 wave slots of the machine)
		int pieces = (int)((hi - lo) >> 17);
		if (pieces > 8) pieces = 8;
		if (pieces < 1) pieces = 1;
		if (const char* e = getenv("TD_MULTI_PIECES")) { const int v = atoi(e); if (v >= 1 && v <= 64) pieces = v; }
		int32_t depth = 1;
		(void)td_get_option(c, "pipeline_depth", &depth);
		int64_t per = ((hi - lo + pieces - 1) / pieces + 63) / 64 * 64;
		if (per < 64) per = 64;
		std::vector<int64_t> tickets;
		int rc = TD_OK;
		size_t waited = 0;
		for (int64_t a = lo; a < hi || a == lo; a += per) {
			const int64_t b = a + per < hi ? a + per : hi;
			if ((int)(tickets.size() - waited) >= depth) { rc = td_wait(c, tickets[waited++]); if (rc != TD_OK) break; }
			if (td_set_batch_window(c, a, n) != TD_OK) { rc = TD_FAIL; break; }
			int64_t ticket = 0;
			// offsets keep the caller's base: read i of this piece starts at offs[a + i] in `bases` and in `seq_out`, and its
			// labels at offs[a + i] + (a + i)
			rc = td_submit(c, bases, is_ascii, offs + a, b - a, mode, res ? res + a : nullptr,
			               labels ? labels + offs[a] + a : nullptr, seq_out ? seq_out + offs[a] : nullptr, &ticket);
			if (rc != TD_OK) break;
			tickets.push_back(ticket);
			if (b >= hi) break;
		}
		for (; waited < tickets.size(); waited++) { const int r2 = td_wait(c, tickets[waited]); if (rc == TD_OK) rc = r2; }
		return rc;
	});
	for (td_ctx* c : m->ctx) (void)td_set_batch_window(c, 0, 0);
	if (bad >= 0) return mfail(m, "device %d: %s", m->devices[(size_t)bad], td_last_error(m->ctx[(size_t)bad]));
	return TD_OK;
}

extern "C" int td_multi_counts_reset(td_multi* m)
{
	if (!m) return TD_FAIL;
	for (td_ctx* c : m->ctx) if (td_counts_reset(c) != TD_OK) return mfail(m, "%s", td_last_error(c));
	return TD_OK;
}

extern "C" int td_multi_counts(td_multi* m, int64_t* counts)
{
	if (!m || !counts) return TD_FAIL;
	const int world = (int)m->ctx.size();
	if (m->comm.empty()) {   // one device (or one device listed several times): read back and add
		memset(counts, 0, sizeof(int64_t) * TD_NUM_COUNTERS);
		for (td_ctx* c : m->ctx) {
			int64_t part[TD_NUM_COUNTERS];
			if (td_counts_get(c, part) != TD_OK) return mfail(m, "%s", td_last_error(c));
			for (int j = 0; j < TD_NUM_COUNTERS; j++) counts[j] += part[j];
		}
		return TD_OK;
	}
	// every device's counters are complete once its compute stream is idle; then one all-reduce over xGMI, sum into a
	// second buffer so that the running counters stay per device
	for (td_ctx* c : m->ctx) if (td_sync(c) != TD_OK) return mfail(m, "%s", td_last_error(c));
	ncclResult_t r = g_rccl.GroupStart();
	for (int k = 0; k < world && r == ncclSuccess; k++)
		r = g_rccl.AllReduce(td_counts_device_ptr(m->ctx[(size_t)k]), m->d_sum[(size_t)k], TD_NUM_COUNTERS, ncclInt64, ncclSum, m->comm[(size_t)k], m->red_stream[(size_t)k]);
	const ncclResult_t r2 = g_rccl.GroupEnd();
	if (r == ncclSuccess) r = r2;
	if (r != ncclSuccess) return mfail(m, "td_multi_counts: ncclAllReduce: %s", g_rccl.GetErrorString(r));
	for (int k = 0; k < world; k++) {
		if (hipSetDevice(m->devices[(size_t)k]) != hipSuccess || hipStreamSynchronize(m->red_stream[(size_t)k]) != hipSuccess)
			return mfail(m, "td_multi_counts: device %d: %s", m->devices[(size_t)k], hipGetErrorString(hipGetLastError()));
	}
	if (hipSetDevice(m->devices[0]) != hipSuccess || hipMemcpy(counts, m->d_sum[0], sizeof(int64_t) * TD_NUM_COUNTERS, hipMemcpyDeviceToHost) != hipSuccess)
		return mfail(m, "td_multi_counts: read-back failed");
	return TD_OK;
}
