// Context, device memory, error reporting and HIP-event profiling for libmnc_hip.so.
#include <atomic>
#include <cstdlib>

#include "mnc_internal.h"

namespace mnc {

static thread_local char g_err[512] = "";

static const char* const kTuneNames[T_COUNT] = {
#define MNC_TUNE_NAME(n) #n,
    MNC_TUNE_KEYS(MNC_TUNE_NAME)
#undef MNC_TUNE_NAME
};

// "5" -> 5; "2,4" (a tile pair) -> 2 * 1000 + 4
static int parse_tune(const char* v) {
  const char* comma = strchr(v, ',');
  return comma ? atoi(v) * 1000 + atoi(comma + 1) : atoi(v);
}

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
void clear_error() { g_err[0] = 0; }

int ensure_scratch(mnc_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->scratch_bytes) return MNC_OK;
  MNC_NO_CAPTURE(ctx, "scratch arena growth");
  MNC_HIP_TRY(hipSetDevice(ctx->device));
  if (ctx->scratch) {
    MNC_HIP_TRY(hipStreamSynchronize(ctx->stream));
    MNC_HIP_TRY(hipFree(ctx->scratch));
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
    ++ctx->arena_gen;
  }
  size_t want = bytes + (bytes >> 2);
  hipError_t e = hipMalloc(&ctx->scratch, want);
  if (e != hipSuccess) {
    set_error("hipMalloc(%zu) for scratch failed: %s", want, hipGetErrorString(e));
    return MNC_ERR_NOMEM;
  }
  ctx->scratch_bytes = want;
  ++ctx->arena_gen;
  return MNC_OK;
}

static hipEvent_t take_event(mnc_ctx* ctx) {
  if (!ctx->event_pool.empty()) {
    hipEvent_t e = ctx->event_pool.back();
    ctx->event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

void prof_begin(mnc_ctx* ctx, const char* name, double flops, double bytes) {
  ProfRecord r{name, take_event(ctx), take_event(ctx), flops, bytes};
  (void)hipEventRecord(r.start, ctx->stream);
  ctx->prof.push_back(r);
}
void prof_end(mnc_ctx* ctx) {
  if (!ctx->prof.empty()) (void)hipEventRecord(ctx->prof.back().stop, ctx->stream);
}

}  // namespace mnc

using namespace mnc;

extern "C" {

const char* mnc_last_error(void) { return g_err; }
#ifdef MNC_TUNING
const char* mnc_version(void) { return "mnc_hip 0.3 (gfx950, tuning build: ablation and superseded kernels included)"; }
#else
const char* mnc_version(void) { return "mnc_hip 0.3 (gfx950)"; }
#endif

int mnc_device_mem_info(int device_id, size_t* free_bytes, size_t* total_bytes) {
  MNC_REQUIRE(free_bytes && total_bytes, "mnc_device_mem_info: null pointer");
  MNC_HIP_TRY(hipSetDevice(device_id));
  MNC_HIP_TRY(hipMemGetInfo(free_bytes, total_bytes));
  clear_error();
  return MNC_OK;
}

int mnc_device_count(int* count) {
  MNC_REQUIRE(count, "mnc_device_count: null pointer");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    n = 0;
  }
  *count = n;
  clear_error();
  return MNC_OK;
}

// Round 6: the HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); a host that keeps more images in flight
// than that has streams sharing a queue, and a shared queue serialises them (profiles/r06_streams.txt: the dip at the fifth stream).
// With 16 queues twelve images in flight measure 280.6 images/s against 270.0 with four (fp32, one box, two runs each; 8 queues / 8
// images: 278.2).  The variable is read when the runtime initialises -- at the process's first HIP call -- so the library sets a
// default when it is loaded, before it makes any (an already exported value wins; a process that initialised HIP earlier keeps its
// own setting).
__attribute__((constructor)) static void mnc_default_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "16", /*overwrite=*/0); }

int mnc_ctx_create(mnc_ctx** out, int device_id) {
  MNC_REQUIRE(out, "mnc_ctx_create: null out pointer");
  *out = nullptr;
  int n = 0;
  MNC_HIP_TRY(hipGetDeviceCount(&n));
  MNC_REQUIRE(device_id >= 0 && device_id < n, "mnc_ctx_create: device %d out of range (have %d)", device_id, n);
  MNC_HIP_TRY(hipSetDevice(device_id));
  mnc_ctx* ctx = new (std::nothrow) mnc_ctx();
  if (!ctx) {
    set_error("mnc_ctx_create: out of host memory");
    return MNC_ERR_NOMEM;
  }
  ctx->device = device_id;
  for (int k = 0; k < T_COUNT; ++k) {                 // the environment is read here, once per context, never on a launch path
    char name[64];
    snprintf(name, sizeof(name), "MNC_%s", kTuneNames[k]);
    const char* v = getenv(name);
    ctx->tune[k] = v && *v ? parse_tune(v) : kTuneUnset;
  }
  hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete ctx;
    set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
    return MNC_ERR_HIP;
  }
  e = hipMalloc((void**)&ctx->tickets, kTickets * sizeof(unsigned));
  if (e == hipSuccess) e = hipMemsetAsync(ctx->tickets, 0, kTickets * sizeof(unsigned), ctx->stream);   // (ordered before the stream's launches)
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) {
    if (ctx->tickets) (void)hipFree(ctx->tickets);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    set_error("mnc_ctx_create: arrival tickets: %s", hipGetErrorString(e));
    return MNC_ERR_HIP;
  }
  *out = ctx;
  clear_error();
  return MNC_OK;
}

int mnc_ctx_destroy(mnc_ctx* ctx) {
  if (!ctx) return MNC_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (auto& r : ctx->prof) {
    (void)hipEventDestroy(r.start);
    (void)hipEventDestroy(r.stop);
  }
  for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  if (ctx->proposal) mnc::proposal_state_free(ctx->proposal);
  if (ctx->vote_ws) (void)hipFree(ctx->vote_ws);
  if (ctx->tickets) (void)hipFree(ctx->tickets);
  if (ctx->comm) mnc::comm_free(ctx);
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  clear_error();
  return MNC_OK;
}

int mnc_ctx_sync(mnc_ctx* ctx) {
  MNC_REQUIRE(ctx, "mnc_ctx_sync: null context");
  MNC_NO_CAPTURE(ctx, "mnc_ctx_sync");
  const hipError_t e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) {
    // A launch that failed or faulted midway may have left arrival tickets of the in-launch K-range reductions non-zero
    // (mnc_internal.h: slab_last_arriver resets a tile's ticket only when its last workgroup arrives); every later launch would
    // then never see "arrivals - 1" and leave its tiles unfinished.  Put the counters back before reporting the error.
    if (ctx->tickets) (void)hipMemsetAsync(ctx->tickets, 0, mnc::kTickets * sizeof(unsigned), ctx->stream);
    mnc::set_error("hipStreamSynchronize failed: %s (%s:%d)", hipGetErrorString(e), __FILE__, __LINE__);
    return MNC_ERR_HIP;
  }
  clear_error();
  return MNC_OK;
}

int mnc_ctx_device(const mnc_ctx* ctx, int* device_id) {
  MNC_REQUIRE(ctx && device_id, "mnc_ctx_device: null pointer");
  *device_id = ctx->device;
  return MNC_OK;
}

int mnc_ctx_arena_generation(const mnc_ctx* ctx, unsigned long* generation) {
  MNC_REQUIRE(ctx && generation, "mnc_ctx_arena_generation: null pointer");
  *generation = ctx->arena_gen;
  return MNC_OK;
}

int mnc_ctx_capture_begin(mnc_ctx* ctx) {
  MNC_REQUIRE(ctx, "mnc_ctx_capture_begin: null context");
  MNC_REQUIRE(!ctx->capturing, "mnc_ctx_capture_begin: a capture is already open on this context");
  if (ctx->profiling != 0) {       // LaunchScope would record event pairs into the captured stream: they never execute
    set_error("mnc_ctx_capture_begin: per-launch profiling is on (mnc_prof_enable); event pairs cannot be captured");
    return MNC_ERR_STATE;
  }
  MNC_HIP_TRY(hipSetDevice(ctx->device));
  MNC_HIP_TRY(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed));
  ctx->capturing = true;
  ctx->capture_gen = ctx->arena_gen;
  clear_error();
  return MNC_OK;
}

int mnc_ctx_capture_end(mnc_ctx* ctx, mnc_graph** out) {
  MNC_REQUIRE(ctx && out, "mnc_ctx_capture_end: null pointer");
  *out = nullptr;
  MNC_REQUIRE(ctx->capturing, "mnc_ctx_capture_end: no capture is open on this context");
  ctx->capturing = false;
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(ctx->stream, &g);
  if (e != hipSuccess || !g) {
    (void)hipGetLastError();
    if (g) (void)hipGraphDestroy(g);
    set_error("mnc_ctx_capture_end: the capture was invalidated (%s): something synchronised or allocated inside it",
              hipGetErrorString(e));
    return MNC_ERR_HIP;
  }
  if (ctx->capture_gen != ctx->arena_gen) {          // an internal arena moved while capturing: earlier nodes hold the old address
    (void)hipGraphDestroy(g);
    set_error("mnc_ctx_capture_end: an internal arena was re-allocated during the capture (run the sequence eagerly once first)");
    return MNC_ERR_STATE;
  }
  hipGraphExec_t exec = nullptr;
  e = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess || !exec) {
    (void)hipGetLastError();
    set_error("mnc_ctx_capture_end: hipGraphInstantiate failed: %s", hipGetErrorString(e));
    return MNC_ERR_HIP;
  }
  mnc_graph* gr = new (std::nothrow) mnc_graph();
  if (!gr) { (void)hipGraphExecDestroy(exec); set_error("mnc_ctx_capture_end: out of host memory"); return MNC_ERR_NOMEM; }
  gr->exec = exec;
  gr->ctx = ctx;
  gr->arena_gen = ctx->arena_gen;
  *out = gr;
  clear_error();
  return MNC_OK;
}

int mnc_graph_launch(mnc_ctx* ctx, mnc_graph* graph) {
  MNC_REQUIRE(ctx && graph && graph->ctx == ctx, "mnc_graph_launch: null pointer, or a graph of another context");
  if (graph->arena_gen != ctx->arena_gen) {
    set_error("mnc_graph_launch: an internal arena of the context was re-allocated since the capture; capture again");
    return MNC_ERR_STATE;
  }
  MNC_HIP_TRY(hipSetDevice(ctx->device));
  MNC_HIP_TRY(hipGraphLaunch(graph->exec, ctx->stream));
  return MNC_OK;
}

int mnc_graph_destroy(mnc_graph* graph) {
  if (!graph) return MNC_OK;
  if (graph->exec) (void)hipGraphExecDestroy(graph->exec);
  delete graph;
  return MNC_OK;
}

int mnc_ctx_set_tuning(mnc_ctx* ctx, const char* name, const char* value) {
  MNC_REQUIRE(ctx && name, "mnc_ctx_set_tuning: null pointer");
  if (!strncmp(name, "MNC_", 4)) name += 4;
  for (int k = 0; k < T_COUNT; ++k)
    if (!strcmp(name, kTuneNames[k])) {
      ctx->tune[k] = value && *value ? parse_tune(value) : kTuneUnset;
      clear_error();
      return MNC_OK;
    }
  set_error("mnc_ctx_set_tuning: unknown key %s", name);
  return MNC_ERR_INVALID;
}

int mnc_ctx_set_layer_conventions(mnc_ctx* ctx, const mnc_layer_conventions* conv) {
  MNC_REQUIRE(ctx, "mnc_ctx_set_layer_conventions: null context");
  const mnc_layer_conventions spec = {0, 0, 0, 0, 0, 0, 0.4f, 0};
  const mnc_layer_conventions c = conv ? *conv : spec;
  MNC_REQUIRE(c.warp_sample >= 0 && c.warp_sample <= 2, "layer conventions: warp_sample %d not in {0,1,2}", c.warp_sample);
  MNC_REQUIRE(c.resize_mode >= 0 && c.resize_mode <= 2, "layer conventions: resize_mode %d not in {0,1,2}", c.resize_mode);
  MNC_REQUIRE((c.warp_round_edges | 1) == 1 && (c.warp_no_plus_one | 1) == 1 && (c.warp_oob | 1) == 1 && (c.maskpool_binary | 1) == 1,
              "layer conventions: warp_round_edges / warp_no_plus_one / warp_oob / maskpool_binary are 0 or 1");
  MNC_REQUIRE(c.maskpool_thresh == c.maskpool_thresh, "layer conventions: maskpool_thresh is NaN");
  ctx->conv = c;
  ctx->conv.inherit = 0;
  clear_error();
  return MNC_OK;
}

int mnc_ctx_get_layer_conventions(const mnc_ctx* ctx, mnc_layer_conventions* conv) {
  MNC_REQUIRE(ctx && conv, "mnc_ctx_get_layer_conventions: null pointer");
  *conv = ctx->conv;
  return MNC_OK;
}

int mnc_dev_alloc(mnc_ctx* ctx, size_t bytes, void** d_ptr) {
  MNC_REQUIRE(ctx && d_ptr, "mnc_dev_alloc: null pointer");
  *d_ptr = nullptr;
  MNC_NO_CAPTURE(ctx, "mnc_dev_alloc");
  MNC_HIP_TRY(hipSetDevice(ctx->device));
  hipError_t e = hipMalloc(d_ptr, bytes ? bytes : 16);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    set_error("mnc_dev_alloc: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return MNC_ERR_NOMEM;
  }
  clear_error();
  return MNC_OK;
}

int mnc_dev_free(mnc_ctx* ctx, void* d_ptr) {
  MNC_REQUIRE(ctx, "mnc_dev_free: null context");
  if (!d_ptr) return MNC_OK;
  MNC_NO_CAPTURE(ctx, "mnc_dev_free");
  MNC_HIP_TRY(hipStreamSynchronize(ctx->stream));
  MNC_HIP_TRY(hipFree(d_ptr));
  return MNC_OK;
}

int mnc_host_alloc(mnc_ctx* ctx, size_t bytes, void** host_ptr) {
  MNC_REQUIRE(ctx && host_ptr, "mnc_host_alloc: null pointer");
  *host_ptr = nullptr;
  MNC_HIP_TRY(hipSetDevice(ctx->device));
  hipError_t e = hipHostMalloc(host_ptr, bytes ? bytes : 16, hipHostMallocDefault);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    set_error("mnc_host_alloc: hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return MNC_ERR_NOMEM;
  }
  clear_error();
  return MNC_OK;
}

int mnc_host_free(mnc_ctx* ctx, void* host_ptr) {
  MNC_REQUIRE(ctx, "mnc_host_free: null context");
  if (!host_ptr) return MNC_OK;
  MNC_NO_CAPTURE(ctx, "mnc_host_free");
  MNC_HIP_TRY(hipStreamSynchronize(ctx->stream));
  MNC_HIP_TRY(hipHostFree(host_ptr));
  return MNC_OK;
}

int mnc_h2d_async(mnc_ctx* ctx, void* d_dst, const void* src_host, size_t bytes) {
  MNC_REQUIRE(ctx && (bytes == 0 || (d_dst && src_host)), "mnc_h2d_async: null pointer");
  if (bytes == 0) return MNC_OK;
  MNC_HIP_TRY(hipMemcpyAsync(d_dst, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
  return MNC_OK;
}

int mnc_d2h_async(mnc_ctx* ctx, void* dst_host, const void* d_src, size_t bytes) {
  MNC_REQUIRE(ctx && (bytes == 0 || (dst_host && d_src)), "mnc_d2h_async: null pointer");
  if (bytes == 0) return MNC_OK;
  MNC_HIP_TRY(hipMemcpyAsync(dst_host, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  return MNC_OK;
}

int mnc_h2d(mnc_ctx* ctx, void* d_dst, const void* src_host, size_t bytes) {
  MNC_REQUIRE(ctx && (bytes == 0 || (d_dst && src_host)), "mnc_h2d: null pointer");
  if (bytes == 0) return MNC_OK;
  MNC_NO_CAPTURE(ctx, "mnc_h2d");
  MNC_HIP_TRY(hipMemcpyAsync(d_dst, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
  MNC_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return MNC_OK;
}

int mnc_d2h(mnc_ctx* ctx, void* dst_host, const void* d_src, size_t bytes) {
  MNC_REQUIRE(ctx && (bytes == 0 || (dst_host && d_src)), "mnc_d2h: null pointer");
  if (bytes == 0) return MNC_OK;
  MNC_NO_CAPTURE(ctx, "mnc_d2h");
  MNC_HIP_TRY(hipMemcpyAsync(dst_host, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  MNC_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return MNC_OK;
}

int mnc_d2d(mnc_ctx* ctx, void* d_dst, const void* d_src, size_t bytes) {
  MNC_REQUIRE(ctx && (bytes == 0 || (d_dst && d_src)), "mnc_d2d: null pointer");
  if (bytes == 0) return MNC_OK;
  MNC_HIP_TRY(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return MNC_OK;
}

int mnc_dev_zero(mnc_ctx* ctx, void* d_ptr, size_t bytes) {
  MNC_REQUIRE(ctx && (bytes == 0 || d_ptr), "mnc_dev_zero: null pointer");
  if (bytes == 0) return MNC_OK;
  MNC_HIP_TRY(hipMemsetAsync(d_ptr, 0, bytes, ctx->stream));
  return MNC_OK;
}

int mnc_prof_enable(mnc_ctx* ctx, int enable) {
  MNC_REQUIRE(ctx, "mnc_prof_enable: null context");
  ctx->profiling = enable < 0 ? 0 : (enable > 2 ? 1 : enable);
  return MNC_OK;
}

int mnc_prof_reset(mnc_ctx* ctx) {
  MNC_REQUIRE(ctx, "mnc_prof_reset: null context");
  MNC_NO_CAPTURE(ctx, "mnc_prof_reset");
  MNC_HIP_TRY(hipStreamSynchronize(ctx->stream));
  for (auto& r : ctx->prof) {
    ctx->event_pool.push_back(r.start);
    ctx->event_pool.push_back(r.stop);
  }
  ctx->prof.clear();
  return MNC_OK;
}

int mnc_prof_count(mnc_ctx* ctx, int* n_records) {
  MNC_REQUIRE(ctx && n_records, "mnc_prof_count: null pointer");
  MNC_NO_CAPTURE(ctx, "mnc_prof_count");
  MNC_HIP_TRY(hipStreamSynchronize(ctx->stream));
  *n_records = (int)ctx->prof.size();
  return MNC_OK;
}

int mnc_prof_get(mnc_ctx* ctx, int index, char* name_buf, int name_cap, float* ms, double* flops, double* bytes) {
  MNC_REQUIRE(ctx && index >= 0 && index < (int)ctx->prof.size(), "mnc_prof_get: index %d out of range", index);
  const ProfRecord& r = ctx->prof[index];
  if (name_buf && name_cap > 0) {
    strncpy(name_buf, r.name, name_cap - 1);
    name_buf[name_cap - 1] = 0;
  }
  if (ms) {
    float t = 0.f;
    MNC_HIP_TRY(hipEventElapsedTime(&t, r.start, r.stop));
    *ms = t;
  }
  if (flops) *flops = r.flops;
  if (bytes) *bytes = r.bytes;
  return MNC_OK;
}

}  // extern "C"
