// Context, memory helpers and the dense column-major matrix object (upload / generate / download).
#include <string>
#include <cstdarg>

#include <cstdlib>

#include "pg_internal.h"

static thread_local std::string g_last_error;

void pg_set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
}

std::mutex& pg_coop_launch_mutex() {
  static std::mutex mu;
  return mu;
}

extern "C" {

int32_t pg_abi_version(void) { return PG_ABI_VERSION; }
const char* pg_last_error(void) { return g_last_error.c_str(); }

pg_status pg_ctx_create(int32_t device, void* stream, pg_ctx** out) {
  PG_REQUIRE(out != nullptr, "out is null");
  *out = nullptr;
  int count = 0;
  PG_HIP(hipGetDeviceCount(&count));
  PG_REQUIRE(device >= 0 && device < count, "device index out of range");
  PG_HIP(hipSetDevice(device));
  pg_ctx* c = new pg_ctx();
  c->device = device;
  c->stream = (hipStream_t)stream;
  hipError_t e = hipGetDeviceProperties(&c->prop, device);
  if (e != hipSuccess) {
    delete c;
    pg_set_error("hipGetDeviceProperties failed: %s", hipGetErrorString(e));
    return PG_ERR_HIP;
  }
  c->num_cu = c->prop.multiProcessorCount > 0 ? c->prop.multiProcessorCount : 256;
  if (strncmp(c->prop.gcnArchName, "gfx950", 6) != 0) {
    pg_set_error("libproxgrad_hip is built for gfx950 (MI355X) only; device %d is %s", device,
                 c->prop.gcnArchName);
    delete c;
    return PG_ERR_UNSUPPORTED;
  }
  if (hipMalloc(&c->red_partials, sizeof(double) * PG_RED_MAX_BLOCKS * PG_RED_MAX_NS) != hipSuccess ||
      hipMalloc(&c->red_counter, sizeof(unsigned) * 8) != hipSuccess ||
      hipHostMalloc(&c->hscal, sizeof(double) * PG_S_COUNT, hipHostMallocMapped) != hipSuccess ||
      hipHostGetDevicePointer((void**)&c->dscal, c->hscal, 0) != hipSuccess) {
    pg_set_error("context workspace allocation failed");
    pg_ctx_destroy(c);
    return PG_ERR_ALLOC;
  }
  PG_HIP(hipMemset(c->red_counter, 0, sizeof(unsigned) * 8));
  memset(c->hscal, 0, sizeof(double) * PG_S_COUNT);
  if (const char* v = getenv("PG_TN_TEAM_PLAIN")) c->team_plain_launch = atoi(v) != 0;
  PG_HIP(hipDeviceSynchronize());
  *out = c;
  return PG_OK;
}

pg_status pg_ctx_destroy(pg_ctx* c) {
  if (!c) return PG_OK;
  if (c->comm) (void)pg_ctx_comm_destroy(c);
  for (hipEvent_t e : c->coop_probe)
    if (e) (void)hipEventDestroy(e);
  for (void* p : c->rteam_imported) (void)hipIpcCloseMemHandle(p);
  if (c->rteam.own) (void)hipFree(c->rteam.own);
  if (c->rteam.f_local) (void)hipFree(c->rteam.f_local);
  if (c->rteam.wait_stats) (void)hipFree(c->rteam.wait_stats);
  if (c->red_partials) (void)hipFree(c->red_partials);
  if (c->red_counter) (void)hipFree(c->red_counter);
  if (c->hscal) (void)hipHostFree(c->hscal);
  if (c->dr_ws) (void)hipFree(c->dr_ws);
  for (hipEvent_t e : c->dr_ev)
    if (e) (void)hipEventDestroy(e);
  if (c->small_out_host) (void)hipHostFree(c->small_out_host);
  if (c->coop_ws) (void)hipFree(c->coop_ws);
  for (int k = 0; k < PG_K_COUNT; ++k)
    for (auto& pr : c->prof_events[k]) {
      (void)hipEventDestroy(pr.first);
      (void)hipEventDestroy(pr.second);
    }
  for (auto e : c->prof_pool) (void)hipEventDestroy(e);
  delete c;
  return PG_OK;
}

pg_status pg_ctx_set_stream(pg_ctx* c, void* stream) {
  PG_REQUIRE(c != nullptr, "ctx is null");
  c->stream = (hipStream_t)stream;
  return PG_OK;
}

pg_status pg_ctx_set_allreduce(pg_ctx* c, pg_allreduce_fn fn, void* user) {
  PG_REQUIRE(c != nullptr, "ctx is null");
  c->allreduce = fn;
  c->allreduce_user = user;
  return PG_OK;
}

pg_status pg_ctx_set_allreduce_async(pg_ctx* c, pg_allreduce_fn begin, pg_allreduce_wait_fn wait, void* user) {
  PG_REQUIRE(c != nullptr, "ctx is null");
  PG_REQUIRE((begin == nullptr) == (wait == nullptr), "begin and wait must be given together");
  c->allreduce_begin = begin;
  c->allreduce_wait = wait;
  if (begin != nullptr) c->allreduce_user = user;
  return PG_OK;
}

pg_status pg_ctx_sync(pg_ctx* c) {
  PG_REQUIRE(c != nullptr, "ctx is null");
  PG_HIP(hipStreamSynchronize(c->stream));
  return PG_OK;
}

// ---- stream capture: a launch-bound iteration body recorded once, replayed as ONE graph launch --------------------
pg_status pg_ctx_capture_begin(pg_ctx* c) {
  PG_REQUIRE(c != nullptr, "ctx is null");
  PG_REQUIRE(!c->capturing, "a capture is already in progress");
  PG_REQUIRE(c->stream != nullptr, "the default (null) stream cannot be captured: create the context on its own stream");
  PG_REQUIRE(c->allreduce == nullptr && c->allreduce_begin == nullptr && c->comm == nullptr,
             "capture is not available on a context with a collective attached");
  PG_HIP(hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed));
  c->capturing = true;
  return PG_OK;
}

pg_status pg_ctx_capture_end(pg_ctx* c, pg_graph** out) {
  PG_REQUIRE(c != nullptr, "ctx is null");
  PG_REQUIRE(c->capturing, "no capture in progress");
  c->capturing = false;
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamEndCapture(c->stream, &graph);
  if (e != hipSuccess || graph == nullptr) {
    (void)hipGetLastError();
    pg_set_error("hipStreamEndCapture failed: %s", hipGetErrorString(e));
    return PG_ERR_HIP;
  }
  if (out == nullptr) {  // abort: drop what was recorded
    (void)hipGraphDestroy(graph);
    return PG_OK;
  }
  hipGraphExec_t exec = nullptr;
  e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGraphDestroy(graph);
    pg_set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e));
    return PG_ERR_HIP;
  }
  pg_graph* g = new pg_graph();
  g->ctx = c;
  g->graph = graph;
  g->exec = exec;
  *out = g;
  return PG_OK;
}

pg_status pg_graph_launch(pg_graph* g) {
  PG_REQUIRE(g != nullptr && g->exec != nullptr, "graph is null");
  PG_REQUIRE(!g->ctx->capturing, "cannot launch a graph while capturing");
  PG_HIP(hipGraphLaunch(g->exec, g->ctx->stream));
  return PG_OK;
}

pg_status pg_graph_destroy(pg_graph* g) {
  if (!g) return PG_OK;
  if (g->exec) (void)hipGraphExecDestroy(g->exec);
  if (g->graph) (void)hipGraphDestroy(g->graph);
  delete g;
  return PG_OK;
}

pg_status pg_ctx_profile_enable(pg_ctx* c, int32_t enable) {
  PG_REQUIRE(c != nullptr, "ctx is null");
  c->profiling = enable != 0;
  return PG_OK;
}

// ---- row teams (pg_gemv_tn4.hip) --------------------------------------------------------------------------------
pg_status pg_ctx_row_team_alloc(pg_ctx* c, void** inbox_out, int64_t* bytes_out) {
  PG_REQUIRE(c != nullptr && inbox_out != nullptr, "null argument");
  const size_t bytes = pgtn::peer_inbox_bytes();
  if (c->rteam.own == nullptr) {
    PG_HIP(hipSetDevice(c->device));
    // uncached / fine-grained device memory: a peer's stores must become visible to a running kernel of this device
    hipError_t e = hipExtMallocWithFlags(&c->rteam.own, bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      e = hipExtMallocWithFlags(&c->rteam.own, bytes, hipDeviceMallocFinegrained);
    }
    if (e != hipSuccess) {
      (void)hipGetLastError();
      pg_set_error("allocation of the row-team inbox (%zu bytes of fine-grained device memory) failed: %s", bytes, hipGetErrorString(e));
      return PG_ERR_ALLOC;
    }
    PG_HIP(hipMemset(c->rteam.own, 0, bytes));
  }
  // (outside the branch above: a call that failed here is completed by the next one)
  if (c->rteam.f_local == nullptr) PG_HIP(hipMalloc((void**)&c->rteam.f_local, 18 * sizeof(double)));
  if (c->rteam.wait_stats == nullptr) {
    PG_HIP(hipMalloc((void**)&c->rteam.wait_stats, 4 * sizeof(unsigned long long)));
    PG_HIP(hipMemset(c->rteam.wait_stats, 0, 4 * sizeof(unsigned long long)));
  }
  *inbox_out = c->rteam.own;
  if (bytes_out) *bytes_out = (int64_t)bytes;
  return PG_OK;
}

pg_status pg_ctx_row_team_export(pg_ctx* c, void* handle_out) {
  PG_REQUIRE(c != nullptr && handle_out != nullptr, "null argument");
  PG_REQUIRE(c->rteam.own != nullptr, "pg_ctx_row_team_alloc has not been called");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "the C ABI passes IPC handles as 64 bytes");
  hipIpcMemHandle_t h;
  PG_HIP(hipIpcGetMemHandle(&h, c->rteam.own));
  memcpy(handle_out, &h, sizeof(h));
  return PG_OK;
}

pg_status pg_ctx_row_team_import(pg_ctx* c, const void* handle, void** inbox_out) {
  PG_REQUIRE(c != nullptr && handle != nullptr && inbox_out != nullptr, "null argument");
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof(h));
  PG_HIP(hipSetDevice(c->device));
  void* p = nullptr;
  PG_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
  c->rteam_imported.push_back(p);
  *inbox_out = p;
  return PG_OK;
}

pg_status pg_ctx_set_row_team(pg_ctx* c, int32_t nranks, int32_t rank, void* const* inboxes, int32_t max_workgroups) {
  PG_REQUIRE(c != nullptr, "ctx is null");
  if (nranks <= 1 && !(nranks == 1 && c->rteam.solo && inboxes != nullptr)) {
    c->rteam.n = 0;
    return PG_OK;
  }
  PG_REQUIRE(nranks <= 16 && rank >= 0 && rank < nranks && inboxes != nullptr, "a row team has 2..16 devices");
  PG_REQUIRE(max_workgroups >= -16, "max_workgroups: a count, 0 (default), or -k when k members share this device");
  for (int q = 0; q < nranks; ++q) PG_REQUIRE(inboxes[q] != nullptr, "an inbox pointer is null");
  PG_REQUIRE(c->rteam.own != nullptr && inboxes[rank] == c->rteam.own, "inboxes[rank] must be this context's own inbox (pg_ctx_row_team_alloc)");
  c->rteam.n = nranks;
  c->rteam.rank = rank;
  c->rteam.max_wgs = max_workgroups;
  for (int q = 0; q < 16; ++q) c->rteam.inbox[q] = q < nranks ? inboxes[q] : nullptr;
  // epochs restart together: every device of the team makes this call at the same point of the program
  c->rteam.epoch = c->rteam.scal_epoch = 0;
  c->rteam.ring_sig = 0;
  c->rteam.sweeps = 0;
  c->rteam.gen++;
  if (c->rteam.wait_stats) PG_HIP(hipMemset(c->rteam.wait_stats, 0, 4 * sizeof(unsigned long long)));
  PG_HIP(hipMemset(c->rteam.own, 0, pgtn::peer_inbox_bytes()));
  return PG_OK;
}

pg_status pg_ctx_row_team_selftest(pg_ctx* c, double* sum_out) {
  PG_REQUIRE(c != nullptr && sum_out != nullptr, "null argument");
  PG_REQUIRE(pg_rteam_active(c) && c->rteam.f_local != nullptr, "the context is not a row team");
  const double mine = (double)(c->rteam.rank + 1);
  PG_HIP(hipMemcpyAsync(c->rteam.f_local, &mine, sizeof(double), hipMemcpyHostToDevice, c->stream));
  PG_HIP(hipStreamSynchronize(c->stream));
  PG_TRY(pg_rteam_sum_scalar(c, c->rteam.f_local, c->rteam.f_local));
  double got = 0.0;
  PG_HIP(hipMemcpyAsync(&got, c->rteam.f_local, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  const pg_status st = pg_read_scalars(c, PG_S_TEAMERR, 1);  // (syncs; a peer that never answered shows as PG_ERR_TIMEOUT)
  c->team_timeout = false;
  *sum_out = got;
  return st;
}

pg_status pg_ctx_row_team_stats(pg_ctx* c, int64_t* sweeps, int64_t* late_waves, int64_t* wait_polls) {
  PG_REQUIRE(c != nullptr, "ctx is null");
  unsigned long long h[2] = {0, 0};
  if (c->rteam.wait_stats) {
    PG_HIP(hipStreamSynchronize(c->stream));
    PG_HIP(hipMemcpy(h, c->rteam.wait_stats, sizeof(h), hipMemcpyDeviceToHost));
  }
  if (sweeps) *sweeps = c->rteam.sweeps;
  if (late_waves) *late_waves = (int64_t)h[0];
  if (wait_polls) *wait_polls = (int64_t)h[1];
  return PG_OK;
}

// The row-team sweep's geometry per context (no PG_TUNE): what a first run on real fabric turns without a rebuild.
pg_status pg_ctx_row_team_tune(pg_ctx* c, const char* key, int64_t value) {
  PG_REQUIRE(c != nullptr && key != nullptr, "null argument");
  pg_row_team::Tune& t = c->rteam.tune;
  const std::string k(key);
  PG_REQUIRE(value >= 0, "a knob's value is >= 0 (0: the default)");
  if (k == "C") { PG_REQUIRE(value == 0 || value == 1 || value == 2 || value == 4, "C: columns per step, 1 / 2 / 4"); t.C = (int)value; }
  else if (k == "LAG") { PG_REQUIRE(value <= 7, "LAG: tiles parked in LDS, <= 7"); t.LAG = (int)value; }
  else if (k == "LAGR") { PG_REQUIRE(value <= 4, "LAGR: value = tiles parked in registers + 1 (1: none), <= 4"); t.LAGR = (int)value; }
  else if (k == "PF") { PG_REQUIRE(value <= 2, "PF: tiles in flight, 1 / 2"); t.PF = (int)value; }
  else if (k == "WGS") { PG_REQUIRE(value <= 4, "WGS: workgroups per compute unit, <= 4"); t.WGS = (int)value; }
  else if (k == "W") { PG_REQUIRE(value == 0 || value == 1 || value == 2 || value == 4, "W: waves per column, 1 / 2 / 4"); t.W = (int)value; }
  else if (k == "K1") { PG_REQUIRE(value <= 2, "K1: 0 default, 1 the one-wave sweep, 2 round 5's kernel"); t.K1 = value == 0 ? -1 : (value == 1 ? 1 : 0); }
  else if (k == "PAIR") { PG_REQUIRE(value <= 2, "PAIR: 0 default, 1 one post per two steps, 2 one post per step"); t.PAIR = value == 0 ? -1 : (value == 1 ? 1 : 0); }
  else if (k == "AHEAD") { PG_REQUIRE(value <= 2, "AHEAD: 0 default, 1 the poll one step ahead of its use, 2 at the start of its own step"); t.AHEAD = value == 0 ? -1 : (value == 1 ? 1 : 0); }
  else if (k == "SPIN") { t.SPIN = (long long)value; }
  else {
    pg_set_error("pg_ctx_row_team_tune: unknown key '%s' (C, LAG, LAGR, PF, WGS, W, K1, PAIR, AHEAD, SPIN)", key);
    return PG_ERR_INVALID;
  }
  return PG_OK;
}

// "W=1 U=8 C=2 LAG=2 LAGR=2 PF=2 WGS=4 K1=1 PAIR=0 AHEAD=1 SPIN=2097152 WG=1024": what the LAST row-team sweep of this context ran with
pg_status pg_ctx_row_team_geometry(pg_ctx* c, char* buf, int64_t buflen) {
  PG_REQUIRE(c != nullptr && buf != nullptr && buflen > 0, "null argument");
  const pg_row_team::Geom& g = c->rteam.last;
  if (g.W == 0) snprintf(buf, (size_t)buflen, "none");
  else
    snprintf(buf, (size_t)buflen, "W=%d U=%d C=%d LAG=%d LAGR=%d PF=%d WGS=%d K1=%d PAIR=%d AHEAD=%d SPIN=%lld WG=%d", g.W, g.U, g.C, g.LAG, g.LAGR, g.PF, g.WGS, g.K1,
             g.PAIR, g.AHEAD, g.SPIN, g.nteams);
  return PG_OK;
}

pg_status pg_ctx_test_team_slack(pg_ctx* c, int64_t* ticks, int64_t* wave_steps) {
  PG_REQUIRE(c != nullptr && ticks != nullptr && wave_steps != nullptr, "null argument");
  unsigned long long h[4] = {0, 0, 0, 0};
  if (c->rteam.wait_stats) {
    PG_HIP(hipStreamSynchronize(c->stream));
    PG_HIP(hipMemcpy(h, c->rteam.wait_stats, sizeof(h), hipMemcpyDeviceToHost));
  }
  *ticks = (int64_t)h[2];
  *wave_steps = (int64_t)h[3];
  return PG_OK;
}

pg_status pg_ctx_test_team_fault(pg_ctx* c, int32_t kth_launch, int32_t kind) {
  PG_REQUIRE(c != nullptr && kth_launch >= 0 && kind >= 0 && kind <= 4, "bad argument");
  if (kind == 4) {  // a team of ONE device counts as a team from the next pg_ctx_set_row_team on (kth_launch != 0): profiling the sweep alone
    c->rteam.solo = kth_launch != 0;
    return PG_OK;
  }
  if (kind == 2 || kind == 3) {  // latency injector of the row-team sweep: kth_launch = nanoseconds (kind 3: injector off again)
    c->test_team_delay_on = kind == 2;
    c->test_team_delay_ticks = kind == 2 ? (unsigned)(kth_launch / 10) : 0u;
    return PG_OK;
  }
  c->test_team_fault = kth_launch;
  c->test_team_fault_kind = kind;
  c->team_launches = 0;
  return PG_OK;
}

pg_status pg_ctx_set_column_sharding(pg_ctx* c, int32_t nranks, int32_t rank) {
  PG_REQUIRE(c != nullptr, "ctx is null");
  PG_REQUIRE(nranks >= 0 && (nranks == 0 || (rank >= 0 && rank < nranks)), "rank out of range");
  c->shard_cols = nranks > 0 ? 1 : 0;
  c->shard_nranks = nranks > 0 ? nranks : 1;
  c->shard_rank = nranks > 0 ? rank : 0;
  return PG_OK;
}

pg_status pg_ctx_profile_select(pg_ctx* c, uint32_t kind_mask) {
  PG_REQUIRE(c != nullptr, "ctx is null");
  c->prof_mask = kind_mask;
  return PG_OK;
}

pg_status pg_ctx_profile_reset(pg_ctx* c) {
  PG_REQUIRE(c != nullptr, "ctx is null");
  PG_HIP(hipStreamSynchronize(c->stream));
  for (int k = 0; k < PG_K_COUNT; ++k) {
    for (auto& pr : c->prof_events[k]) {
      c->prof_pool.push_back(pr.first);
      c->prof_pool.push_back(pr.second);
    }
    c->prof_events[k].clear();
  }
  return PG_OK;
}

pg_status pg_ctx_profile_read(pg_ctx* c, int32_t kernel, int64_t* launches, double* total_ms) {
  PG_REQUIRE(c != nullptr, "ctx is null");
  PG_REQUIRE(kernel >= 0 && kernel < PG_K_COUNT, "unknown kernel id");
  PG_HIP(hipStreamSynchronize(c->stream));
  double tot = 0.0;
  for (auto& pr : c->prof_events[kernel]) {
    float ms = 0.f;
    PG_HIP(hipEventElapsedTime(&ms, pr.first, pr.second));
    tot += ms;
  }
  if (launches) *launches = (int64_t)c->prof_events[kernel].size();
  if (total_ms) *total_ms = tot;
  return PG_OK;
}

pg_status pg_ctx_device_info(pg_ctx* c, pg_device_info* out) {
  PG_REQUIRE(c != nullptr && out != nullptr, "null argument");
  memset(out, 0, sizeof(*out));
  out->device = c->device;
  out->compute_units = c->prop.multiProcessorCount;
  out->wavefront_size = c->prop.warpSize;
  out->lds_bytes_per_cu = (int32_t)c->prop.maxSharedMemoryPerMultiProcessor;
  out->global_mem_bytes = (int64_t)c->prop.totalGlobalMem;
  out->clock_khz = c->prop.clockRate;
  strncpy(out->arch, c->prop.gcnArchName, sizeof(out->arch) - 1);
  strncpy(out->name, c->prop.name, sizeof(out->name) - 1);
  return PG_OK;
}

pg_status pg_malloc(pg_ctx* c, size_t bytes, void** dptr) {
  PG_REQUIRE(c != nullptr && dptr != nullptr, "null argument");
  *dptr = nullptr;
  if (bytes == 0) return PG_OK;
  PG_HIP(hipSetDevice(c->device));
  hipError_t e = hipMalloc(dptr, bytes);
  if (e != hipSuccess) {
    pg_set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return PG_ERR_ALLOC;
  }
  return PG_OK;
}

pg_status pg_free(pg_ctx* c, void* dptr) {
  PG_REQUIRE(c != nullptr, "ctx is null");
  if (dptr) PG_HIP(hipFree(dptr));
  return PG_OK;
}

pg_status pg_memcpy_h2d(pg_ctx* c, void* dst, const void* src, size_t bytes) {
  PG_REQUIRE(c != nullptr, "ctx is null");
  if (bytes == 0) return PG_OK;
  PG_REQUIRE(dst != nullptr && src != nullptr, "null pointer");
  PG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  PG_HIP(hipStreamSynchronize(c->stream));  // host buffer is only pinned for the duration of the call
  return PG_OK;
}

pg_status pg_memcpy_d2h(pg_ctx* c, void* dst, const void* src, size_t bytes) {
  PG_REQUIRE(c != nullptr, "ctx is null");
  if (bytes == 0) return PG_OK;
  PG_REQUIRE(dst != nullptr && src != nullptr, "null pointer");
  PG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  PG_HIP(hipStreamSynchronize(c->stream));
  return PG_OK;
}

pg_status pg_memcpy_d2d(pg_ctx* c, void* dst, const void* src, size_t bytes) {
  PG_REQUIRE(c != nullptr, "ctx is null");
  if (bytes == 0) return PG_OK;
  PG_REQUIRE(dst != nullptr && src != nullptr, "null pointer");
  PG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
  return PG_OK;
}

pg_status pg_memset_zero(pg_ctx* c, void* dst, size_t bytes) {
  PG_REQUIRE(c != nullptr, "ctx is null");
  if (bytes == 0) return PG_OK;
  PG_REQUIRE(dst != nullptr, "null pointer");
  PG_HIP(hipMemsetAsync(dst, 0, bytes, c->stream));
  return PG_OK;
}

}  // extern "C"

static hipEvent_t prof_get_event(pg_ctx* c) {
  if (!c->prof_pool.empty()) {
    hipEvent_t e = c->prof_pool.back();
    c->prof_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

pg_prof_scope::pg_prof_scope(pg_ctx* ctx, int k) : c(ctx), kind(k) {
  if (!c->profiling || c->capturing || !((c->prof_mask >> k) & 1u)) return;
  start = prof_get_event(c);
  stop = prof_get_event(c);
  if (start && stop) (void)hipEventRecord(start, c->stream);
}

pg_prof_scope::~pg_prof_scope() {
  if (!c->profiling || !start || !stop) return;
  (void)hipEventRecord(stop, c->stream);
  c->prof_events[kind].emplace_back(start, stop);
}

pg_status pg_read_scalars(pg_ctx* c, int first, int count) {
  // the scalar block lives in mapped pinned host memory: kernels store into it directly, so reading it back
  // is one stream synchronisation (no copy kernel, no extra launch)
  (void)first;
  (void)count;
  if (c->capturing) return PG_OK;  // recorded, not run: the caller's scalars are placeholders
  PG_HIP(hipStreamSynchronize(c->stream));
  if (c->hscal[PG_S_TEAMERR] == 2.0) {  // column shards: some rank's sweep was refused at launch (pg_gemv.hip)
    c->hscal[PG_S_TEAMERR] = 0.0;
    pg_set_error("single-sweep pass: the sweep was refused at launch on one of the column shards; every rank leaves the single-sweep mode");
    return PG_ERR_UNSUPPORTED;
  }
  if (c->hscal[PG_S_TEAMERR] != 0.0) {  // a workgroup team of the long-column sweep gave up waiting (pg_gemv_tn2.hip)
    c->hscal[PG_S_TEAMERR] = 0.0;
    c->team_timeout = true;
    pg_set_error("single-sweep pass: a workgroup team timed out waiting for a member (device shared with another job?)");
    return PG_ERR_TIMEOUT;
  }
  return PG_OK;
}

// ---------------------------------------------------------------------------------------------
// matrix
// ---------------------------------------------------------------------------------------------

// murmur3 fmix32 -- must match oracle/proxgrad_oracle.py::_mix32 bit for bit
__device__ __forceinline__ uint32_t pg_mix32(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}

// centred sum of eight hashed 16-bit uniforms (Irwin-Hall(8)); oracle: counter_ih8
__device__ __forceinline__ int pg_ih8(uint32_t hseed, uint32_t i, uint32_t j) {
  uint32_t h = pg_mix32(hseed ^ j);
  h = pg_mix32(h ^ (i * 0x9E3779B1u));
  int s = 0;
#pragma unroll
  for (uint32_t t = 0; t < 4; ++t) {
    uint32_t w = pg_mix32(h + t * 0x632BE5ABu);
    s += (int)(w & 0xFFFFu) + (int)(w >> 16);
  }
  return s - 4 * 65535;
}

template <typename T>
__global__ __launch_bounds__(256) void generate_kernel(T* __restrict__ A, int64_t m, int64_t n, int64_t ld,
                                                       uint32_t hseed, uint32_t row_offset, uint32_t col_offset, float scale) {
  using V = typename VecOf<T>::type;
  constexpr int VEC = VecOf<T>::N;
  const int64_t vec_per_col = ld / VEC;
  const int64_t total = vec_per_col * n;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t j = idx / vec_per_col;
    const int64_t i0 = (idx - j * vec_per_col) * VEC;
    V v;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const int64_t i = i0 + e;
      float val = 0.0f;
      if (i < m) val = (float)pg_ih8(hseed, row_offset + (uint32_t)i, col_offset + (uint32_t)j) * scale;
      v[e] = (T)val;
    }
    *reinterpret_cast<V*>(A + j * ld + i0) = v;
  }
}

static inline uint32_t host_mix32(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}

// A[i, j] += alpha * u[i] * w[j] on the padded column-major store (rows >= m stay zero): the rank-one update of the
// Broyden operator (src/accel/broyden.jl:18-28).  HBM-bound: one read and one write of A, 16-byte accesses along columns.
template <typename T>
__global__ void rank1_update_kernel(T* __restrict__ A, int64_t m, int64_t n, int64_t ld, T alpha, const T* __restrict__ u,
                                    const T* __restrict__ w) {
  constexpr int V = 16 / (int)sizeof(T);
  const int64_t vec_per_col = ld / V;
  const int64_t total = vec_per_col * n;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = t / vec_per_col;
    const int64_t i0 = (t - j * vec_per_col) * V;
    if (i0 >= m) continue;
    const T s = alpha * w[j];
    T* p = A + j * ld + i0;
#pragma unroll
    for (int k = 0; k < V; ++k)
      if (i0 + k < m) p[k] += s * u[i0 + k];
  }
}

extern "C" {

pg_status pg_mat_create(pg_ctx* c, int32_t dtype, int64_t m, int64_t n, pg_mat** out) {
  PG_REQUIRE(c != nullptr && out != nullptr, "null argument");
  PG_REQUIRE(dtype == PG_F32 || dtype == PG_F64, "dtype must be PG_F32 or PG_F64");
  PG_REQUIRE(m >= 0 && n >= 0, "negative dimension");
  PG_REQUIRE(m < (int64_t)1 << 32 && n < (int64_t)1 << 32, "dimension >= 2^32");
  *out = nullptr;
  pg_mat* A = new pg_mat();
  A->ctx = c;
  A->dtype = dtype;
  A->m = m;
  A->n = n;
  const int64_t row_quantum = 1024 / (int64_t)pg_sizeof(dtype);  // one 64-lane x 16 B wave load
  A->ld = pg_round_up(m > 0 ? m : 1, row_quantum);
  const size_t bytes = (size_t)A->ld * (size_t)(n > 0 ? n : 1) * pg_sizeof(dtype);
  PG_HIP(hipSetDevice(c->device));
  hipError_t e = hipMalloc(&A->data, bytes);
  if (e != hipSuccess) {
    pg_set_error("hipMalloc(%zu bytes) for a %lld x %lld matrix failed: %s", bytes, (long long)m, (long long)n,
                 hipGetErrorString(e));
    delete A;
    return PG_ERR_ALLOC;
  }
  if (A->ld != m) {  // padding rows must read as zero
    e = hipMemsetAsync(A->data, 0, bytes, c->stream);
    if (e != hipSuccess) {
      (void)hipFree(A->data);
      delete A;
      pg_set_error("hipMemsetAsync failed: %s", hipGetErrorString(e));
      return PG_ERR_HIP;
    }
  }
  *out = A;
  return PG_OK;
}

pg_status pg_mat_destroy(pg_mat* A) {
  if (!A) return PG_OK;
  if (A->data) (void)hipFree(A->data);
  if (A->partials) (void)hipFree(A->partials);
  for (void* q : A->retired) (void)hipFree(q);
  if (A->rpad) (void)hipFree(A->rpad);
  if (A->rpad2) (void)hipFree(A->rpad2);
  if (A->rpad3) (void)hipFree(A->rpad3);
  if (A->xch) (void)hipFree(A->xch);
  delete A;
  return PG_OK;
}

// host <-> device copy of the matrix body: one linear copy per 1 GiB when both sides are dense in the leading dimension
// (the headline matrix is 64 GiB; a 2-D copy of 2^20 "rows" is needlessly slow), a 2-D copy otherwise
static pg_status mat_copy(pg_mat* A, void* dst, size_t dst_pitch, const void* src, size_t src_pitch, hipMemcpyKind kind) {
  const size_t es = pg_sizeof(A->dtype), width = (size_t)A->m * es;
  if (dst_pitch == width && src_pitch == width) {
    const size_t total = width * (size_t)A->n, chunk = (size_t)1 << 30;
    for (size_t off = 0; off < total; off += chunk) {
      const size_t nb = total - off < chunk ? total - off : chunk;
      PG_HIP(hipMemcpyAsync((char*)dst + off, (const char*)src + off, nb, kind, A->ctx->stream));
    }
    return PG_OK;
  }
  PG_HIP(hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, width, (size_t)A->n, kind, A->ctx->stream));
  return PG_OK;
}

pg_status pg_mat_upload(pg_mat* A, const void* host, int64_t ld_host) {
  PG_REQUIRE(A != nullptr, "matrix is null");
  if (A->m == 0 || A->n == 0) return PG_OK;
  PG_REQUIRE(host != nullptr, "host pointer is null");
  PG_REQUIRE(ld_host >= A->m, "ld_host < m");
  const size_t es = pg_sizeof(A->dtype);
  PG_TRY(mat_copy(A, A->data, (size_t)A->ld * es, host, (size_t)ld_host * es, hipMemcpyHostToDevice));
  PG_HIP(hipStreamSynchronize(A->ctx->stream));
  return PG_OK;
}

pg_status pg_mat_set_from_device(pg_mat* A, const void* dev, int64_t ld_dev) {
  PG_REQUIRE(A != nullptr, "matrix is null");
  if (A->m == 0 || A->n == 0) return PG_OK;
  PG_REQUIRE(dev != nullptr, "device pointer is null");
  PG_REQUIRE(ld_dev >= A->m, "ld_dev < m");
  const size_t es = pg_sizeof(A->dtype);
  PG_HIP(hipMemcpy2DAsync(A->data, (size_t)A->ld * es, dev, (size_t)ld_dev * es, (size_t)A->m * es, (size_t)A->n,
                          hipMemcpyDeviceToDevice, A->ctx->stream));
  return PG_OK;
}

pg_status pg_mat_download(pg_mat* A, void* host, int64_t ld_host) {
  PG_REQUIRE(A != nullptr, "matrix is null");
  if (A->m == 0 || A->n == 0) return PG_OK;
  PG_REQUIRE(host != nullptr, "host pointer is null");
  PG_REQUIRE(ld_host >= A->m, "ld_host < m");
  const size_t es = pg_sizeof(A->dtype);
  PG_TRY(mat_copy(A, host, (size_t)ld_host * es, A->data, (size_t)A->ld * es, hipMemcpyDeviceToHost));
  PG_HIP(hipStreamSynchronize(A->ctx->stream));
  return PG_OK;
}

pg_status pg_mat_generate_block(pg_mat* A, uint32_t seed, int64_t row_offset, int64_t col_offset, double scale) {
  PG_REQUIRE(A != nullptr, "matrix is null");
  PG_REQUIRE(row_offset >= 0 && row_offset + A->m <= (int64_t)1 << 32, "row_offset out of range");
  PG_REQUIRE(col_offset >= 0 && col_offset + A->n <= (int64_t)1 << 32, "col_offset out of range");
  if (A->m == 0 || A->n == 0) return PG_OK;
  const uint32_t hseed = host_mix32(seed ^ 0x9E3779B9u);
  const int64_t vecs = A->ld / (16 / (int64_t)pg_sizeof(A->dtype)) * A->n;
  int64_t blocks = (vecs + 255) / 256;
  const int64_t cap = (int64_t)A->ctx->num_cu * 32;
  if (blocks > cap) blocks = cap;
  if (A->dtype == PG_F32)
    hipLaunchKernelGGL(generate_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, A->ctx->stream, (float*)A->data, A->m,
                       A->n, A->ld, hseed, (uint32_t)row_offset, (uint32_t)col_offset, (float)scale);
  else
    hipLaunchKernelGGL(generate_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, A->ctx->stream, (double*)A->data, A->m,
                       A->n, A->ld, hseed, (uint32_t)row_offset, (uint32_t)col_offset, (float)scale);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

pg_status pg_mat_rank1_update(pg_mat* A, double alpha, const void* u, const void* w) {
  PG_REQUIRE(A != nullptr && u != nullptr && w != nullptr, "null argument");
  if (A->m == 0 || A->n == 0) return PG_OK;
  const int64_t vecs = A->ld / (16 / (int64_t)pg_sizeof(A->dtype)) * A->n;
  int64_t blocks = (vecs + 255) / 256;
  const int64_t cap = (int64_t)A->ctx->num_cu * 32;
  if (blocks > cap) blocks = cap;
  if (A->dtype == PG_F32)
    hipLaunchKernelGGL(rank1_update_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, A->ctx->stream, (float*)A->data, A->m,
                       A->n, A->ld, (float)alpha, (const float*)u, (const float*)w);
  else
    hipLaunchKernelGGL(rank1_update_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, A->ctx->stream, (double*)A->data, A->m,
                       A->n, A->ld, alpha, (const double*)u, (const double*)w);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

pg_status pg_mat_generate(pg_mat* A, uint32_t seed, int64_t row_offset, double scale) {
  return pg_mat_generate_block(A, seed, row_offset, 0, scale);
}

pg_status pg_mat_info(const pg_mat* A, int64_t* m, int64_t* n, int64_t* ld, int32_t* dtype, void** dptr) {
  PG_REQUIRE(A != nullptr, "matrix is null");
  if (m) *m = A->m;
  if (n) *n = A->n;
  if (ld) *ld = A->ld;
  if (dtype) *dtype = A->dtype;
  if (dptr) *dptr = A->data;
  return PG_OK;
}

}  // extern "C"
