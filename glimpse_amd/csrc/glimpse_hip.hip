// glimpse_hip.hip -- C ABI (include/glimpse_hip.h) over the kernels in glh_kernels.h.
// Host side: context, HBM residency, launches on one HIP stream, stage timers.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/glimpse_hip.h"
#include "glh_comm.h"
#include "glh_host.h"
#include "glh_kernels.h"
#include "glh_point.h"
#include "glh_point_variants.h"

using namespace glh;

// The fused kernel's instantiations live in their own translation units (glh_point_inst.hip, glh_point_variants.h).
namespace glh {
#define GLH_PT_DECL(TB, PPT, NOBS, S, F, C) const void* GLH_PT_NAME(TB, PPT, NOBS, S, F, C)();
#define GLH_PT_DECL_SHAPE(TB, PPT, NOBS) GLH_PT_CODES(GLH_PT_DECL, TB, PPT, NOBS)
GLH_PT_SHAPES(GLH_PT_DECL_SHAPE)
#undef GLH_PT_DECL_SHAPE
#undef GLH_PT_DECL
}  // namespace glh

static const void* pt_kernel(int tb, int ppt, int nobs, int surf, bool fast, bool con) {
#define GLH_PT_PICK(TB, PPT, NOBS, S, F, C) \
  if (tb == TB && ppt == PPT && nobs == NOBS && surf == S && fast == (bool)F && con == (bool)C) \
    return GLH_PT_NAME(TB, PPT, NOBS, S, F, C)();
#define GLH_PT_PICK_SHAPE(TB, PPT, NOBS) GLH_PT_CODES(GLH_PT_PICK, TB, PPT, NOBS)
  GLH_PT_SHAPES(GLH_PT_PICK_SHAPE)
#undef GLH_PT_PICK_SHAPE
#undef GLH_PT_PICK
  return nullptr;
}

// (every instantiation with its surface code: code 2 holds the raster windows in static LDS)
static std::vector<std::pair<const void*, int>> pt_all_kernels() {
  std::vector<std::pair<const void*, int>> v;
#define GLH_PT_PUSH(TB, PPT, NOBS, S, F, C) v.push_back({GLH_PT_NAME(TB, PPT, NOBS, S, F, C)(), S});
#define GLH_PT_PUSH_SHAPE(TB, PPT, NOBS) GLH_PT_CODES(GLH_PT_PUSH, TB, PPT, NOBS)
  GLH_PT_SHAPES(GLH_PT_PUSH_SHAPE)
#undef GLH_PT_PUSH_SHAPE
#undef GLH_PT_PUSH
  return v;
}

// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIPCHK(expr)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return fail(GLH_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                  __LINE__);                                                                 \
  } while (0)

#define CHK(expr)          \
  do {                     \
    int rc_ = (expr);      \
    if (rc_ != GLH_OK) return rc_; \
  } while (0)

// LDS plan of the fused kernel (glh_point.h): c[N] + region 2
constexpr int PT_LDS_MAX = 152 * 1024;   // dynamic LDS of one workgroup (static <= 5 KB on top, 160 KB per CU)
constexpr int PT_LDS_HALF = 75 * 1024;   // dynamic LDS that still lets two workgroups share a CU
constexpr int PT_PATCH_LDS = 3328;       // static LDS of the raster windows (code 2: glh_point.h, PtPatches), taken off both
static_assert(PT_PATCH_LDS >= 2 * (int)sizeof(RasterPatch), "the raster windows fit their share");

// ------------------------------------------------------------------------------------------
// stages (for the event timers)
// ------------------------------------------------------------------------------------------
enum Stage {
  ST_INIT = 0,
  ST_EVOLVE_PROJECT,
  ST_MOMENTS,
  ST_TEMPLATE,
  ST_TILEPREP,
  ST_SSD,
  ST_SPLINE_FIT,
  ST_WEIGHTS,
  ST_RESAMPLE,
  ST_POINT_STEP,
  ST_COUNT
};
static const char* kStageNames[ST_COUNT] = {"init_particles", "evolve_project", "moments",
                                            "template_init",  "tileprep",       "ssd",
                                            "spline_fit",     "weights",        "resample",
                                            "point_step"};

// ------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------
struct Observer {
  int n_images = 0, width = 0, height = 0, channels = 0;
  int bits = 8;  // bits per sample of the frames: 8 or 16 (unsigned integers), 32 (float32), 64 (float64): glh_observer_set_depth
  size_t frame_bytes() const { return (size_t)width * height * channels * (bits / 8); }
  double sigma = 0.3;
  CamDev* cams = nullptr;               // device [n_images]
  std::vector<CamDev> cams_host;        // same, for kernels that take the camera by value
  std::vector<const uint8_t*> frames;   // device pointers (owned or borrowed)
  std::vector<uint8_t*> owned;          // owned allocations (same indexing; null if borrowed)
};

struct glh_ctx {
  glh_config cfg{};
  hipStream_t stream = nullptr;
  // frame ingest (glh_observer_upload_frame_async): a copy stream and a ring of pinned staging buffers
  static constexpr int NSTAGE = 4;
  hipStream_t copy_stream = nullptr;
  uint8_t* stage[NSTAGE] = {nullptr, nullptr, nullptr, nullptr};
  size_t stage_bytes[NSTAGE] = {0, 0, 0, 0};
  hipEvent_t stage_done[NSTAGE] = {nullptr, nullptr, nullptr, nullptr};
  bool stage_busy[NSTAGE] = {false, false, false, false};
  int stage_next = 0;
  // uploads from caller-registered host memory (glh_observer_upload_frame_pinned): ticket t has event pin_ev[t % NPIN]
  static constexpr int NPIN = 64;
  hipEvent_t pin_ev[NPIN] = {nullptr};
  int64_t pin_next = 0, pin_completed = 0;  // tickets handed out; every ticket below pin_completed has finished
  hipEvent_t upload_done = nullptr;  // recorded on copy_stream after the latest upload
  bool uploads_pending = false;      // the compute stream has not yet been ordered after upload_done
  int P = 0, N = 0, tw = 0, th = 0, NB = 0;
  int cur = 0;  // current particle/weight buffer
  int frame = 0;
  bool have_mask = false, have_active = false, keep_sse = false, keep_idx = false, has_dem = false;
  int fused = 1;    // glh_step: 0 staged kernels, 1 fused per-point kernel, 2 fused with tiles forced to HBM (test)
  int pt_base = 0;  // global index of this context's point 0
  bool all_cartesian = true;  // every point is CartesianMotion (the common fused instantiation; else the general one)
  bool fast_math = false;     // GLH_MATH_FAST (glh_set_math)
  int hp_rx = 2, hp_ry = 2;   // half sizes of the median high-pass window (glh_set_highpass; 5 x 5 by default)
  int tile_cap = 0, search_cap = 0, sse_cap = 0, ssd_blocks = 2;
  Observer obs[MAX_OBS];
  // device buffers
  double *particles[2] = {nullptr, nullptr}, *weights[2] = {nullptr, nullptr};
  double *motion = nullptr, *uv = nullptr, *bbox_part = nullptr, *normals = nullptr, *u = nullptr;
  double *mean6 = nullptr, *moments = nullptr;
  struct RasterBuf {
    double *z = nullptr, *gx = nullptr, *gy = nullptr;
    RasterDev dev{};
    std::vector<double> hgx, hgy;  // host copies of the coordinates (same_grid below)
  } rasters[3];  // GLH_RASTER_DEM, _DEM_SIGMA, _VIEWSHED
  bool same_grid = false;  // the dem and dem_sigma rasters have bit-identical coordinate arrays (Surfaces::same_grid)
  double* covariances = nullptr;  // [max_frames][P][36], allocated on first use
  double* uj = nullptr;           // [P][N] host-fed per-particle uniforms (stratified / choice)
  double* extra_ll = nullptr;     // [P][N] caller-computed log-likelihood term of the next glh_update_weights, or null
  bool have_extra = false;
  uint8_t *obs_mask = nullptr, *active = nullptr;
  uint32_t* pt_status = nullptr;
  int32_t *pt_err_frame = nullptr, *box = nullptr, *idx = nullptr;
  int32_t* obs_status_all = nullptr;  // [max_frames][O][P]: the per-(observer, point) status of every frame
  int32_t *tmpl_box = nullptr, *tmpl_hist_n = nullptr, *tmpl_valid = nullptr;
  double *tmpl_duv = nullptr, *tmpl_tile64 = nullptr, *tmpl_hist_v = nullptr, *tmpl_hist_q = nullptr;
  float *tmpl_tile32 = nullptr, *search = nullptr;
  unsigned long long* stamps = nullptr;  // phase stamps of the fused kernel (diagnostic)
  uint16_t* uidx[2] = {nullptr, nullptr};  // [P][N] record of every particle in particles[b] / weights[b] when compact
  bool compact = false;  // particles[cur] / weights[cur] are run-length compact (left by the fused step)
  int32_t* resid_draws = nullptr;  // [P] uniforms consumed by the last residual resampling
  uint32_t* bins16 = nullptr;   // [P][65535 * 3 + 1] key histograms of 16-bit frames (staged tile kernels), on first use
  double* fwork = nullptr;      // [P][2 D^2] tile workspace of float64 frames (staged tile kernels), on first use
  double* tracks_tmp = nullptr; // means | sigmas in the caller's layout (glh_get_tracks), on first use
  size_t tracks_tmp_n = 0;
  uint16_t* ws_keys = nullptr;  // raw-key workspace of the fused kernel for tiles that do not fit in LDS
  int keys_cap = 0;
  int plan_N = 0;  // the N the pairwise-sum plan on the device was made for (0: none)
  bool track_covariances = false;  // glh_track_covariances
  // glh_track on two streams (glh_set_track_streams): 0 = automatic (two when the batch is at least two rounds of
  // workgroups per half), 1 = always one, 2 = two whenever the fused step runs
  int track_streams = 0;
  hipStream_t extra_streams[3] = {nullptr, nullptr, nullptr};  // streams 2 .. 4 of glh_track
  hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
  int last_track_streams = 1;  // streams the last glh_track used
  int n_cus = 256;             // compute units of the device (hipDeviceProp: glh_create)
  bool capturing = false;      // glh_track is recording its frame loop into a hipGraph (no event timers meanwhile)
  hipGraphExec_t track_graph = nullptr;  // the last captured frame loop (kept until the next one or the context's end)
  int force_tb = 0;            // experiment (GLH_PT_BIG_FRAMES): this launch runs the 1 024-thread instantiation
  double *sse = nullptr, *sse_copy = nullptr, *ll_dbg = nullptr;
  double* lu = nullptr;
  double* poly = nullptr;
  int64_t* lu_off = nullptr;
  double* spl_inv = nullptr;  // explicit inverses of the collocation matrices, sides 4 .. GLH_SPL_DENSE_MAX
  int32_t *leaf_off = nullptr, *leaf_len = nullptr, *sum_ops = nullptr, *level_off = nullptr, *roots = nullptr;
  int nleaves = 0, nnodes = 0, nlevels = 0, nroots = 0;
  int moments_frame = -1;  // history slot already filled by the fused resample kernel
  int interp_k = 3;  // interpolation order of the surface sampling: 3 (bicubic, the default), 1 (bilinear), or 0: the
                     // general orders interp_kx (rows axis) / interp_ky (columns axis) (glh_set_interpolation)
  int interp_kx = 3, interp_ky = 3;
  double* glu[2] = {nullptr, nullptr};       // general orders: spline_lu_general factors for degree kx / ky by size
  int64_t* glu_off[2] = {nullptr, nullptr};  // [max_search_dim + 1]
  int32_t last_variant[4] = {0, 0, 0, 0};  // fused kernel instantiation of the last step: TB, PPT, NOBS, fast | general << 1
  size_t normals_cap = 0;
  // profiling
  bool profiling = false;
  struct Ev {
    hipEvent_t a, b;
    int stage;
  };
  std::vector<Ev> pending;
  std::vector<hipEvent_t> pool;
  double ms[ST_COUNT] = {0};
  int64_t launches[ST_COUNT] = {0};
  std::vector<float> launch_ms[ST_COUNT];  // duration of every timed launch since the last reset (bounded)
  // GPU time a stage spans since the last reset: from the start of its first timed launch to the end of its last one,
  // on whichever stream (launches of glh_track's two streams overlap: their durations do not add up to it)
  hipEvent_t span_a[ST_COUNT] = {nullptr};
  float span_ms[ST_COUNT] = {0};
  // multi-GPU (glh_comm.h)
  Comm* comm = nullptr;
};

// status words [O][P] of the current frame
static int32_t* cur_status(const glh_ctx* c) {
  return c->obs_status_all + (size_t)c->frame * c->cfg.n_observers * c->P;
}

template <typename T>
static int dalloc(T** p, size_t count) {
  *p = nullptr;
  if (count == 0) count = 1;
  hipError_t e = hipMalloc((void**)p, count * sizeof(T));
  if (e != hipSuccess)
    return fail(GLH_E_NOMEM, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
  return GLH_OK;
}
template <typename T>
static void dfree(T*& p) {
  if (p) (void)hipFree((void*)p);
  p = nullptr;
}

struct StageTimer {
  glh_ctx* c;
  int stage;
  hipEvent_t a = nullptr, b = nullptr;
  hipStream_t s;
  StageTimer(glh_ctx* ctx, int st, hipStream_t on = nullptr) : c(ctx), stage(st), s(on ? on : ctx->stream) {
    c->launches[st]++;
    if (!c->profiling || c->capturing) return;
    auto get = [&]() {
      hipEvent_t e;
      if (!c->pool.empty()) {
        e = c->pool.back();
        c->pool.pop_back();
      } else {
        (void)hipEventCreate(&e);
      }
      return e;
    };
    a = get();
    b = get();
    (void)hipEventRecord(a, s);
  }
  ~StageTimer() {
    if (!a) return;
    (void)hipEventRecord(b, s);
    c->pending.push_back({a, b, stage});
  }
};

static int drain_profile(glh_ctx* c) {
  if (c->pending.empty()) return GLH_OK;
  HIPCHK(hipStreamSynchronize(c->stream));
  for (auto& e : c->pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
      c->ms[e.stage] += ms;
      if (c->launch_ms[e.stage].size() < (1u << 16)) c->launch_ms[e.stage].push_back(ms);
    }
    if (!c->span_a[e.stage]) {
      c->span_a[e.stage] = e.a;  // (kept until the next reset)
    }
    float t = 0.f;
    if (hipEventElapsedTime(&t, c->span_a[e.stage], e.b) == hipSuccess && t > c->span_ms[e.stage]) c->span_ms[e.stage] = t;
    if (c->span_a[e.stage] != e.a) c->pool.push_back(e.a);
    c->pool.push_back(e.b);
  }
  c->pending.clear();
  return GLH_OK;
}

// ------------------------------------------------------------------------------------------
// library / context
// ------------------------------------------------------------------------------------------
extern "C" int glh_version(void) { return GLH_VERSION; }
extern "C" const char* glh_last_error(void) { return g_err.c_str(); }
extern "C" int glh_stage_count(void) { return ST_COUNT; }
extern "C" const char* glh_stage_name(int s) { return (s >= 0 && s < ST_COUNT) ? kStageNames[s] : ""; }

extern "C" int glh_device_count(int* count) {
  if (!count) return fail(GLH_E_INVALID, "count is null");
  HIPCHK(hipGetDeviceCount(count));
  return GLH_OK;
}

extern "C" int glh_device_memory(int device_id, uint64_t* free_bytes, uint64_t* total_bytes) {
  if (!free_bytes || !total_bytes) return fail(GLH_E_INVALID, "null argument");
  HIPCHK(hipSetDevice(device_id));
  size_t f = 0, t = 0;
  HIPCHK(hipMemGetInfo(&f, &t));
  *free_bytes = (uint64_t)f;
  *total_bytes = (uint64_t)t;
  return GLH_OK;
}

extern "C" int glh_device_compute_units(int device_id, int* count) {
  if (!count) return fail(GLH_E_INVALID, "count is null");
  HIPCHK(hipDeviceGetAttribute(count, hipDeviceAttributeMultiprocessorCount, device_id));
  return GLH_OK;
}

extern "C" int glh_destroy(glh_ctx* c) {
  if (!c) return GLH_OK;
  (void)hipSetDevice(c->cfg.device_id);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) (void)glh_comm_destroy(c);
  if (c->track_graph) (void)hipGraphExecDestroy(c->track_graph);
  if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
  for (auto e : c->pin_ev)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : c->pending) {
    (void)hipEventDestroy(e.a);
    (void)hipEventDestroy(e.b);
  }
  for (auto e : c->pool) (void)hipEventDestroy(e);
  for (int i = 0; i < ST_COUNT; ++i)
    if (c->span_a[i]) (void)hipEventDestroy(c->span_a[i]);  // (kept out of the pool since the last reset)
  for (int q = 0; q < 3; ++q)
    if (c->extra_streams[q]) (void)hipStreamSynchronize(c->extra_streams[q]);
  for (int o = 0; o < MAX_OBS; ++o) {
    dfree(c->obs[o].cams);
    for (auto& p : c->obs[o].owned) dfree(p);
  }
  for (int i = 0; i < 2; ++i) {
    dfree(c->particles[i]);
    dfree(c->weights[i]);
  }
  dfree(c->motion); dfree(c->uv); dfree(c->bbox_part); dfree(c->normals); dfree(c->u);
  dfree(c->mean6); dfree(c->moments); dfree(c->covariances); dfree(c->uj); dfree(c->extra_ll);
  for (auto& r : c->rasters) {
    dfree(r.z); dfree(r.gx); dfree(r.gy);
  } dfree(c->obs_mask); dfree(c->active); dfree(c->pt_status);
  dfree(c->pt_err_frame); dfree(c->obs_status_all); dfree(c->box); dfree(c->idx); dfree(c->tmpl_box);
  dfree(c->tmpl_hist_n); dfree(c->tmpl_valid); dfree(c->tmpl_duv); dfree(c->tmpl_tile64);
  dfree(c->tmpl_hist_v); dfree(c->tmpl_hist_q); dfree(c->tmpl_tile32); dfree(c->search); dfree(c->ws_keys); dfree(c->bins16); dfree(c->fwork); dfree(c->tracks_tmp); dfree(c->resid_draws); dfree(c->uidx[0]); dfree(c->uidx[1]); dfree(c->stamps);
  dfree(c->sse); dfree(c->sse_copy); dfree(c->ll_dbg); dfree(c->lu); dfree(c->poly); dfree(c->lu_off); dfree(c->spl_inv); dfree(c->glu[0]); dfree(c->glu[1]); dfree(c->glu_off[0]); dfree(c->glu_off[1]); dfree(c->leaf_off);
  dfree(c->leaf_len); dfree(c->sum_ops); dfree(c->level_off); dfree(c->roots);
  if (c->copy_stream) {
    (void)hipStreamSynchronize(c->copy_stream);
    (void)hipStreamDestroy(c->copy_stream);
  }
  for (int k = 0; k < glh_ctx::NSTAGE; ++k) {
    if (c->stage[k]) (void)hipHostFree(c->stage[k]);
    if (c->stage_done[k]) (void)hipEventDestroy(c->stage_done[k]);
  }
  if (c->upload_done) (void)hipEventDestroy(c->upload_done);
  for (int q = 0; q < 3; ++q) {
    if (c->extra_streams[q]) (void)hipStreamDestroy(c->extra_streams[q]);
    if (c->ev_join[q]) (void)hipEventDestroy(c->ev_join[q]);
  }
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return GLH_OK;
}

extern "C" int glh_create(const glh_config* cfg, glh_ctx** out) {
  if (!cfg || !out) return fail(GLH_E_INVALID, "null argument");
  *out = nullptr;
  glh_config k = *cfg;
  if (k.max_tile <= 0) k.max_tile = 31;
  if (k.max_search_dim <= 0) k.max_search_dim = 320;
  if (k.max_frames <= 0) k.max_frames = 128;
  if (k.max_points <= 0 || k.max_particles <= 0 || k.n_observers <= 0 || k.n_observers > MAX_OBS)
    return fail(GLH_E_INVALID, "max_points/max_particles must be > 0 and 1 <= n_observers <= %d", MAX_OBS);
  if (k.max_points > 65535)  // the staged kernels put the point index in gridDim.y
    return fail(GLH_E_UNSUPPORTED, "max_points %d exceeds 65535 points per context: shard the points", k.max_points);
  if (k.max_tile < 5 || k.max_tile > 127) return fail(GLH_E_INVALID, "max_tile must be in [5, 127]");
  if (k.max_search_dim < k.max_tile + 3 || k.max_search_dim > 2000)
    return fail(GLH_E_INVALID, "max_search_dim must be in [max_tile + 3, 2000]");
  if ((size_t)k.max_particles * 10 + 4096 > 150 * 1024)
    return fail(GLH_E_UNSUPPORTED, "max_particles %d exceeds the LDS-resident scan (<= 14900)", k.max_particles);
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (k.device_id < 0 || k.device_id >= ndev)
    return fail(GLH_E_INVALID, "device_id %d out of range (%d devices)", k.device_id, ndev);
  HIPCHK(hipSetDevice(k.device_id));
  glh_ctx* c = new (std::nothrow) glh_ctx();
  if (!c) return fail(GLH_E_NOMEM, "out of host memory");
  c->cfg = k;
  {
    int cus = 0;  // (what decides whether a batch is two rounds of workgroups per half: glh_track's two streams)
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, k.device_id) == hipSuccess && cus > 0) c->n_cus = cus;
  }
  const size_t P = k.max_points, N = k.max_particles, O = k.n_observers;
  c->tile_cap = k.max_tile * k.max_tile;
  c->search_cap = k.max_search_dim * (k.max_search_dim + 16);  // rows padded for the fused kernel
  c->keys_cap = pt_keys_count(k.max_search_dim, k.max_search_dim);  // with the reflected border
  c->sse_cap = c->search_cap;
  const size_t NBmax = (N + BLK - 1) / BLK;
  int rc = GLH_OK;
  auto A = [&](int r) {
    if (rc == GLH_OK) rc = r;
  };
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete c;
    return fail(GLH_E_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
  }
  for (int i = 0; i < 2; ++i) {
    A(dalloc(&c->particles[i], P * N * 6));
    A(dalloc(&c->weights[i], P * N));
  }
  A(dalloc(&c->motion, P * GLH_MOTION_FULL_LEN));
  A(dalloc(&c->uv, O * P * N * 2));
  A(dalloc(&c->bbox_part, O * P * NBmax * 5));
  A(dalloc(&c->u, P));
  A(dalloc(&c->mean6, P * 6));
  A(dalloc(&c->moments, (size_t)k.max_frames * P * 12));
  A(dalloc(&c->obs_mask, P * O));
  A(dalloc(&c->active, P));
  A(dalloc(&c->pt_status, P));
  A(dalloc(&c->pt_err_frame, P));
  A(dalloc(&c->obs_status_all, (size_t)k.max_frames * O * P));
  A(dalloc(&c->box, O * P * 4));
  A(dalloc(&c->tmpl_box, O * P * 4));
  A(dalloc(&c->tmpl_hist_n, O * P));
  A(dalloc(&c->tmpl_valid, O * P));
  A(dalloc(&c->tmpl_duv, O * P * 2));
  A(dalloc(&c->tmpl_tile64, O * P * c->tile_cap));
  A(dalloc(&c->tmpl_tile32, O * P * c->tile_cap));
  A(dalloc(&c->tmpl_hist_v, O * P * c->tile_cap));
  A(dalloc(&c->tmpl_hist_q, O * P * c->tile_cap));
  A(dalloc(&c->search, O * P * (size_t)c->search_cap));
  A(dalloc(&c->ws_keys, O * P * (size_t)c->keys_cap));
  A(dalloc(&c->sse, O * P * (size_t)c->sse_cap));
  // spline LU table for every surface side 4..max_search_dim
  if (rc == GLH_OK) {
    const int maxn = k.max_search_dim;
    std::vector<int64_t> off(maxn + 1, 0);
    int64_t total = 0;
    for (int n = 4; n <= maxn; ++n) {
      off[n] = total;
      total += 5 * (int64_t)n;
    }
    std::vector<double> lu((size_t)total);
    for (int n = 4; n <= maxn; ++n) spline_lu(n, lu.data() + off[n]);
    A(dalloc(&c->lu, (size_t)total));
    A(dalloc(&c->lu_off, (size_t)maxn + 1));
    if (rc == GLH_OK) {
      if (hipMemcpy(c->lu, lu.data(), total * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
          hipMemcpy(c->lu_off, off.data(), (maxn + 1) * sizeof(int64_t), hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(GLH_E_HIP, "upload of the spline LU table failed");
    }
  }
  if (rc == GLH_OK) {
    std::vector<double> inv((size_t)spline_inverse_off(GLH_SPL_DENSE_MAX + 1));
    for (int n = 4; n <= GLH_SPL_DENSE_MAX; ++n) spline_inverse(n, inv.data() + spline_inverse_off(n));
    A(dalloc(&c->spl_inv, inv.size()));
    if (rc == GLH_OK && hipMemcpy(c->spl_inv, inv.data(), inv.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
      rc = fail(GLH_E_HIP, "upload of the spline inverse table failed");
  }
  if (rc == GLH_OK) {
    std::vector<double> tab(16 * GLH_NPOLY);
    basis_poly_table(tab.data());
    A(dalloc(&c->poly, tab.size()));
    if (rc == GLH_OK && hipMemcpy(c->poly, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
      rc = fail(GLH_E_HIP, "upload of the spline basis table failed");
  }
  if (rc == GLH_OK) {
    // kernels that may use more than the default 64 KB of dynamic LDS
    hipError_t e1 = hipFuncSetAttribute((const void*)k_resample, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    hipError_t e2 = hipFuncSetAttribute((const void*)k_ssd, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipError_t e3 = hipFuncSetAttribute((const void*)k_tileprep, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipError_t e4 = hipSuccess;
    for (const auto& f : pt_all_kernels()) {
      hipError_t e = hipFuncSetAttribute(f.first, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         PT_LDS_MAX - (f.second == 2 ? PT_PATCH_LDS : 0));
      if (e != hipSuccess) e4 = e;
    }
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess)
      rc = fail(GLH_E_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
  }
  if (rc != GLH_OK) {
    std::string keep = g_err;
    glh_destroy(c);
    g_err = keep;
    return rc;
  }
  *out = c;
  return GLH_OK;
}

// Order the compute stream after the frame uploads enqueued so far (no host wait).
static int join_uploads(glh_ctx* c) {
  if (c->uploads_pending) {
    HIPCHK(hipStreamWaitEvent(c->stream, c->upload_done, 0));
    c->uploads_pending = false;
  }
  return GLH_OK;
}

extern "C" int glh_sync(glh_ctx* c) {
  if (!c) return fail(GLH_E_INVALID, "null context");
  CHK(join_uploads(c));
  HIPCHK(hipStreamSynchronize(c->stream));
  return GLH_OK;
}

extern "C" int glh_get_stream(glh_ctx* c, void** stream) {
  if (!c || !stream) return fail(GLH_E_INVALID, "null argument");
  *stream = (void*)c->stream;
  return GLH_OK;
}

// ------------------------------------------------------------------------------------------
// observers
// ------------------------------------------------------------------------------------------
static int check_obs(glh_ctx* c, int o) {
  if (!c) return fail(GLH_E_INVALID, "null context");
  if (o < 0 || o >= c->cfg.n_observers) return fail(GLH_E_INVALID, "observer %d out of range", o);
  return GLH_OK;
}

extern "C" int glh_observer_init(glh_ctx* c, int o, int n_images, int width, int height, int channels,
                                 double sigma) {
  CHK(check_obs(c, o));
  if (n_images <= 0 || width <= 0 || height <= 0) return fail(GLH_E_INVALID, "bad observer geometry");
  if (channels != 1 && channels != 3)
    return fail(GLH_E_UNSUPPORTED, "frames must be uint8 with 1 or 3 channels (got %d)", channels);
  if (!(sigma > 0.0)) return fail(GLH_E_INVALID, "sigma must be > 0");
  HIPCHK(hipSetDevice(c->cfg.device_id));
  Observer& ob = c->obs[o];
  dfree(ob.cams);
  for (auto& p : ob.owned) dfree(p);
  ob.n_images = n_images;
  ob.width = width;
  ob.height = height;
  ob.channels = channels;
  ob.bits = 8;
  ob.sigma = sigma;
  ob.frames.assign(n_images, nullptr);
  ob.owned.assign(n_images, nullptr);
  CHK(dalloc(&ob.cams, (size_t)n_images));
  return GLH_OK;
}

extern "C" int glh_observer_set_depth(glh_ctx* c, int o, int bits) {
  CHK(check_obs(c, o));
  if (bits != 8 && bits != 16 && bits != 32 && bits != 64)
    return fail(GLH_E_UNSUPPORTED, "frames are 8 or 16 bits per sample (unsigned), 32 = float32 or 64 = float64 samples "
                                   "(got %d)", bits);
  Observer& ob = c->obs[o];
  if (ob.n_images <= 0) return fail(GLH_E_STATE, "glh_observer_init first");
  if (bits >= 32 && ob.channels != 1 && ob.channels != 3)
    return fail(GLH_E_UNSUPPORTED, "float frames have one or three channels (observer %d has %d)", o, ob.channels);
  for (auto& p : ob.owned)
    if (p) return fail(GLH_E_STATE, "observer %d: set the depth before uploading frames", o);
  // what the wider tile kernels are sized for (their LDS requests grow with the context's limits)
  if (bits == 16 && (size_t)(BAND_H + 6) * c->cfg.max_search_dim * 4 > 96 * 1024)
    return fail(GLH_E_UNSUPPORTED, "16-bit frames: max_search_dim %d exceeds %d (one median band must fit 96 KiB of LDS)",
                c->cfg.max_search_dim, (int)(96 * 1024 / ((BAND_H + 6) * 4)));
  if (bits >= 32 && (size_t)c->cfg.max_tile * c->cfg.max_tile * 12 > 64 * 1024)
    return fail(GLH_E_UNSUPPORTED, "float frames: max_tile %d exceeds 73 (template workspace of 64 KiB of LDS)",
                c->cfg.max_tile);
  ob.bits = bits;
  return GLH_OK;
}

extern "C" int glh_observer_set_cameras(glh_ctx* c, int o, int first, int n, const double* cams) {
  CHK(check_obs(c, o));
  Observer& ob = c->obs[o];
  if (!cams || first < 0 || n <= 0 || first + n > ob.n_images)
    return fail(GLH_E_INVALID, "camera range [%d, %d) outside the observer's %d images", first, first + n, ob.n_images);
  std::vector<CamDev> tmp(n);
  for (int i = 0; i < n; ++i) {
    expand_camera(cams + (size_t)i * GLH_CAM_LEN, &tmp[i]);
    if ((int)tmp[i].imgsz[0] != ob.width || (int)tmp[i].imgsz[1] != ob.height)
      return fail(GLH_E_INVALID, "camera imgsz (%g, %g) != frame size (%d, %d): resized reads are out of scope",
                  tmp[i].imgsz[0], tmp[i].imgsz[1], ob.width, ob.height);
  }
  HIPCHK(hipSetDevice(c->cfg.device_id));
  if ((int)ob.cams_host.size() != ob.n_images) ob.cams_host.resize(ob.n_images);
  for (int i = 0; i < n; ++i) ob.cams_host[first + i] = tmp[i];
  HIPCHK(hipMemcpyAsync(ob.cams + first, tmp.data(), n * sizeof(CamDev), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return GLH_OK;
}

extern "C" int glh_observer_upload_frame(glh_ctx* c, int o, int image, const uint8_t* pixels) {
  CHK(check_obs(c, o));
  Observer& ob = c->obs[o];
  if (!pixels || image < 0 || image >= ob.n_images) return fail(GLH_E_INVALID, "bad frame index %d", image);
  HIPCHK(hipSetDevice(c->cfg.device_id));
  size_t bytes = ob.frame_bytes();
  if (!ob.owned[image]) CHK(dalloc(&ob.owned[image], bytes));
  HIPCHK(hipMemcpyAsync(ob.owned[image], pixels, bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  ob.frames[image] = ob.owned[image];
  return GLH_OK;
}

// The same upload without waiting for the device: `pixels` is copied into a pinned staging buffer before the call
// returns (the caller may reuse it at once), the host-to-device copy runs on a copy stream, and the next call that
// reads frames is ordered after it on the device.  A decoder pool can so keep the PCIe link busy while it decodes.
extern "C" int glh_observer_upload_frame_async(glh_ctx* c, int o, int image, const uint8_t* pixels) {
  CHK(check_obs(c, o));
  Observer& ob = c->obs[o];
  if (!pixels || image < 0 || image >= ob.n_images) return fail(GLH_E_INVALID, "bad frame index %d", image);
  HIPCHK(hipSetDevice(c->cfg.device_id));
  const size_t bytes = ob.frame_bytes();
  if (!c->copy_stream) {
    HIPCHK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&c->upload_done, hipEventDisableTiming));
  }
  const int k = c->stage_next;
  c->stage_next = (k + 1) % glh_ctx::NSTAGE;
  if (c->stage_busy[k]) {  // the copy that last used this slot has to be over before it is overwritten
    HIPCHK(hipEventSynchronize(c->stage_done[k]));
    c->stage_busy[k] = false;
  }
  if (c->stage_bytes[k] < bytes) {
    if (c->stage[k]) HIPCHK(hipHostFree(c->stage[k]));
    c->stage[k] = nullptr;
    c->stage_bytes[k] = 0;
    HIPCHK(hipHostMalloc((void**)&c->stage[k], bytes, hipHostMallocDefault));
    c->stage_bytes[k] = bytes;
  }
  if (!c->stage_done[k]) HIPCHK(hipEventCreateWithFlags(&c->stage_done[k], hipEventDisableTiming));
  memcpy(c->stage[k], pixels, bytes);
  if (!ob.owned[image]) CHK(dalloc(&ob.owned[image], bytes));
  HIPCHK(hipMemcpyAsync(ob.owned[image], c->stage[k], bytes, hipMemcpyHostToDevice, c->copy_stream));
  HIPCHK(hipEventRecord(c->stage_done[k], c->copy_stream));
  HIPCHK(hipEventRecord(c->upload_done, c->copy_stream));
  c->stage_busy[k] = true;
  c->uploads_pending = true;
  ob.frames[image] = ob.owned[image];
  return GLH_OK;
}

extern "C" int glh_host_register(void* ptr, uint64_t bytes) {
  if (!ptr || bytes == 0) return fail(GLH_E_INVALID, "null buffer");
  HIPCHK(hipHostRegister(ptr, (size_t)bytes, hipHostRegisterPortable));  // (for every device of the process)
  return GLH_OK;
}

extern "C" int glh_host_unregister(void* ptr) {
  if (!ptr) return fail(GLH_E_INVALID, "null buffer");
  HIPCHK(hipHostUnregister(ptr));
  return GLH_OK;
}

// advance pin_completed over every ticket whose event has passed (in order: the copy stream runs them in order)
static int pin_poll(glh_ctx* c, int64_t wait_for) {
  while (c->pin_completed < c->pin_next) {
    hipEvent_t e = c->pin_ev[c->pin_completed % glh_ctx::NPIN];
    if (c->pin_completed <= wait_for) {
      HIPCHK(hipEventSynchronize(e));
    } else {
      const hipError_t q = hipEventQuery(e);
      if (q == hipErrorNotReady) break;
      HIPCHK(q);
    }
    ++c->pin_completed;
  }
  return GLH_OK;
}

extern "C" int glh_observer_upload_frame_pinned(glh_ctx* c, int o, int image, const uint8_t* pixels, int64_t* ticket) {
  CHK(check_obs(c, o));
  Observer& ob = c->obs[o];
  if (!pixels || !ticket || image < 0 || image >= ob.n_images) return fail(GLH_E_INVALID, "bad frame index %d", image);
  HIPCHK(hipSetDevice(c->cfg.device_id));
  const size_t bytes = ob.frame_bytes();
  if (!c->copy_stream) {
    HIPCHK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&c->upload_done, hipEventDisableTiming));
  }
  const int64_t t = c->pin_next;
  if (t - c->pin_completed >= glh_ctx::NPIN) CHK(pin_poll(c, t - glh_ctx::NPIN));  // (its event is about to be reused)
  hipEvent_t& e = c->pin_ev[t % glh_ctx::NPIN];
  if (!e) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  if (!ob.owned[image]) CHK(dalloc(&ob.owned[image], bytes));
  HIPCHK(hipMemcpyAsync(ob.owned[image], pixels, bytes, hipMemcpyHostToDevice, c->copy_stream));
  HIPCHK(hipEventRecord(e, c->copy_stream));
  HIPCHK(hipEventRecord(c->upload_done, c->copy_stream));
  c->pin_next = t + 1;
  c->uploads_pending = true;
  ob.frames[image] = ob.owned[image];
  *ticket = t;
  return GLH_OK;
}

extern "C" int glh_upload_done(glh_ctx* c, int64_t ticket, int wait, int* done) {
  if (!c || !done) return fail(GLH_E_INVALID, "null argument");
  if (ticket < 0 || ticket >= c->pin_next) return fail(GLH_E_INVALID, "unknown upload ticket %lld", (long long)ticket);
  HIPCHK(hipSetDevice(c->cfg.device_id));
  CHK(pin_poll(c, wait ? ticket : -1));
  *done = ticket < c->pin_completed ? 1 : 0;
  return GLH_OK;
}

extern "C" int glh_observer_set_frame_device(glh_ctx* c, int o, int image, const void* dev) {
  CHK(check_obs(c, o));
  Observer& ob = c->obs[o];
  if (!dev || image < 0 || image >= ob.n_images) return fail(GLH_E_INVALID, "bad frame index %d", image);
  dfree(ob.owned[image]);
  ob.frames[image] = (const uint8_t*)dev;
  return GLH_OK;
}

// ------------------------------------------------------------------------------------------
// sequence state
// ------------------------------------------------------------------------------------------
extern "C" int glh_begin_sequence(glh_ctx* c, int P, int N, int tw, int th) {
  if (!c) return fail(GLH_E_INVALID, "null context");
  if (P <= 0 || P > c->cfg.max_points || N <= 0 || N > c->cfg.max_particles)
    return fail(GLH_E_INVALID, "n_points %d / n_particles %d exceed the context capacity (%d, %d)", P, N,
                c->cfg.max_points, c->cfg.max_particles);
  if (tw < 5 || th < 5 || tw > c->cfg.max_tile || th > c->cfg.max_tile)
    return fail(GLH_E_INVALID, "tile_size (%d, %d) must be in [5, max_tile=%d]", tw, th, c->cfg.max_tile);
  HIPCHK(hipSetDevice(c->cfg.device_id));
  c->P = P;
  c->N = N;
  c->tw = tw;
  c->th = th;
  c->NB = (N + BLK - 1) / BLK;
  c->cur = 0;
  c->compact = false;
  c->frame = 0;
  c->have_mask = c->have_active = false;
  c->pt_base = 0;
  const size_t O = c->cfg.n_observers;
  HIPCHK(hipMemsetAsync(c->pt_status, 0, P * sizeof(uint32_t), c->stream));
  HIPCHK(hipMemsetAsync(c->pt_err_frame, 0x7f, P * sizeof(int32_t), c->stream));
  HIPCHK(hipMemsetAsync(c->tmpl_valid, 0, O * P * sizeof(int32_t), c->stream));
  // (the fused kernel fetches a template's CDF length before it knows the template is valid: never garbage)
  HIPCHK(hipMemsetAsync(c->tmpl_hist_n, 0, O * P * sizeof(int32_t), c->stream));
  {
    // GLH_OBS_SKIPPED everywhere, for every frame
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)c->obs_status_all, GLH_OBS_SKIPPED, (size_t)c->cfg.max_frames * O * P,
                             c->stream));
  }
  size_t nm = (size_t)c->cfg.max_frames * P * 12;
  hipLaunchKernelGGL(k_fill_f64, dim3(256), dim3(256), 0, c->stream, c->moments, nm, (double)NAN);
  HIPCHK(hipGetLastError());
  if (c->covariances) {  // a reused context must not return the previous sequence's covariances
    hipLaunchKernelGGL(k_fill_f64, dim3(256), dim3(256), 0, c->stream, c->covariances,
                       (size_t)c->cfg.max_frames * c->cfg.max_points * 36, (double)NAN);
    HIPCHK(hipGetLastError());
  }
  // NumPy pairwise-sum plan for this N (kept from the last sequence when N is the same: five frees, allocations and
  // blocking copies -- ~10 ms -- that a tracker reusing its context for the next run does not pay again)
  if (c->plan_N != N) {
    PairwisePlan pl;
    pairwise_plan(N, pl);
    dfree(c->leaf_off); dfree(c->leaf_len); dfree(c->sum_ops); dfree(c->level_off); dfree(c->roots);
    c->plan_N = 0;
    CHK(dalloc(&c->leaf_off, pl.leaf_off.size()));
    CHK(dalloc(&c->leaf_len, pl.leaf_len.size()));
    CHK(dalloc(&c->sum_ops, pl.ops.size()));
    CHK(dalloc(&c->level_off, pl.level_off.size()));
    CHK(dalloc(&c->roots, pl.roots.size()));
    HIPCHK(hipMemcpy(c->leaf_off, pl.leaf_off.data(), pl.leaf_off.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->leaf_len, pl.leaf_len.data(), pl.leaf_len.size() * 4, hipMemcpyHostToDevice));
    if (!pl.ops.empty()) HIPCHK(hipMemcpy(c->sum_ops, pl.ops.data(), pl.ops.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->level_off, pl.level_off.data(), pl.level_off.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->roots, pl.roots.data(), pl.roots.size() * 4, hipMemcpyHostToDevice));
    c->nleaves = (int)pl.leaf_off.size();
    c->nnodes = pl.nnodes;
    c->nlevels = (int)pl.level_off.size() - 1;
    c->nroots = (int)pl.roots.size();
    c->plan_N = N;
  }
  c->moments_frame = -1;
  return GLH_OK;
}

extern "C" int glh_set_frame(glh_ctx* c, int frame) {
  if (!c) return fail(GLH_E_INVALID, "null context");
  if (frame < 0 || frame >= c->cfg.max_frames) return fail(GLH_E_INVALID, "frame %d outside [0, max_frames)", frame);
  c->frame = frame;
  return GLH_OK;
}

static int need_seq(glh_ctx* c) {
  if (!c) return fail(GLH_E_INVALID, "null context");
  if (c->P <= 0) return fail(GLH_E_STATE, "glh_begin_sequence has not been called");
  return GLH_OK;
}

// The fused step leaves the state run-length compact; everything else works on one record per particle.
static int ensure_expanded(glh_ctx* c) {
  if (!c->compact) return GLH_OK;
  HIPCHK(hipSetDevice(c->cfg.device_id));
  const dim3 grid((c->N + BLK - 1) / BLK, c->P);
  hipLaunchKernelGGL(k_expand_state, grid, dim3(BLK), 0, c->stream, c->particles[c->cur], c->weights[c->cur],
                     c->uidx[c->cur], c->particles[c->cur ^ 1], c->weights[c->cur ^ 1], c->N);
  HIPCHK(hipGetLastError());
  c->cur ^= 1;
  c->compact = false;
  return GLH_OK;
}

#define UPLOAD(dst, src, count, type)                                                                 \
  do {                                                                                                \
    HIPCHK(hipSetDevice(c->cfg.device_id));                                                           \
    HIPCHK(hipMemcpyAsync((dst), (src), (size_t)(count) * sizeof(type), hipMemcpyHostToDevice, c->stream)); \
    HIPCHK(hipStreamSynchronize(c->stream));                                                          \
  } while (0)
#define DOWNLOAD(dst, src, count, type)                                                               \
  do {                                                                                                \
    HIPCHK(hipSetDevice(c->cfg.device_id));                                                           \
    HIPCHK(hipMemcpyAsync((dst), (src), (size_t)(count) * sizeof(type), hipMemcpyDeviceToHost, c->stream)); \
    HIPCHK(hipStreamSynchronize(c->stream));                                                          \
  } while (0)

// Fast arithmetic (glh_set_math): every kernel has it.  In the general instantiation (gridded surfaces, the other motion
// models) it changes the projection, the sampling, the weights and the resampling, and (round 5) the surface lookups of
// the evolve step and of the DEM term (glh_math.h: raster_bilinear_fast) and the tangent models' step.
static bool use_fast(const glh_ctx* c) { return c->fast_math; }

static Surfaces surfaces(const glh_ctx* c) {
  Surfaces s{};
  s.dem = c->rasters[GLH_RASTER_DEM].dev;
  s.dem_sigma = c->rasters[GLH_RASTER_DEM_SIGMA].dev;
  s.viewshed = c->rasters[GLH_RASTER_VIEWSHED].dev;
  s.same_grid = c->same_grid ? 1 : 0;
  return s;
}

static int check_raster_args(int nx, int ny, const double* gx, const double* gy, int sx, int sy) {
  if (nx < 2 || ny < 2) return fail(GLH_E_UNSUPPORTED, "rasters need at least 2 x 2 cells (got %d x %d)", nx, ny);
  if (!gx || !gy || (sx != 1 && sx != -1) || (sy != 1 && sy != -1)) return fail(GLH_E_INVALID, "bad raster geometry");
  for (int i = 1; i < nx; ++i)
    if (!(gx[i] > gx[i - 1])) return fail(GLH_E_INVALID, "gx must be strictly ascending");
  for (int i = 1; i < ny; ++i)
    if (!(gy[i] > gy[i - 1])) return fail(GLH_E_INVALID, "gy must be strictly ascending");
  return GLH_OK;
}
// (the kernels find a sample's cell from the cell size: glh_math.h, raster_interval)
static int check_raster_uniform(int nx, int ny, const double* gx, const double* gy, double xmin, double xmax, double ymin,
                                double ymax) {
  if (!raster_coordinates_uniform(gx, nx, xmin, xmax) || !raster_coordinates_uniform(gy, ny, ymin, ymax))
    return fail(GLH_E_UNSUPPORTED, "raster coordinates must be the cell centres of a uniform grid over the outer limits "
                                   "(glimpse.Grid.x / .y); these are further than a quarter cell from it");
  return GLH_OK;
}

extern "C" int glh_set_raster(glh_ctx* c, int which, const double* z, int nx, int ny, const double* gx,
                              const double* gy, int sx, int sy, double xmin, double xmax, double ymin,
                              double ymax) {
  if (!c) return fail(GLH_E_INVALID, "null context");
  if (which < GLH_RASTER_DEM || which > GLH_RASTER_VIEWSHED) return fail(GLH_E_INVALID, "unknown raster slot %d", which);
  auto& r = c->rasters[which];
  if (!z && !r.z) return GLH_OK;  // (no raster before, none now: nothing to wait for -- the usual call of every run)
  HIPCHK(hipSetDevice(c->cfg.device_id));
  HIPCHK(hipStreamSynchronize(c->stream));
  dfree(r.z); dfree(r.gx); dfree(r.gy);
  r.dev = RasterDev{};
  r.hgx.clear();
  r.hgy.clear();
  c->same_grid = false;
  if (!z) return GLH_OK;
  CHK(check_raster_args(nx, ny, gx, gy, sx, sy));
  CHK(check_raster_uniform(nx, ny, gx, gy, xmin, xmax, ymin, ymax));
  CHK(dalloc(&r.z, (size_t)nx * ny));
  CHK(dalloc(&r.gx, (size_t)nx));
  CHK(dalloc(&r.gy, (size_t)ny));
  HIPCHK(hipMemcpy(r.z, z, (size_t)nx * ny * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(r.gx, gx, (size_t)nx * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(r.gy, gy, (size_t)ny * sizeof(double), hipMemcpyHostToDevice));
  r.dev = raster_dev(r.z, r.gx, r.gy, nx, ny, sx, sy, xmin, xmax, ymin, ymax);
  r.hgx.assign(gx, gx + nx);
  r.hgy.assign(gy, gy + ny);
  // a DEM and its uncertainty on ONE grid share the cell and the weights of a sample (raster_sample_pair): only when the
  // coordinate arrays are the same numbers -- equal limits and sizes alone allow a quarter cell of difference
  const auto &d0 = c->rasters[GLH_RASTER_DEM], &d1 = c->rasters[GLH_RASTER_DEM_SIGMA];
  c->same_grid = d0.z && d1.z && d0.hgx == d1.hgx && d0.hgy == d1.hgy;
  return GLH_OK;
}

extern "C" int glh_set_motion(glh_ctx* c, const double* params) {
  CHK(need_seq(c));
  if (!params) return fail(GLH_E_INVALID, "params is null");
  c->has_dem = false;
  c->all_cartesian = true;
  for (int p = 0; p < c->P; ++p) {
    const double* m = params + (size_t)p * GLH_MOTION_FULL_LEN;
    const int kind = (int)m[18];
    if (m[18] != (double)kind || kind < GLH_MOTION_CARTESIAN || kind > GLH_MOTION_EXTERNAL)
      return fail(GLH_E_INVALID, "point %d: unknown motion kind %g", p, m[18]);
    if (kind != GLH_MOTION_CARTESIAN) c->all_cartesian = false;
    if (m[20] != 0.0 && !c->rasters[GLH_RASTER_DEM].z)
      return fail(GLH_E_STATE, "point %d uses a dem raster but glh_set_raster(GLH_RASTER_DEM) was not called", p);
    if (m[21] != 0.0 && !c->rasters[GLH_RASTER_DEM_SIGMA].z)
      return fail(GLH_E_STATE, "point %d uses a dem_sigma raster but glh_set_raster(GLH_RASTER_DEM_SIGMA) was not called", p);
    if (kind <= GLH_MOTION_CYLINDRICAL && (m[17] != 0.0 || m[21] != 0.0)) c->has_dem = true;
  }
  UPLOAD(c->motion, params, (size_t)c->P * GLH_MOTION_FULL_LEN, double);
  return GLH_OK;
}

extern "C" int glh_set_motion_cartesian(glh_ctx* c, const double* params) {
  CHK(need_seq(c));
  if (!params) return fail(GLH_E_INVALID, "params is null");
  std::vector<double> full((size_t)c->P * GLH_MOTION_FULL_LEN, 0.0);
  for (int p = 0; p < c->P; ++p)
    std::copy(params + (size_t)p * GLH_MOTION_LEN, params + (size_t)(p + 1) * GLH_MOTION_LEN,
              full.begin() + (size_t)p * GLH_MOTION_FULL_LEN);
  return glh_set_motion(c, full.data());
}
extern "C" int glh_set_point_offset(glh_ctx* c, int offset) {
  CHK(need_seq(c));
  if (offset < 0) return fail(GLH_E_INVALID, "offset must be >= 0");
  c->pt_base = offset;
  return GLH_OK;
}
extern "C" int glh_set_observer_mask(glh_ctx* c, const uint8_t* mask) {
  CHK(need_seq(c));
  c->have_mask = mask != nullptr;
  if (mask) UPLOAD(c->obs_mask, mask, (size_t)c->P * c->cfg.n_observers, uint8_t);
  return GLH_OK;
}
extern "C" int glh_set_active(glh_ctx* c, const uint8_t* active) {
  CHK(need_seq(c));
  c->have_active = active != nullptr;
  if (active) UPLOAD(c->active, active, (size_t)c->P, uint8_t);
  return GLH_OK;
}
extern "C" int glh_set_extra_log_likelihoods(glh_ctx* c, const double* ll) {
  CHK(need_seq(c));
  c->have_extra = ll != nullptr;
  if (!ll) return GLH_OK;
  if (!c->extra_ll) CHK(dalloc(&c->extra_ll, (size_t)c->cfg.max_points * c->cfg.max_particles));
  UPLOAD(c->extra_ll, ll, (size_t)c->P * c->N, double);
  return GLH_OK;
}
extern "C" int glh_set_particles(glh_ctx* c, const double* p) {
  CHK(need_seq(c));
  CHK(ensure_expanded(c));
  c->moments_frame = -1;
  if (!p) return fail(GLH_E_INVALID, "particles is null");
  UPLOAD(c->particles[c->cur], p, (size_t)c->P * c->N * 6, double);
  return GLH_OK;
}
extern "C" int glh_get_particles(glh_ctx* c, double* p) {
  CHK(need_seq(c));
  CHK(ensure_expanded(c));
  if (!p) return fail(GLH_E_INVALID, "particles is null");
  DOWNLOAD(p, c->particles[c->cur], (size_t)c->P * c->N * 6, double);
  return GLH_OK;
}
extern "C" int glh_set_weights(glh_ctx* c, const double* w) {
  CHK(need_seq(c));
  CHK(ensure_expanded(c));
  c->moments_frame = -1;
  if (!w) return fail(GLH_E_INVALID, "weights is null");
  UPLOAD(c->weights[c->cur], w, (size_t)c->P * c->N, double);
  return GLH_OK;
}
extern "C" int glh_get_weights(glh_ctx* c, double* w) {
  CHK(need_seq(c));
  CHK(ensure_expanded(c));
  if (!w) return fail(GLH_E_INVALID, "weights is null");
  DOWNLOAD(w, c->weights[c->cur], (size_t)c->P * c->N, double);
  return GLH_OK;
}
extern "C" int glh_get_point_status(glh_ctx* c, uint32_t* st) {
  CHK(need_seq(c));
  if (!st) return fail(GLH_E_INVALID, "status is null");
  DOWNLOAD(st, c->pt_status, (size_t)c->P, uint32_t);
  return GLH_OK;
}
extern "C" int glh_get_point_error_frame(glh_ctx* c, int32_t* fr) {
  CHK(need_seq(c));
  if (!fr) return fail(GLH_E_INVALID, "frames is null");
  DOWNLOAD(fr, c->pt_err_frame, (size_t)c->P, int32_t);
  return GLH_OK;
}
extern "C" int glh_get_observer_status(glh_ctx* c, int32_t* st) {
  CHK(need_seq(c));
  if (!st) return fail(GLH_E_INVALID, "status is null");
  DOWNLOAD(st, cur_status(c), (size_t)c->cfg.n_observers * c->P, int32_t);
  return GLH_OK;
}
extern "C" int glh_get_observer_status_frames(glh_ctx* c, int frame0, int n_frames, int32_t* st) {
  CHK(need_seq(c));
  if (!st || frame0 < 0 || n_frames <= 0 || frame0 + n_frames > c->cfg.max_frames)
    return fail(GLH_E_INVALID, "bad frame range");
  const size_t per = (size_t)c->cfg.n_observers * c->P;
  DOWNLOAD(st, c->obs_status_all + (size_t)frame0 * per, (size_t)n_frames * per, int32_t);
  return GLH_OK;
}
extern "C" int glh_get_point_state(glh_ctx* c, int point, double* particles, double* weights) {
  CHK(need_seq(c));
  CHK(ensure_expanded(c));
  if (point < 0 || point >= c->P) return fail(GLH_E_INVALID, "point %d out of range", point);
  if (particles) DOWNLOAD(particles, c->particles[c->cur] + (size_t)point * c->N * 6, (size_t)c->N * 6, double);
  if (weights) DOWNLOAD(weights, c->weights[c->cur] + (size_t)point * c->N, (size_t)c->N, double);
  return GLH_OK;
}
extern "C" int glh_get_log_likelihoods(glh_ctx* c, int o, double* ll) {
  CHK(need_seq(c));
  CHK(check_obs(c, o));
  if (!ll || !c->keep_sse || !c->ll_dbg) return fail(GLH_E_STATE, "log-likelihood capture is off (glh_set_debug)");
  DOWNLOAD(ll, c->ll_dbg + (size_t)o * c->P * c->N, (size_t)c->P * c->N, double);
  return GLH_OK;
}
extern "C" int glh_get_search_boxes(glh_ctx* c, int32_t* boxes) {
  CHK(need_seq(c));
  if (!boxes) return fail(GLH_E_INVALID, "boxes is null");
  DOWNLOAD(boxes, c->box, (size_t)c->cfg.n_observers * c->P * 4, int32_t);
  return GLH_OK;
}
extern "C" int glh_get_resample_indices(glh_ctx* c, int32_t* idx) {
  CHK(need_seq(c));
  if (!idx || !c->idx) return fail(GLH_E_STATE, "index capture is off (glh_set_debug)");
  DOWNLOAD(idx, c->idx, (size_t)c->P * c->N, int32_t);
  return GLH_OK;
}

extern "C" int glh_set_debug(glh_ctx* c, int keep) {
  if (!c) return fail(GLH_E_INVALID, "null context");
  HIPCHK(hipSetDevice(c->cfg.device_id));
  if (keep < 0 || keep > 2) return fail(GLH_E_INVALID, "keep must be 0, 1 or 2");
  c->keep_sse = keep == 1;
  c->keep_idx = keep != 0;
  if (keep == 2 && !c->idx) CHK(dalloc(&c->idx, (size_t)c->cfg.max_points * c->cfg.max_particles));
  if (keep == 1) {
    if (!c->sse_copy)
      CHK(dalloc(&c->sse_copy, (size_t)c->cfg.n_observers * c->cfg.max_points * (size_t)c->sse_cap));
    if (!c->idx) CHK(dalloc(&c->idx, (size_t)c->cfg.max_points * c->cfg.max_particles));
    if (!c->ll_dbg)
      CHK(dalloc(&c->ll_dbg, (size_t)c->cfg.n_observers * c->cfg.max_points * c->cfg.max_particles));
  }
  return GLH_OK;
}

// staging of host-fed random numbers (parity mode)
static int stage_normals(glh_ctx* c, const double* host, size_t count) {
  if (count > c->normals_cap) {
    dfree(c->normals);
    CHK(dalloc(&c->normals, count));
    c->normals_cap = count;
  }
  HIPCHK(hipMemcpyAsync(c->normals, host, count * sizeof(double), hipMemcpyHostToDevice, c->stream));
  // the host buffer may be reused by the caller right away
  HIPCHK(hipStreamSynchronize(c->stream));
  return GLH_OK;
}

// ------------------------------------------------------------------------------------------
// stages
// ------------------------------------------------------------------------------------------
extern "C" int glh_init_particles(glh_ctx* c, int rng_mode, const double* normals, uint64_t seed) {
  CHK(need_seq(c));
  CHK(ensure_expanded(c));
  HIPCHK(hipSetDevice(c->cfg.device_id));
  if (rng_mode == GLH_RNG_HOST) {
    if (!normals) return fail(GLH_E_INVALID, "GLH_RNG_HOST needs normals [P][N][6]");
    CHK(stage_normals(c, normals, (size_t)c->P * c->N * 6));
  } else if (rng_mode != GLH_RNG_PHILOX) {
    return fail(GLH_E_INVALID, "unknown rng_mode %d", rng_mode);
  }
  c->moments_frame = -1;
  InitArgs a{};
  a.particles = c->particles[c->cur];
  a.weights = c->weights[c->cur];
  a.motion = c->motion;
  a.active = c->have_active ? c->active : nullptr;
  a.normals = c->normals;
  a.seed = seed;
  a.rng_mode = rng_mode;
  a.N = c->N;
  a.pt_base = c->pt_base;
  a.frame = c->frame;
  a.pt_status = c->pt_status;
  a.pt_err_frame = c->pt_err_frame;
  a.surf = surfaces(c);
  {
    StageTimer t(c, ST_INIT);
    hipLaunchKernelGGL(k_init_particles, dim3(c->NB, c->P), dim3(BLK), 0, c->stream, a);
  }
  HIPCHK(hipGetLastError());
  return GLH_OK;
}

static void fill_obs(glh_ctx* c, int o, int image, ObsFrame* f) {
  const Observer& ob = c->obs[o];
  f->on = image >= 0;
  f->cam = ob.cams + (image >= 0 ? image : 0);
  f->frame = image >= 0 ? ob.frames[image] : nullptr;
  f->width = ob.width;
  f->height = ob.height;
  f->channels = ob.channels;
  f->bits = ob.bits;
  f->bins = c->bins16;
  f->fwork = c->fwork;
  f->fwork_cap = 2 * (int64_t)c->cfg.max_search_dim * c->cfg.max_search_dim;
}

// 16-bit frames: the zeroed per-point key histograms the staged tile kernels of observer `o` count into
static int prepare_bins16(glh_ctx* c, int o, bool search_tiles = false) {
  if (c->obs[o].bits >= 16 && !c->fwork)  // 16-bit and float frames: two tile-sized arrays of doubles per point
    CHK(dalloc(&c->fwork, (size_t)c->cfg.max_points * 2 * c->cfg.max_search_dim * c->cfg.max_search_dim));
  if (c->obs[o].bits != 16 || search_tiles) return GLH_OK;  // (search tiles are ranked; only templates count into bins)
  const size_t per = (size_t)(65535 * 3 + 1);
  if (!c->bins16) CHK(dalloc(&c->bins16, (size_t)c->cfg.max_points * per));
  HIPCHK(hipMemsetAsync(c->bins16, 0, (size_t)c->P * (65535 * c->obs[o].channels + 1) * sizeof(uint32_t), c->stream));
  return GLH_OK;
}

static int check_images(glh_ctx* c, const int32_t* images) {
  if (!images) return fail(GLH_E_INVALID, "images is null");
  for (int o = 0; o < c->cfg.n_observers; ++o) {
    if (images[o] < 0) continue;
    const Observer& ob = c->obs[o];
    if (images[o] >= ob.n_images) return fail(GLH_E_INVALID, "observer %d: image %d out of range", o, images[o]);
    if (!ob.frames[images[o]]) return fail(GLH_E_STATE, "observer %d: image %d has not been uploaded", o, images[o]);
  }
  return join_uploads(c);  // the kernels that follow read the frames
}

// evolve (optional) + project into the given images + bbox partials
static int launch_evolve_project(glh_ctx* c, bool do_evolve, double tau, int rng_mode, uint64_t seed,
                                 uint64_t step, const int32_t* images, bool store = true) {
  CHK(ensure_expanded(c));
  if (do_evolve) c->moments_frame = -1;
  EvolveArgs a{};
  a.particles = c->particles[c->cur];
  a.store = store;
  a.surf = surfaces(c);
  a.motion = c->motion;
  a.active = c->have_active ? c->active : nullptr;
  a.obs_mask = c->have_mask ? c->obs_mask : nullptr;
  a.normals = c->normals;
  a.uv = c->uv;
  a.bbox_part = c->bbox_part;
  a.pt_status = c->pt_status;
  a.pt_err_frame = c->pt_err_frame;
  a.seed = seed;
  a.step = step;
  a.tau = tau;
  a.do_evolve = do_evolve;
  a.rng_mode = rng_mode;
  a.N = c->N;
  a.P = c->P;
  a.O = c->cfg.n_observers;
  a.NB = c->NB;
  a.frame = c->frame;
  a.pt_base = c->pt_base;
  a.fast = use_fast(c);
  for (int o = 0; o < a.O; ++o) fill_obs(c, o, images ? images[o] : -1, &a.obs[o]);
  {
    StageTimer t(c, ST_EVOLVE_PROJECT);
    hipLaunchKernelGGL(k_evolve_project, dim3(c->NB, c->P), dim3(BLK), 0, c->stream, a);
  }
  HIPCHK(hipGetLastError());
  return GLH_OK;
}

extern "C" int glh_evolve(glh_ctx* c, double tau, int rng_mode, const double* normals, uint64_t seed,
                          uint64_t step) {
  CHK(need_seq(c));
  HIPCHK(hipSetDevice(c->cfg.device_id));
  if (rng_mode == GLH_RNG_HOST) {
    if (!normals) return fail(GLH_E_INVALID, "GLH_RNG_HOST needs normals [P][N][3]");
    CHK(stage_normals(c, normals, (size_t)c->P * c->N * 3));
  } else if (rng_mode != GLH_RNG_PHILOX) {
    return fail(GLH_E_INVALID, "unknown rng_mode %d", rng_mode);
  }
  return launch_evolve_project(c, true, tau, rng_mode, seed, step, nullptr);
}

static int launch_moments(glh_ctx* c, double* out, int ld, int with_sigma) {
  CHK(ensure_expanded(c));
  MomentsArgs a{};
  a.particles = c->particles[c->cur];
  a.weights = c->weights[c->cur];
  a.active = c->have_active ? c->active : nullptr;
  a.out = out;
  a.N = c->N;
  a.ld = ld;
  a.with_sigma = with_sigma;
  {
    StageTimer t(c, ST_MOMENTS);
    hipLaunchKernelGGL(k_moments, dim3(c->P), dim3(BLK), 0, c->stream, a);
  }
  HIPCHK(hipGetLastError());
  return GLH_OK;
}

extern "C" int glh_init_templates(glh_ctx* c, int o, int image) {
  CHK(need_seq(c));
  CHK(check_obs(c, o));
  HIPCHK(hipSetDevice(c->cfg.device_id));
  const Observer& ob = c->obs[o];
  if (image < 0 || image >= ob.n_images || !ob.frames[image])
    return fail(GLH_E_STATE, "observer %d: image %d is not resident", o, image);
  CHK(join_uploads(c));
  CHK(launch_moments(c, c->mean6, 6, 0));  // particle_mean with the carried weights (tracker.py:548)
  TemplateArgs a{};
  a.mean6 = c->mean6;
  a.active = c->have_active ? c->active : nullptr;
  a.obs_mask = c->have_mask ? c->obs_mask : nullptr;
  fill_obs(c, o, image, &a.obs);
  a.o = o;
  a.O = c->cfg.n_observers;
  a.P = c->P;
  a.tw = c->tw;
  a.th = c->th;
  a.tile_cap = c->tile_cap;
  a.frame = c->frame;
  a.hp_rx = c->hp_rx;
  a.hp_ry = c->hp_ry;
  a.tmpl_box = c->tmpl_box;
  a.tmpl_duv = c->tmpl_duv;
  a.tmpl_tile64 = c->tmpl_tile64;
  a.tmpl_tile32 = c->tmpl_tile32;
  a.tmpl_hist_v = c->tmpl_hist_v;
  a.tmpl_hist_q = c->tmpl_hist_q;
  a.tmpl_hist_n = c->tmpl_hist_n;
  a.tmpl_valid = c->tmpl_valid;
  a.pt_status = c->pt_status;
  a.pt_err_frame = c->pt_err_frame;
  CHK(prepare_bins16(c, o));
  a.obs.bins = c->bins16;
  a.obs.fwork = c->fwork;
  {
    StageTimer t(c, ST_TEMPLATE);
    hipLaunchKernelGGL(k_template_init, dim3(c->P), dim3(BLK),
                       (size_t)c->tw * c->th * (ob.bits >= 32 ? 12 : (ob.bits == 16 ? 4 : 2)), c->stream, a);
  }
  HIPCHK(hipGetLastError());
  return GLH_OK;
}

// search tile -> SSD surface -> spline coefficients of every observer with an image
static int launch_tile_stages(glh_ctx* c, const int32_t* images) {
  const int O = c->cfg.n_observers;
  for (int o = 0; o < O; ++o) {
    if (images[o] < 0) {
      // no image for this observer at this frame (tracker.py:577): the status must not keep the previous frame's
      HIPCHK(hipMemsetD32Async((hipDeviceptr_t)(cur_status(c) + (size_t)o * c->P), GLH_OBS_SKIPPED, (size_t)c->P,
                               c->stream));
      continue;
    }
    TilePrepArgs tp{};
    tp.active = c->have_active ? c->active : nullptr;
    tp.obs_mask = c->have_mask ? c->obs_mask : nullptr;
    fill_obs(c, o, images[o], &tp.obs);
    tp.o = o;
    tp.O = O;
    tp.P = c->P;
    tp.NB = c->NB;
    tp.tw = c->tw;
    tp.th = c->th;
    tp.tile_cap = c->tile_cap;
    tp.search_cap = c->search_cap;
    tp.max_dim = c->cfg.max_search_dim;
    tp.hp_rx = c->hp_rx;
    tp.hp_ry = c->hp_ry;
    tp.kcols = c->interp_ky;  // ("ky" widens the columns, "kx" the rows: tracker.py:585-590)
    tp.krows = c->interp_kx;
    tp.bbox_part = c->bbox_part;
    tp.tmpl_valid = c->tmpl_valid;
    tp.tmpl_hist_v = c->tmpl_hist_v;
    tp.tmpl_hist_q = c->tmpl_hist_q;
    tp.tmpl_hist_n = c->tmpl_hist_n;
    tp.box = c->box;
    tp.obs_status = cur_status(c);
    tp.search = c->search;
    CHK(prepare_bins16(c, o, true));
    tp.obs.bins = c->bins16;
    tp.obs.fwork = c->fwork;
    {
      StageTimer t(c, ST_TILEPREP);
      size_t lds = (size_t)(BAND_H + 6) * c->cfg.max_search_dim * (c->obs[o].bits == 16 ? 4 : 2);  // (halo of up to 3 rows)
      // float frames: the template CDF (up to tw x th values and quantiles) is searched twice per pixel -- from LDS
      // ... and the float32 scratch of a typical tile's normalisation (2 n floats) is read by one thread -- from LDS
      if (c->obs[o].bits >= 16)
        lds = std::max({lds, (size_t)2 * c->tile_cap * sizeof(double),
                        std::min((size_t)8 * c->cfg.max_search_dim * c->cfg.max_search_dim, (size_t)40 * 1024)});
      tp.lds_bytes = (int32_t)lds;
      hipLaunchKernelGGL(k_tileprep, dim3(c->P), dim3(BLK), lds, c->stream, tp);
    }
    HIPCHK(hipGetLastError());
    SsdArgs sa{};
    sa.o = o;
    sa.P = c->P;
    sa.tw = c->tw;
    sa.th = c->th;
    sa.tile_cap = c->tile_cap;
    sa.search_cap = c->search_cap;
    sa.sse_cap = c->sse_cap;
    sa.box = c->box;
    sa.obs_status = cur_status(c);
    sa.search = c->search;
    sa.tmpl = c->tmpl_tile32;
    sa.sse = c->sse;
    {
      StageTimer t(c, ST_SSD);
      hipLaunchKernelGGL(k_ssd, dim3(c->ssd_blocks, c->P), dim3(BLK), ssd_lds_bytes(c->tw, c->th), c->stream, sa);
    }
    HIPCHK(hipGetLastError());
    SplineFitArgs sf{};
    sf.o = o;
    sf.P = c->P;
    sf.tw = c->tw;
    sf.th = c->th;
    sf.sse_cap = c->sse_cap;
    sf.max_n = c->cfg.max_search_dim;
    sf.box = c->box;
    sf.obs_status = cur_status(c);
    sf.lu = c->lu;
    sf.lu_off = c->lu_off;
    sf.inv = c->spl_inv;
    sf.sse = c->sse;
    sf.sse_copy = c->keep_sse ? c->sse_copy : nullptr;
    sf.linear = c->interp_k == 1;
    sf.kx = c->interp_kx;
    sf.ky = c->interp_ky;
    if (c->interp_k == 0) {
      sf.glu_v = c->glu[0];
      sf.glu_v_off = c->glu_off[0];
      sf.glu_u = c->glu[1];
      sf.glu_u_off = c->glu_off[1];
    }
    {
      StageTimer t(c, ST_SPLINE_FIT);
      hipLaunchKernelGGL(k_spline_fit, dim3(c->P), dim3(BLK), 0, c->stream, sf);
    }
    HIPCHK(hipGetLastError());
  }
  return GLH_OK;
}

static int cell_cap(const glh_ctx* c);  // (defined with the fused kernel's LDS plan below)
static int update_weights_impl(glh_ctx* c, const int32_t* images, bool projected) {
  CHK(ensure_expanded(c));
  const int O = c->cfg.n_observers;
  // uv + bbox of the (already evolved) particles in the matched images
  if (!projected) CHK(launch_evolve_project(c, false, 0.0, GLH_RNG_PHILOX, 0, 0, images));
  CHK(launch_tile_stages(c, images));
  c->moments_frame = -1;
  WeightArgs wa{};
  wa.particles = c->particles[c->cur];
  wa.weights = c->weights[c->cur];
  wa.motion = c->motion;
  wa.active = c->have_active ? c->active : nullptr;
  wa.uv = c->uv;
  wa.box = c->box;
  wa.obs_status = cur_status(c);
  wa.tmpl_duv = c->tmpl_duv;
  wa.coef = c->sse;
  wa.poly = c->poly;
  wa.ll_out = c->keep_sse ? c->ll_dbg : nullptr;
  wa.extra = c->have_extra ? c->extra_ll : nullptr;
  wa.pt_status = c->pt_status;
  wa.pt_err_frame = c->pt_err_frame;
  for (int o = 0; o < O; ++o) {
    wa.on[o] = images[o] >= 0;
    wa.inv2s2[o] = 1.0 / (2.0 * (c->obs[o].sigma * c->obs[o].sigma));  // 1 / (2 * sigma ** 2)
  }
  wa.N = c->N;
  wa.P = c->P;
  wa.O = O;
  wa.tw = c->tw;
  wa.th = c->th;
  wa.sse_cap = c->sse_cap;
  wa.frame = c->frame;
  wa.fast = use_fast(c);
  wa.cell_cap = cell_cap(c);
  wa.linear = c->interp_k == 1;
  wa.general = c->interp_k == 0;
  wa.kx = c->interp_kx;
  wa.ky = c->interp_ky;
  wa.surf = surfaces(c);
  {
    StageTimer t(c, ST_WEIGHTS);
    hipLaunchKernelGGL(k_weights, dim3((c->NB + WEIGHTS_PER_THREAD - 1) / WEIGHTS_PER_THREAD, c->P), dim3(BLK), 0, c->stream, wa);
  }
  HIPCHK(hipGetLastError());
  return GLH_OK;
}

extern "C" int glh_update_weights(glh_ctx* c, const int32_t* images) {
  CHK(need_seq(c));
  CHK(check_images(c, images));
  HIPCHK(hipSetDevice(c->cfg.device_id));
  return update_weights_impl(c, images, false);
}

extern "C" int glh_resample(glh_ctx* c, int rng_mode, const double* u, uint64_t seed, uint64_t step) {
  return glh_resample_method(c, GLH_RESAMPLE_SYSTEMATIC, rng_mode, u, seed, step);
}

extern "C" int glh_resample_method(glh_ctx* c, int method, int rng_mode, const double* u, uint64_t seed,
                                   uint64_t step) {
  CHK(need_seq(c));
  CHK(ensure_expanded(c));
  HIPCHK(hipSetDevice(c->cfg.device_id));
  if (method != GLH_RESAMPLE_SYSTEMATIC && method != GLH_RESAMPLE_STRATIFIED && method != GLH_RESAMPLE_CHOICE &&
      method != GLH_RESAMPLE_RESIDUAL)
    return fail(GLH_E_INVALID, "unknown resampling method %d", method);
  if (method == GLH_RESAMPLE_RESIDUAL && !c->resid_draws) CHK(dalloc(&c->resid_draws, (size_t)c->cfg.max_points));
  const bool per_particle = method != GLH_RESAMPLE_SYSTEMATIC;
  if (rng_mode == GLH_RNG_HOST) {
    if (!u) return fail(GLH_E_INVALID, "GLH_RNG_HOST needs u (%s)", per_particle ? "[P][N]" : "[P]");
    if (per_particle) {
      if (!c->uj) CHK(dalloc(&c->uj, (size_t)c->cfg.max_points * c->cfg.max_particles));
      HIPCHK(hipMemcpyAsync(c->uj, u, (size_t)c->P * c->N * sizeof(double), hipMemcpyHostToDevice, c->stream));
    } else {
      HIPCHK(hipMemcpyAsync(c->u, u, (size_t)c->P * sizeof(double), hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
  } else if (rng_mode != GLH_RNG_PHILOX) {
    return fail(GLH_E_INVALID, "unknown rng_mode %d", rng_mode);
  }
  ResampleArgs a{};
  a.method = method;
  a.particles_in = c->particles[c->cur];
  a.weights_in = c->weights[c->cur];
  a.particles_out = c->particles[c->cur ^ 1];
  a.weights_out = c->weights[c->cur ^ 1];
  a.active = c->have_active ? c->active : nullptr;
  a.u = per_particle ? c->uj : c->u;
  a.idx_out = c->keep_idx ? c->idx : nullptr;
  a.n_draws = method == GLH_RESAMPLE_RESIDUAL ? c->resid_draws : nullptr;
  a.pt_status = c->pt_status;
  a.pt_err_frame = c->pt_err_frame;
  a.moments = c->moments + (size_t)c->frame * c->P * 12;  // fused particle_mean / sigma of frame c->frame
  a.leaf_off = c->leaf_off;
  a.leaf_len = c->leaf_len;
  a.ops = c->sum_ops;
  a.level_off = c->level_off;
  a.roots = c->roots;
  a.seed = seed;
  a.step = step;
  a.N = c->N;
  a.nleaves = c->nleaves;
  a.nnodes = c->nnodes;
  a.nlevels = c->nlevels;
  a.nroots = c->nroots;
  a.rng_mode = rng_mode;
  a.frame = c->frame;
  a.pt_base = c->pt_base;
  a.fast = use_fast(c);
  if (c->have_active) {
    // inactive points keep their state: copy their rows across before swapping buffers
    HIPCHK(hipMemcpyAsync(c->particles[c->cur ^ 1], c->particles[c->cur], (size_t)c->P * c->N * 6 * sizeof(double),
                          hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->weights[c->cur ^ 1], c->weights[c->cur], (size_t)c->P * c->N * sizeof(double),
                          hipMemcpyDeviceToDevice, c->stream));
  }
  {
    StageTimer t(c, ST_RESAMPLE);
    // c[N] | tree nodes | indices [N] u16 | repetitions [N] u16 (residual)
    size_t lds = ((size_t)c->N + c->nnodes) * sizeof(double) +
                 (method == GLH_RESAMPLE_RESIDUAL ? 2 : 1) * (size_t)c->N * sizeof(uint16_t);
    if (lds > 156 * 1024) return fail(GLH_E_UNSUPPORTED, "residual resampling: %d particles exceed the LDS-resident scan", c->N);
    hipLaunchKernelGGL(k_resample, dim3(c->P), dim3(BLK), lds, c->stream, a);
  }
  HIPCHK(hipGetLastError());
  c->cur ^= 1;
  // the fused moments cover every point only when no active mask was in force
  c->moments_frame = c->have_active ? -1 : c->frame;
  return GLH_OK;
}

extern "C" int glh_record_moments(glh_ctx* c, int frame) {
  CHK(need_seq(c));
  HIPCHK(hipSetDevice(c->cfg.device_id));
  if (frame < 0 || frame >= c->cfg.max_frames) return fail(GLH_E_INVALID, "frame %d outside [0, max_frames)", frame);
  if (c->moments_frame == frame && !c->have_active) return GLH_OK;  // written by the fused resample kernel
  return launch_moments(c, c->moments + (size_t)frame * c->P * 12, 12, 1);
}

// LDS plan of the fused kernel (glh_point.h): c[N] + region 2.  Two workgroups per CU when the
// state and a typical tile fit in half the LDS, otherwise one.

static bool fused_plan(const glh_ctx* c, int* r2_bytes, int mode = -1) {
  if (mode < 0) mode = c->fused;
  const int O = c->cfg.n_observers;
  if (O > PT_MAX_OBS || c->tw > PT_MAX_TILE || c->th > PT_MAX_TILE) return false;
  // (other median windows and bilinear sampling run on the general instantiations: fused_step)
  if (c->interp_k == 0) return false;  // orders other than (3, 3) / (1, 1): staged kernels

  int nb = 256;
  for (int o = 0; o < O; ++o) {
    if (c->obs[o].channels != 1 && c->obs[o].channels != 3) return false;
    // 16-bit and float frames: ranked in LDS / by linear buckets (glh_point.h: pt_tile_prep_wide, the float branch of
    // the observer pass) while a tile's pixel count fits a 16-bit key; wider workspaces: staged kernels
    if (c->obs[o].bits >= 16 && c->cfg.max_search_dim > 255) return false;
    if (c->obs[o].channels == 3 || c->obs[o].bits >= 16) nb = 766;  // (16-bit / float: the bucket table is no larger)
  }
  // c[N] and, behind region 2, the pairwise-sum plan
  const int cN = pt_align16(c->N * 8) + pt_align16(4 * pt_plan_ints(c->nleaves, c->nnodes, c->nlevels, c->nroots));
  // phase D/E: tree nodes | clast, then (over them) three rank tables of N uint16
  const int plan = std::max(pt_align16(c->nnodes * 8) + PT_BLK_BIG * 8, 3 * pt_align16(c->N * 2));
  // (the big-tile path parks the scratch surface of a dense spline fit behind the template tile)
  // (... and the scratch of a dense spline fit of the largest surface fitted that way, when that surface itself stays in its
  // HBM workspace: glh_point.h, the last branch of the observer pass)
  const int r2_min = std::max(plan, std::max(pt_small_bytes(c->tw, c->th, nb) + GLH_SPL_DENSE_NINV / 2 * 8,
                                             pt_align16(GLH_SPL_DENSE_MAX * GLH_SPL_DENSE_MAX * 8)));
  // a 48 x 48 search tile of this template in LDS (what a ~2 px cloud needs)
  // (the template CDF lies over the search tile while the LUT is made: no bytes of its own)
  const int typical = pt_small_bytes(c->tw, c->th, nb) + 48 * pt_search_ld(48) * 4 + pt_keys_count(48, 48) * 2;
  // (a context with rasters runs the instantiations that keep windows of them in static LDS)
  const int patch = c->rasters[0].z || c->rasters[1].z || c->rasters[2].z ? PT_PATCH_LDS : 0;
  // experiment (GLH_PT_LDS_HALF=bytes): another bound for the dynamic LDS of a workgroup that shares its compute unit --
  // 49 152 lets THREE 512-thread workgroups in (with a build at 6 waves per SIMD: -DPT_MINW=6)
  int lds_shared = PT_LDS_HALF;
  if (const char* e = getenv("GLH_PT_LDS_HALF")) lds_shared = std::max(16 * 1024, std::min(atoi(e), PT_LDS_HALF));
  const int lds_half = lds_shared - patch, lds_max = PT_LDS_MAX - patch;
  int r2;
  if (cN + std::max(r2_min, typical) <= lds_half)
    r2 = lds_half - cN;
  else
    r2 = std::min(lds_max - cN, 72 * 1024);
  if (getenv("GLH_PT_ONE_BLOCK")) r2 = std::min(lds_max - cN, 100 * 1024);  // experiment: 1 workgroup / CU
  if (mode == 2) r2 = r2_min;  // test hook: typical tiles no longer fit -> HBM workspaces
  if (r2 < r2_min || cN + r2 > lds_max) return false;  // (N beyond ~10 900: the staged kernels take the step)
  *r2_bytes = r2;
  return true;
}

// Fast arithmetic samples small fitted surfaces in per-cell power form (glh_math.h: spline_cell_row).  The fused
// kernel holds the cell table in region 2 and converts at most two rows per thread; the staged kernels apply the
// same bound, so that a given surface is evaluated by the same formula on either path (bit-identical results).
static int cell_cap(const glh_ctx* c) {
  int r2 = 0;
  if (!fused_plan(c, &r2, c->fused == 2 ? 2 : 1)) return 0;
  const int tb = c->N > 10 * PT_BLK ? PT_BLK_BIG : PT_BLK;
  return std::min(r2 / (GLH_CELL_LD * 8), 2 * tb / 4);
}

// The fused frame step (glh_point.h): ONE launch, one workgroup per point.
// `pt0`, `npts`, `on`: the block of points this launch updates and the stream it is enqueued on (glh_track runs the two
// halves of a large batch on two streams); `flip`: the last launch of the frame -- the state buffers change roles.
static int fused_step(glh_ctx* c, int frame, double tau, const int32_t* images, int rng_mode, const double* u,
                      uint64_t seed, int r2_bytes, int pt0 = 0, int npts = -1, hipStream_t on = nullptr,
                      bool flip = true) {
  const int O = c->cfg.n_observers;
  if (npts < 0) npts = c->P;
  if (!on) on = c->stream;
  if (rng_mode == GLH_RNG_HOST) {
    if (!u) return fail(GLH_E_INVALID, "GLH_RNG_HOST needs u [P]");
    HIPCHK(hipMemcpyAsync(c->u, u, (size_t)c->P * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  PointArgs a{};
  a.particles_in = c->particles[c->cur];
  a.particles_out = c->particles[c->cur ^ 1];
  a.weights_tmp = c->weights[c->cur];
  a.weights_out = c->weights[c->cur ^ 1];
  a.motion = c->motion;
  a.obs_mask = c->have_mask ? c->obs_mask : nullptr;
  a.normals = c->normals;
  a.u = c->u;
  a.uv = c->uv;
  a.box = c->box;
  a.obs_status = cur_status(c);
  a.tmpl_valid = c->tmpl_valid;
  a.tmpl_duv = c->tmpl_duv;
  a.tmpl_tile32 = c->tmpl_tile32;
  a.tmpl_hist_v = c->tmpl_hist_v;
  a.tmpl_hist_q = c->tmpl_hist_q;
  a.tmpl_hist_n = c->tmpl_hist_n;
  a.ws_search = c->search;
  a.ws_keys = c->ws_keys;
  a.ws_sse = c->sse;
  a.lu = c->lu;
  a.lu_off = c->lu_off;
  a.inv = c->spl_inv;
  a.poly = c->poly;
  a.idx_out = c->keep_idx ? c->idx : nullptr;
  for (int b = 0; b < 2; ++b)
    if (!c->uidx[b]) CHK(dalloc(&c->uidx[b], (size_t)c->cfg.max_points * c->cfg.max_particles));
  a.uidx_in = c->compact ? c->uidx[c->cur] : nullptr;
  a.uidx_out = c->uidx[c->cur ^ 1];
  a.stamps = c->stamps ? c->stamps + (size_t)pt0 * PT_NSTAMP : nullptr;  // (indexed by the launch's own block index)
  a.pt0 = pt0;
  a.moments = c->moments + (size_t)frame * c->P * 12;
  a.pt_status = c->pt_status;
  a.pt_err_frame = c->pt_err_frame;
  a.leaf_off = c->leaf_off;
  a.leaf_len = c->leaf_len;
  a.ops = c->sum_ops;
  a.level_off = c->level_off;
  a.roots = c->roots;
  a.seed = seed;
  a.step = (uint64_t)frame;
  a.tau = tau;
  for (int o = 0; o < O; ++o) {
    if (c->obs[o].bits >= 32) CHK(prepare_bins16(c, o, true));  // (the float workspace, on first use)
    fill_obs(c, o, images[o], &a.obs[o]);
    if ((int)c->obs[o].cams_host.size() != c->obs[o].n_images)
      return fail(GLH_E_STATE, "observer %d: cameras have not been set", o);
    a.cam[o] = c->obs[o].cams_host[images[o] >= 0 ? images[o] : 0];
    a.cam_flags[o] = cam_flags(a.cam[o]);
    a.inv2s2[o] = 1.0 / (2.0 * (c->obs[o].sigma * c->obs[o].sigma));
  }
  a.N = c->N;
  a.P = c->P;
  a.O = O;
  a.tw = c->tw;
  a.th = c->th;
  a.tile_cap = c->tile_cap;
  a.search_cap = c->search_cap;
  a.keys_cap = c->keys_cap;
  a.sse_cap = c->sse_cap;
  a.max_dim = c->cfg.max_search_dim;
  a.frame = frame;
  a.rng_mode = rng_mode;
  a.has_dem = c->has_dem;
  a.r2_bytes = r2_bytes;
  a.cell_cap = cell_cap(c);
  a.hp_rx = c->hp_rx;
  a.hp_ry = c->hp_ry;
  a.interp_k = c->interp_k;
  a.pt_base = c->pt_base;
  a.stop_at = -1;
  if (const char* stop = getenv("GLH_PT_STOP")) {  // diagnostic: "stamp:frame" -- that launch ends at that stamp
    int k = -1, f = -1;
    if (sscanf(stop, "%d:%d", &k, &f) == 2 && f == frame) a.stop_at = k;
  }
  a.surf = surfaces(c);
  a.nleaves = c->nleaves;
  a.nnodes = c->nnodes;
  a.nlevels = c->nlevels;
  a.nroots = c->nroots;
  {
    StageTimer t(c, ST_POINT_STEP, on);
    // (at least what phase F parks its partial sums in: every plan but fused_plan's small-LDS test hook has more)
    const size_t lds = std::max((size_t)pt_align16(c->N * 8) + r2_bytes +
                                    pt_align16(4 * pt_plan_ints(c->nleaves, c->nnodes, c->nlevels, c->nroots)),
                                (size_t)pt_park_bytes());
    const dim3 grid(npts);
    // N <= 5120: 512 threads, two workgroups per CU; larger N: 1024 threads, one per CU.  u of observer 0 in registers
    // (PPT per thread) and its v in c[] up to 10240 particles, both parked in LDS / the uv scratch beyond that and with
    // three or four observers.  Two observers keep observer 0 in registers as well (round 4: the first observer's pass is
    // peeled off the observer loop, so the registers are dead during the second observer's tile pipeline).
    const bool big = c->N > 10 * PT_BLK || c->force_tb == PT_BLK_BIG;
    const int tb = big ? PT_BLK_BIG : PT_BLK;
    const dim3 block(tb);
    int ppt = c->N <= 4 * tb ? 4 : (c->N <= 10 * tb ? 10 : 0);
    if (big && ppt == 4) ppt = 10;
    if (O == 2 && ppt == 4) ppt = 10;  // (the library carries <.., 10, 2> only)
    if (getenv("GLH_PT_UVLDS") || O >= 3) ppt = 0;
    if (O == 2 && getenv("GLH_PT_PPT0")) ppt = 0;  // diagnostic: the round-3 form (everything through the scratch)
    // the general instantiation: gridded surfaces and / or motion models other than CartesianMotion
    const bool fast = use_fast(c);
    // ... and, in fast arithmetic, everything the common instantiation is not compiled for (glh_point.h: COMMON): it
    // takes device Philox draws, compact input records, every observer on and unmasked, and cameras within
    // perspective + radial numerator (project_simple_fast)
    bool common = fast && rng_mode == GLH_RNG_PHILOX && c->compact && !c->have_mask;
    for (int o = 0; o < O; ++o) common &= a.obs[o].on && !(a.cam_flags[o] & CAM_F_NOT_SIMPLE);
    // (the contract is independent of the surfaces and motion models: the general code has its instantiation too)
    bool plain = c->hp_rx == 2 && c->hp_ry == 2 && c->interp_k == 3;  // the 5 x 5 median, bicubic sampling ...
    for (int o = 0; o < O; ++o) plain &= c->obs[o].bits == 8;            // ... of 8-bit frames
    const bool rast = c->rasters[0].z || c->rasters[1].z || c->rasters[2].z;  // (the instantiations with the raster samples)
    const bool surf = rast || !c->all_cartesian || (fast && !common) || !plain;
    int tbv = 512, nobsv = O;
    if (big) {
      tbv = 1024;
      if (ppt != 10) ppt = 0;
    }
    c->last_variant[0] = tbv; c->last_variant[1] = ppt; c->last_variant[2] = nobsv;
    c->last_variant[3] = (fast ? 1 : 0) | (surf ? 2 : 0) | (common ? 4 : 0) | (rast ? 8 : 0);
    // codes the library carries (glh_point_variants.h): exact / exact general / fast common / fast general / fast
    // general under the contract
    const void* kern = pt_kernel(tbv, ppt, nobsv, rast ? 2 : (surf ? 1 : 0), fast, surf ? common : fast);
    if (!kern) return fail(GLH_E_STATE, "no instantiation of the fused kernel for <%d, %d, %d>", tbv, ppt, nobsv);
    void* kargs[] = {(void*)&a};
    HIPCHK(hipLaunchKernel(kern, grid, block, kargs, lds, on));
  }
  HIPCHK(hipGetLastError());
  if (flip) {
    c->cur ^= 1;
    c->compact = true;
    c->moments_frame = frame;
  }
  return GLH_OK;
}

extern "C" int glh_get_residual_draws(glh_ctx* c, int32_t* draws) {
  CHK(need_seq(c));
  if (!draws) return fail(GLH_E_INVALID, "null argument");
  if (!c->resid_draws) return fail(GLH_E_STATE, "no residual resampling has run");
  DOWNLOAD(draws, c->resid_draws, (size_t)c->P, int32_t);
  return GLH_OK;
}

extern "C" int glh_record_covariances(glh_ctx* c, int frame) {
  CHK(need_seq(c));
  // (a run-length compact state -- what the fused step leaves -- is read through its record indices: expanding it first
  // would cost a pass over the state and send the next frame to the general instantiation)
  HIPCHK(hipSetDevice(c->cfg.device_id));
  if (frame < 0 || frame >= c->cfg.max_frames) return fail(GLH_E_INVALID, "frame %d outside [0, max_frames)", frame);
  if (!c->covariances) {
    const size_t n = (size_t)c->cfg.max_frames * c->cfg.max_points * 36;
    CHK(dalloc(&c->covariances, n));
    hipLaunchKernelGGL(k_fill_f64, dim3(256), dim3(256), 0, c->stream, c->covariances, n, (double)NAN);
    HIPCHK(hipGetLastError());
  }
  CovArgs a{};
  a.particles = c->particles[c->cur];
  a.weights = c->weights[c->cur];
  a.uidx = c->compact ? c->uidx[c->cur] : nullptr;
  a.active = c->have_active ? c->active : nullptr;
  a.out = c->covariances + (size_t)frame * c->P * 36;
  a.N = c->N;
  {
    StageTimer t(c, ST_MOMENTS);
    hipLaunchKernelGGL(k_covariance, dim3(c->P), dim3(BLK), 0, c->stream, a);
  }
  HIPCHK(hipGetLastError());
  return GLH_OK;
}

extern "C" int glh_get_covariances(glh_ctx* c, int frame0, int n_frames, double* out) {
  CHK(need_seq(c));
  if (!out || frame0 < 0 || n_frames <= 0 || frame0 + n_frames > c->cfg.max_frames)
    return fail(GLH_E_INVALID, "bad frame range");
  if (!c->covariances) return fail(GLH_E_STATE, "glh_record_covariances has not been called");
  DOWNLOAD(out, c->covariances + (size_t)frame0 * c->P * 36, (size_t)n_frames * c->P * 36, double);
  return GLH_OK;
}

extern "C" int glh_step(glh_ctx* c, int frame, double tau, const int32_t* images, int rng_mode,
                        const double* normals, const double* u, uint64_t seed) {
  CHK(need_seq(c));
  CHK(glh_set_frame(c, frame));
  CHK(check_images(c, images));
  HIPCHK(hipSetDevice(c->cfg.device_id));
  if (rng_mode == GLH_RNG_HOST) {
    if (!normals) return fail(GLH_E_INVALID, "GLH_RNG_HOST needs normals [P][N][3]");
    CHK(stage_normals(c, normals, (size_t)c->P * c->N * 3));
  } else if (rng_mode != GLH_RNG_PHILOX) {
    return fail(GLH_E_INVALID, "unknown rng_mode %d", rng_mode);
  }
  int r2_bytes = 0;
  if (c->fused && !c->have_active && !c->keep_sse && !c->have_extra && fused_plan(c, &r2_bytes))
    return fused_step(c, frame, tau, images, rng_mode, u, seed, r2_bytes);
  // one pass over the particle state: evolve, NaN test, project, bbox partials
  CHK(launch_evolve_project(c, true, tau, rng_mode, seed, (uint64_t)frame, images));
  CHK(update_weights_impl(c, images, true));
  CHK(glh_resample(c, rng_mode, u, seed, (uint64_t)frame));
  return glh_record_moments(c, frame);
}

// The frame loop of every track (tracker.py:326-357) in one call: n_frames consecutive glh_step updates with the
// device RNG, enqueued back to back on the context's stream (no host synchronisation in between).
extern "C" int glh_track(glh_ctx* c, int n_frames, const int32_t* frames, const double* taus, const int32_t* images,
                         uint64_t seed) {
  CHK(need_seq(c));
  if (n_frames <= 0 || !frames || !taus || !images)
    return fail(GLH_E_INVALID, "n_frames must be > 0 and the arrays non-null");
  const int O = c->cfg.n_observers;
  for (int k = 0; k < n_frames; ++k) {
    if (frames[k] < 0 || frames[k] >= c->cfg.max_frames)
      return fail(GLH_E_INVALID, "frame %d outside [0, max_frames)", frames[k]);
    CHK(check_images(c, images + (size_t)k * O));
  }
  // Two streams (round 4).  A launch of the fused step is rounds of workgroups -- 512 (or 256 of 1 024 threads) at a
  // time -- and between two launches the chip drains and refills: at C3 t(P) = 0.059 ms + 0.107 us x P, the constant
  // is 12 % of a frame.  The points are independent, so the batch is cut in two halves whose frame loops run on two
  // streams: while one half's launch drains, the other half's launch fills the idle compute units (C3 -6 %, C4 shard
  // -4 %; four ways: worse).  Same kernel, same per-point arithmetic: results are bit for bit those of one stream.
  int r2_bytes = 0;
  const bool fused_ok = c->fused && !c->have_active && !c->keep_sse && !c->have_extra && fused_plan(c, &r2_bytes);
  // Automatic (round 5): two streams as soon as the batch has more points than the chip has compute units.  Round 4 asked
  // for two full rounds of workgroups per half; but a launch ends with its slowest point (their lifetimes spread by 30 %),
  // the halves' launches drift apart and fill each other's tails, and a batch of 1.5 rounds no longer runs a half-empty
  // second one: same box, one -> two streams, C3 x 320 points -10.7 %, x 512 -9.4 %, x 768 -26 %, C5's per-GPU share (512
  // points) -8.5 %, C4 x 384 -26 %, C2 x 512 -9 %; at one workgroup per CU or fewer (C2's 256 points: +2 %, C4 x 256: +-0)
  // one stream (profiles/ab_r05/r5j23_streams_small.txt, r5j24_streams_threshold.txt).  Three streams: the same; four: worse.
  int ns = 1;
  if (fused_ok && !c->track_covariances) {
    if (c->track_streams >= 2) ns = c->track_streams;
    else if (c->track_streams == 0 && c->P > c->n_cus) ns = 2;
    if (const char* e = getenv("GLH_TRACK_STREAMS")) ns = atoi(e) >= 1 && atoi(e) <= 4 ? atoi(e) : ns;
    ns = std::min(ns, c->P);
  }
  c->last_track_streams = ns;
  // experiment (GLH_PT_BIG_FRAMES=n): the first n frames of the call run the 1 024-thread instantiation with the whole
  // CU's LDS (the wide search tiles after the prior fit without the HBM workspaces)
  int big_frames = 0, r2_big = r2_bytes;
  if (const char* e = getenv("GLH_PT_BIG_FRAMES")) {
    if (fused_ok && c->N <= 10 * PT_BLK && O <= 2) {
      big_frames = atoi(e);
      const int cN = pt_align16(c->N * 8) + pt_align16(4 * pt_plan_ints(c->nleaves, c->nnodes, c->nlevels, c->nroots));
      const int patch = c->rasters[0].z || c->rasters[1].z || c->rasters[2].z ? PT_PATCH_LDS : 0;
      r2_big = std::max(r2_bytes, std::min(PT_LDS_MAX - patch - cN, 100 * 1024));
    }
  }
  if (ns == 1) {
    // experiment (GLH_TRACK_GRAPH=1): the frame loop of a one-stream run as ONE hipGraph launch -- small batches (C2,
    // a GPU's share of C5) spend 15 % of a frame between two launches
    const bool graph = fused_ok && !c->track_covariances && getenv("GLH_TRACK_GRAPH") && n_frames > 1;
    if (graph) {
      HIPCHK(hipSetDevice(c->cfg.device_id));
      for (int b = 0; b < 2; ++b)  // (nothing may be allocated while the stream is being captured)
        if (!c->uidx[b]) CHK(dalloc(&c->uidx[b], (size_t)c->cfg.max_points * c->cfg.max_particles));
      for (int o = 0; o < O; ++o)
        if (c->obs[o].bits >= 32) CHK(prepare_bins16(c, o, true));
      CHK(drain_profile(c));
      if (c->track_graph) {
        HIPCHK(hipStreamSynchronize(c->stream));
        (void)hipGraphExecDestroy(c->track_graph);
        c->track_graph = nullptr;
      }
      HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed));
      c->capturing = true;
    }
    int rc = GLH_OK;
    for (int k = 0; k < n_frames && rc == GLH_OK; ++k) {
      c->force_tb = k < big_frames ? PT_BLK_BIG : 0;
      if (fused_ok && (graph || big_frames > 0)) {
        rc = glh_set_frame(c, frames[k]);
        if (rc == GLH_OK)
          rc = fused_step(c, frames[k], taus[k], images + (size_t)k * O, GLH_RNG_PHILOX, nullptr, seed,
                          k < big_frames ? r2_big : r2_bytes);
      } else {
        rc = glh_step(c, frames[k], taus[k], images + (size_t)k * O, GLH_RNG_PHILOX, nullptr, nullptr, seed);
      }
      if (rc == GLH_OK && c->track_covariances) rc = glh_record_covariances(c, frames[k]);
    }
    c->force_tb = 0;
    if (graph) {
      c->capturing = false;
      hipGraph_t g = nullptr;
      const hipError_t e1 = hipStreamEndCapture(c->stream, &g);
      if (rc != GLH_OK) {
        if (g) (void)hipGraphDestroy(g);
        return rc;
      }
      HIPCHK(e1);
      hipError_t e2 = hipGraphInstantiate(&c->track_graph, g, nullptr, nullptr, 0);
      (void)hipGraphDestroy(g);
      HIPCHK(e2);
      HIPCHK(hipGraphLaunch(c->track_graph, c->stream));
    }
    return rc;
  }
  HIPCHK(hipSetDevice(c->cfg.device_id));
  if (!c->ev_fork) HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
  hipStream_t on[4] = {c->stream, nullptr, nullptr, nullptr};
  for (int q = 1; q < ns; ++q) {
    if (!c->extra_streams[q - 1]) {
      HIPCHK(hipStreamCreateWithFlags(&c->extra_streams[q - 1], hipStreamNonBlocking));
      HIPCHK(hipEventCreateWithFlags(&c->ev_join[q - 1], hipEventDisableTiming));
    }
    on[q] = c->extra_streams[q - 1];
  }
  // the other streams start behind everything enqueued so far, and the context's stream ends behind them
  HIPCHK(hipEventRecord(c->ev_fork, c->stream));
  for (int q = 1; q < ns; ++q) HIPCHK(hipStreamWaitEvent(on[q], c->ev_fork, 0));
  int rc = GLH_OK;
  for (int k = 0; k < n_frames && rc == GLH_OK; ++k) {
    rc = glh_set_frame(c, frames[k]);
    for (int q = 0; q < ns && rc == GLH_OK; ++q) {
      const int p0 = (int)((int64_t)c->P * q / ns), p1 = (int)((int64_t)c->P * (q + 1) / ns);
      c->force_tb = k < big_frames ? PT_BLK_BIG : 0;
      rc = fused_step(c, frames[k], taus[k], images + (size_t)k * O, GLH_RNG_PHILOX, nullptr, seed,
                      k < big_frames ? r2_big : r2_bytes, p0, p1 - p0, on[q], q == ns - 1);
    }
  }
  c->force_tb = 0;
  for (int q = 1; q < ns; ++q) {
    (void)hipEventRecord(c->ev_join[q - 1], on[q]);
    (void)hipStreamWaitEvent(c->stream, c->ev_join[q - 1], 0);
  }
  return rc;
}

// Streams of glh_track's frame loop: 0 automatic, 1 one stream, 2 two streams whenever the fused step runs.
extern "C" int glh_set_track_streams(glh_ctx* c, int n) {
  if (!c) return fail(GLH_E_INVALID, "null context");
  if (n < 0 || n > 4) return fail(GLH_E_INVALID, "streams must be 0 (automatic) or 1 .. 4");
  c->track_streams = n;
  return GLH_OK;
}

extern "C" int glh_debug_last_track_streams(glh_ctx* c, int* n) {
  if (!c || !n) return fail(GLH_E_INVALID, "null argument");
  *n = c->last_track_streams;
  return GLH_OK;
}

// glh_track also records the particle covariances of every frame it runs (Tracker.track(return_covariances=True),
// tracker.py:307-308, :352): 0 (default) or 1.
extern "C" int glh_track_covariances(glh_ctx* c, int on) {
  if (!c) return fail(GLH_E_INVALID, "null context");
  c->track_covariances = on != 0;
  return GLH_OK;
}

extern "C" int glh_measure_copy_bandwidth(glh_ctx* c, uint64_t bytes, int iters, double* gbps) {
  if (!c || !gbps || bytes == 0 || iters <= 0) return fail(GLH_E_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->cfg.device_id));
  uint8_t *src = nullptr, *dst = nullptr;
  CHK(dalloc(&src, (size_t)bytes));
  if (dalloc(&dst, (size_t)bytes) != GLH_OK) {
    dfree(src);
    return GLH_E_NOMEM;
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  float ms = 0.0f;
  hipError_t err = hipMemsetAsync(src, 1, bytes, c->stream);
  if (err == hipSuccess) err = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream);  // warm-up
  if (err == hipSuccess) err = hipEventCreate(&e0);
  if (err == hipSuccess) err = hipEventCreate(&e1);
  if (err == hipSuccess) err = hipEventRecord(e0, c->stream);
  for (int k = 0; k < iters && err == hipSuccess; ++k)
    err = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream);
  if (err == hipSuccess) err = hipEventRecord(e1, c->stream);
  if (err == hipSuccess) err = hipEventSynchronize(e1);
  if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  dfree(src);
  dfree(dst);
  if (err != hipSuccess) return fail(GLH_E_HIP, "copy bandwidth measurement failed: %s", hipGetErrorString(err));
  *gbps = 2.0 * (double)bytes * iters / ((double)ms * 1e-3) / 1e9;
  return GLH_OK;
}

extern "C" int glh_debug_last_variant(glh_ctx* c, int32_t* variant) {
  if (!c || !variant) return fail(GLH_E_INVALID, "null argument");
  for (int k = 0; k < 4; ++k) variant[k] = c->last_variant[k];
  return GLH_OK;
}

extern "C" int glh_debug_draws(glh_ctx* c, int kind, uint64_t seed, uint64_t step, double* out) {
  CHK(need_seq(c));
  if (!out || kind < 0 || kind > 2) return fail(GLH_E_INVALID, "kind must be 0 (init), 1 (evolve) or 2 (resample offset)");
  HIPCHK(hipSetDevice(c->cfg.device_id));
  const size_t per = kind == 0 ? 6 : (kind == 1 ? 3 : 0);
  const size_t count = kind == 2 ? (size_t)c->P : (size_t)c->P * c->N * per;
  double* buf = nullptr;
  CHK(dalloc(&buf, count));
  DrawsArgs a{};
  a.out = buf;
  a.seed = seed;
  a.step = step;
  a.kind = kind;
  a.N = c->N;
  a.P = c->P;
  a.pt_base = c->pt_base;
  hipLaunchKernelGGL(k_debug_draws, dim3(kind == 2 ? 1 : (c->N + BLK - 1) / BLK, c->P), dim3(BLK), 0, c->stream, a);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(out, buf, count * sizeof(double), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  dfree(buf);
  if (e != hipSuccess) return fail(GLH_E_HIP, "glh_debug_draws failed: %s", hipGetErrorString(e));
  return GLH_OK;
}

extern "C" int glh_debug_phase_stamps(glh_ctx* c, uint64_t* stamps) {
  CHK(need_seq(c));
  if (!stamps) return fail(GLH_E_INVALID, "null argument");
  HIPCHK(hipSetDevice(c->cfg.device_id));
  const size_t n = (size_t)c->P * PT_NSTAMP;
  if (!c->stamps) {
    // first call arms the stamps; the next fused steps record them
    CHK(dalloc(&c->stamps, (size_t)c->cfg.max_points * PT_NSTAMP));
    HIPCHK(hipMemset(c->stamps, 0, (size_t)c->cfg.max_points * PT_NSTAMP * sizeof(unsigned long long)));
    memset(stamps, 0, n * sizeof(uint64_t));
    return GLH_OK;
  }
  DOWNLOAD(stamps, c->stamps, n, unsigned long long);
  return GLH_OK;
}

extern "C" int glh_set_highpass(glh_ctx* c, int size_x, int size_y) {
  if (!c) return fail(GLH_E_INVALID, "null context");
  if (size_x < 1 || size_y < 1 || size_x > 7 || size_y > 7 || !(size_x & 1) || !(size_y & 1))
    return fail(GLH_E_UNSUPPORTED, "high-pass window %d x %d: sizes must be odd and at most 7", size_x, size_y);
  c->hp_rx = size_x / 2;
  c->hp_ry = size_y / 2 | (GLH_HP_MODE(c->hp_ry) << 4);  // (the boundary mode rides in the half height: glh_math.h)
  return GLH_OK;
}

extern "C" int glh_set_highpass_mode(glh_ctx* c, int mode) {
  if (!c) return fail(GLH_E_INVALID, "null context");
  if (mode < GLH_HP_REFLECT || mode > GLH_HP_WRAP)
    return fail(GLH_E_UNSUPPORTED, "high-pass boundary mode %d: 0 reflect, 1 nearest, 2 mirror, 3 wrap", mode);
  c->hp_ry = GLH_HP_RY(c->hp_ry) | (mode << 4);
  return GLH_OK;
}

extern "C" int glh_set_interpolation(glh_ctx* c, int kx, int ky) {
  if (!c) return fail(GLH_E_INVALID, "null context");
  if (kx < 1 || kx > GLH_SPL_KMAX || ky < 1 || ky > GLH_SPL_KMAX)
    return fail(GLH_E_UNSUPPORTED, "interpolation orders (%d, %d): each of 1 .. %d (RectBivariateSpline)", kx, ky,
                GLH_SPL_KMAX);
  HIPCHK(hipSetDevice(c->cfg.device_id));
  for (int q = 0; q < 2; ++q) {
    dfree(c->glu[q]);
    dfree(c->glu_off[q]);
  }
  c->interp_kx = kx;
  c->interp_ky = ky;
  c->interp_k = (kx == 3 && ky == 3) ? 3 : ((kx == 1 && ky == 1) ? 1 : 0);
  if (c->interp_k == 0) {
    // any other orders: banded factors of the degree-k collocation matrices for every surface side up to the workspace's
    const int maxn = c->cfg.max_search_dim;
    for (int q = 0; q < 2; ++q) {
      const int k = q == 0 ? kx : ky;
      std::vector<int64_t> off(maxn + 1, 0);
      int64_t total = 0;
      for (int n = k + 1; n <= maxn; ++n) {
        off[n] = total;
        total += (int64_t)(2 * k + 1) * n;
      }
      std::vector<double> lu((size_t)total);
      for (int n = k + 1; n <= maxn; ++n)
        if (!spline_lu_general(n, k, lu.data() + off[n]))
          return fail(GLH_E_STATE, "spline collocation matrix of size %d, degree %d has support outside its band", n, k);
      CHK(dalloc(&c->glu[q], (size_t)total));
      CHK(dalloc(&c->glu_off[q], (size_t)maxn + 1));
      HIPCHK(hipMemcpy(c->glu[q], lu.data(), (size_t)total * sizeof(double), hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(c->glu_off[q], off.data(), (size_t)(maxn + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    }
  }
  return GLH_OK;
}

extern "C" int glh_set_math(glh_ctx* c, int mode) {
  if (!c) return fail(GLH_E_INVALID, "null context");
  if (mode != GLH_MATH_EXACT && mode != GLH_MATH_FAST) return fail(GLH_E_INVALID, "mode must be GLH_MATH_EXACT or GLH_MATH_FAST");
  c->fast_math = mode == GLH_MATH_FAST;
  return GLH_OK;
}

extern "C" int glh_set_fused(glh_ctx* c, int on) {
  if (!c) return fail(GLH_E_INVALID, "null context");
  if (on < 0 || on > 2) return fail(GLH_E_INVALID, "mode must be 0, 1 or 2");
  c->fused = on;
  return GLH_OK;
}

// ------------------------------------------------------------------------------------------
// results
// ------------------------------------------------------------------------------------------
extern "C" int glh_get_moments(glh_ctx* c, int frame0, int n_frames, double* out) {
  CHK(need_seq(c));
  if (!out || frame0 < 0 || n_frames <= 0 || frame0 + n_frames > c->cfg.max_frames)
    return fail(GLH_E_INVALID, "bad frame range");
  DOWNLOAD(out, c->moments + (size_t)frame0 * c->P * 12, (size_t)n_frames * c->P * 12, double);
  return GLH_OK;
}

extern "C" int glh_get_tracks(glh_ctx* c, int frame0, int n_frames, double* means, double* sigmas) {
  CHK(need_seq(c));
  if (!means || !sigmas || frame0 < 0 || n_frames <= 0 || frame0 + n_frames > c->cfg.max_frames)
    return fail(GLH_E_INVALID, "bad frame range or null output");
  HIPCHK(hipSetDevice(c->cfg.device_id));
  const size_t n = (size_t)n_frames * c->P * 6;
  if (c->tracks_tmp_n < 2 * n) {
    dfree(c->tracks_tmp);
    c->tracks_tmp = nullptr;
    c->tracks_tmp_n = 0;
    CHK(dalloc(&c->tracks_tmp, 2 * n));
    c->tracks_tmp_n = 2 * n;
  }
  hipLaunchKernelGGL(k_tracks_layout, dim3((unsigned)((n + BLK - 1) / BLK)), dim3(BLK), 0, c->stream,
                     c->moments + (size_t)frame0 * c->P * 12, n_frames, c->P, c->tracks_tmp, c->tracks_tmp + n);
  HIPCHK(hipGetLastError());
  DOWNLOAD(means, c->tracks_tmp, n, double);
  DOWNLOAD(sigmas, c->tracks_tmp + n, n, double);
  return GLH_OK;
}

extern "C" int glh_get_moments_device(glh_ctx* c, void** p, uint64_t* bytes) {
  CHK(need_seq(c));
  if (!p || !bytes) return fail(GLH_E_INVALID, "null argument");
  *p = c->moments;
  *bytes = (uint64_t)c->cfg.max_frames * c->P * 12 * sizeof(double);
  return GLH_OK;
}

extern "C" int glh_get_template(glh_ctx* c, int o, int pt, int32_t* box, double* duv, double* tile,
                                double* hv, double* hq, int32_t* hn) {
  CHK(need_seq(c));
  CHK(check_obs(c, o));
  if (pt < 0 || pt >= c->P) return fail(GLH_E_INVALID, "point %d out of range", pt);
  const size_t slot = (size_t)o * c->P + pt;
  int32_t valid = 0;
  DOWNLOAD(&valid, c->tmpl_valid + slot, 1, int32_t);
  if (!valid) return fail(GLH_E_STATE, "no template for observer %d, point %d", o, pt);
  int32_t n = 0;
  DOWNLOAD(&n, c->tmpl_hist_n + slot, 1, int32_t);
  if (hn) *hn = n;
  if (box) DOWNLOAD(box, c->tmpl_box + slot * 4, 4, int32_t);
  if (duv) DOWNLOAD(duv, c->tmpl_duv + slot * 2, 2, double);
  if (tile) DOWNLOAD(tile, c->tmpl_tile64 + slot * c->tile_cap, (size_t)c->tw * c->th, double);
  if (hv) DOWNLOAD(hv, c->tmpl_hist_v + slot * c->tile_cap, (size_t)n, double);
  if (hq) DOWNLOAD(hq, c->tmpl_hist_q + slot * c->tile_cap, (size_t)n, double);
  return GLH_OK;
}

extern "C" int glh_get_likelihood_debug(glh_ctx* c, int o, int pt, double* uv, int32_t* box, float* search,
                                        double* sse) {
  CHK(need_seq(c));
  CHK(check_obs(c, o));
  if (pt < 0 || pt >= c->P) return fail(GLH_E_INVALID, "point %d out of range", pt);
  const size_t slot = (size_t)o * c->P + pt;
  int32_t st = 0;
  DOWNLOAD(&st, cur_status(c) + slot, 1, int32_t);
  if (uv) DOWNLOAD(uv, c->uv + slot * c->N * 2, (size_t)c->N * 2, double);
  if (st != GLH_OBS_OK) {
    if (box) box[0] = box[1] = box[2] = box[3] = -1;
    return GLH_OK;
  }
  int32_t b[4];
  DOWNLOAD(b, c->box + slot * 4, 4, int32_t);
  if (box) memcpy(box, b, sizeof b);
  const int ws = b[2] - b[0], hs = b[3] - b[1];
  if (search) DOWNLOAD(search, c->search + slot * (size_t)c->search_cap, (size_t)ws * hs, float);
  if (sse) {
    if (!c->keep_sse || !c->sse_copy) return fail(GLH_E_STATE, "SSE capture is off (glh_set_debug)");
    DOWNLOAD(sse, c->sse_copy + slot * (size_t)c->sse_cap, (size_t)(ws - c->tw + 1) * (hs - c->th + 1), double);
  }
  return GLH_OK;
}

extern "C" int glh_profile_enable(glh_ctx* c, int on) {
  if (!c) return fail(GLH_E_INVALID, "null context");
  CHK(drain_profile(c));
  c->profiling = on != 0;
  return GLH_OK;
}
extern "C" int glh_profile_reset(glh_ctx* c) {
  if (!c) return fail(GLH_E_INVALID, "null context");
  CHK(drain_profile(c));
  for (int i = 0; i < ST_COUNT; ++i) {
    c->ms[i] = 0;
    c->launches[i] = 0;
    c->launch_ms[i].clear();
    if (c->span_a[i]) c->pool.push_back(c->span_a[i]);
    c->span_a[i] = nullptr;
    c->span_ms[i] = 0.f;
  }
  return GLH_OK;
}
extern "C" int glh_profile_get(glh_ctx* c, double* ms, int64_t* launches) {
  if (!c) return fail(GLH_E_INVALID, "null context");
  CHK(drain_profile(c));
  for (int i = 0; i < ST_COUNT; ++i) {
    if (ms) ms[i] = c->ms[i];
    if (launches) launches[i] = c->launches[i];
  }
  return GLH_OK;
}

extern "C" int glh_profile_get_span(glh_ctx* c, int stage, double* ms) {
  if (!c || !ms || stage < 0 || stage >= ST_COUNT) return fail(GLH_E_INVALID, "bad argument");
  CHK(drain_profile(c));
  *ms = (double)c->span_ms[stage];
  return GLH_OK;
}

extern "C" int glh_profile_get_launches(glh_ctx* c, int stage, double* ms, int cap, int* n) {
  if (!c || !n || stage < 0 || stage >= ST_COUNT || cap < 0 || (cap > 0 && !ms))
    return fail(GLH_E_INVALID, "bad argument");
  CHK(drain_profile(c));
  const auto& v = c->launch_ms[stage];
  *n = (int)v.size();
  for (int i = 0; i < cap && i < (int)v.size(); ++i) ms[i] = (double)v[i];
  return GLH_OK;
}

// ------------------------------------------------------------------------------------------
// multi-GPU: one process per GPU, one gather at the end of a sequence (glh_comm.h)
// ------------------------------------------------------------------------------------------
#define NCCLCHK(api, expr)                                                                         \
  do {                                                                                             \
    ncclResult_t r_ = (expr);                                                                      \
    if (r_ != ncclSuccess)                                                                         \
      return fail(GLH_E_COMM, "%s failed: %s (%s:%d)", #expr, (api)->GetErrorString(r_), __FILE__, __LINE__); \
  } while (0)

extern "C" int glh_comm_unique_id(char* id) {
  if (!id) return fail(GLH_E_INVALID, "id is null");
  RcclApi* api = rccl_api();
  if (!api) return fail(GLH_E_COMM, "%s", rccl_load_error());
  static_assert(GLH_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id is RCCL's ncclUniqueId");
  ncclUniqueId u;
  NCCLCHK(api, api->GetUniqueId(&u));
  memcpy(id, u.internal, GLH_COMM_ID_BYTES);
  return GLH_OK;
}

static int comm_free(glh_ctx* c) {
  if (!c->comm) return GLH_OK;
  Comm* k = c->comm;
  c->comm = nullptr;
  (void)hipSetDevice(c->cfg.device_id);
  (void)hipStreamSynchronize(c->stream);
  RcclApi* api = rccl_api();
  if (k->comm && api) (void)api->CommDestroy(k->comm);
  dfree(k->stage);
  dfree(k->scalar);
  delete k;
  return GLH_OK;
}

extern "C" int glh_comm_destroy(glh_ctx* c) {
  if (!c) return fail(GLH_E_INVALID, "null context");
  return comm_free(c);
}

extern "C" int glh_comm_init(glh_ctx* c, const char* id, int rank, int world) {
  if (!c || !id) return fail(GLH_E_INVALID, "null argument");
  if (world < 1 || rank < 0 || rank >= world) return fail(GLH_E_INVALID, "need 0 <= rank < world (got %d, %d)", rank, world);
  RcclApi* api = rccl_api();
  if (!api) return fail(GLH_E_COMM, "%s", rccl_load_error());
  CHK(comm_free(c));
  HIPCHK(hipSetDevice(c->cfg.device_id));
  Comm* k = new (std::nothrow) Comm();
  if (!k) return fail(GLH_E_NOMEM, "out of host memory");
  k->rank = rank;
  k->world = world;
  ncclUniqueId u;
  memcpy(u.internal, id, GLH_COMM_ID_BYTES);
  ncclResult_t r = api->CommInitRank(&k->comm, world, u, rank);
  if (r != ncclSuccess) {
    delete k;
    return fail(GLH_E_COMM, "ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, world, c->cfg.device_id,
                api->GetErrorString(r));
  }
  if (dalloc(&k->scalar, 2) != GLH_OK) {
    (void)api->CommDestroy(k->comm);
    delete k;
    return GLH_E_NOMEM;
  }
  c->comm = k;
  return GLH_OK;
}

static int need_comm(glh_ctx* c) {
  if (!c) return fail(GLH_E_INVALID, "null context");
  if (!c->comm) return fail(GLH_E_STATE, "glh_comm_init has not been called");
  return GLH_OK;
}

// max over the ranks of one host double (the bench's max-over-ranks wall time); doubles as a barrier: it
// returns once every rank's stream has reached the reduction
extern "C" int glh_comm_max_f64(glh_ctx* c, double* value) {
  CHK(need_comm(c));
  if (!value) return fail(GLH_E_INVALID, "value is null");
  RcclApi* api = rccl_api();
  Comm* k = c->comm;
  HIPCHK(hipSetDevice(c->cfg.device_id));
  HIPCHK(hipMemcpyAsync(k->scalar, value, sizeof(double), hipMemcpyHostToDevice, c->stream));
  NCCLCHK(api, api->AllReduce(k->scalar, k->scalar + 1, 1, ncclDouble, ncclMax, k->comm, c->stream));
  HIPCHK(hipMemcpyAsync(value, k->scalar + 1, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return GLH_OK;
}

extern "C" int glh_comm_barrier(glh_ctx* c) {
  double v = 0.0;
  CHK(join_uploads(c));
  return glh_comm_max_f64(c, &v);
}

// Gather of the posterior moments (and the per-point status words) to rank `root`:
//   every rank sends frames [frame0, frame0 + n_frames) of its moments history, a contiguous block of
//   n_frames * P_rank * 12 doubles, and its P_rank status words; the root receives the blocks rank after rank,
//   out = | rank 0: [n_frames][P_0][12] | rank 1: [n_frames][P_1][12] | ... (doubles), status = | P_0 | P_1 | ...
// One ncclGroup of sends / receives on the context's stream (the root's own block goes through the same
// send / receive pair); the root then downloads to the host buffers.  points_per_rank [world].
extern "C" int glh_gather_moments(glh_ctx* c, int root, int frame0, int n_frames, const int32_t* points_per_rank,
                                  double* out, uint32_t* status) {
  CHK(need_comm(c));
  CHK(need_seq(c));
  Comm* k = c->comm;
  RcclApi* api = rccl_api();
  if (root < 0 || root >= k->world || !points_per_rank) return fail(GLH_E_INVALID, "bad root / points_per_rank");
  if (frame0 < 0 || n_frames <= 0 || frame0 + n_frames > c->cfg.max_frames) return fail(GLH_E_INVALID, "bad frame range");
  if (points_per_rank[k->rank] != c->P)
    return fail(GLH_E_INVALID, "points_per_rank[%d] = %d, but this context tracks %d points", k->rank,
                points_per_rank[k->rank], c->P);
  const bool is_root = k->rank == root;
  HIPCHK(hipSetDevice(c->cfg.device_id));
  size_t total_pts = 0;
  for (int r = 0; r < k->world; ++r) {
    if (points_per_rank[r] < 0) return fail(GLH_E_INVALID, "points_per_rank[%d] < 0", r);
    total_pts += (size_t)points_per_rank[r];
  }
  const size_t mom_doubles = total_pts * (size_t)n_frames * 12;
  const size_t st_off = (mom_doubles * 8 + 15) & ~(size_t)15;  // bytes
  if (is_root) {
    const size_t need = st_off + total_pts * 4;
    if (k->stage_bytes < need) {
      dfree(k->stage);
      k->stage_bytes = 0;
      CHK(dalloc((uint8_t**)&k->stage, need));
      k->stage_bytes = need;
    }
  }
  const double* mine = c->moments + (size_t)frame0 * c->P * 12;
  NCCLCHK(api, api->GroupStart());
  ncclResult_t r1 = api->Send(mine, (size_t)n_frames * c->P * 12, ncclDouble, root, k->comm, c->stream);
  ncclResult_t r2 = api->Send(c->pt_status, (size_t)c->P, ncclUint32, root, k->comm, c->stream);
  ncclResult_t r3 = ncclSuccess;
  if (is_root) {
    double* dst = k->stage;
    uint32_t* sdst = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(k->stage) + st_off);
    for (int r = 0; r < k->world && r3 == ncclSuccess; ++r) {
      const size_t pr = (size_t)points_per_rank[r];
      r3 = api->Recv(dst, pr * n_frames * 12, ncclDouble, r, k->comm, c->stream);
      if (r3 == ncclSuccess) r3 = api->Recv(sdst, pr, ncclUint32, r, k->comm, c->stream);
      dst += pr * n_frames * 12;
      sdst += pr;
    }
  }
  ncclResult_t r4 = api->GroupEnd();
  for (ncclResult_t r : {r1, r2, r3, r4})
    if (r != ncclSuccess) return fail(GLH_E_COMM, "moments gather failed: %s", api->GetErrorString(r));
  if (is_root) {
    k->gathered_doubles = mom_doubles;
    k->gathered_points = total_pts;
    k->gathered_status_off = st_off;
    if (out) HIPCHK(hipMemcpyAsync(out, k->stage, mom_doubles * 8, hipMemcpyDeviceToHost, c->stream));
    if (out && status)
      HIPCHK(hipMemcpyAsync(status, reinterpret_cast<uint8_t*>(k->stage) + st_off, total_pts * 4, hipMemcpyDeviceToHost,
                            c->stream));
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  return GLH_OK;
}

// Host copy of what the last glh_gather_moments left on the root's device (when it was called with out = NULL: the
// exchange itself is then all that a caller times).
extern "C" int glh_get_gathered(glh_ctx* c, double* out, uint32_t* status) {
  CHK(need_comm(c));
  Comm* k = c->comm;
  if (!k->stage || !k->gathered_doubles) return fail(GLH_E_STATE, "no gathered moments on this rank");
  if (!out) return fail(GLH_E_INVALID, "out is null");
  HIPCHK(hipSetDevice(c->cfg.device_id));
  HIPCHK(hipMemcpyAsync(out, k->stage, k->gathered_doubles * 8, hipMemcpyDeviceToHost, c->stream));
  if (status)
    HIPCHK(hipMemcpyAsync(status, reinterpret_cast<uint8_t*>(k->stage) + k->gathered_status_off, k->gathered_points * 4,
                          hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return GLH_OK;
}

// ------------------------------------------------------------------------------------------
// stateless stage hooks (parity tests): explicit inputs -> one kernel -> outputs
// ------------------------------------------------------------------------------------------
namespace {
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  int alloc(size_t bytes) {
    if (bytes == 0) bytes = 1;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return fail(GLH_E_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return GLH_OK;
  }
  int up(const void* src, size_t bytes) {
    CHK(alloc(bytes));
    HIPCHK(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
    return GLH_OK;
  }
  int down(void* dst, size_t bytes) {
    HIPCHK(hipMemcpy(dst, p, bytes, hipMemcpyDeviceToHost));
    return GLH_OK;
  }
  template <typename T>
  T* as() {
    return (T*)p;
  }
};
int finish() {
  HIPCHK(hipGetLastError());
  HIPCHK(hipDeviceSynchronize());
  return GLH_OK;
}
}  // namespace

static int stage_project_impl(int dev, const double* cam, const double* xyz, int n, double* uv, int directions,
                              double* depth = nullptr) {
  if (!cam || !xyz || !uv || n <= 0) return fail(GLH_E_INVALID, "bad argument");
  HIPCHK(hipSetDevice(dev));
  CamDev cd;
  expand_camera(cam, &cd);
  DevBuf dc, dx, du, dd;
  CHK(dc.up(&cd, sizeof cd));
  CHK(dx.up(xyz, (size_t)n * 3 * sizeof(double)));
  CHK(du.alloc((size_t)n * 2 * sizeof(double)));
  if (depth) CHK(dd.alloc((size_t)n * sizeof(double)));
  hipLaunchKernelGGL(k_project_points, dim3((n + BLK - 1) / BLK), dim3(BLK), 0, 0, dc.as<CamDev>(),
                     dx.as<double>(), n, du.as<double>(), directions, depth ? dd.as<double>() : nullptr);
  CHK(finish());
  if (depth) CHK(dd.down(depth, (size_t)n * sizeof(double)));
  return du.down(uv, (size_t)n * 2 * sizeof(double));
}

extern "C" int glh_stage_project_depth(int dev, const double* cam, const double* xyz, int n, int directions,
                                       double* uv, double* depth) {
  if (!depth) return fail(GLH_E_INVALID, "bad argument");
  if (cam && cam[23] != 0.0) return fail(GLH_E_INVALID, "depth is a camera notion (not a raster grid)");
  return stage_project_impl(dev, cam, xyz, n, uv, directions ? 1 : 0, depth);
}

extern "C" int glh_stage_project(int dev, const double* cam, const double* xyz, int n, double* uv) {
  return stage_project_impl(dev, cam, xyz, n, uv, 0);
}

extern "C" int glh_stage_project_directions(int dev, const double* cam, const double* xyz, int n, double* uv) {
  if (cam && cam[23] != 0.0) return fail(GLH_E_INVALID, "ray directions are a camera notion (not a raster grid)");
  return stage_project_impl(dev, cam, xyz, n, uv, 1);
}

extern "C" int glh_stage_unproject(int dev, const double* cam, const double* uv, int n, const double* depth,
                                   int n_depth, int directions, double* xyz) {
  if (!cam || !uv || !xyz || n <= 0) return fail(GLH_E_INVALID, "bad argument");
  if (cam[23] != 0.0) return fail(GLH_E_UNSUPPORTED, "uv_to_xyz of a raster grid is not built");
  if (depth && n_depth != 1 && n_depth != n) return fail(GLH_E_INVALID, "depth must have 1 or n entries");
  HIPCHK(hipSetDevice(dev));
  CamDev cd;
  expand_camera(cam, &cd);
  DevBuf dc, du, dd, dx;
  CHK(dc.up(&cd, sizeof cd));
  CHK(du.up(uv, (size_t)n * 2 * sizeof(double)));
  if (depth) CHK(dd.up(depth, (size_t)n_depth * sizeof(double)));
  CHK(dx.alloc((size_t)n * 3 * sizeof(double)));
  hipLaunchKernelGGL(k_unproject_points, dim3((n + BLK - 1) / BLK), dim3(BLK), 0, 0, dc.as<CamDev>(),
                     du.as<double>(), n, depth ? dd.as<double>() : nullptr, n_depth, directions, dx.as<double>());
  CHK(finish());
  return dx.down(xyz, (size_t)n * 3 * sizeof(double));
}

static int check_box(const int32_t* box, int width, int height) {
  if (!box || box[0] < 0 || box[1] < 0 || box[2] > width || box[3] > height || box[2] <= box[0] || box[3] <= box[1])
    return fail(GLH_E_INVALID, "box outside the frame");
  return GLH_OK;
}

static int check_highpass(int size_x, int size_y) {
  if (size_x < 1 || size_y < 1 || size_x > 7 || size_y > 7 || !(size_x & 1) || !(size_y & 1))
    return fail(GLH_E_UNSUPPORTED, "high-pass window %d x %d: sizes must be odd and at most 7", size_x, size_y);
  return GLH_OK;
}

extern "C" int glh_stage_template(int dev, const uint8_t* frame, int width, int height, int channels,
                                  const int32_t* box, double* tile, double* hv, double* hq, int32_t* hn) {
  return glh_stage_template_highpass(dev, frame, width, height, channels, box, 5, 5, GLH_HP_REFLECT, tile, hv, hq, hn);
}

extern "C" int glh_stage_template_highpass(int dev, const uint8_t* frame, int width, int height, int channels,
                                           const int32_t* box, int size_x, int size_y, int mode, double* tile, double* hv,
                                           double* hq, int32_t* hn) {
  if (!frame || !tile || !hv || !hq || !hn) return fail(GLH_E_INVALID, "null argument");
  CHK(check_highpass(size_x, size_y));
  if (mode < GLH_HP_REFLECT || mode > GLH_HP_WRAP) return fail(GLH_E_UNSUPPORTED, "high-pass boundary mode %d", mode);
  if (channels != 1 && channels != 3) return fail(GLH_E_UNSUPPORTED, "1 or 3 channels");
  CHK(check_box(box, width, height));
  HIPCHK(hipSetDevice(dev));
  const size_t n = (size_t)(box[2] - box[0]) * (box[3] - box[1]);
  if (n * 2 > 60000) return fail(GLH_E_INVALID, "template too large for the test hook");
  DevBuf df, t64, t32, dv, dq, dn;
  CHK(df.up(frame, (size_t)width * height * channels));
  CHK(t64.alloc(n * 8)); CHK(t32.alloc(n * 4)); CHK(dv.alloc(n * 8)); CHK(dq.alloc(n * 8)); CHK(dn.alloc(4));
  TemplateBoxArgs a{};
  a.frame = df.as<uint8_t>();
  a.width = width;
  a.channels = channels;
  for (int k = 0; k < 4; ++k) a.box[k] = box[k];
  a.out.tile64 = t64.as<double>();
  a.out.tile32 = t32.as<float>();
  a.out.hist_v = dv.as<double>();
  a.out.hist_q = dq.as<double>();
  a.out.hist_n = dn.as<int32_t>();
  a.hp_rx = size_x / 2;
  a.hp_ry = size_y / 2 | (mode << 4);
  hipLaunchKernelGGL(k_template_from_box, dim3(1), dim3(BLK), n * sizeof(uint16_t), 0, a);
  CHK(finish());
  CHK(dn.down(hn, 4));
  CHK(t64.down(tile, n * 8));
  CHK(dv.down(hv, (size_t)*hn * 8));
  return dq.down(hq, (size_t)*hn * 8);
}

extern "C" int glh_stage_search_tile(int dev, const uint8_t* frame, int width, int height, int channels,
                                     const int32_t* box, const double* hv, const double* hq, int hn, float* tile) {
  return glh_stage_search_tile_highpass(dev, frame, width, height, channels, box, hv, hq, hn, 5, 5, GLH_HP_REFLECT, tile);
}

extern "C" int glh_stage_search_tile_highpass(int dev, const uint8_t* frame, int width, int height, int channels,
                                              const int32_t* box, const double* hv, const double* hq, int hn,
                                              int size_x, int size_y, int mode, float* tile) {
  if (!frame || !tile || !hv || !hq || hn <= 0) return fail(GLH_E_INVALID, "bad argument");
  CHK(check_highpass(size_x, size_y));
  if (mode < GLH_HP_REFLECT || mode > GLH_HP_WRAP) return fail(GLH_E_UNSUPPORTED, "high-pass boundary mode %d", mode);
  if (channels != 1 && channels != 3) return fail(GLH_E_UNSUPPORTED, "1 or 3 channels");
  CHK(check_box(box, width, height));
  HIPCHK(hipSetDevice(dev));
  const int w = box[2] - box[0], h = box[3] - box[1];
  const size_t n = (size_t)w * h;
  if ((size_t)(BAND_H + 6) * w * 2 > 60000) return fail(GLH_E_INVALID, "tile too wide for the test hook");
  DevBuf df, dv, dq, dout;
  CHK(df.up(frame, (size_t)width * height * channels));
  CHK(dv.up(hv, (size_t)hn * 8));
  CHK(dq.up(hq, (size_t)hn * 8));
  CHK(dout.alloc(n * 4));
  SearchBoxArgs a{};
  a.frame = df.as<uint8_t>();
  a.width = width;
  a.channels = channels;
  for (int k = 0; k < 4; ++k) a.box[k] = box[k];
  a.hist_v = dv.as<double>();
  a.hist_q = dq.as<double>();
  a.hist_n = hn;
  a.hp_rx = size_x / 2;
  a.hp_ry = size_y / 2 | (mode << 4);
  a.out = dout.as<float>();
  hipLaunchKernelGGL(k_search_from_box, dim3(1), dim3(BLK), (size_t)(BAND_H + 6) * w * sizeof(uint16_t), 0, a);
  CHK(finish());
  return dout.down(tile, n * 4);
}

extern "C" int glh_stage_ssd(int dev, const float* search, int hs, int ws, const float* templ, int th, int tw,
                             float* sse) {
  if (!search || !templ || !sse || th <= 0 || tw <= 0 || hs < th || ws < tw) return fail(GLH_E_INVALID, "bad argument");
  HIPCHK(hipSetDevice(dev));
  const int ho = hs - th + 1, wo = ws - tw + 1;
  DevBuf ds, dt, dbox, dst, dout;
  CHK(ds.up(search, (size_t)hs * ws * 4));
  CHK(dt.up(templ, (size_t)th * tw * 4));
  int32_t box[4] = {0, 0, ws, hs};
  int32_t st = GLH_OBS_OK;
  CHK(dbox.up(box, sizeof box));
  CHK(dst.up(&st, sizeof st));
  CHK(dout.alloc((size_t)ho * wo * 8));
  SsdArgs a{};
  a.o = 0;
  a.P = 1;
  a.tw = tw;
  a.th = th;
  a.tile_cap = th * tw;
  a.search_cap = hs * ws;
  a.sse_cap = ho * wo;
  size_t lds = ssd_lds_bytes(tw, th);
  HIPCHK(hipFuncSetAttribute((const void*)k_ssd, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  a.box = dbox.as<int32_t>();
  a.obs_status = dst.as<int32_t>();
  a.search = ds.as<float>();
  a.tmpl = dt.as<float>();
  a.sse = dout.as<double>();
  hipLaunchKernelGGL(k_ssd, dim3(4, 1), dim3(BLK), lds, 0, a);
  CHK(finish());
  std::vector<double> tmp((size_t)ho * wo);
  CHK(dout.down(tmp.data(), tmp.size() * 8));
  for (size_t i = 0; i < tmp.size(); ++i) sse[i] = (float)tmp[i];  // exact: values are float32 widened
  return GLH_OK;
}

static int stage_sample_impl(int dev, const float* sse, int ho, int wo, int kx, int ky, const double* box,
                             const double* uv, int n, double* values, uint8_t* outside);

extern "C" int glh_stage_sample(int dev, const float* sse, int ho, int wo, const double* box, const double* uv, int n,
                                double* values, uint8_t* outside) {
  return stage_sample_impl(dev, sse, ho, wo, 0, 0, box, uv, n, values, outside);
}

extern "C" int glh_stage_sample_orders(int dev, const float* sse, int ho, int wo, int kx, int ky, const double* box,
                                       const double* uv, int n, double* values, uint8_t* outside) {
  if (kx < 1 || kx > GLH_SPL_KMAX || ky < 1 || ky > GLH_SPL_KMAX)
    return fail(GLH_E_UNSUPPORTED, "interpolation orders (%d, %d): each of 1 .. %d", kx, ky, GLH_SPL_KMAX);
  return stage_sample_impl(dev, sse, ho, wo, kx, ky, box, uv, n, values, outside);
}

// kx = ky = 0: the bicubic default (k_spline_fit's own path + spline_eval); else the general orders
static int stage_sample_impl(int dev, const float* sse, int ho, int wo, int kx, int ky, const double* box,
                             const double* uv, int n, double* values, uint8_t* outside) {
  const int need_h = kx ? kx + 1 : 4, need_w = ky ? ky + 1 : 4;
  if (!sse || !box || !uv || !values || !outside || ho < need_h || wo < need_w || n <= 0)
    return fail(GLH_E_INVALID, "bad argument");
  HIPCHK(hipSetDevice(dev));
  if (kx) {
    std::vector<double> z((size_t)ho * wo);
    for (size_t i = 0; i < z.size(); ++i) z[i] = (double)sse[i];
    std::vector<double> fv((size_t)(2 * kx + 1) * ho), fu((size_t)(2 * ky + 1) * wo);
    if (!spline_lu_general(ho, kx, fv.data()) || !spline_lu_general(wo, ky, fu.data()))
      return fail(GLH_E_STATE, "spline collocation matrix has support outside its band");
    const int maxn = ho > wo ? ho : wo;
    std::vector<int64_t> zero(maxn + 1, 0);
    DevBuf dz, dfv, dfu, doff, dbox, dst, duv, dval, dout;
    CHK(dz.up(z.data(), z.size() * 8));
    CHK(dfv.up(fv.data(), fv.size() * 8));
    CHK(dfu.up(fu.data(), fu.size() * 8));
    CHK(doff.up(zero.data(), zero.size() * 8));
    int32_t ibox[4] = {0, 0, wo, ho};
    int32_t st = GLH_OBS_OK;
    CHK(dbox.up(ibox, sizeof ibox));
    CHK(dst.up(&st, sizeof st));
    SplineFitArgs sf{};
    sf.o = 0; sf.P = 1; sf.tw = 1; sf.th = 1; sf.sse_cap = ho * wo; sf.max_n = maxn;
    sf.box = dbox.as<int32_t>();
    sf.obs_status = dst.as<int32_t>();
    sf.sse = dz.as<double>();
    sf.kx = kx; sf.ky = ky;
    sf.glu_v = dfv.as<double>(); sf.glu_v_off = doff.as<int64_t>();
    sf.glu_u = dfu.as<double>(); sf.glu_u_off = doff.as<int64_t>();
    hipLaunchKernelGGL(k_spline_fit, dim3(1), dim3(BLK), 0, 0, sf);
    CHK(duv.up(uv, (size_t)n * 16));
    CHK(dval.alloc((size_t)n * 8));
    CHK(dout.alloc((size_t)n));
    SampleArgs sa{};
    sa.coef = dz.as<double>();
    sa.ho = ho; sa.wo = wo; sa.n = n; sa.kx = kx; sa.ky = ky;
    for (int k = 0; k < 4; ++k) sa.sb[k] = box[k];
    sa.uv = duv.as<double>();
    sa.values = dval.as<double>();
    sa.outside = dout.as<uint8_t>();
    hipLaunchKernelGGL(k_sample, dim3((n + BLK - 1) / BLK), dim3(BLK), 0, 0, sa);
    CHK(finish());
    CHK(dval.down(values, (size_t)n * 8));
    return dout.down(outside, (size_t)n);
  }
  const int maxn = ho > wo ? ho : wo;
  std::vector<int64_t> off(maxn + 1, 0);
  std::vector<double> lu;
  for (int m : {ho, wo}) {
    off[m] = (int64_t)lu.size();
    lu.resize(lu.size() + 5 * (size_t)m);
    spline_lu(m, lu.data() + off[m]);
  }
  std::vector<double> z((size_t)ho * wo);
  for (size_t i = 0; i < z.size(); ++i) z[i] = (double)sse[i];
  DevBuf dz, dlu, doff, dbox, dst, duv, dval, dout, dinv;
  {
    std::vector<double> inv((size_t)spline_inverse_off(GLH_SPL_DENSE_MAX + 1));
    for (int m = 4; m <= GLH_SPL_DENSE_MAX; ++m)
      if (m == ho || m == wo) spline_inverse(m, inv.data() + spline_inverse_off(m));
    CHK(dinv.up(inv.data(), inv.size() * 8));
  }
  CHK(dz.up(z.data(), z.size() * 8));
  CHK(dlu.up(lu.data(), lu.size() * 8));
  CHK(doff.up(off.data(), off.size() * 8));
  // k_spline_fit derives (wo, ho) from box and (tw, th): use tw = th = 1
  int32_t ibox[4] = {0, 0, wo, ho};
  int32_t st = GLH_OBS_OK;
  CHK(dbox.up(ibox, sizeof ibox));
  CHK(dst.up(&st, sizeof st));
  SplineFitArgs sf{};
  sf.o = 0; sf.P = 1; sf.tw = 1; sf.th = 1; sf.sse_cap = ho * wo; sf.max_n = maxn;
  sf.box = dbox.as<int32_t>();
  sf.obs_status = dst.as<int32_t>();
  sf.lu = dlu.as<double>();
  sf.lu_off = doff.as<int64_t>();
  sf.inv = dinv.as<double>();
  sf.sse = dz.as<double>();
  sf.sse_copy = nullptr;
  hipLaunchKernelGGL(k_spline_fit, dim3(1), dim3(BLK), 0, 0, sf);
  CHK(duv.up(uv, (size_t)n * 16));
  CHK(dval.alloc((size_t)n * 8));
  CHK(dout.alloc((size_t)n));
  SampleArgs sa{};
  sa.coef = dz.as<double>();
  sa.ho = ho; sa.wo = wo; sa.n = n;
  for (int k = 0; k < 4; ++k) sa.sb[k] = box[k];
  sa.uv = duv.as<double>();
  sa.values = dval.as<double>();
  sa.outside = dout.as<uint8_t>();
  hipLaunchKernelGGL(k_sample, dim3((n + BLK - 1) / BLK), dim3(BLK), 0, 0, sa);
  CHK(finish());
  CHK(dval.down(values, (size_t)n * 8));
  return dout.down(outside, (size_t)n);
}

extern "C" int glh_stage_raster_sample(int dev, const double* z, int nx, int ny, const double* gx, const double* gy,
                                       int sx, int sy, double xmin, double xmax, double ymin, double ymax,
                                       const double* xy, int n, int order, double* values, uint8_t* oob) {
  if (!z || !xy || !values || !oob || n <= 0 || (order != 0 && order != 1)) return fail(GLH_E_INVALID, "bad argument");
  CHK(check_raster_args(nx, ny, gx, gy, sx, sy));
  CHK(check_raster_uniform(nx, ny, gx, gy, xmin, xmax, ymin, ymax));
  HIPCHK(hipSetDevice(dev));
  DevBuf dz, dgx, dgy, dxy, dv, do_;
  CHK(dz.up(z, (size_t)nx * ny * 8));
  CHK(dgx.up(gx, (size_t)nx * 8));
  CHK(dgy.up(gy, (size_t)ny * 8));
  CHK(dxy.up(xy, (size_t)n * 16));
  CHK(dv.alloc((size_t)n * 8));
  CHK(do_.alloc((size_t)n));
  const RasterDev r = raster_dev(dz.as<double>(), dgx.as<double>(), dgy.as<double>(), nx, ny, sx, sy, xmin, xmax, ymin, ymax);
  hipLaunchKernelGGL(k_raster_sample, dim3((n + BLK - 1) / BLK), dim3(BLK), 0, 0, r, dxy.as<double>(), n, order,
                     dv.as<double>(), do_.as<uint8_t>());
  CHK(finish());
  CHK(dv.down(values, (size_t)n * 8));
  return do_.down(oob, (size_t)n);
}

extern "C" int glh_stage_resample(int dev, const double* weights, int n, double u, int64_t* idx) {
  if (!weights || !idx || n <= 0) return fail(GLH_E_INVALID, "bad argument");
  if ((size_t)n * 10 + 4096 > 150 * 1024) return fail(GLH_E_UNSUPPORTED, "n too large for the LDS-resident scan");
  HIPCHK(hipSetDevice(dev));
  PairwisePlan pl;
  pairwise_plan(n, pl);
  std::vector<double> pin((size_t)n * 6, 0.0);
  DevBuf dw, dpi, dpo, dwo, du, didx, dst, def, doff, dlen, dops, dlev, droot;
  CHK(dw.up(weights, (size_t)n * 8));
  CHK(dpi.up(pin.data(), pin.size() * 8));
  CHK(dpo.alloc(pin.size() * 8));
  CHK(dwo.alloc((size_t)n * 8));
  CHK(du.up(&u, 8));
  CHK(didx.alloc((size_t)n * 4));
  uint32_t st0 = 0;
  int32_t ef = 0x7f7f7f7f;
  CHK(dst.up(&st0, 4));
  CHK(def.up(&ef, 4));
  CHK(doff.up(pl.leaf_off.data(), pl.leaf_off.size() * 4));
  CHK(dlen.up(pl.leaf_len.data(), pl.leaf_len.size() * 4));
  CHK(dops.up(pl.ops.data(), pl.ops.size() * 4));
  CHK(dlev.up(pl.level_off.data(), pl.level_off.size() * 4));
  CHK(droot.up(pl.roots.data(), pl.roots.size() * 4));
  ResampleArgs a{};
  a.particles_in = dpi.as<double>();
  a.weights_in = dw.as<double>();
  a.particles_out = dpo.as<double>();
  a.weights_out = dwo.as<double>();
  a.active = nullptr;
  a.u = du.as<double>();
  a.idx_out = didx.as<int32_t>();
  a.moments = nullptr;
  a.pt_status = dst.as<uint32_t>();
  a.pt_err_frame = def.as<int32_t>();
  a.leaf_off = doff.as<int32_t>();
  a.leaf_len = dlen.as<int32_t>();
  a.ops = dops.as<int32_t>();
  a.level_off = dlev.as<int32_t>();
  a.roots = droot.as<int32_t>();
  a.N = n;
  a.nleaves = (int)pl.leaf_off.size();
  a.nnodes = pl.nnodes;
  a.nlevels = (int)pl.level_off.size() - 1;
  a.nroots = (int)pl.roots.size();
  a.rng_mode = GLH_RNG_HOST;
  a.frame = 0;
  HIPCHK(hipFuncSetAttribute((const void*)k_resample, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
  hipLaunchKernelGGL(k_resample, dim3(1), dim3(BLK), ((size_t)n + pl.nnodes) * 8 + (size_t)n * 2, 0, a);
  CHK(finish());
  std::vector<int32_t> tmp(n);
  CHK(didx.down(tmp.data(), (size_t)n * 4));
  for (int i = 0; i < n; ++i) idx[i] = tmp[i];
  return GLH_OK;
}
