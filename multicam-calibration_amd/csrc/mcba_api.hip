// mcba_api.hip -- the C ABI of include/mcba.h: handle, device buffers, kernel sequencing.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <dlfcn.h>
#include <map>
#include <mutex>
#include <rccl/rccl.h>
#include <string>
#include <vector>

#include "../../include/mcba.h"
#include "mcba_kernels.h"
#include "mcba_math.h"

namespace {

thread_local std::string g_err;

enum KernelId { K_TRANSPOSE = 0, K_GRAM, K_COST, K_SYRK, K_REDUCE, K_BACKSUB, K_SUM_TRIAL, K_JACOBIAN, K_DECIDE, K_SOLVE, K_COUNT };
const char* kKernelNames = "k_transpose_obs\nk_gram\nk_cost\nk_syrk\nk_reduce_system\nk_backsub\nk_sum_trial\nk_jacobian\nk_decide\nk_solve_cam";
constexpr int kRing = 16;  // host-mapped LM state slots (device-resident loop): the host may run at most kRing - 1 ticks ahead

struct EvRec { int kid; hipEvent_t a, b; };
struct DevBuf { void** slot; size_t bytes; };  // a pooled device buffer of a handle: where its pointer lives, its size

}  // namespace

struct mcba_handle {
  int C = 0, F = 0, N = 0, Fpad = 0, nfb = 0, n = 0;
  int device = 0;
  hipStream_t stream = nullptr;
  int loss = MCBA_LOSS_SOFT_L1;
  double f_scale = 1.0;
  bool have_obs = false, have_lin = false, have_red = false, have_jac = false;
  // device buffers
  double *obs_t = nullptr, *obs_raw = nullptr, *obj = nullptr, *x[2] = {nullptr, nullptr};
  double *rec2[2] = {nullptr, nullptr}, *gpart2[2] = {nullptr, nullptr}, *fbuf = nullptr, *fpart = nullptr;
  int lin = 0;          // which of the two linearisation buffers holds the accepted point
  bool have_spec = false;  // the other one holds a speculative linearisation of the last trial point
  double *spart = nullptr, *cpart = nullptr, *bpart = nullptr;
  double *red_own = nullptr, *red = nullptr;
  double *jac = nullptr, *res = nullptr;
  double *err = nullptr, *dmean = nullptr, *dfull = nullptr, *repro = nullptr, *trans = nullptr, *und = nullptr;  // pre-filter / diagnostics (lazy)
  unsigned char *sel = nullptr, *fmask = nullptr;
  // mcba_prefilter (the selection on the device): scratch state, per-frame status / worst mean error, the packed result and its pinned landing place
  unsigned char *pf_state = nullptr, *pf_status = nullptr, *pf_packed = nullptr, *pf_host = nullptr;
  double* pf_worst = nullptr;
  size_t pf_host_bytes = 0;
  double* core_arena = nullptr;            // mcba_create: x[0] | x[1] | obj
  unsigned char* solver_arena = nullptr;   // ensure_solver: the one allocation the solver buffers below are pieces of
  int* sub_frames = nullptr;   // mcba_create_subset: the frame indices on the device (kept with the handle: no synchronisation to free them)
  double* outbuf = nullptr;    // mcba_lm_result: [x | gradient] packed for one device-to-host copy
  // calibrate() on the device (mcba_calib_*): intrinsics [C][9], every view's board pose [C][6][Fpad] (NaN = none), per-view flags, and
  // scratch that grows with the call (view lists, outputs, pairwise transforms, select states, world-frame poses)
  double *cal_intr = nullptr, *cal_poses_t = nullptr, *cal_out = nullptr, *cal_rel = nullptr, *cal_world = nullptr;
  unsigned char *cal_valid = nullptr, *cal_nit = nullptr, *cal_sel = nullptr;
  int* cal_views = nullptr;
  size_t cal_out_cap = 0, cal_rel_cap = 0, cal_sel_cap = 0, cal_views_cap = 0;
  bool have_cal_poses = false;
  double* obj_host = nullptr;  // board points as uploaded (diagnostics normalise them on the host)
  int planar = 0;              // every board point has z = 0 exactly (the fused k_gram then runs its planar instance)
  int *tile_i = nullptr, *tile_j = nullptr;
  int NT = 0, NP = 0, G = 0, sq = 0, sr = 0, FS = 0, ppw = 4, nfblocks = 0, nbblocks = 0, nch = 1;  // k_syrk: G workgroups, sq stages of FS frames each, the first sr one more
  int gram_nchunk = 0;  // gram_split == 3: point chunks per (camera, frame block) of the tail
  double* gchunk = nullptr;  // ... and their raw sums
  int gram_split = 0;  // 0: both accumulator sets in one lane (1 wave/SIMD); 1: two roles, two waves/SIMD (few frames); 2 / 3: fused rounds + split-role / point-chunk tail;
                       // 4: point split inside the workgroup (gram_npw wavefronts per (camera, frame block)); 5: fused rounds + point-split tail
  int gram_npw = 4;
  double curv_floor = 1.0;  // curvature weight of the NEXT linearisations: max(Triggs, curv_floor rho') -- 1 = IRLS (mcba_set_curvature_floor; csrc/mcba_math.h)
  int cw = 12;         // camera block width: 12, or 6 = the intrinsics of every camera are held fixed (mcba_set_camera_block; BASELINE configs[1]): n = cw C
  size_t nx = 0, nsys = 0;
  double* pinned = nullptr;  // nsys + 8 doubles, + 12C for dc
  ncclComm_t comm = nullptr;  // direct RCCL communicator (optional)
  // device-resident LM loop (mcba_lm_auto_*)
  double *dcbuf = nullptr, *swork = nullptr;
  double* dscale = nullptr;   // numeric x_scale (least_squares): D = 1 / x_scale^2 in the layout of x; have_xscale says whether it is in use
  bool have_xscale = false;   // the XS kernel instances run: a numeric x_scale and / or frozen coordinates are in `dscale`
  std::vector<double> xs_host;             // numeric x_scale as D = 1 / x_scale^2 (nx entries; empty = 'jac')
  std::vector<unsigned char> frozen_host;  // coordinates taken out of the system (mcba_set_frozen; empty = none)
  double *blo = nullptr, *bhi = nullptr;   // box constraints (mcba_set_bounds), in the layout of x
  bool have_bounds = false;
  double* loss_tab = nullptr;   // loss == LOSS_TABLE (mcba_set_loss_table): [3][C][N][Fpad] (u, v) pairs, laid out as obs_t
  int fuse_max_polls = 200000;
  bool strict_sync = true;    // the fused back-substitution's readers ACQUIRE the release word with an agent-scope fence: the HIP memory model's form, the default since round 6 (MCBA_STRICT_SYNC=0 / mcba_set_strict_sync(h, 0): relaxed loads + gfx950's in-order issue, ~1.3 us per iteration faster)
  unsigned char* fixed = nullptr;
  bool have_fixed = false, auto_ready = false;
  bool speculate = true;       // frame-sharded ticks: one collective (speculative Schur reduction) instead of two
  double* ring = nullptr;      // kRing x MCBA_LMS doubles, host-coherent pinned memory the GPU writes directly
  double* ring_dev = nullptr;  // the same memory as the device sees it
  int npad = 0, solve_lds = 0;
  // k_solve_backsub (single-GPU ticks, factor in LDS): the solve's launch also runs the back-substitution of the NEXT trial step;
  // trial_ready = the last tick did so, the next one must not back-substitute again.  The flag word sits behind the camera step.
  bool fuse_backsub = false, trial_ready = false;
  unsigned long long last_solve_seq = 0;  // sequence number of the last mcba_lm_auto_solve / tick (what a timed-out back-substitution of that launch stamps)
  unsigned long long waited_seq = 0;      // the last tick whose posted state the host has read (mcba_lm_auto_wait): equal to last_solve_seq = nothing posts into the ring any more
  unsigned long long solve_launches = 0;  // k_solve_cam launches so far (SolveArgs.stage_tag)
  int slots = 1024;  // wavefront slots of the device (4 x CUs): where k_gram's launch variants cut this shard into rounds
  int ncu = 256, lds_optin = 160 * 1024;  // compute units and the LDS a workgroup may ask for (hipGetDeviceProperties at create; MI355X: 256 / 160 KiB)
  bool spec_copy_ready = false;  // the last k_reduce_system was a speculative one: the pre-decision state copy is in place
  double ftol = 1e-8, xtol = 1e-8, gtol = 1e-8, lam_min = 1e-12, lam_max = 1e12;
  double dec_floor = 0.0;      // floor of Nielsen's damping factor on accepted steps (0 = the classical 1/3): mcba_lm_set_decrease_floor
  // profiling
  bool prof = false;
  unsigned prof_mask = ~0u;
  int prof_stride = 1;             // bracket every prof_stride-th launch of a selected kernel
  bool prof_exact = false;         // k_gram: events on the dispatch itself (mcba_profile_exact)
  unsigned prof_count[32] = {};
  std::vector<EvRec> evs;
  std::vector<hipEvent_t> pool;
  std::vector<DevBuf> bufs;        // every pooled device buffer (mcba_destroy parks them)
  std::vector<double> hist;        // mcba_lm_run: the state every retired tick posted (MCBA_LMS doubles each; row 0 = the solve of the start point)
  bool have_solver = false;        // solver buffers are allocated on first use (ensure_solver): a pre-filter handle never needs them
  size_t ring_bytes = 0, pinned_bytes = 0;
  unsigned ring_flags = 0;
};

// a device array that outlives its handle (mcba_residuals_detach, mcba_lm_result)
struct mcba_buffer { double* dev; size_t count; int device; hipStream_t stream; double* base; size_t base_count; };  // dev / count: what a download delivers; base / base_count: the pooled allocation it lies in

namespace {

#define HIPCHK(expr)                                                                                     \
  do {                                                                                                   \
    hipError_t e_ = (expr);                                                                              \
    if (e_ != hipSuccess) {                                                                              \
      g_err = std::string(#expr) + ": " + hipGetErrorString(e_);                                         \
      return MCBA_ERR_HIP;                                                                               \
    }                                                                                                    \
  } while (0)

int fail(int code, const char* msg) { g_err = msg; return code; }

hipEvent_t get_event(mcba_handle* h) {
  if (!h->pool.empty()) { hipEvent_t e = h->pool.back(); h->pool.pop_back(); return e; }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}

struct Scope {  // brackets one launch with events when profiling
  mcba_handle* h; int kid; hipEvent_t a{}, b{};
  bool on, exact;
  Scope(mcba_handle* h_, int k) : h(h_), kid(k), on(h_->prof && ((h_->prof_mask >> k) & 1u) && (h_->prof_count[k]++ % (unsigned)h_->prof_stride) == 0), exact(false) {
    if (!on) return;
    a = get_event(h); b = get_event(h);
    // k_gram with exact timing asked for (mcba_profile_exact): the events ride on the kernel's dispatch (its own begin / end timestamps,
    // what rocprofv3 reports); everything else: event records around the launch (which read ~2.5 us more than the kernel takes)
    exact = h->prof_exact && k == K_GRAM;
    if (exact) mcba::gram_time_next_launch(a, b);
    else (void)hipEventRecord(a, h->stream);
  }
  ~Scope() {
    if (!on) return;
    if (exact && mcba::gram_time_pending()) {   // another launch variant than the fused kernel ran: no exact timing for it
      mcba::gram_time_next_launch(nullptr, nullptr);
      h->pool.push_back(a); h->pool.push_back(b);
      return;
    }
    if (!exact) (void)hipEventRecord(b, h->stream);
    h->evs.push_back({kid, a, b});
  }
};

mcba::Sel host_sel(int idx, double lam = 0.0) { return mcba::Sel{nullptr, idx, lam, 0, 0.0}; }
mcba::Sel dev_sel(const mcba_handle* h, int flip) { return mcba::Sel{h->red + h->nsys + 8, flip, 0.0, 0, 0.0}; }  // LM state lives behind the trial scalars
mcba::Sel spec_sel(const mcba_handle* h) { return mcba::Sel{h->red + h->nsys + 8, 0, h->lam_min, 1, h->dec_floor}; }
// the state AFTER the decision k_syrk took itself (single-GPU ticks): a second buffer behind the first
double* post_state(const mcba_handle* h) { return h->red + h->nsys + 8 + MCBA_LMS; }
mcba::Sel post_sel(const mcba_handle* h) { return mcba::Sel{post_state(h), 0, 0.0, 0, 0.0}; }
// a linearisation's selector: + the curvature floor this handle linearises with (mcba_set_curvature_floor)
mcba::Sel gram_sel(const mcba_handle* h, mcba::Sel s) { s.cfl = h->curv_floor; return s; }
mcba::SyrkFuse no_fuse() { mcba::SyrkFuse z{}; return z; }

int check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { g_err = std::string("kernel launch: ") + hipGetErrorString(e); return MCBA_ERR_HIP; }
  return MCBA_OK;
}

// ---- buffer pool.  hipMalloc / hipFree / hipHostMalloc are synchronising driver calls of 50-400 us each, and a
// bundle_adjust() call creates and destroys three handles of ~25 buffers: 4 ms of its 16 ms at 6 x 10 000 x 54 was hipFree alone.
// Freed buffers are parked here (per device, keyed by their exact size -- repeated calls ask for the same sizes) and handed out
// again; MCBA_POOL_MB caps what is parked (default 2048 MiB of the 288 GB -- other allocators of the process, torch's or RCCL's, cannot
// see parked memory --; 0 switches the pool off; pinned host memory: 256 MiB), mcba_pool_trim() returns everything to the driver, and
// so does an allocation of this library that the driver answers with out-of-memory, before it asks once more.  Re-use is safe without
// events: every kernel and copy of this library is enqueued on the handle's stream, so a buffer that goes to a handle on the SAME stream
// is ordered behind whatever its previous owner still had in flight (round 5: mcba_destroy no longer waits for the stream -- a handle
// was closed while the next stage's kernels were still running, and the host stood still for them); a buffer that goes to another
// stream, or back to the driver, waits for the stream it was parked with first.
struct ParkedBuf {
  void* p;
  hipStream_t stream;  // work enqueued on this stream may still read / write the buffer (busy) ...
  bool busy;           // ... or nothing can (the owner synchronised before parking it)
};
struct BufferPool {
  std::mutex mu;
  std::multimap<std::pair<int, size_t>, ParkedBuf> dev;     // (device, bytes) -> pointer + who may still be using it
  std::multimap<std::pair<unsigned, size_t>, void*> host;  // (hipHostMalloc flags, bytes) -> pointer
  size_t parked = 0, parked_host = 0;
  const size_t host_cap = (size_t)256 << 20;  // pinned host memory parked at most (state rings and staging buffers: a few hundred KB each)
  size_t cap() {
    static size_t c = [] { const char* e = getenv("MCBA_POOL_MB"); return (size_t)(e ? atoll(e) : 2048) << 20; }();
    return c;
  }
};
BufferPool g_pool;

void pool_release_all();
hipError_t pool_malloc(void** p, size_t bytes, int device, hipStream_t stream = nullptr, bool any_stream = false) {
  {
    ParkedBuf got{nullptr, nullptr, false};
    {
      std::lock_guard<std::mutex> lk(g_pool.mu);
      auto it = g_pool.dev.find({device, bytes});
      if (it != g_pool.dev.end()) { got = it->second; g_pool.dev.erase(it); g_pool.parked -= bytes; }
    }
    if (got.p) {
      // (any_stream: the caller cannot say which stream will touch the buffer)
      if (got.busy && (any_stream || got.stream != stream)) (void)hipStreamSynchronize(got.stream);
      *p = got.p;
      return hipSuccess;
    }
  }
  hipError_t e = hipMalloc(p, bytes);
  if (e == hipErrorOutOfMemory) {  // what the pool has parked is memory too: give it back to the driver and ask once more
    (void)hipGetLastError();
    pool_release_all();
    e = hipMalloc(p, bytes);
  }
  return e;
}
void pool_free(void* p, size_t bytes, int device, hipStream_t stream = nullptr, bool busy = false) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> lk(g_pool.mu);
    if (g_pool.parked + bytes <= g_pool.cap()) { g_pool.dev.insert({{device, bytes}, ParkedBuf{p, stream, busy}}); g_pool.parked += bytes; return; }
  }
  if (busy) (void)hipStreamSynchronize(stream);
  (void)hipFree(p);
}
hipError_t pool_host_malloc(void** p, size_t bytes, unsigned flags) {
  {
    std::lock_guard<std::mutex> lk(g_pool.mu);
    auto it = g_pool.host.find({flags, bytes});
    if (it != g_pool.host.end()) { *p = it->second; g_pool.host.erase(it); g_pool.parked_host -= bytes; return hipSuccess; }
  }
  return hipHostMalloc(p, bytes, flags);
}
void pool_host_free(void* p, size_t bytes, unsigned flags) {
  if (!p) return;
  std::lock_guard<std::mutex> lk(g_pool.mu);
  if (g_pool.cap() == 0 || g_pool.parked_host + bytes > g_pool.host_cap) { (void)hipHostFree(p); return; }
  g_pool.host.insert({{flags, bytes}, p});
  g_pool.parked_host += bytes;
}
void pool_release_all() {
  std::lock_guard<std::mutex> lk(g_pool.mu);
  int cur = 0;
  (void)hipGetDevice(&cur);
  for (auto& kv : g_pool.dev) { (void)hipSetDevice(kv.first.first); if (kv.second.busy) (void)hipStreamSynchronize(kv.second.stream); (void)hipFree(kv.second.p); }
  for (auto& kv : g_pool.host) (void)hipHostFree(kv.second);
  g_pool.dev.clear();
  g_pool.host.clear();
  g_pool.parked = g_pool.parked_host = 0;
  (void)hipSetDevice(cur);
}

// MCBA_POISON (tests): buffers that are handed out WITHOUT a zero fill are filled with a byte pattern instead -- 1: 0xFF (every double a NaN,
// every byte mask "set"); 2: 0x3F (finite garbage: every double 4.8e-4, every 32-bit count 1 061 109 567 -- what recycled pool memory looks
// like; NaN is the benign value for several of these buffers, e.g. the errors the median skips: ADVICE r5).  0 / unset: no fill.
int poison_byte() {
  static const int b = [] { const char* e = getenv("MCBA_POISON"); const int v = e ? atoi(e) : 0; return v == 1 ? 0xFF : (v == 2 ? 0x3F : 0); }();
  return b;
}

// zero-filled device buffer from the pool, registered with the handle (mcba_destroy parks it again).  The fill is
// enqueued on the handle's stream (no host synchronisation); `zero = false` for buffers a kernel overwrites completely
// before anything reads them (observation layouts, Jacobian blocks).
template <class T>
int dalloc(mcba_handle* h, T** p, size_t count, bool zero = true) {
  const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  HIPCHK(pool_malloc(reinterpret_cast<void**>(p), bytes, h->device, h->stream));
  h->bufs.push_back({reinterpret_cast<void**>(p), bytes});
  if (zero) HIPCHK(hipMemsetAsync(*p, 0, bytes, h->stream));
  else if (int pz = poison_byte()) HIPCHK(hipMemsetAsync(*p, pz, bytes, h->stream));  // (tests: whatever relies on a fill that is no longer made shows)
  return MCBA_OK;
}

}  // namespace

// RCCL entry points resolved at run time from the copy already loaded in the process (torch's librccl.so)
struct RcclApi {
  bool ok = false;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;

static int load_rccl() {
  if (g_rccl.ok) return MCBA_OK;
  void* lib = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
  if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
  if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) return fail(MCBA_ERR_ARG, "RCCL library not found (dlopen librccl.so)");
  g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
  g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
  g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(dlsym(lib, "ncclAllReduce"));
  g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
  g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
  g_rccl.CommCount = reinterpret_cast<decltype(g_rccl.CommCount)>(dlsym(lib, "ncclCommCount"));
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy) return fail(MCBA_ERR_ARG, "RCCL symbols missing");
  g_rccl.ok = true;
  return MCBA_OK;
}
static int rccl_fail(const char* what, ncclResult_t r) {
  g_err = std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
  return MCBA_ERR_HIP;
}

// The tile-pair tables of k_syrk / k_reduce_system (pair p of the NT x NT tile grid's upper triangle -> (i, j)) depend on NT alone:
// uploaded once per (device, NT) and process, shared by every handle and never freed (at most a few KB each) -- so the first solver
// call of a handle neither copies nor synchronises for them.
static int tile_tables(int device, int NT, int** ti, int** tj) {
  static std::mutex mu;
  static std::map<std::pair<int, int>, std::pair<int*, int*>> cache;
  std::lock_guard<std::mutex> lk(mu);
  auto it = cache.find({device, NT});
  if (it == cache.end()) {
    std::vector<int> ci, cj;
    for (int a = 0; a < NT; ++a) for (int b = a; b < NT; ++b) { ci.push_back(a); cj.push_back(b); }
    int *di = nullptr, *dj = nullptr;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&di), ci.size() * sizeof(int)));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&dj), cj.size() * sizeof(int)));
    HIPCHK(hipMemcpy(di, ci.data(), ci.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dj, cj.data(), cj.size() * sizeof(int), hipMemcpyHostToDevice));
    it = cache.insert({{device, NT}, {di, dj}}).first;
  }
  *ti = it->second.first;
  *tj = it->second.second;
  return MCBA_OK;
}

// Launch geometry of the solver kernels, derived from the problem's shape, the device and the camera block width (h->cw): called by
// mcba_create and again by mcba_set_camera_block.
static int derive_geometry(mcba_handle* h) {
  const int C = h->C, F = h->F, N = h->N, device = h->device;
  (void)N;
  const int ncu = h->ncu, slots = 4 * h->ncu;  // wavefront slots at one wavefront per SIMD
  h->slots = slots;
  h->n = h->cw * C;
  h->nsys = (size_t)h->n * h->n + 3 * h->n + 16;
  // k_syrk geometry: (12C + 1) rows of [Y ; z^T] padded to NT tiles of 16, NP tile pairs (ti <= tj);
  // FS frames per LDS stage (a divisor of 64; <= 96 KiB so that two workgroups fit a CU when C is small)
  h->NT = (h->n + 1 + 15) / 16;
  h->NP = h->NT * (h->NT + 1) / 2;
  h->ppw = h->NP <= 64 ? 4 : 16;
  h->FS = 8;  // (measured at 6 x 10k: 8 frames per stage and up to 512 workgroups -- two per CU, one building Y while the
              //  other is in its MFMA phase -- 24.2 us; 16 frames / 256 workgroups 27.4 us)
  if (const char* e = getenv("MCBA_SYRK_FS")) { int v = atoi(e); if (v == 2 || v == 4 || v == 8 || v == 16) h->FS = v; }  // tuning knob
  while (h->FS > 2 && (mcba::syrk_lds_bytes(C, h->FS, h->cw) > (size_t)h->lds_optin * 3 / 5 || (h->n + 1) * h->FS > 256 * mcba::syrk_items_per_thread())) h->FS /= 2;
  if ((h->n + 1) * h->FS > 256 * mcba::syrk_items_per_thread()) { return fail(MCBA_ERR_ARG, "too many cameras for k_syrk's per-thread item budget"); }
  if (mcba::syrk_lds_bytes(C, h->FS, h->cw) > (size_t)h->lds_optin) { return fail(MCBA_ERR_ARG, "too many cameras for the LDS staging of k_syrk"); }
  {
    int nstage = (F + h->FS - 1) / h->FS;
    // the 16-tile variant (> 13 cameras) runs ONE workgroup per CU and grid.y = ceil(NP / 64) of them share a set of frames:
    // one round of the 256 CUs, every workgroup as many stages as that allows -- 24 x 6250 x 200: 83 x 3 workgroups of 19 stages
    // 776 us per tick (33 MB of partial tiles) against 839 us for 391 x 3 of 4 stages (152 MB), 806 / 796 / 822 / 809 / 874 us
    // for G = 256 / 171 / 128 / 64 / 43
    int gmax = h->ppw == 4 ? std::min(512, 2 * ncu) : std::max(1, ncu / ((h->NP + 63) / 64));
    if (const char* e = getenv("MCBA_SYRK_G")) gmax = std::max(1, std::min(512, atoi(e)));  // tuning knob (k_reduce_system holds <= 512 / 16 partial rows per wavefront)
    int g = std::min(nstage, gmax);
    // Two workgroups per CU (ppw == 4: <= 10 cameras): deal the stages out evenly over g workgroups -- measured at 6 x 10 000:
    // 1250 stages as 226 x 3 + 286 x 2 (no CU above five stages) 107.5 us per tick against 109.0 us for 417 x 3 (161 CUs with six).
    // One workgroup per CU (the 16-tile variant): every workgroup costs a prologue of its own, fewer and equal ones win
    // (24 x 6250 x 200: 873 us against 918 us).
    bool balance = h->ppw == 4;
    if (const char* e = getenv("MCBA_SYRK_BALANCE")) balance = atoi(e) != 0;  // development knob
    if (balance) {
      h->G = g;
      h->sq = nstage / g;
      h->sr = nstage % g;
    } else {
      h->sq = (nstage + g - 1) / g;
      h->sr = 0;
      h->G = (nstage + h->sq - 1) / h->sq;
    }
  }
  h->nbblocks = h->Fpad / 64;
  // fused k_gram needs >= ~1 wavefront per SIMD (1024) to fill the chip; with fewer (camera, frame-block) pairs the
  // split-role variant doubles the number of wavefronts.  MCBA_GRAM_SPLIT=0/1 overrides (tuning knob).
  // Beyond one round (1024 wavefront slots) a small second round costs the fused variant a full pass; the split roles
  // share SIMDs and fill the tail better (measured 6 x 12500 x 54: 102 us split vs 132 us fused; equal at 1.84 and 2.3 rounds).
  {
    const int items = C * h->nfb;
    h->gram_split = items < 3 * slots / 4 || (items > slots && items <= 3 * slots / 2);
    // more than one round with a short last round: fused for the whole rounds + split roles for the tail (mode 2)
    // (measured 24 x 6250 x 200, 2.3 rounds: 536 us vs 561 us fused, 612 us split; at 1.15 rounds plain split roles win)
    if (items > 2 * slots && (items % slots) > 0 && (items % slots) <= slots / 2) h->gram_split = 2;
    // Round 3: more than one round with a short last round -> whole rounds fused + the tail's POINTS in chunks over the idle SIMDs
    // (mode 3; measured 24 x 6250 x 200: k_gram 381 -> 360 us -- profiles/round3/NOTES_round3.md section 6)
    if (items > slots) {
      const int fba = ((items / slots) * slots / C) & ~3, tail = C * (h->nfb - fba);
      // (worth it for large boards only: the chunk launch + the combine launch cost ~25 us whatever the board, the split-role
      //  tail 0.73 of a fused pass -- break-even near 90 points; 6 x 12500 x 54: 88 us against 79 us with plain split roles)
      if (fba > 0 && tail > 0 && tail <= slots / 2 && N >= 128) {
        const int nch = std::min(std::min(8, slots / tail), N / 8);
        if (nch >= 2) { h->gram_split = 3; h->gram_nchunk = nch; }
      }
    }
  }
  // Round 4: point split INSIDE the workgroup (k_gram_psplit: the 4 or 2 wavefronts of a (camera, frame block) take a part of the board's
  // points each and meet in LDS) -- for shards with at most half a round of items, and for a short last round behind whole fused rounds.
  {
    static bool ps_ready[64] = {};  // the dynamic-LDS limit of its instances is raised once per device
    static std::mutex ps_mu;        // (handles may be created from several threads: solver.InProcessShards)
    bool ps_ok = (size_t)h->lds_optin >= mcba::gram_psplit_lds_bytes(4) + 2048;
    {
      std::lock_guard<std::mutex> lk(ps_mu);
      if (ps_ok && !ps_ready[device & 63]) { ps_ok = mcba::gram_psplit_set_lds_limit() == 0; ps_ready[device & 63] = ps_ok; (void)hipGetLastError(); }
    }
    const int items = C * h->nfb;
    if (ps_ok) {
      if (items <= slots / 4) { h->gram_split = 4; h->gram_npw = 4; }
      else if (items <= slots / 2) { h->gram_split = 4; h->gram_npw = 2; }
      else if (items <= slots) h->gram_split = 0;  // (measured in round 4 at 6 x 7 000 x 54, 660 items: fused 47.4 us, split roles 53.2 us)
      else if (items > slots) {
        const int fba = mcba::gram_round_blocks(C, h->nfb, slots), tail = C * (h->nfb - fba);
        if (fba > 0 && tail > 0 && tail <= slots / 4) { h->gram_split = 5; h->gram_npw = 4; }
        else if (fba > 0 && tail > 0 && tail <= slots / 2 && h->gram_split != 3) { h->gram_split = 5; h->gram_npw = 2; }
      }
    }
    if (const char* e = getenv("MCBA_GRAM_SPLIT")) {  // 0 fused, 1 split roles, 2 fused + split-role tail, 3 fused + point-chunk tail, 4 point split, 5 fused + point-split tail
      h->gram_split = std::max(0, std::min(5, atoi(e)));
      if (h->gram_split == 3 && h->gram_nchunk < 2) h->gram_nchunk = std::max(2, std::min(4, N / 8));
      if (h->gram_split >= 4 && !ps_ok) { return fail(MCBA_ERR_ARG, "MCBA_GRAM_SPLIT=4/5: this device cannot give k_gram_psplit its LDS"); }
    }
    if (const char* e = getenv("MCBA_GRAM_NPW")) h->gram_npw = atoi(e) == 2 ? 2 : 4;
  }
  if (h->cw == 6 && h->gram_split != 4) h->gram_split = 1;  // intrinsics held fixed: role A alone -- the point split (4), else the role-A half of the split roles
  if (const char* e = getenv("MCBA_GRAM_NCHUNK")) h->gram_nchunk = std::max(2, std::min(8, atoi(e)));
  // k_cost: split the board points so that ~4 waves per SIMD (1024 SIMDs) are in flight
  h->nch = std::max(1, std::min(std::min(8, N / 8), (4 * slots + C * h->nfb - 1) / (C * h->nfb)));
  h->nfblocks = h->G;  // (kept: k_syrk's workgroups factorise their own frames: one (max |g_f|, #failures) pair each)
  h->npad = 16 * h->NT;
  h->solve_lds = mcba::solve_fits_lds(h->npad, h->lds_optin);
  return MCBA_OK;
}

extern "C" {

int mcba_abi_version(void) { return 7; }  // 7 (round 6): calibrate() on the device (mcba_calib_*, mcba_pose_*, mcba_create_views); 6 (round 5): mcba_prefilter, mcba_lm_run / _history / _result -- whole stages of bundle_adjust() per crossing; additions only: every ABI-5 entry point is unchanged
const char* mcba_last_error(void) { return g_err.c_str(); }
const char* mcba_profile_names(void) { return kKernelNames; }

int mcba_device_count(int* count) {
  if (!count) return fail(MCBA_ERR_ARG, "count is NULL");
  hipError_t e = hipGetDeviceCount(count);
  if (e != hipSuccess) { *count = 0; g_err = hipGetErrorString(e); return MCBA_ERR_NODEVICE; }
  return MCBA_OK;
}

int mcba_create(mcba_handle** out, int C, int F, int N, int device) {
  if (!out || C < 1 || F < 1 || N < 1) return fail(MCBA_ERR_ARG, "mcba_create: need cameras >= 1, frames >= 1, points >= 1");
  if (C > 40) {
    g_err = "mcba_create: " + std::to_string(C) + " cameras -- this build handles at most 40 per handle: the (12 C + 1)-row reduced camera system is factorised by ONE workgroup "
            "(k_solve_cam: tile tables, LDS and scratch are sized for 481 rows) and k_backsub takes the camera step through the kernel arguments (480 doubles); reduced_solver=\"host\" does not "
            "lift it (the Schur product's tile grid has the same bound).  The reference has no such limit (scipy's sparse LSMR); split the rig, or calibrate the cameras in overlapping groups of <= 40";
    return MCBA_ERR_ARG;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(MCBA_ERR_NODEVICE, "no HIP device visible");
  if (device < 0 || device >= ndev) return fail(MCBA_ERR_ARG, "device ordinal out of range");
  HIPCHK(hipSetDevice(device));
  mcba_handle* h = new mcba_handle();
  h->C = C; h->F = F; h->N = N; h->device = device;
  {  // the launch geometry below is derived from the device, not from MI355X constants (a part with fewer CUs or less LDS gets its own deal)
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
      if (prop.multiProcessorCount > 0) h->ncu = prop.multiProcessorCount;
      if (prop.sharedMemPerBlockOptin > 0) h->lds_optin = (int)std::min<size_t>(prop.sharedMemPerBlockOptin, 160 * 1024);
    }
  }
  if (const char* e = getenv("MCBA_STRICT_SYNC")) h->strict_sync = atoi(e) != 0;
  h->Fpad = (F + 63) / 64 * 64;
  h->nfb = h->Fpad / 64;
  h->nx = (size_t)12 * C + (size_t)6 * h->Fpad;
  {
    const int grc = derive_geometry(h);
    if (grc != MCBA_OK) { delete h; return grc; }
  }
  int rc;
#define DA(p, cnt, zero) if ((rc = dalloc(h, &h->p, (cnt), zero)) != MCBA_OK) { mcba_destroy(h); return rc; }
  // what every handle needs (a pre-filter handle needs nothing else): the two observation layouts, the board, the parameter slots
  DA(obs_t, (size_t)2 * C * N * h->Fpad, false);
  DA(obs_raw, (size_t)2 * C * F * N, false);
  {  // the two parameter slots and the board: one allocation, one fill (zero: the poses of the padding frames must be finite)
    const size_t nxp = (h->nx + 31) / 32 * 32;
    DA(core_arena, 2 * nxp + (size_t)3 * N, true);
    h->x[0] = h->core_arena; h->x[1] = h->core_arena + nxp; h->obj = h->core_arena + 2 * nxp;
  }
#undef DA
  // (no synchronisation: the zero fills above are ordered before everything else on this stream, and mcba_set_stream waits for the old stream when it changes it)
  *out = h;
  return MCBA_OK;
}

// Solver buffers, allocated on first use (linearise / reduce / LM entry points): records, partial sums, reduce buffer, the host-
// mapped state ring.  A handle that only runs the pre-filter (api.select_frames over ALL frames of a long recording) never gets
// here, so its footprint is the observations alone.
static int compose_dscale(mcba_handle* h);
static int ensure_solver(mcba_handle* h) {
  if (h->have_solver) return MCBA_OK;
  const int C = h->C;
  int rc;
  // ONE device allocation for the whole set (round 5: sixteen pool look-ups and sixteen fill launches -- ~50 us of host time and as
  // much of the stream in front of the first linearisation -- became one of each), cut into 256-byte aligned pieces.  The pieces a
  // kernel overwrites completely before anything reads them (linearisation records, k_syrk's partial tiles, the point-chunk scratch:
  // 110 of the 116 MB at 6 x 10 000 x 54) sit behind the ones that start from zero and are not filled.
  if (!h->solver_arena) {
    struct Piece { double** p; size_t count; };
    double** none = nullptr;
    (void)none;
    const size_t n_swork = h->solve_lds ? 16 : 2 * (size_t)h->npad * h->npad + 64 * (size_t)h->npad;  // two sets of 16 x 16 tiles of the lower triangle (mcba_solve.hip, right-looking variant)
    Piece zeroed[] = {{&h->gpart2[0], (size_t)C * h->nfb * MCBA_GP}, {&h->gpart2[1], (size_t)C * h->nfb * MCBA_GP}, {&h->fbuf, (size_t)h->Fpad * MCBA_FB}, {&h->fpart, (size_t)2 * h->nfblocks},
                      {&h->cpart, (size_t)2 * C * h->nfb * h->nch}, {&h->bpart, (size_t)3 * h->nbblocks}, {&h->red_own, h->nsys + 8 + 2 * MCBA_LMS},
                      {&h->dcbuf, (size_t)h->n + 8},  // + the word k_solve_backsub's solve releases, + the poll-timeout stamp
                      {&h->swork, n_swork}, {&h->dscale, h->nx}, {reinterpret_cast<double**>(&h->fixed), ((size_t)h->n + 7) / 8}};
    Piece plain[] = {{&h->rec2[0], (size_t)h->Fpad * C * MCBA_REC}, {&h->rec2[1], (size_t)h->Fpad * C * MCBA_REC},
                     {&h->spart, (size_t)h->G * h->NP * 256 + 64},
                     {&h->gchunk, h->gram_split == 3 ? mcba::gram_chunk_doubles(C, h->nfb, h->gram_nchunk, h->slots) : 0}};
    auto padded = [](size_t count) { return (std::max<size_t>(count, 1) * sizeof(double) + 255) / 256 * 256; };
    size_t zero_bytes = 0, total = 0;
    for (auto& pc : zeroed) zero_bytes += padded(pc.count);
    total = zero_bytes;
    for (auto& pc : plain) if (pc.count) total += padded(pc.count);
    if ((rc = dalloc(h, &h->solver_arena, total, false)) != MCBA_OK) return rc;
    HIPCHK(hipMemsetAsync(h->solver_arena, 0, zero_bytes, h->stream));
    if (poison_byte()) HIPCHK(hipMemsetAsync(h->solver_arena + zero_bytes, poison_byte(), total - zero_bytes, h->stream));
    size_t off = 0;
    for (auto& pc : zeroed) { *pc.p = reinterpret_cast<double*>(h->solver_arena + off); off += padded(pc.count); }
    for (auto& pc : plain) if (pc.count) { *pc.p = reinterpret_cast<double*>(h->solver_arena + off); off += padded(pc.count); }
  }
  if (mcba::solve_set_lds_limit(h->npad, h->solve_lds) != 0) return fail(MCBA_ERR_HIP, "cannot raise the dynamic LDS limit of k_solve_cam");
  h->fuse_backsub = h->solve_lds != 0;
  if (const char* e = getenv("MCBA_FUSE_BACKSUB")) h->fuse_backsub = h->fuse_backsub && atoi(e) != 0;  // tuning knob
  // (a part whose LDS limit cannot be raised keeps the two-launch k_solve_cam + k_backsub path instead of failing)
  if (h->fuse_backsub && mcba::solve_backsub_set_lds_limit(h->npad, h->cw) != 0) h->fuse_backsub = false;
  h->fuse_max_polls = 200000;
  if (const char* e = getenv("MCBA_FUSE_MAX_POLLS")) h->fuse_max_polls = std::max(0, atoi(e));  // test knob: 0 forces every poll to time out
  if (!h->ring) {
    h->ring_bytes = ((size_t)kRing * MCBA_LMS + 8) * sizeof(double);  // + the poll-timeout counter the GPU bumps
    h->ring_flags = hipHostMallocMapped | hipHostMallocCoherent;
    hipError_t e = pool_host_malloc(reinterpret_cast<void**>(&h->ring), h->ring_bytes, h->ring_flags);
    if (e != hipSuccess) { (void)hipGetLastError(); h->ring_flags = hipHostMallocDefault; e = pool_host_malloc(reinterpret_cast<void**>(&h->ring), h->ring_bytes, h->ring_flags); }
    if (e != hipSuccess) return fail(MCBA_ERR_HIP, "cannot allocate the host-mapped LM state ring");
    memset(h->ring, 0, h->ring_bytes);
    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&h->ring_dev), h->ring, 0) != hipSuccess) return fail(MCBA_ERR_HIP, "hipHostGetDevicePointer failed for the LM state ring");
  }
  h->red = h->red_own;
  if ((rc = tile_tables(h->device, h->NT, &h->tile_i, &h->tile_j)) != MCBA_OK) return rc;
  if (!h->pinned) {
    h->pinned_bytes = (h->nsys + 8 + MCBA_LMS + h->n) * sizeof(double);
    HIPCHK(pool_host_malloc(reinterpret_cast<void**>(&h->pinned), h->pinned_bytes, hipHostMallocDefault));
  }
  size_t lds = mcba::syrk_lds_bytes(C, h->FS, h->cw);
  if (lds > 64 * 1024) {
    if (mcba::syrk_set_lds_limit(lds) != 0) return fail(MCBA_ERR_HIP, "cannot raise the dynamic LDS limit of k_syrk");
  }
  h->have_solver = true;
  // a numeric x_scale / a frozen set given before an mcba_trim are still the caller's: put them back into the new buffers
  if (!h->xs_host.empty() || !h->frozen_host.empty()) return compose_dscale(h);
  return MCBA_OK;
}
#define NEED_SOLVER(h) do { int rc_ = ensure_solver(h); if (rc_) return rc_; } while (0)

int mcba_destroy(mcba_handle* h) {
  if (!h) return MCBA_OK;
  (void)hipSetDevice(h->device);
  // Device buffers are parked as "busy on h->stream" and NOT waited for (a later owner on the same stream is ordered behind whatever is
  // still in flight; pool_malloc waits for anybody else).  What the GPU writes into HOST memory must be quiet, though: the state ring
  // (ticks that were enqueued but never waited for), profiling events, an RCCL communicator -- then the stream is waited for as before.
  const bool quiet = h->last_solve_seq == h->waited_seq && h->evs.empty() && !h->comm && !h->prof;
  if (!quiet) (void)hipStreamSynchronize(h->stream);
  if (h->comm && g_rccl.ok) { g_rccl.CommDestroy(h->comm); h->comm = nullptr; }
  for (auto& b : h->bufs) { pool_free(*b.slot, b.bytes, h->device, h->stream, quiet); *b.slot = nullptr; }
  free(h->obj_host);
  pool_host_free(h->pinned, h->pinned_bytes, hipHostMallocDefault);
  pool_host_free(h->ring, h->ring_bytes, h->ring_flags);
  pool_host_free(h->pf_host, h->pf_host_bytes, hipHostMallocDefault);
  for (auto& e : h->evs) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  for (auto& e : h->pool) (void)hipEventDestroy(e);
  delete h;
  return MCBA_OK;
}

// Device memory a handle holds right now (bytes of its pooled buffers): what a caller that parks handles (api.py: the lazily attached
// result.jac) accounts for.
size_t mcba_device_bytes(const mcba_handle* h) {
  size_t tot = 0;
  if (h) for (auto& b : h->bufs) tot += b.bytes;
  return tot;
}

// Gives back (to the pool) every device buffer except what defines the problem: the observations in both layouts, the board, the two
// parameter slots (and a subset's index list).  Solver buffers (116 MB at 6 x 10 000 x 54), pre-filter scores, Jacobian / residual blocks
// go; the next call that needs them allocates them again.  For a handle that is kept only so that a Jacobian can be produced from it
// later: 0.26 GB -> 0.11 GB at that size.  Synchronises.
int mcba_trim(mcba_handle* h) {
  if (!h) return fail(MCBA_ERR_ARG, "NULL handle");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  // (the box of mcba_set_bounds stays as well: it is part of what defines the problem, and the steps after a trim must still be projected onto it)
  void** keep[] = {reinterpret_cast<void**>(&h->obs_t), reinterpret_cast<void**>(&h->obs_raw), reinterpret_cast<void**>(&h->core_arena), reinterpret_cast<void**>(&h->sub_frames),
                   reinterpret_cast<void**>(&h->blo), reinterpret_cast<void**>(&h->bhi)};
  std::vector<DevBuf> kept;
  for (auto& b : h->bufs) {
    bool k = false;
    for (void** s : keep) k = k || b.slot == s;
    if (k) { kept.push_back(b); continue; }
    pool_free(*b.slot, b.bytes, h->device);
    *b.slot = nullptr;
  }
  h->bufs.swap(kept);
  // the pieces of the solver arena
  h->rec2[0] = h->rec2[1] = h->gpart2[0] = h->gpart2[1] = h->fbuf = h->fpart = h->spart = h->cpart = h->bpart = h->red_own = h->red = h->dcbuf = h->swork = h->dscale = h->gchunk = nullptr;
  h->fixed = nullptr;
  h->cal_out_cap = h->cal_rel_cap = h->cal_sel_cap = h->cal_views_cap = 0;
  h->have_cal_poses = false;
  h->have_solver = h->have_lin = h->have_red = h->have_spec = h->have_jac = h->auto_ready = h->have_xscale = h->have_fixed = h->trial_ready = false;
  return MCBA_OK;
}

// The observations as the handle holds them ((C,F,N,2), what was uploaded or gathered), back to the host: a caller that must let go of a
// parked handle keeps the VALUES the solve saw (api.py: a released result.jac re-creates its handle from them, not from the caller's
// array, which may have changed since).
int mcba_download_observations(mcba_handle* h, double* uvs) {
  if (!h || !uvs) return fail(MCBA_ERR_ARG, "mcba_download_observations: bad argument");
  if (!h->have_obs) return fail(MCBA_ERR_ARG, "mcba_download_observations: no observations");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipMemcpyAsync(uvs, h->obs_raw, (size_t)2 * h->C * h->F * h->N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return MCBA_OK;
}

int mcba_pool_trim(void) {
  pool_release_all();
  return MCBA_OK;
}

int mcba_set_stream(mcba_handle* h, void* s) {
  if (!h) return fail(MCBA_ERR_ARG, "NULL handle");
  if (h->stream != reinterpret_cast<hipStream_t>(s)) HIPCHK(hipStreamSynchronize(h->stream));  // work (and zero fills) enqueued so far
  h->stream = reinterpret_cast<hipStream_t>(s);
  return MCBA_OK;
}

int mcba_synchronize(mcba_handle* h) {
  if (!h) return fail(MCBA_ERR_ARG, "NULL handle");
  HIPCHK(hipStreamSynchronize(h->stream));
  return MCBA_OK;
}

static int upload_impl(mcba_handle* h, const double* uvs, const double* objpoints, bool sync) {
  size_t raw_count = (size_t)2 * h->C * h->F * h->N;
  // both layouts live in HBM: raw (C,F,N) for k_jacobian (lane = point), [C][N][Fpad] for k_gram / k_cost (lane = frame)
  hipError_t e = hipMemcpyAsync(h->obs_raw, uvs, raw_count * sizeof(double), hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess && !mcba::launch_store_small(h->stream, h->obj, objpoints, (size_t)3 * h->N))   // (boards of up to 160 points: through the kernel arguments)
    e = hipMemcpyAsync(h->obj, objpoints, (size_t)3 * h->N * sizeof(double), hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess) {
    Scope sc(h, K_TRANSPOSE);
    mcba::launch_transpose_obs(h->stream, h->obs_raw, h->obs_t, h->C, h->F, h->N, h->Fpad);
  }
  if (e == hipSuccess) e = hipGetLastError();
  // (the sources are pageable host memory: the copies are staged / done when the calls return; callers that go on enqueueing need no wait)
  hipError_t e2 = sync ? hipStreamSynchronize(h->stream) : hipSuccess;
  if (e != hipSuccess || e2 != hipSuccess) { g_err = std::string("upload: ") + hipGetErrorString(e != hipSuccess ? e : e2); return MCBA_ERR_HIP; }
  if (!h->obj_host) h->obj_host = static_cast<double*>(malloc((size_t)3 * h->N * sizeof(double)));
  if (h->obj_host) memcpy(h->obj_host, objpoints, (size_t)3 * h->N * sizeof(double));
  h->planar = 1;
  for (int p = 0; p < h->N; ++p) if (objpoints[3 * p + 2] != 0.0) h->planar = 0;
  if (const char* e = getenv("MCBA_GRAM_FAST")) { if (atoi(e) == 0) h->planar = 0; }  // development knob: the general instance
  h->have_obs = true;
  h->have_lin = h->have_red = h->have_jac = false;
  return MCBA_OK;
}

int mcba_upload_observations(mcba_handle* h, const double* uvs, const double* objpoints) {
  if (!h || !uvs || !objpoints) return fail(MCBA_ERR_ARG, "mcba_upload_observations: NULL argument");
  HIPCHK(hipSetDevice(h->device));
  return upload_impl(h, uvs, objpoints, true);
}

int mcba_set_loss(mcba_handle* h, int loss, double f_scale) {
  if (!h || loss < 0 || loss > 4 || !(f_scale > 0.0)) return fail(MCBA_ERR_ARG, "mcba_set_loss: loss in 0..4 and f_scale > 0 required");
  h->loss = loss; h->f_scale = f_scale;
  h->have_lin = h->have_red = false;
  return MCBA_OK;
}

static int ensure_res(mcba_handle* h);

// least_squares' CALLABLE `loss` (least_squares.py:160-227: rho(z) -> (rho, rho', rho''); the reference forwards it untouched,
// bundle_adjustment.py:301-313): the function is the caller's, so its values are -- tab3 = three (C,F,N,2) arrays in the order of the
// observations, evaluated by the caller at the residuals of the point it is about to linearise:
//   [0] this scalar's share of the cost, 0.5 f_scale^2 rho(z)      [1] rho'(z)      [2] scipy's J_scale^2 = max(rho' + 2 rho'' z, EPS)
// (entries of missing observations are ignored).  From this call on the handle's loss is MCBA_LOSS_TABLE: mcba_linearize builds the normal
// equations with these weights (k_gram_table), the trial cost of mcba_step is NOT evaluated with it (the caller evaluates its function
// on mcba_residuals of the trial point), mcba_jacobian_eval returns unscaled rows, and the device-resident loops refuse to run -- the host
// must call the function between a step and the next linearisation.  mcba_set_loss with one of the five names switches back.
int mcba_set_loss_table(mcba_handle* h, const double* tab3) {
  if (!h || !tab3) return fail(MCBA_ERR_ARG, "mcba_set_loss_table: NULL argument");
  if (!h->have_obs) return fail(MCBA_ERR_ARG, "mcba_set_loss_table: upload observations first");
  HIPCHK(hipSetDevice(h->device));
  int rc = ensure_res(h);
  if (rc) return rc;
  const size_t raw = (size_t)2 * h->C * h->F * h->N, plane = (size_t)2 * h->C * h->N * h->Fpad;
  if (!h->loss_tab && (rc = dalloc(h, &h->loss_tab, 3 * plane, false))) return rc;
  for (int k = 0; k < 3; ++k) {   // through the residual buffer as staging (the stream orders copy -> re-layout -> next copy)
    HIPCHK(hipMemcpyAsync(h->res, tab3 + k * raw, raw * sizeof(double), hipMemcpyHostToDevice, h->stream));
    mcba::launch_transpose_obs(h->stream, h->res, h->loss_tab + k * plane, h->C, h->F, h->N, h->Fpad);
  }
  if ((rc = check_launch())) return rc;
  HIPCHK(hipStreamSynchronize(h->stream));   // (tab3 is the caller's pageable memory)
  h->loss = mcba::LOSS_TABLE;
  h->f_scale = 1.0;   // (folded into the table by the caller)
  h->have_lin = h->have_red = false;
  return MCBA_OK;
}

// Camera block width: 12 (every camera parameter is a variable: the reference, bundle_adjustment.py:149-155) or 6 = the intrinsics
// (fx fy cx cy k1 k2) of EVERY camera are held fixed -- BASELINE configs[1]; SURVEY section 8c-8's wrapper around the reference's
// residuals().  With 6 the solver kernels run role A of the linearisation alone, the camera system is 6C x 6C (row i = parameter
// 6 + i % 6 of camera i / 6: rho, t), every camera-system vector of this ABI (reduced system, camera step, `fixed` flags of
// mcba_lm_auto_config) has 6 entries per camera; the parameter vector keeps the reference's layout.  Call it before the first
// solver entry point of the handle (the solver buffers' geometry follows from it).  Partially frozen cameras: keep 12 and pass flags.
int mcba_set_camera_block(mcba_handle* h, int width) {
  if (!h || (width != 6 && width != 12)) return fail(MCBA_ERR_ARG, "mcba_set_camera_block: width 6 or 12");
  if (width == h->cw) return MCBA_OK;
  // (any solver buffer, not just a complete set: a first attempt that ran out of memory part-way leaves buffers sized for the old width behind)
  if (h->have_solver || h->solver_arena) return fail(MCBA_ERR_ARG, "mcba_set_camera_block: call it before the first solver entry point of the handle");
  if (width == 6) {
    const int nt = (6 * h->C + 1 + 15) / 16;
    if (nt * (nt + 1) / 2 > 64) return fail(MCBA_ERR_ARG, "mcba_set_camera_block: more than 26 cameras -- hold the intrinsics with the flags of mcba_lm_auto_config instead");
  }
  const int old = h->cw;
  h->cw = width;
  HIPCHK(hipSetDevice(h->device));
  const int rc = derive_geometry(h);
  if (rc != MCBA_OK) { h->cw = old; (void)derive_geometry(h); return rc; }
  h->have_lin = h->have_red = h->have_spec = false;
  h->auto_ready = false;
  return MCBA_OK;
}
int mcba_get_camera_block(const mcba_handle* h) { return h ? h->cw : 0; }

// Curvature weight of the linearisations this handle launches from now on: w = max(rho' + 2 rho'' f^2, floor rho') (csrc/mcba_math.h).
// 1 (the default) = the IRLS weight rho': monotone, the model to be far from the optimum with; 0.1 = Triggs' second-order term with a
// safety floor: the model to finish with.  A kernel argument: takes effect with the next linearisation that is ENQUEUED (the caller's
// LM driver switches between ticks; solver.py), costs nothing, changes no buffer.
int mcba_set_curvature_floor(mcba_handle* h, double floor) {
  if (!h || !(floor > 0.0) || floor > 1.0) return fail(MCBA_ERR_ARG, "mcba_set_curvature_floor: 0 < floor <= 1");
  h->curv_floor = floor;
  return MCBA_OK;
}
double mcba_get_curvature_floor(const mcba_handle* h) { return h ? h->curv_floor : 0.0; }

static int slot_ok(mcba_handle* h, int slot) { return h && (slot == 0 || slot == 1); }

int mcba_set_params(mcba_handle* h, int slot, const double* x) {
  if (!slot_ok(h, slot) || !x) return fail(MCBA_ERR_ARG, "mcba_set_params: bad handle/slot/pointer");
  HIPCHK(hipSetDevice(h->device));
  // pageable host memory: the async copy is staged by the runtime before it returns
  HIPCHK(hipMemcpyAsync(h->x[slot], x, ((size_t)12 * h->C + (size_t)6 * h->F) * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return MCBA_OK;
}

int mcba_get_params(mcba_handle* h, int slot, double* x) {
  if (!slot_ok(h, slot) || !x) return fail(MCBA_ERR_ARG, "mcba_get_params: bad handle/slot/pointer");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipMemcpyAsync(x, h->x[slot], ((size_t)12 * h->C + (size_t)6 * h->F) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return MCBA_OK;
}

int mcba_copy_params(mcba_handle* h, int dst, int src) {
  if (!slot_ok(h, dst) || !slot_ok(h, src)) return fail(MCBA_ERR_ARG, "mcba_copy_params: bad slot");
  if (dst == src) return MCBA_OK;
  HIPCHK(hipMemcpyAsync(h->x[dst], h->x[src], h->nx * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  return MCBA_OK;
}

static int run_cost(mcba_handle* h, int slot, double* res_dev, const double* bpart, int nbp) {
  {
    Scope sc(h, K_COST);
    mcba::launch_cost(h->stream, h->loss, h->f_scale, h->obs_t, h->obj, h->x[slot], h->cpart, res_dev, h->C, h->F, h->N, h->Fpad, h->nch);
  }
  int rc = check_launch();
  if (rc) return rc;
  {
    Scope sc(h, K_SUM_TRIAL);
    mcba::launch_sum_trial(h->stream, host_sel(0), h->cpart, h->cpart, 2, h->C * h->nfb * h->nch, 0, h->C * h->nfb * h->nch, bpart, nbp, h->red + h->nsys, mcba::DecideArgs{0, 0.0, 0.0, 0.0, 0.0, 0.0, nullptr});
  }
  return check_launch();
}

static int fetch_trial(mcba_handle* h, double* host8) {
  HIPCHK(hipMemcpyAsync(h->pinned + h->nsys, h->red + h->nsys, 8 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  memcpy(host8, h->pinned + h->nsys, 8 * sizeof(double));
  return MCBA_OK;
}

int mcba_cost(mcba_handle* h, int slot, double* cost, double* n_residuals) {
  if (!slot_ok(h, slot) || !cost) return fail(MCBA_ERR_ARG, "mcba_cost: bad argument");
  if (!h->have_obs) return fail(MCBA_ERR_ARG, "mcba_cost: upload observations first");
  HIPCHK(hipSetDevice(h->device));
  NEED_SOLVER(h);
  int rc = run_cost(h, slot, nullptr, nullptr, 0);
  if (rc) return rc;
  double t[8];
  rc = fetch_trial(h, t);
  if (rc) return rc;
  *cost = t[0];
  if (n_residuals) *n_residuals = t[4];
  if (!isfinite(t[0])) return fail(MCBA_ERR_NONFINITE, "Residuals are not finite");
  return MCBA_OK;
}

static int ensure_res(mcba_handle* h) {
  if (!h->res) return dalloc(h, &h->res, (size_t)2 * h->C * h->F * h->N, false);
  return MCBA_OK;
}

int mcba_residuals(mcba_handle* h, int slot, double* res) {
  if (!slot_ok(h, slot) || !res) return fail(MCBA_ERR_ARG, "mcba_residuals: bad argument");
  if (!h->have_obs) return fail(MCBA_ERR_ARG, "mcba_residuals: upload observations first");
  HIPCHK(hipSetDevice(h->device));
  NEED_SOLVER(h);
  int rc = ensure_res(h);
  if (rc) return rc;
  rc = run_cost(h, slot, h->res, nullptr, 0);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(res, h->res, (size_t)2 * h->C * h->F * h->N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return MCBA_OK;
}

int mcba_jacobian_eval(mcba_handle* h, int slot, int robust_scaled) {
  if (!slot_ok(h, slot)) return fail(MCBA_ERR_ARG, "mcba_jacobian_eval: bad argument");
  if (!h->have_obs) return fail(MCBA_ERR_ARG, "mcba_jacobian_eval: upload observations first");
  HIPCHK(hipSetDevice(h->device));
  int rc = ensure_res(h);
  if (rc) return rc;
  if (!h->jac) {
    rc = dalloc(h, &h->jac, (size_t)36 * h->C * h->F * h->N, false);
    if (rc) return rc;
  }
  {
    Scope sc(h, K_JACOBIAN);
    mcba::launch_jacobian(h->stream, h->loss, h->f_scale, h->obs_raw, h->obj, h->x[slot], h->jac, h->res, h->C, h->F, h->N, h->Fpad, robust_scaled);
  }
  rc = check_launch();
  if (rc) return rc;
  h->have_jac = true;
  return MCBA_OK;
}

int mcba_jacobian_download(mcba_handle* h, double* jac, double* res) {
  if (!h) return fail(MCBA_ERR_ARG, "NULL handle");
  if (!h->have_jac) return fail(MCBA_ERR_ARG, "mcba_jacobian_download: call mcba_jacobian_eval first");
  size_t cnt = (size_t)h->C * h->F * h->N;
  if (jac) HIPCHK(hipMemcpyAsync(jac, h->jac, cnt * 36 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (res) HIPCHK(hipMemcpyAsync(res, h->res, cnt * 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return MCBA_OK;
}

int mcba_linearize(mcba_handle* h, int slot) {
  if (!slot_ok(h, slot)) return fail(MCBA_ERR_ARG, "mcba_linearize: bad argument");
  if (!h->have_obs) return fail(MCBA_ERR_ARG, "mcba_linearize: upload observations first");
  if (h->loss == mcba::LOSS_TABLE && !h->loss_tab) return fail(MCBA_ERR_ARG, "mcba_linearize: the loss table was released (mcba_trim): call mcba_set_loss_table again");
  HIPCHK(hipSetDevice(h->device));
  NEED_SOLVER(h);
  {
    Scope sc(h, K_GRAM);
    mcba::launch_gram(h->stream, h->loss, h->f_scale, h->obs_t, h->obj, gram_sel(h, host_sel(0)), h->x[slot], h->x[slot], h->rec2[h->lin], h->rec2[h->lin], h->gpart2[h->lin], h->gpart2[h->lin], h->C, h->N, h->Fpad, h->gram_split, h->planar, h->gchunk, h->gram_nchunk, h->gram_npw, h->cw, h->slots, h->loss_tab);
  }
  int rc = check_launch();
  if (rc) return rc;
  h->have_lin = true;
  h->have_red = false;
  h->have_spec = false;
  return MCBA_OK;
}

int mcba_build_reduced(mcba_handle* h, double lambda, int rank_slot) {
  if (!h || !(lambda >= 0.0) || rank_slot < 0 || rank_slot > 11) return fail(MCBA_ERR_ARG, "mcba_build_reduced: lambda >= 0 and rank_slot in 0..11 required");
  if (!h->have_lin) return fail(MCBA_ERR_ARG, "mcba_build_reduced: call mcba_linearize first");
  HIPCHK(hipSetDevice(h->device));
  int rc;
  {
    Scope sc(h, K_SYRK);
    mcba::launch_syrk(h->stream, host_sel(h->lin, lambda), no_fuse(), h->rec2[0], h->rec2[1], h->fbuf, h->fpart, h->tile_i, h->tile_j, h->spart, h->C, h->F, h->Fpad, h->NT, h->NP, h->G, h->sq, h->sr, h->FS, h->ppw, h->have_xscale ? h->dscale : nullptr, h->cw);
  }
  if ((rc = check_launch())) return rc;
  {
    Scope sc(h, K_REDUCE);
    mcba::launch_reduce_system(h->stream, host_sel(h->lin), h->gpart2[0], h->gpart2[1], h->spart, h->fpart, h->tile_i, h->tile_j, h->red, h->C, h->nfb, h->G, h->NT, h->NP, h->nfblocks, rank_slot, nullptr, 0, nullptr, h->cw);
  }
  if ((rc = check_launch())) return rc;
  h->have_red = true;
  return MCBA_OK;
}

size_t mcba_reduced_size(const mcba_handle* h) { return h ? h->nsys + 8 + 2 * MCBA_LMS : 0; }

int mcba_bind_reduce_buffer(mcba_handle* h, double* p) {
  if (!h) return fail(MCBA_ERR_ARG, "NULL handle");
  HIPCHK(hipSetDevice(h->device));
  NEED_SOLVER(h);
  h->red = p ? p : h->red_own;
  h->have_red = false;
  return MCBA_OK;
}

int mcba_get_reduced(mcba_handle* h, double* host) {
  if (!h || !host) return fail(MCBA_ERR_ARG, "mcba_get_reduced: bad argument");
  if (!h->have_red) return fail(MCBA_ERR_ARG, "mcba_get_reduced: call mcba_build_reduced first");
  HIPCHK(hipMemcpyAsync(h->pinned, h->red, h->nsys * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  memcpy(host, h->pinned, h->nsys * sizeof(double));
  return MCBA_OK;
}

static int step_common(mcba_handle* h, const double* delta_cam, double lambda, int src, int dst) {
  mcba::CamStep cs;
  memcpy(cs.v, delta_cam, h->n * sizeof(double));
  {
    Scope sc(h, K_BACKSUB);
    // host-selected: 'current' operands are passed in position 0, the destination slot in position 1
    mcba::launch_backsub(h->stream, host_sel(0, lambda), h->rec2[h->lin], h->rec2[h->lin], h->fbuf, cs, h->x[src], h->x[dst], h->bpart, h->C, h->F, h->Fpad, h->cw);
  }
  if (h->have_bounds) mcba::launch_clip(h->stream, h->x[dst], h->blo, h->bhi, (size_t)12 * h->C + (size_t)6 * h->F);   // the trial point, projected onto the box
  return check_launch();
}

int mcba_step(mcba_handle* h, const double* delta_cam, double lambda, int src, int dst) {
  if (!slot_ok(h, src) || !slot_ok(h, dst) || src == dst || !delta_cam) return fail(MCBA_ERR_ARG, "mcba_step: bad argument (slots must differ)");
  if (!h->have_red) return fail(MCBA_ERR_ARG, "mcba_step: call mcba_build_reduced first");
  HIPCHK(hipSetDevice(h->device));
  int rc = step_common(h, delta_cam, lambda, src, dst);
  if (rc) return rc;
  h->have_spec = false;
  return run_cost(h, dst, nullptr, h->bpart, h->nbblocks);
}

int mcba_step_linearize(mcba_handle* h, const double* delta_cam, double lambda, int src, int dst) {
  if (!slot_ok(h, src) || !slot_ok(h, dst) || src == dst || !delta_cam) return fail(MCBA_ERR_ARG, "mcba_step_linearize: bad argument (slots must differ)");
  if (!h->have_red) return fail(MCBA_ERR_ARG, "mcba_step_linearize: call mcba_build_reduced first");
  if (h->loss == mcba::LOSS_TABLE) return fail(MCBA_ERR_ARG, "mcba_step_linearize: a tabulated loss is set -- step (mcba_step), evaluate the function at the trial residuals, mcba_set_loss_table, then mcba_linearize");
  HIPCHK(hipSetDevice(h->device));
  int rc = step_common(h, delta_cam, lambda, src, dst);
  if (rc) return rc;
  const int alt = 1 - h->lin;
  {
    Scope sc(h, K_GRAM);
    mcba::launch_gram(h->stream, h->loss, h->f_scale, h->obs_t, h->obj, gram_sel(h, host_sel(0)), h->x[dst], h->x[dst], h->rec2[alt], h->rec2[alt], h->gpart2[alt], h->gpart2[alt], h->C, h->N, h->Fpad, h->gram_split, h->planar, h->gchunk, h->gram_nchunk, h->gram_npw, h->cw, h->slots, h->loss_tab);
  }
  if ((rc = check_launch())) return rc;
  {
    Scope sc(h, K_SUM_TRIAL);
    mcba::launch_sum_trial(h->stream, host_sel(0), h->gpart2[alt] + (size_t)90 * h->nfb, h->gpart2[alt] + (size_t)90 * h->nfb, 1, h->nfb, (size_t)MCBA_GP * h->nfb, h->C * h->nfb, h->bpart, h->nbblocks, h->red + h->nsys, mcba::DecideArgs{0, 0.0, 0.0, 0.0, 0.0, 0.0, nullptr});
  }
  if ((rc = check_launch())) return rc;
  h->have_spec = true;
  return MCBA_OK;
}

int mcba_accept_linearization(mcba_handle* h) {
  if (!h) return fail(MCBA_ERR_ARG, "NULL handle");
  if (!h->have_spec) return fail(MCBA_ERR_ARG, "mcba_accept_linearization: no speculative linearisation (call mcba_step_linearize first)");
  h->lin = 1 - h->lin;
  h->have_spec = false;
  h->have_red = false;
  return MCBA_OK;
}

int mcba_get_trial(mcba_handle* h, double* host8) {
  if (!h || !host8) return fail(MCBA_ERR_ARG, "mcba_get_trial: bad argument");
  return fetch_trial(h, host8);
}

// Write the eight trial scalars back into the reduce buffer: with a tabulated loss (mcba_set_loss_table) the cost of this shard's trial point
// is the caller's function on its residuals -- a frame-sharded run fetches the scalars of mcba_step (mcba_get_trial), evaluates the function
// (mcba_residuals, which reuses the scalars' slots), and puts [its cost, the step's other scalars] here BEFORE the all-reduce.
int mcba_set_trial(mcba_handle* h, const double* host8) {
  if (!h || !host8 || !h->red) return fail(MCBA_ERR_ARG, "mcba_set_trial: bad argument, or no reduce buffer yet (call mcba_step first)");
  HIPCHK(hipSetDevice(h->device));
  memcpy(h->pinned + h->nsys, host8, 8 * sizeof(double));
  HIPCHK(hipMemcpyAsync(h->red + h->nsys, h->pinned + h->nsys, 8 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return MCBA_OK;
}

int mcba_reduce_fetch(mcba_handle* h, double lambda, int rank_slot, double* host) {
  int rc = mcba_build_reduced(h, lambda, rank_slot);
  if (rc) return rc;
  return mcba_get_reduced(h, host);
}

int mcba_step_fetch(mcba_handle* h, const double* delta_cam, double lambda, int src, int dst, int linearize, double* host8) {
  int rc = linearize ? mcba_step_linearize(h, delta_cam, lambda, src, dst) : mcba_step(h, delta_cam, lambda, src, dst);
  if (rc) return rc;
  return mcba_get_trial(h, host8);
}

// ---------------------------------------------------------------------------------------------------------
// Device-resident LM iteration: the accept/reject decision and the damping update happen on the GPU (k_decide), so
// backsub -> gram(trial) -> sum -> decide -> frame_factor -> syrk -> reduce is ONE stream-ordered chain.
// Convention while it is in use: parameter slot i and linearisation buffer i belong together; state[3] = current i.
int mcba_lm_set_state(mcba_handle* h, const double* state) {
  if (!h || !state) return fail(MCBA_ERR_ARG, "mcba_lm_set_state: bad argument");
  int sel = (int)state[3];
  if (sel != 0 && sel != 1) return fail(MCBA_ERR_ARG, "mcba_lm_set_state: state[3] must be 0 or 1");
  HIPCHK(hipSetDevice(h->device));
  NEED_SOLVER(h);
  double* stage = h->pinned + h->nsys + 8;
  memcpy(stage, state, MCBA_LMS * sizeof(double));
  if (!(stage[MCBA_LM_CFL] > 0.0)) stage[MCBA_LM_CFL] = h->curv_floor;  // (a caller that fills the first four entries only: the handle's model, fixed)
  HIPCHK(hipMemcpyAsync(h->red + h->nsys + 8, stage, MCBA_LMS * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (sel != h->lin) {
    // the accepted linearisation must live in buffer `sel`: swap the buffer pointers instead of copying 48 MB
    std::swap(h->rec2[0], h->rec2[1]);
    std::swap(h->gpart2[0], h->gpart2[1]);
    h->lin = sel;
  }
  h->trial_ready = false;
  return MCBA_OK;
}

static int lm_trial_impl(mcba_handle* h, const double* delta_cam, const mcba::DecideArgs& da) {
  if (!h || !delta_cam) return fail(MCBA_ERR_ARG, "mcba_lm_trial: bad argument");
  if (h->have_bounds) return fail(MCBA_ERR_ARG, "box constraints are set (mcba_set_bounds): drive the steps with mcba_step / mcba_step_linearize (the host-driven loop)");
  if (h->loss == mcba::LOSS_TABLE) return fail(MCBA_ERR_ARG, "a tabulated loss is set (mcba_set_loss_table): the device-resident loops cannot call the caller's function -- use the host-driven loop");
  if (!h->have_lin) return fail(MCBA_ERR_ARG, "mcba_lm_trial: no linearisation");
  HIPCHK(hipSetDevice(h->device));
  mcba::CamStep cs;
  memcpy(cs.v, delta_cam, h->n * sizeof(double));
  int rc;
  {
    Scope sc(h, K_BACKSUB);
    mcba::launch_backsub(h->stream, dev_sel(h, 0), h->rec2[0], h->rec2[1], h->fbuf, cs, h->x[0], h->x[1], h->bpart, h->C, h->F, h->Fpad, h->cw);
  }
  if ((rc = check_launch())) return rc;
  {
    Scope sc(h, K_GRAM);  // trial point = the OTHER slot / buffer
    mcba::launch_gram(h->stream, h->loss, h->f_scale, h->obs_t, h->obj, gram_sel(h, dev_sel(h, 1)), h->x[0], h->x[1], h->rec2[0], h->rec2[1], h->gpart2[0], h->gpart2[1], h->C, h->N, h->Fpad, h->gram_split, h->planar, h->gchunk, h->gram_nchunk, h->gram_npw, h->cw, h->slots, h->loss_tab);
  }
  if ((rc = check_launch())) return rc;
  {
    Scope sc(h, K_SUM_TRIAL);
    mcba::launch_sum_trial(h->stream, dev_sel(h, 1), h->gpart2[0] + (size_t)90 * h->nfb, h->gpart2[1] + (size_t)90 * h->nfb, 1, h->nfb, (size_t)MCBA_GP * h->nfb, h->C * h->nfb, h->bpart, h->nbblocks, h->red + h->nsys, da);
  }
  return check_launch();
}

int mcba_lm_trial(mcba_handle* h, const double* delta_cam) {
  return lm_trial_impl(h, delta_cam, mcba::DecideArgs{0, 0.0, 0.0, 0.0, 0.0, 0.0, nullptr});
}

// decide_here: k_syrk itself sums the trial scalars and takes the accept / reject decision (single-GPU ticks); it reads the
// state the previous tick left and publishes the decided state to the second buffer, which the rest of the tick reads.
static double* timeout_word(const mcba_handle* h) { return h->dcbuf + h->n + 1; }  // behind the camera step and the release word
static int lm_reduce_chain(mcba_handle* h, int rank_slot, bool spec = false, bool decide_here = false, unsigned long long seq = 0) {
  int rc;
  const mcba::Sel sl = spec ? spec_sel(h) : dev_sel(h, 0);
  mcba::SyrkFuse fz = no_fuse();
  if (decide_here) {
    fz.decide = 1;
    fz.cp0 = h->gpart2[0] + (size_t)90 * h->nfb;
    fz.cp1 = h->gpart2[1] + (size_t)90 * h->nfb;
    fz.cstride = 4;  // k_gram: per-workgroup sums in every fourth frame block's slot ...
    fz.cinner = (h->nfb + 3) / 4;
    fz.cdense = 1 << 30;
    // ... except where its point-split variant ran (one frame block per wavefront group: every slot holds its own cost)
    if (h->gram_split == 4) { fz.cdense = 0; fz.cinner = h->nfb; }
    else if (h->gram_split == 5) {
      const int fba = mcba::gram_round_blocks(h->C, h->nfb, h->slots);
      if (fba > 0 && fba < h->nfb) { fz.cdense = fba / 4; fz.cinner = fba / 4 + (h->nfb - fba); }
    }
    fz.couter = (size_t)MCBA_GP * h->nfb;
    fz.ncp = h->C * fz.cinner;
    fz.bpart = h->bpart;
    fz.nbp = h->nbblocks;
    fz.trial_out = h->red + h->nsys;
    fz.lms_post = post_state(h);
    fz.da = mcba::DecideArgs{2, 0.0, 0.0, 0.0, h->lam_min, h->lam_max, nullptr, h->ftol, h->xtol, h->dec_floor};
    fz.timeout_word = timeout_word(h);
    fz.seq_prev = seq > 0 ? (double)(seq - 1) : 0.0;
  }
  {
    Scope sc(h, K_SYRK);
    mcba::launch_syrk(h->stream, sl, fz, h->rec2[0], h->rec2[1], h->fbuf, h->fpart, h->tile_i, h->tile_j, h->spart, h->C, h->F, h->Fpad, h->NT, h->NP, h->G, h->sq, h->sr, h->FS, h->ppw, h->have_xscale ? h->dscale : nullptr, h->cw);
  }
  if ((rc = check_launch())) return rc;
  {
    Scope sc(h, K_REDUCE);
    mcba::launch_reduce_system(h->stream, decide_here ? post_sel(h) : sl, h->gpart2[0], h->gpart2[1], h->spart, h->fpart, h->tile_i, h->tile_j, h->red, h->C, h->nfb, h->G, h->NT, h->NP, h->nfblocks, rank_slot,
                               spec ? h->bpart : nullptr, h->nbblocks, spec ? post_state(h) : nullptr, h->cw, spec ? timeout_word(h) : nullptr, (double)h->last_solve_seq);
  }
  if ((rc = check_launch())) return rc;
  h->have_red = true;
  h->spec_copy_ready = spec;
  return MCBA_OK;
}

int mcba_lm_decide_reduce(mcba_handle* h, double pred_cam, double dcn2, double xcn2, double lam_min, double lam_max, int rank_slot) {
  if (!h || rank_slot < 0 || rank_slot > 11) return fail(MCBA_ERR_ARG, "mcba_lm_decide_reduce: bad argument");
  HIPCHK(hipSetDevice(h->device));
  {
    Scope sc(h, K_DECIDE);
    mcba::launch_decide(h->stream, h->red + h->nsys, mcba::DecideArgs{1, pred_cam, dcn2, xcn2, lam_min, lam_max, h->red + h->nsys + 8, 0.0, 0.0, h->dec_floor});
  }
  int rc = check_launch();
  if (rc) return rc;
  return lm_reduce_chain(h, rank_slot);
}

int mcba_lm_rebuild(mcba_handle* h, int rank_slot) {
  if (!h || rank_slot < 0 || rank_slot > 11) return fail(MCBA_ERR_ARG, "mcba_lm_rebuild: bad argument");
  HIPCHK(hipSetDevice(h->device));
  return lm_reduce_chain(h, rank_slot);
}

int mcba_lm_fetch(mcba_handle* h, double* host) {
  if (!h || !host) return fail(MCBA_ERR_ARG, "mcba_lm_fetch: bad argument");
  size_t cnt = h->nsys + 8 + MCBA_LMS;
  HIPCHK(hipMemcpyAsync(h->pinned, h->red, cnt * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  memcpy(host, h->pinned, cnt * sizeof(double));
  int sel = (int)host[h->nsys + 8 + 3];
  if (sel == 0 || sel == 1) h->lin = sel;  // keep the host-selected entry points consistent with the device state
  h->have_spec = false;
  return MCBA_OK;
}

int mcba_lm_iterate(mcba_handle* h, const double* delta_cam, double pred_cam, double dcn2, double xcn2, double lam_min, double lam_max, double* host) {
  if (!h) return fail(MCBA_ERR_ARG, "NULL handle");
  // single rank: the decision rides on k_sum_trial (no separate launch, nothing to all-reduce in between)
  int rc = lm_trial_impl(h, delta_cam, mcba::DecideArgs{1, pred_cam, dcn2, xcn2, lam_min, lam_max, h->red + h->nsys + 8, 0.0, 0.0, h->dec_floor});
  if (rc) return rc;
  rc = lm_reduce_chain(h, 0);
  if (rc) return rc;
  return mcba_lm_fetch(h, host);
}


// ---------------------------------------------------------------------------------------------------------
// Device-resident LM loop: the reduced camera system is solved on the GPU too (k_solve_cam), the termination tests run
// there, and the host only enqueues "ticks" and reads the 32-double state each one posts to a host-mapped ring:
//   one GPU:        tick = k_gram(trial) -> k_syrk (trial sums + decision + frame factors + SYRK) -> k_reduce_system -> k_solve_backsub (solve + the
//                   back-substitution of the next trial step; k_backsub / k_solve_cam apart for the first tick, > 9 cameras, MCBA_FUSE_BACKSUB=0)
//   frame-sharded:  tick = k_backsub -> k_gram(trial) -> k_syrk (speculative) -> k_reduce_system (+ trial scalars) -> all-reduce -> k_solve_cam (decides)
//                   (MCBA_SPECULATE=0: k_sum_trial -> all-reduce -> k_decide -> k_syrk -> k_reduce_system -> all-reduce -> k_solve_cam)
// No host synchronisation inside or between ticks; after termination the remaining ticks return immediately.
int mcba_lm_auto_config(mcba_handle* h, double ftol, double xtol, double gtol, double lam_min, double lam_max, const unsigned char* fixed) {
  if (!h || !(lam_min > 0.0) || !(lam_max > lam_min)) return fail(MCBA_ERR_ARG, "mcba_lm_auto_config: bad argument");
  HIPCHK(hipSetDevice(h->device));
  NEED_SOLVER(h);
  h->ftol = ftol; h->xtol = xtol; h->gtol = gtol; h->lam_min = lam_min; h->lam_max = lam_max;
  h->have_fixed = fixed != nullptr;
  if (fixed) {
    HIPCHK(hipMemcpyAsync(h->fixed, fixed, (size_t)h->n, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  memset(h->ring, 0, (size_t)kRing * MCBA_LMS * sizeof(double));
  HIPCHK(hipMemsetAsync(h->dcbuf + h->n, 0, 8 * sizeof(double), h->stream));  // sequence numbers restart: no stale release word
  HIPCHK(hipStreamSynchronize(h->stream));
  h->trial_ready = false;
  h->last_solve_seq = 0;
  h->waited_seq = 0;
  h->auto_ready = true;
  if (const char* e = getenv("MCBA_SPECULATE")) h->speculate = atoi(e) != 0;
  return MCBA_OK;
}

static int auto_solve_impl(mcba_handle* h, unsigned long long seq, int decide, bool decided_by_syrk, bool fuse_next = false) {
  // frame-sharded ticks with one collective (the decision is taken here, after a speculative reduction that left a copy of the
  // pre-decision state behind): the back-substitution of the next trial step rides along as well
  if (decide && h && h->fuse_backsub && h->spec_copy_ready) fuse_next = true;
  if (!h || !h->auto_ready || seq == 0) return fail(MCBA_ERR_ARG, "mcba_lm_auto_solve: call mcba_lm_auto_config first; seq >= 1");
  if (!h->have_red) return fail(MCBA_ERR_ARG, "mcba_lm_auto_solve: no reduced system");
  HIPCHK(hipSetDevice(h->device));
  mcba::SolveArgs a;
  a.red = h->red; a.lms = h->red + h->nsys + 8; a.lms_in = decided_by_syrk ? post_state(h) : a.lms; a.work = h->swork; a.dc = h->dcbuf; a.x0 = h->x[0]; a.x1 = h->x[1];
  a.fixed = h->have_fixed ? h->fixed : nullptr;
  a.dscale = h->have_xscale ? h->dscale : nullptr;
  a.host_state = h->ring_dev + (size_t)(seq % kRing) * MCBA_LMS;
  a.flag = fuse_next ? h->dcbuf + h->n : nullptr;
  a.timeout_word = timeout_word(h);
  a.seq = (double)seq; a.gtol = h->gtol; a.lam_max = h->lam_max;
  h->last_solve_seq = seq;
  a.stage_tag = (double)(++h->solve_launches);
  a.n = h->n; a.npad = h->npad; a.use_lds = h->solve_lds; a.cw = h->cw;
  a.decide = decide ? 1 : 0; a.lam_min = h->lam_min; a.ftol = h->ftol; a.xtol = h->xtol; a.dec_floor = h->dec_floor;
  {
    Scope sc(h, K_SOLVE);
    if (fuse_next)  // + the back-substitution of the next tick's trial step, overlapped with the solve (polls bounded: ~0.5 s)
      mcba::launch_solve_backsub(h->stream, a, dev_sel(h, 0), h->rec2[0], h->rec2[1], h->fbuf, h->x[0], h->x[1], h->bpart, h->C, h->F, h->Fpad, decide ? post_state(h) : a.lms_in, h->fuse_max_polls, decide ? 1 : 0,
                                 timeout_word(h), h->ring_dev + (size_t)kRing * MCBA_LMS, h->strict_sync ? 1 : 0);
    else
      mcba::launch_solve_cam(h->stream, a);
  }
  h->trial_ready = fuse_next;
  return check_launch();
}

int mcba_lm_auto_solve(mcba_handle* h, unsigned long long seq, int decide) { return auto_solve_impl(h, seq, decide, false); }

// The release-word protocol between the solve and the back-substitution workgroups of k_solve_backsub (csrc/mcba_backsub.h): by default
// (round 6) the readers ACQUIRE the word with an agent-scope fence behind the poll -- the form the HIP memory model asks for.  on == 0 selects,
// at run time for this handle, the relaxed reader (agent-scope relaxed loads that bypass the per-XCD L2 + in-order issue): ~1.3 us per
// iteration faster, stress-tested, but a data race by the model.  A new handle starts from MCBA_STRICT_SYNC in the environment (unset = 1).
// Same results to the bit either way.
int mcba_set_strict_sync(mcba_handle* h, int on) {
  if (!h) return fail(MCBA_ERR_ARG, "NULL handle");
  h->strict_sync = on != 0;
  return MCBA_OK;
}
int mcba_get_strict_sync(const mcba_handle* h) { return h && h->strict_sync ? 1 : 0; }

int mcba_lm_set_decrease_floor(mcba_handle* h, double dec_floor) {
  if (!h || !(dec_floor >= 0.0) || dec_floor >= 1.0) return fail(MCBA_ERR_ARG, "mcba_lm_set_decrease_floor: 0 <= floor < 1 required (0 = 1/3)");
  h->dec_floor = dec_floor;
  return MCBA_OK;
}

// sum_here: k_sum_trial follows (frame-sharded ticks: the trial scalars are all-reduced); otherwise k_syrk sums and decides
static int auto_trial_impl(mcba_handle* h, int decide, bool sum_here) {
  if (!h || !h->auto_ready) return fail(MCBA_ERR_ARG, "mcba_lm_auto_trial: call mcba_lm_auto_config first");
  if (h->have_bounds) return fail(MCBA_ERR_ARG, "box constraints are set (mcba_set_bounds): the device-resident loop does not project its trial points -- use the host-driven loop");
  if (h->loss == mcba::LOSS_TABLE) return fail(MCBA_ERR_ARG, "a tabulated loss is set (mcba_set_loss_table): the device-resident loops cannot call the caller's function -- use the host-driven loop");
  if (!h->have_lin) return fail(MCBA_ERR_ARG, "mcba_lm_auto_trial: no linearisation");
  HIPCHK(hipSetDevice(h->device));
  int rc;
  if (!h->trial_ready) {  // (else the previous tick's k_solve_backsub has already produced this trial step)
    Scope sc(h, K_BACKSUB);
    mcba::launch_backsub_dev(h->stream, dev_sel(h, 0), h->rec2[0], h->rec2[1], h->fbuf, h->dcbuf, h->x[0], h->x[1], h->bpart, h->C, h->F, h->Fpad, h->cw);
  }
  h->trial_ready = false;
  if ((rc = check_launch())) return rc;
  {
    Scope sc(h, K_GRAM);
    mcba::launch_gram(h->stream, h->loss, h->f_scale, h->obs_t, h->obj, gram_sel(h, dev_sel(h, 1)), h->x[0], h->x[1], h->rec2[0], h->rec2[1], h->gpart2[0], h->gpart2[1], h->C, h->N, h->Fpad, h->gram_split, h->planar, h->gchunk, h->gram_nchunk, h->gram_npw, h->cw, h->slots, h->loss_tab);
  }
  if ((rc = check_launch())) return rc;
  if (!sum_here) return MCBA_OK;
  {
    Scope sc(h, K_SUM_TRIAL);
    mcba::DecideArgs da{decide ? 2 : 0, 0.0, 0.0, 0.0, h->lam_min, h->lam_max, h->red + h->nsys + 8, h->ftol, h->xtol, h->dec_floor};
    mcba::launch_sum_trial(h->stream, dev_sel(h, 1), h->gpart2[0] + (size_t)90 * h->nfb, h->gpart2[1] + (size_t)90 * h->nfb, 1, h->nfb, (size_t)MCBA_GP * h->nfb, h->C * h->nfb, h->bpart, h->nbblocks, h->red + h->nsys, da);
  }
  return check_launch();
}

// decide: 0 k_sum_trial follows (the trial scalars are all-reduced on their own), != 0 it also decides, -1 no k_sum_trial:
// the speculative reduction (mcba_lm_auto_reduce(h, 2, .)) sums the trial scalars itself
int mcba_lm_auto_trial(mcba_handle* h, int decide) { return auto_trial_impl(h, decide < 0 ? 0 : decide, decide >= 0); }

int mcba_lm_auto_reduce(mcba_handle* h, int decide, int rank_slot) {
  if (!h || !h->auto_ready || rank_slot < 0 || rank_slot > 11 || decide < 0 || decide > 2) return fail(MCBA_ERR_ARG, "mcba_lm_auto_reduce: bad argument");
  HIPCHK(hipSetDevice(h->device));
  if (decide == 1) {
    {
      Scope sc(h, K_DECIDE);
      mcba::launch_decide(h->stream, h->red + h->nsys, mcba::DecideArgs{2, 0.0, 0.0, 0.0, h->lam_min, h->lam_max, h->red + h->nsys + 8, h->ftol, h->xtol, h->dec_floor});
    }
    int rc = check_launch();
    if (rc) return rc;
  }
  return lm_reduce_chain(h, rank_slot, decide == 2);
}

int mcba_lm_auto_tick(mcba_handle* h, unsigned long long seq, int rank_slot) {
  if (!h) return fail(MCBA_ERR_ARG, "NULL handle");
  const bool coll = h->comm != nullptr;
  int rc;
  if (!coll) {  // one GPU: [k_backsub ->] k_gram -> k_syrk (trial sums + decision + frame factors + SYRK) -> k_reduce_system -> k_solve_backsub
    if (rank_slot < 0 || rank_slot > 11) return fail(MCBA_ERR_ARG, "mcba_lm_auto_tick: bad rank slot");
    if ((rc = auto_trial_impl(h, 0, false))) return rc;
    if ((rc = lm_reduce_chain(h, rank_slot, false, true, seq))) return rc;
    return auto_solve_impl(h, seq, 0, true, h->fuse_backsub);
  }
  if (h->speculate) {  // ONE collective: speculative reduction, [system | trial scalars] all-reduced together, decision in k_solve_cam
    if ((rc = mcba_lm_auto_trial(h, -1))) return rc;
    if ((rc = mcba_lm_auto_reduce(h, 2, rank_slot))) return rc;
    if ((rc = mcba_comm_allreduce(h, 0, h->nsys + 8))) return rc;
    return mcba_lm_auto_solve(h, seq, 1);
  }
  if ((rc = mcba_lm_auto_trial(h, 0))) return rc;
  if ((rc = mcba_comm_allreduce(h, h->nsys, 8))) return rc;
  if ((rc = mcba_lm_auto_reduce(h, 1, rank_slot))) return rc;
  if ((rc = mcba_comm_allreduce(h, 0, h->nsys))) return rc;
  return mcba_lm_auto_solve(h, seq, 0);
}

int mcba_get_cam_step(mcba_handle* h, double* host) {
  if (!h || !host) return fail(MCBA_ERR_ARG, "mcba_get_cam_step: bad argument");
  HIPCHK(hipSetDevice(h->device));
  NEED_SOLVER(h);
  HIPCHK(hipMemcpyAsync(host, h->dcbuf, (size_t)h->n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return MCBA_OK;
}

int mcba_lm_auto_wait(mcba_handle* h, unsigned long long seq, double* state) {
  if (!h || !state || !h->auto_ready || seq == 0) return fail(MCBA_ERR_ARG, "mcba_lm_auto_wait: bad argument");
  volatile double* slot = h->ring + (size_t)(seq % kRing) * MCBA_LMS;
  const double want = (double)seq;
  auto t0 = std::chrono::steady_clock::now();
  bool synced = false;
  for (unsigned spin = 0;; ++spin) {
    if (slot[MCBA_LM_SEQ] == want) break;
    __builtin_ia32_pause();
    if ((spin & 0xFFF) == 0xFFF && !synced) {
      double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (el > 0.05) {  // not the fast path any more: block on the stream, then look once more
        HIPCHK(hipSetDevice(h->device));
        HIPCHK(hipStreamSynchronize(h->stream));
        synced = true;
        if (slot[MCBA_LM_SEQ] != want) return fail(MCBA_ERR_ARG, "mcba_lm_auto_wait: that tick was never enqueued (or the ring slot was overwritten)");
      }
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  for (int i = 0; i < MCBA_LMS; ++i) state[i] = slot[i];
  if (seq > h->waited_seq) h->waited_seq = seq;
  if (h->fuse_backsub && const_cast<volatile double*>(h->ring)[(size_t)kRing * MCBA_LMS] != 0.0) {
    // a back-substitution workgroup of k_solve_backsub gave up waiting for the solve (mcba_backsub.h): the ticks already in
    // flight discard their stale trial points on the device; from here on the solve and the back-substitution are two launches
    h->fuse_backsub = false;
    h->trial_ready = false;
  }
  int sel = (int)state[3];
  if (sel == 0 || sel == 1) h->lin = sel;
  h->have_spec = false;
  return MCBA_OK;
}

// ---------------------------------------------------------------------------------------------------------
// The whole device-resident LM loop in ONE call (round 5): what solver.LevenbergMarquardt.start() + its iterate() loop + finalize() do
// through a dozen crossings and six host synchronisations before the first tick -- upload x0, linearise, reduce, read the cost back,
// write the state, configure, solve -- is enqueued here without a single wait: the start state is written ON THE DEVICE from the reduced
// system (k_lm_init), the first ticks are enqueued behind the first solve at once, and the host then only polls the ring.  Same ticks,
// same decisions, same order as the Python loop (the device decides; this loop only chooses how many ticks are in flight, by the same
// rule).  opt: 0 ftol 1 xtol 2 gtol 3 lam0 4 lam_min 5 lam_max 6 dec_floor 7 curvature floor 8 curvature switch (0 = fixed model)
// 9 max_nfev 10 max ticks (< 0: no limit) 11 ticks in flight (depth) 12 rank slot.  x0 NULL = start from what slot 0 holds.
// summary: 0 status (scipy's; 0 = a limit was reached) 1 rows recorded 2 rows the main loop consumed (the rest were retired by the
// final drain) -- fetch the rows with mcba_lm_history.
int mcba_lm_run(mcba_handle* h, const double* x0, const double* opt, const unsigned char* fixed, double* summary) {
  if (!h || !opt || !summary) return fail(MCBA_ERR_ARG, "mcba_lm_run: bad argument");
  if (!h->have_obs) return fail(MCBA_ERR_ARG, "mcba_lm_run: upload observations first");
  if (h->have_bounds) return fail(MCBA_ERR_ARG, "mcba_lm_run: box constraints are set (mcba_set_bounds) -- use the host-driven loop");
  if (h->loss == mcba::LOSS_TABLE) return fail(MCBA_ERR_ARG, "mcba_lm_run: a tabulated loss is set (mcba_set_loss_table) -- use the host-driven loop");
  const double lam0 = opt[3], lam_min = opt[4], lam_max = opt[5], cfl = opt[7], cfl_switch = opt[8];
  const int depth = std::max(1, std::min((int)opt[11], 12)), rank_slot = (int)opt[12];
  const double max_nfev = opt[9], max_ticks = opt[10];
  if (!(lam0 > 0.0) || !(lam_min > 0.0) || !(lam_max > lam_min) || !(cfl > 0.0) || cfl > 1.0 || rank_slot < 0 || rank_slot > 11 || !(opt[6] >= 0.0) || opt[6] >= 1.0)
    return fail(MCBA_ERR_ARG, "mcba_lm_run: bad option");
  HIPCHK(hipSetDevice(h->device));
  NEED_SOLVER(h);
  int rc;
  if (h->auto_ready) HIPCHK(hipStreamSynchronize(h->stream));  // no tick of an earlier run may still be posting into the ring
  if (x0) HIPCHK(hipMemcpyAsync(h->x[0], x0, ((size_t)12 * h->C + (size_t)6 * h->F) * sizeof(double), hipMemcpyHostToDevice, h->stream));  // (pageable: staged when the call returns)
  h->curv_floor = cfl;
  h->dec_floor = opt[6];
  h->lin = 0;   // parameter slot 0 and linearisation buffer 0 belong together (mcba_lm_set_state's convention)
  {
    Scope sc(h, K_GRAM);
    mcba::launch_gram(h->stream, h->loss, h->f_scale, h->obs_t, h->obj, gram_sel(h, host_sel(0)), h->x[0], h->x[0], h->rec2[0], h->rec2[0], h->gpart2[0], h->gpart2[0], h->C, h->N, h->Fpad, h->gram_split, h->planar, h->gchunk, h->gram_nchunk, h->gram_npw, h->cw, h->slots, h->loss_tab);
  }
  if ((rc = check_launch())) return rc;
  h->have_lin = true; h->have_spec = false;
  if ((rc = mcba_build_reduced(h, lam0, rank_slot))) return rc;
  if (h->comm && (rc = mcba_comm_allreduce(h, 0, h->nsys))) return rc;
  mcba::launch_lm_init(h->stream, h->red + (size_t)h->n * h->n + 3 * (size_t)h->n, h->red + h->nsys + 8, lam0, 0, cfl, cfl_switch, h->dcbuf + h->n);
  if ((rc = check_launch())) return rc;
  // mcba_lm_auto_config, without its two waits
  h->ftol = opt[0]; h->xtol = opt[1]; h->gtol = opt[2]; h->lam_min = lam_min; h->lam_max = lam_max;
  h->have_fixed = fixed != nullptr;
  if (fixed) HIPCHK(hipMemcpyAsync(h->fixed, fixed, (size_t)h->n, hipMemcpyHostToDevice, h->stream));
  memset(h->ring, 0, (size_t)kRing * MCBA_LMS * sizeof(double));
  h->trial_ready = false;
  h->last_solve_seq = 0;
  h->waited_seq = 0;
  h->auto_ready = true;
  if (const char* e = getenv("MCBA_SPECULATE")) h->speculate = atoi(e) != 0;
  // (with <= 9 cameras the first solve's launch already carries the back-substitution of the first trial step, like every later one)
  if ((rc = auto_solve_impl(h, 1, 0, false, h->fuse_backsub))) return rc;

  h->hist.clear();
  unsigned long long issued = 1, retired = 1;
  double nfev = 1.0, st[MCBA_LMS];
  auto top_up = [&]() -> int {
    while (issued - retired < (unsigned long long)depth) {
      const double inflight = (double)(issued - retired);
      if (nfev + inflight >= max_nfev) break;
      if (max_ticks >= 0.0 && (double)(issued - 1) >= max_ticks) break;
      ++issued;
      int r = mcba_lm_auto_tick(h, issued, rank_slot);
      if (r) return r;
    }
    return MCBA_OK;
  };
  if ((rc = top_up())) return rc;            // the first ticks go in behind the first solve: nobody waits for it on the way
  if ((rc = mcba_lm_auto_wait(h, 1, st))) return rc;
  h->hist.insert(h->hist.end(), st, st + MCBA_LMS);
  if (!std::isfinite(st[0])) return fail(MCBA_ERR_NONFINITE, "Residuals are not finite in the initial point.");
  const int status0 = (int)st[MCBA_LM_DONE];
  int status = -1;
  double steps = 0.0;
  for (;;) {
    if (nfev >= max_nfev || (max_ticks >= 0.0 && steps >= max_ticks)) { status = 0; break; }
    steps += 1.0;
    if (status0) { status = status0; break; }
    if ((rc = top_up())) return rc;
    if (issued == retired) { status = 0; break; }
    ++retired;
    if ((rc = mcba_lm_auto_wait(h, retired, st))) return rc;
    h->hist.insert(h->hist.end(), st, st + MCBA_LMS);
    if (st[MCBA_LM_REBUILD] == 0.0) nfev = 1.0 + st[MCBA_LM_NFEV];
    if (st[MCBA_LM_DONE] != 0.0) { status = (int)st[MCBA_LM_DONE]; break; }
  }
  const size_t n_main = h->hist.size() / MCBA_LMS;
  while (retired < issued) {   // stopped with ticks in flight: retire them (their accepted steps count unless the loop had terminated)
    ++retired;
    if ((rc = mcba_lm_auto_wait(h, retired, st))) return rc;
    h->hist.insert(h->hist.end(), st, st + MCBA_LMS);
  }
  summary[0] = (double)status;
  summary[1] = (double)(h->hist.size() / MCBA_LMS);
  summary[2] = (double)n_main;
  summary[3] = steps;
  return MCBA_OK;
}

int mcba_lm_history(mcba_handle* h, double* rows, size_t capacity_rows) {
  if (!h || !rows) return fail(MCBA_ERR_ARG, "mcba_lm_history: bad argument");
  const size_t n = h->hist.size() / MCBA_LMS;
  if (capacity_rows < n) return fail(MCBA_ERR_ARG, "mcba_lm_history: buffer too small (summary[1] of mcba_lm_run rows)");
  if (n) memcpy(rows, h->hist.data(), h->hist.size() * sizeof(double));
  return MCBA_OK;
}

// Solution and gradient of the current point in ONE device-to-host copy: out = [x (12C + 6F) | gradient (12C + 6F)] -- x of `slot`, the
// camera gradient of the reduced system in the reduce buffer (scattered to the parameter layout, zero where a parameter is held fixed by
// the camera block width or by mcba_lm_auto_config's / mcba_lm_run's flags), the frame gradients.  The reduced system must be that of
// the current point (after a terminated loop it is; solver.LevenbergMarquardt.finalize rebuilds it otherwise).
int mcba_lm_result(mcba_handle* h, int slot, double* x_out, double* grad_out, mcba_buffer** grad_dev) {
  if (!slot_ok(h, slot) || !x_out || (grad_out && grad_dev)) return fail(MCBA_ERR_ARG, "mcba_lm_result: bad argument");
  if (!h->have_red) return fail(MCBA_ERR_ARG, "mcba_lm_result: no reduced system");
  HIPCHK(hipSetDevice(h->device));
  const size_t nx = (size_t)12 * h->C + (size_t)6 * h->F;
  if (grad_out && grad_out != x_out + nx) return fail(MCBA_ERR_ARG, "mcba_lm_result: grad_out must directly follow x_out (x_out + 12C + 6F): both arrive in one copy");
  int rc;
  if (!h->outbuf && (rc = dalloc(h, &h->outbuf, 2 * nx, false))) return rc;
  mcba::launch_pack_result(h->stream, h->x[slot], h->red + (size_t)h->n * h->n + 2 * (size_t)h->n, h->fbuf, h->have_fixed ? h->fixed : nullptr, h->outbuf, h->C, h->F, h->cw);
  if ((rc = check_launch())) return rc;
  HIPCHK(hipMemcpyAsync(x_out, h->outbuf, (grad_out ? 2 : 1) * nx * sizeof(double), hipMemcpyDeviceToHost, h->stream));  // (grad_out, if given, must directly follow x_out: one copy)
  HIPCHK(hipStreamSynchronize(h->stream));
  if (grad_dev) {   // the gradient stays on the device as an object of its own (OptimizeResult.grad is rarely read: 0.48 MB of D2H at 6 x 10 000 x 54)
    *grad_dev = new mcba_buffer{h->outbuf + nx, nx, h->device, h->stream, h->outbuf, 2 * nx};
    for (size_t i = 0; i < h->bufs.size(); ++i)
      if (h->bufs[i].slot == reinterpret_cast<void**>(&h->outbuf)) { h->bufs.erase(h->bufs.begin() + i); break; }
    h->outbuf = nullptr;
  }
  return MCBA_OK;
}

// ---------------------------------------------------------------------------------------------------------
// bundle_adjust()'s frame pre-filter on the GPU (reference bundle_adjustment.py:265-285) and frame subsets without a second upload
static int ensure_diag(mcba_handle* h) {
  int rc;
  // (none of them is filled: k_frame_err / k_reproj_diag write every error and statistic, launch_select clears its states, the mask is
  //  written whole by whoever uses it)
  if (!h->err && (rc = dalloc(h, &h->err, (size_t)h->C * h->N * h->Fpad, false))) return rc;
  if (!h->dmean && (rc = dalloc(h, &h->dmean, std::max<size_t>((size_t)h->C * h->F, 8), false))) return rc;
  if (!h->dfull && (rc = dalloc(h, &h->dfull, (size_t)h->C * h->F, false))) return rc;
  if (!h->sel && (rc = dalloc(h, &h->sel, mcba::select_state_bytes(2 * h->C), false))) return rc;
  if (!h->fmask && (rc = dalloc(h, &h->fmask, (size_t)h->Fpad, false))) return rc;
  return MCBA_OK;
}

// ---------------------------------------------------------------------------------------------------------
// bundle_adjust()'s pre-filter in ONE call and ONE host synchronisation (round 5): upload (observations, board, parameters of every
// frame), re-layout, k_frame_err, and the whole selection on the device (mcba_diag.hip: frames complete in two cameras, worst camera's
// mean error, 5 x nanmedian by a three-pass radix select, the comparison) -- the host reads 80 + F bytes.  The transfer is NOT cut
// into chunks with kernels in between: measured on the MI355X box (scripts/micro/h2d_pipeline.hip, profiles/round5/h2d_pipeline.txt)
// one hipMemcpyAsync of the 51.8 MB takes 0.92 ms (56 GB/s, the call blocks: pageable source), six chunks 1.10 ms, twelve 1.23 ms
// (~30 us per extra call), a pinned staging ring 1.93 ms -- while everything the GPU does behind the copy is ~60 us.
static int median_of_err(mcba_handle* h, size_t per_group, int groups, bool use_mask, double* median, double* count);
static int ensure_prefilter(mcba_handle* h) {
  int rc = ensure_diag(h);
  if (rc) return rc;
  if (!h->pf_state && (rc = dalloc(h, &h->pf_state, mcba::prefilter_state_bytes(), false))) return rc;
  if (!h->pf_status && (rc = dalloc(h, &h->pf_status, (size_t)h->Fpad, false))) return rc;
  if (!h->pf_worst && (rc = dalloc(h, &h->pf_worst, (size_t)h->Fpad, false))) return rc;
  if (!h->pf_packed && (rc = dalloc(h, &h->pf_packed, (size_t)h->Fpad + 128, false))) return rc;
  if (!h->pf_host) {
    h->pf_host_bytes = (size_t)h->Fpad + 128;
    HIPCHK(pool_host_malloc(reinterpret_cast<void**>(&h->pf_host), h->pf_host_bytes, hipHostMallocDefault));
  }
  return MCBA_OK;
}

int mcba_prefilter(mcba_handle* h, const double* uvs, const double* objpoints, const double* x, double outlier_threshold, unsigned char* status, double* info8) {
  if (!h || !x || !status || !info8 || (uvs == nullptr) != (objpoints == nullptr)) return fail(MCBA_ERR_ARG, "mcba_prefilter: bad argument");
  if (!uvs && !h->have_obs) return fail(MCBA_ERR_ARG, "mcba_prefilter: no observations (pass them, or upload them first)");
  HIPCHK(hipSetDevice(h->device));
  int rc = ensure_prefilter(h);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(h->x[0], x, ((size_t)12 * h->C + (size_t)6 * h->F) * sizeof(double), hipMemcpyHostToDevice, h->stream));
  if (uvs && (rc = upload_impl(h, uvs, objpoints, false))) return rc;
  mcba::launch_frame_err(h->stream, h->obs_t, h->obj, h->x[0], h->err, h->dmean, h->dfull, h->C, h->F, h->N, h->Fpad, h->pf_state);
  mcba::launch_prefilter_select(h->stream, h->err, h->dmean, h->dfull, h->fmask, h->pf_status, h->pf_worst, h->pf_state, h->pf_packed, h->C, h->F, h->N, h->Fpad, outlier_threshold, true);
  if ((rc = check_launch())) return rc;
  const size_t nb = 64 + (size_t)h->F;
  auto fetch = [&]() -> int {
    HIPCHK(hipMemcpyAsync(h->pf_host, h->pf_packed, nb, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return MCBA_OK;
  };
  if ((rc = fetch())) return rc;
  double info[8];
  memcpy(info, h->pf_host, sizeof(info));
  const char* force = getenv("MCBA_PREFILTER_FALLBACK");   // test knob: take the eight-pass select whatever the candidate count
  if (outlier_threshold != outlier_threshold && (info[3] != 0.0 || (force && atoi(force) != 0))) {
    // more values share the median's 24 leading bits than the candidate list holds: the eight-pass radix select on the same mask
    double med = 0.0, cnt = 0.0;
    if ((rc = median_of_err(h, (size_t)h->C * h->N * h->Fpad, 1, true, &med, &cnt))) return rc;
    mcba::launch_prefilter_status(h->stream, h->pf_status, h->pf_worst, h->pf_state, h->pf_packed, h->F, 5.0 * med, 0);
    if ((rc = check_launch())) return rc;
    if ((rc = fetch())) return rc;
    memcpy(info, h->pf_host, sizeof(info));
    info[1] = med; info[2] = cnt; info[3] = 1.0;
  }
  memcpy(status, h->pf_host + 64, (size_t)h->F);
  {  // frames used / excluded / kept but incomplete in some camera (the counts of the printed line)
    double used = 0, excl = 0, inc = 0;
    for (int f = 0; f < h->F; ++f) { const unsigned char sf = status[f]; used += sf & 1; excl += (sf >> 1) & 1; inc += ((sf & 7) == 1) ? 1 : 0; }
    info[4] = used; info[5] = excl; info[6] = inc;
  }
  memcpy(info8, info, sizeof(info));
  return MCBA_OK;
}

// mcba_prefilter + what bundle_adjust does with its answer when no random draw stands in between (bundle_adjustment.py:292-296: the
// subsample is drawn from the caller's global numpy RNG only if n_frames <= the number of frames kept): the kept frames are gathered into a
// new handle right here, with the status bytes still warm -- the host round trip between "the selection is known" and "its gather is enqueued"
// was a Python function and a second crossing.  info8[7]: 0 nothing kept, 1 the caller must draw (no handle made), 2 every frame kept in
// order (solve on h itself), 3 *sub holds the kept frames (mcba_create_subset of them, in order).  n_frames < 0: no cap (None).
int mcba_prefilter_subset(mcba_handle* h, const double* uvs, const double* objpoints, const double* x, double outlier_threshold, int n_frames, unsigned char* status, double* info8,
                          mcba_handle** sub) {
  if (!sub) return fail(MCBA_ERR_ARG, "mcba_prefilter_subset: bad argument");
  *sub = nullptr;
  int rc = mcba_prefilter(h, uvs, objpoints, x, outlier_threshold, status, info8);
  if (rc) return rc;
  const int kept = (int)(info8[4] - info8[5]);
  if (kept == 0) { info8[7] = 0.0; return MCBA_OK; }
  if (n_frames >= 0 && n_frames <= kept) { info8[7] = 1.0; return MCBA_OK; }
  if (kept == h->F) { info8[7] = 2.0; return MCBA_OK; }
  std::vector<int> frames;
  frames.reserve((size_t)kept);
  for (int f = 0; f < h->F; ++f)
    if ((status[f] & 3) == 1) frames.push_back(f);
  if ((rc = mcba_create_subset(sub, h, frames.data(), (int)frames.size()))) return rc;
  info8[7] = 3.0;
  return MCBA_OK;
}

int mcba_frame_errors(mcba_handle* h, int slot, double* mean_cf, double* full_cf) {
  if (!slot_ok(h, slot) || !mean_cf || !full_cf) return fail(MCBA_ERR_ARG, "mcba_frame_errors: bad argument");
  if (!h->have_obs) return fail(MCBA_ERR_ARG, "mcba_frame_errors: upload observations first");
  HIPCHK(hipSetDevice(h->device));
  int rc = ensure_diag(h);
  if (rc) return rc;
  mcba::launch_frame_err(h->stream, h->obs_t, h->obj, h->x[slot], h->err, h->dmean, h->dfull, h->C, h->F, h->N, h->Fpad);
  if ((rc = check_launch())) return rc;
  const size_t cnt = (size_t)h->C * h->F * sizeof(double);
  HIPCHK(hipMemcpyAsync(mean_cf, h->dmean, cnt, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(full_cf, h->dfull, cnt, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return MCBA_OK;
}

// nan-median (exact order statistics) of `groups` equal slices of h->err restricted to the frames of h->fmask
static int median_of_err(mcba_handle* h, size_t per_group, int groups, bool use_mask, double* median, double* count) {
  struct Sel { unsigned long long prefix, rank, count, value; unsigned int hist[256]; };
  std::vector<Sel> both(2 * (size_t)groups);  // state 2 g: rank (n - 1) / 2, state 2 g + 1: rank n / 2 -- found in the same eight passes
  mcba::launch_select(h->stream, h->err, use_mask ? h->fmask : nullptr, per_group, groups, h->Fpad, h->sel, 2);
  int rc = check_launch();
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(both.data(), h->sel, both.size() * sizeof(Sel), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  for (int g = 0; g < groups; ++g) {
    double a, b;
    memcpy(&a, &both[2 * g].value, 8);
    memcpy(&b, &both[2 * g + 1].value, 8);
    median[g] = both[2 * g].count ? 0.5 * (a + b) : NAN;  // np.median / np.nanmedian: mean of the two middle values
    if (count) count[g] = (double)both[2 * g].count;
  }
  return MCBA_OK;
}

int mcba_error_median(mcba_handle* h, const unsigned char* frame_mask, double* median, double* count) {
  if (!h || !median) return fail(MCBA_ERR_ARG, "mcba_error_median: bad argument");
  if (!h->err) return fail(MCBA_ERR_ARG, "mcba_error_median: call mcba_frame_errors first");
  HIPCHK(hipSetDevice(h->device));
  if (frame_mask) {
    HIPCHK(hipMemsetAsync(h->fmask, 0, (size_t)h->Fpad, h->stream));
    HIPCHK(hipMemcpyAsync(h->fmask, frame_mask, (size_t)h->F, hipMemcpyHostToDevice, h->stream));  // (pageable source: staged before the call returns)
  }
  return median_of_err(h, (size_t)h->C * h->N * h->Fpad, 1, frame_mask != nullptr, median, count);
}

// only_cam != nullptr (mcba_create_views): destination frame j keeps the detection of camera only_cam[j] alone
static int create_subset_impl(mcba_handle** out, mcba_handle* src, const int* frames, const int* only_cam, int n_frames) {
  if (!out || !src || !frames || n_frames < 1) return fail(MCBA_ERR_ARG, "mcba_create_subset: bad argument");
  if (!src->have_obs) return fail(MCBA_ERR_ARG, "mcba_create_subset: the source handle has no observations");
  for (int i = 0; i < n_frames; ++i)
    if (frames[i] < 0 || frames[i] >= src->F || (only_cam && (only_cam[i] < 0 || only_cam[i] >= src->C))) return fail(MCBA_ERR_ARG, "mcba_create_subset: frame / camera index out of range");
  int rc = mcba_create(out, src->C, n_frames, src->N, src->device);
  if (rc) return rc;
  mcba_handle* h = *out;
  if (src->stream != h->stream) HIPCHK(hipStreamSynchronize(h->stream));  // mcba_create's zero fills ran on the creation stream
  h->stream = src->stream;
  h->loss = src->loss == mcba::LOSS_TABLE ? MCBA_LOSS_SOFT_L1 : src->loss;   // (a table belongs to its frames: the subset starts from the default)
  h->f_scale = src->f_scale;
  h->strict_sync = src->strict_sync;
  // (the index list lives and dies with the new handle: nothing to free here, so nothing to wait for)
  if ((rc = dalloc(h, &h->sub_frames, (size_t)n_frames * (only_cam ? 2 : 1), false)) != MCBA_OK) { mcba_destroy(h); *out = nullptr; return rc; }
  int* d_frames = h->sub_frames;
  hipError_t e = hipMemcpyAsync(d_frames, frames, (size_t)n_frames * sizeof(int), hipMemcpyHostToDevice, h->stream);  // (pageable source: staged before the call returns)
  if (e == hipSuccess && only_cam) e = hipMemcpyAsync(d_frames + n_frames, only_cam, (size_t)n_frames * sizeof(int), hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess) {
    mcba::launch_gather_frames(h->stream, src->obs_raw, d_frames, h->obs_raw, h->C, src->F, h->F, h->N, only_cam ? d_frames + n_frames : nullptr);
    // ... and the parameters of the source's slot 0: the camera blocks + the poses of the chosen frames (what bundle_adjust starts from)
    mcba::launch_gather_params(h->stream, src->x[0], d_frames, h->x[0], h->C, h->F);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(h->obj, src->obj, (size_t)3 * h->N * sizeof(double), hipMemcpyDeviceToDevice, h->stream);
  if (e == hipSuccess) {
    Scope sc(h, K_TRANSPOSE);
    mcba::launch_transpose_obs(h->stream, h->obs_raw, h->obs_t, h->C, h->F, h->N, h->Fpad);
    e = hipGetLastError();
  }
  if (e != hipSuccess) {
    g_err = std::string("mcba_create_subset: ") + hipGetErrorString(e);
    mcba_destroy(h);
    *out = nullptr;
    return MCBA_ERR_HIP;
  }
  if (src->obj_host) {
    h->obj_host = static_cast<double*>(malloc((size_t)3 * h->N * sizeof(double)));
    if (h->obj_host) memcpy(h->obj_host, src->obj_host, (size_t)3 * h->N * sizeof(double));
    h->planar = src->planar;
  }
  h->have_obs = true;
  return MCBA_OK;
}

int mcba_create_subset(mcba_handle** out, mcba_handle* src, const int* frames, int n_frames) { return create_subset_impl(out, src, frames, nullptr, n_frames); }

// A handle of C cameras x n_views frames whose frame j holds the detection of view j = (camera, frame) of `src` in ITS camera alone (NaN in
// the others): the <= 100 sampled views of every camera side by side, so that ONE device-resident LM run refines every camera's intrinsics
// with its own views' poses (get_intrinsics, reference calibration.py:11-71) -- the normal equations are block-diagonal over the cameras.
int mcba_create_views(mcba_handle** out, mcba_handle* src, const int* views, int n_views) {
  if (!out || !src || !views || n_views < 1) return fail(MCBA_ERR_ARG, "mcba_create_views: bad argument");
  std::vector<int> frames((size_t)n_views), cams((size_t)n_views);
  for (int i = 0; i < n_views; ++i) { cams[i] = views[2 * i]; frames[i] = views[2 * i + 1]; }
  return create_subset_impl(out, src, frames.data(), cams.data(), n_views);
}

// ---------------------------------------------------------------------------------------------------------
// calibrate() on the device (reference calibration.py:11-113 -- the two OpenCV calls per view -- and :116-277 -- the pose graph).
// The handle holds every detection (mcba_upload_observations); a call of calibrate() is: mcba_calib_complete -> [host: the reference's
// RNG draw] -> mcba_calib_homographies (sampled views) -> [host: Zhang's closed form, a 6-vector] -> mcba_calib_view_poses ->
// mcba_create_views + mcba_lm_run (all cameras' intrinsics in one run) -> mcba_calib_poses (every view, ONE launch; the poses stay on the
// device) -> [host: spanning tree] -> mcba_calib_pairwise -> [host: chain C - 1 transforms] -> mcba_calib_consensus.
}  // extern "C"
namespace {
// a scratch buffer of the handle that grows with the call
template <class T>
int dgrow(mcba_handle* h, T** p, size_t* cap, size_t count) {
  if (*p && *cap >= count) return MCBA_OK;
  if (*p) {
    for (size_t i = 0; i < h->bufs.size(); ++i)
      if (h->bufs[i].slot == reinterpret_cast<void**>(p)) { pool_free(*p, h->bufs[i].bytes, h->device, h->stream, true); h->bufs.erase(h->bufs.begin() + i); break; }
    *p = nullptr;
  }
  *cap = 0;
  int rc = dalloc(h, p, count, false);
  if (rc == MCBA_OK) *cap = count;
  return rc;
}
// Hartley normalisation of the board's XY as calibration.py's closed-form start uses it: centroid, sqrt(2) / rms distance
void board_normalisation(const double* obj, int N, double* bn) {
  bn[0] = bn[1] = 0.0; bn[2] = 1.0;
  for (int p = 0; p < N; ++p) { bn[0] += obj[3 * p]; bn[1] += obj[3 * p + 1]; }
  bn[0] /= N; bn[1] /= N;
  double ms = 0.0;
  for (int p = 0; p < N; ++p) ms += (obj[3 * p] - bn[0]) * (obj[3 * p] - bn[0]) + (obj[3 * p + 1] - bn[1]) * (obj[3 * p + 1] - bn[1]);
  if (ms > 0.0) bn[2] = sqrt(2.0) / sqrt(ms / N);
}
int calib_ready(mcba_handle* h, const char* who) {
  if (!h) return fail(MCBA_ERR_ARG, "NULL handle");
  if (!h->have_obs || !h->obj_host) { g_err = std::string(who) + ": upload observations first"; return MCBA_ERR_ARG; }
  for (int p = 0; p < h->N; ++p)
    if (h->obj_host[3 * p + 2] != 0.0) { g_err = std::string(who) + ": the closed-form start needs a planar calibration board (z = 0)"; return MCBA_ERR_ARG; }
  HIPCHK(hipSetDevice(h->device));
  return MCBA_OK;
}
int upload_views(mcba_handle* h, const int* views, int n_views, const char* who) {
  for (int i = 0; i < n_views; ++i)
    if (views[2 * i] < 0 || views[2 * i] >= h->C || views[2 * i + 1] < 0 || views[2 * i + 1] >= h->F) { g_err = std::string(who) + ": view (camera, frame) out of range"; return MCBA_ERR_ARG; }
  int rc = dgrow(h, &h->cal_views, &h->cal_views_cap, (size_t)2 * n_views);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(h->cal_views, views, (size_t)2 * n_views * sizeof(int), hipMemcpyHostToDevice, h->stream));
  return MCBA_OK;
}
int upload_intr(mcba_handle* h, const double* intr9) {
  int rc;
  if (!h->cal_intr && (rc = dalloc(h, &h->cal_intr, (size_t)9 * h->C, false))) return rc;
  HIPCHK(hipMemcpyAsync(h->cal_intr, intr9, (size_t)9 * h->C * sizeof(double), hipMemcpyHostToDevice, h->stream));
  return MCBA_OK;
}
struct SelRecord { unsigned long long prefix, rank, count, value; unsigned int hist[256]; };   // = SelState of mcba_diag.hip

// medians of the pairwise transforms (calibration.py:143): rel [E][6][Fpad] -> out (E, 6), counts (E) = frames the pair shares
int pairwise_medians(hipStream_t st, const double* rel, int n_edges, int Fpad, unsigned char* sel_dev, double* out, double* counts) {
  const int groups = 6 * n_edges;
  mcba::launch_select(st, rel, nullptr, (size_t)Fpad, groups, Fpad, sel_dev, 2, 1);
  int rc = check_launch();
  if (rc) return rc;
  std::vector<unsigned long long> head((size_t)4 * 2 * groups);   // prefix rank count value of every state
  HIPCHK(hipMemcpy2DAsync(head.data(), 32, sel_dev, sizeof(SelRecord), 32, (size_t)2 * groups, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  for (int g = 0; g < groups; ++g) {
    double a, b;
    memcpy(&a, &head[4 * (2 * g) + 3], 8);
    memcpy(&b, &head[4 * (2 * g + 1) + 3], 8);
    out[g] = head[4 * (2 * g) + 2] ? 0.5 * (a + b) : NAN;   // np.median: the mean of the two middle values; no common frame: NaN
    if (counts && g % 6 == 0) counts[g / 6] = (double)head[4 * (2 * g) + 2];
  }
  return MCBA_OK;
}
struct TempDevice {   // one device allocation for a stateless call, freed on every way out
  unsigned char* p = nullptr;
  ~TempDevice() { if (p) (void)hipFree(p); }
};
int stateless_device(int device, const char* who) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(MCBA_ERR_NODEVICE, "no HIP device visible");
  if (device < 0 || device >= ndev) { g_err = std::string(who) + ": device ordinal out of range"; return MCBA_ERR_ARG; }
  HIPCHK(hipSetDevice(device));
  return MCBA_OK;
}
size_t up256(size_t b) { return (b + 255) / 256 * 256; }
}  // namespace
extern "C" {

// complete_cf (C, F) bytes: 1 = every scalar of the detection is present (what get_intrinsics samples from and estimate_pose solves: :55, :107)
int mcba_calib_complete(mcba_handle* h, unsigned char* complete_cf) {
  int rc = calib_ready(h, "mcba_calib_complete");
  if (rc) return rc;
  if (!complete_cf) return fail(MCBA_ERR_ARG, "mcba_calib_complete: NULL output");
  if (!h->cal_valid && (rc = dalloc(h, &h->cal_valid, (size_t)h->C * h->F, false))) return rc;
  mcba::launch_view_complete(h->stream, h->obs_t, h->cal_valid, h->C, h->F, h->N, h->Fpad);
  if ((rc = check_launch())) return rc;
  HIPCHK(hipMemcpyAsync(complete_cf, h->cal_valid, (size_t)h->C * h->F, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return MCBA_OK;
}

// Board-plane -> pixel homographies (H[2][2] = 1) of the listed (camera, frame) views: normalised DLT.  H_out n_views x 9; ok_out n_views bytes or NULL
int mcba_calib_homographies(mcba_handle* h, const int* views, int n_views, double* H_out, unsigned char* ok_out) {
  int rc = calib_ready(h, "mcba_calib_homographies");
  if (rc) return rc;
  if (!views || n_views < 1 || !H_out) return fail(MCBA_ERR_ARG, "mcba_calib_homographies: bad argument");
  if ((rc = upload_views(h, views, n_views, "mcba_calib_homographies"))) return rc;
  if ((rc = dgrow(h, &h->cal_out, &h->cal_out_cap, (size_t)10 * n_views + 8))) return rc;
  unsigned char* okd = reinterpret_cast<unsigned char*>(h->cal_out + (size_t)9 * n_views);
  double bn[3];
  board_normalisation(h->obj_host, h->N, bn);
  mcba::launch_pnp(h->stream, 0, h->obs_t, h->obj, nullptr, h->cal_views, n_views, bn, h->C, h->F, h->N, h->Fpad, 0, 0, h->cal_out, nullptr, okd, nullptr);
  if ((rc = check_launch())) return rc;
  HIPCHK(hipMemcpyAsync(H_out, h->cal_out, (size_t)9 * n_views * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (ok_out) HIPCHK(hipMemcpyAsync(ok_out, okd, (size_t)n_views, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return MCBA_OK;
}

// cv2.solvePnP's job (calibration.py:108) for the listed views: intr9 = C x (fx fy cx cy k1 k2 p1 p2 k3); poses_out n_views x 6 (NaN = none)
int mcba_calib_view_poses(mcba_handle* h, const int* views, int n_views, const double* intr9, int undistort_iterations, int max_evaluations, double* poses_out, unsigned char* ok_out) {
  int rc = calib_ready(h, "mcba_calib_view_poses");
  if (rc) return rc;
  if (!views || n_views < 1 || !intr9 || !poses_out || undistort_iterations < 0 || max_evaluations < 1) return fail(MCBA_ERR_ARG, "mcba_calib_view_poses: bad argument");
  if ((rc = upload_views(h, views, n_views, "mcba_calib_view_poses"))) return rc;
  if ((rc = upload_intr(h, intr9))) return rc;
  if ((rc = dgrow(h, &h->cal_out, &h->cal_out_cap, (size_t)10 * n_views + 8))) return rc;
  unsigned char* okd = reinterpret_cast<unsigned char*>(h->cal_out + (size_t)9 * n_views);
  double bn[3];
  board_normalisation(h->obj_host, h->N, bn);
  mcba::launch_pnp(h->stream, 1, h->obs_t, h->obj, h->cal_intr, h->cal_views, n_views, bn, h->C, h->F, h->N, h->Fpad, undistort_iterations, max_evaluations, h->cal_out, nullptr, okd, nullptr);
  if ((rc = check_launch())) return rc;
  HIPCHK(hipMemcpyAsync(poses_out, h->cal_out, (size_t)6 * n_views * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (ok_out) HIPCHK(hipMemcpyAsync(ok_out, okd, (size_t)n_views, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return MCBA_OK;
}

// The closed-form start of get_intrinsics for every camera at once, ONE crossing (what cv2.calibrateCamera does before it refines, calibration.py:68):
// homographies of the listed views -> Zhang's K per camera from its views (k_zhang; image_sizes = C x (width, height)) -> cv2.solvePnP's job
// for the same views with that K and no distortion.  k4_out C x (fx fy cx cy); closed_out (C bytes, optional) = 1 where the closed form was used
// (0: the fallback f = max(w, h), c = the image centre); poses_out n_views x 6 (NaN = none); ok_out (n_views bytes, optional).
int mcba_calib_start(mcba_handle* h, const int* views, int n_views, const double* image_sizes, int undistort_iterations, int max_evaluations, double* k4_out, unsigned char* closed_out,
                     double* poses_out, unsigned char* ok_out) {
  int rc = calib_ready(h, "mcba_calib_start");
  if (rc) return rc;
  if (!views || n_views < 1 || !image_sizes || !k4_out || !poses_out || undistort_iterations < 0 || max_evaluations < 1) return fail(MCBA_ERR_ARG, "mcba_calib_start: bad argument");
  for (int c = 0; c < h->C; ++c)
    if (!(image_sizes[2 * c] >= 1.0 && image_sizes[2 * c + 1] >= 1.0 && image_sizes[2 * c] < 1e9 && image_sizes[2 * c + 1] < 1e9)) return fail(MCBA_ERR_ARG, "mcba_calib_start: image sizes must be finite and >= 1");
  if ((rc = upload_views(h, views, n_views, "mcba_calib_start"))) return rc;
  if (!h->cal_intr && (rc = dalloc(h, &h->cal_intr, (size_t)9 * h->C, false))) return rc;
  const size_t tail = (size_t)10 * n_views + 8;   // [0, 9 n) homographies, then poses; [9 n, 10 n) the views' valid bytes; then sizes (2 C) and closed bytes (C)
  if ((rc = dgrow(h, &h->cal_out, &h->cal_out_cap, tail + (size_t)3 * h->C + 8))) return rc;
  unsigned char* okd = reinterpret_cast<unsigned char*>(h->cal_out + (size_t)9 * n_views);
  double* sizes_d = h->cal_out + tail;
  unsigned char* closed_d = reinterpret_cast<unsigned char*>(sizes_d + (size_t)2 * h->C);
  HIPCHK(hipMemcpyAsync(sizes_d, image_sizes, (size_t)2 * h->C * sizeof(double), hipMemcpyHostToDevice, h->stream));
  double bn[3];
  board_normalisation(h->obj_host, h->N, bn);
  mcba::launch_pnp(h->stream, 0, h->obs_t, h->obj, nullptr, h->cal_views, n_views, bn, h->C, h->F, h->N, h->Fpad, 0, 0, h->cal_out, nullptr, okd, nullptr);
  mcba::launch_zhang(h->stream, h->cal_out, okd, h->cal_views, n_views, sizes_d, h->C, h->cal_intr, closed_d);
  mcba::launch_pnp(h->stream, 1, h->obs_t, h->obj, h->cal_intr, h->cal_views, n_views, bn, h->C, h->F, h->N, h->Fpad, undistort_iterations, max_evaluations, h->cal_out, nullptr, okd, nullptr);
  if ((rc = check_launch())) return rc;
  std::vector<double> intr((size_t)9 * h->C);
  HIPCHK(hipMemcpyAsync(intr.data(), h->cal_intr, intr.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(poses_out, h->cal_out, (size_t)6 * n_views * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (ok_out) HIPCHK(hipMemcpyAsync(ok_out, okd, (size_t)n_views, hipMemcpyDeviceToHost, h->stream));
  if (closed_out) HIPCHK(hipMemcpyAsync(closed_out, closed_d, (size_t)h->C, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  for (int c = 0; c < h->C; ++c)
    for (int k = 0; k < 4; ++k) k4_out[4 * c + k] = intr[(size_t)9 * c + k];
  return MCBA_OK;
}

// estimate_pose (calibration.py:74-113) of EVERY camera in one launch: the board pose of every (camera, frame) with a complete detection.
// The poses stay on the device for mcba_calib_pairwise / mcba_calib_consensus; poses_out (C, F, 6) (NaN rows = no pose), ok_out (C, F) bytes
// and evals_out (C, F) bytes (LM evaluations a view took) are optional.
int mcba_calib_poses(mcba_handle* h, const double* intr9, int undistort_iterations, int max_evaluations, double* poses_out, unsigned char* ok_out, unsigned char* evals_out) {
  int rc = calib_ready(h, "mcba_calib_poses");
  if (rc) return rc;
  if (!intr9 || undistort_iterations < 0 || max_evaluations < 1) return fail(MCBA_ERR_ARG, "mcba_calib_poses: bad argument");
  if ((rc = upload_intr(h, intr9))) return rc;
  const size_t CF = (size_t)h->C * h->F;
  if (!h->cal_poses_t && (rc = dalloc(h, &h->cal_poses_t, (size_t)6 * h->C * h->Fpad, false))) return rc;
  if (!h->cal_valid && (rc = dalloc(h, &h->cal_valid, CF, false))) return rc;
  if (!h->cal_nit && (rc = dalloc(h, &h->cal_nit, CF, false))) return rc;
  if (poses_out && (rc = dgrow(h, &h->cal_out, &h->cal_out_cap, 6 * CF))) return rc;
  double bn[3];
  board_normalisation(h->obj_host, h->N, bn);
  mcba::launch_pnp(h->stream, 1, h->obs_t, h->obj, h->cal_intr, nullptr, 0, bn, h->C, h->F, h->N, h->Fpad, undistort_iterations, max_evaluations, poses_out ? h->cal_out : nullptr, h->cal_poses_t, h->cal_valid,
                   h->cal_nit);
  if ((rc = check_launch())) return rc;
  h->have_cal_poses = true;
  if (poses_out) HIPCHK(hipMemcpyAsync(poses_out, h->cal_out, 6 * CF * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (ok_out) HIPCHK(hipMemcpyAsync(ok_out, h->cal_valid, CF, hipMemcpyDeviceToHost, h->stream));
  if (evals_out) HIPCHK(hipMemcpyAsync(evals_out, h->cal_nit, CF, hipMemcpyDeviceToHost, h->stream));
  if (poses_out || ok_out || evals_out) HIPCHK(hipStreamSynchronize(h->stream));
  return MCBA_OK;
}

// estimate_pairwise_camera_transform (calibration.py:116-143) for a list of camera pairs (c1, c2) over the poses mcba_calib_poses left on
// the device: transforms_out (n_edges, 6) = component-wise median over the common frames of T2 T1^-1; counts_out (n_edges) or NULL
int mcba_calib_pairwise(mcba_handle* h, const int* edges, int n_edges, double* transforms_out, double* counts_out) {
  int rc = calib_ready(h, "mcba_calib_pairwise");
  if (rc) return rc;
  if (!edges || n_edges < 1 || !transforms_out) return fail(MCBA_ERR_ARG, "mcba_calib_pairwise: bad argument");
  if (!h->have_cal_poses) return fail(MCBA_ERR_ARG, "mcba_calib_pairwise: call mcba_calib_poses first");
  for (int i = 0; i < 2 * n_edges; ++i)
    if (edges[i] < 0 || edges[i] >= h->C) return fail(MCBA_ERR_ARG, "mcba_calib_pairwise: camera index out of range");
  if ((rc = dgrow(h, &h->cal_views, &h->cal_views_cap, (size_t)2 * n_edges))) return rc;
  if ((rc = dgrow(h, &h->cal_rel, &h->cal_rel_cap, (size_t)6 * n_edges * h->Fpad))) return rc;
  if ((rc = dgrow(h, &h->cal_sel, &h->cal_sel_cap, mcba::select_state_bytes(12 * n_edges)))) return rc;
  HIPCHK(hipMemcpyAsync(h->cal_views, edges, (size_t)2 * n_edges * sizeof(int), hipMemcpyHostToDevice, h->stream));
  mcba::launch_pose_pairs(h->stream, h->cal_poses_t, (size_t)6 * h->Fpad, 1, (size_t)h->Fpad, h->cal_views, n_edges, h->F, h->Fpad, h->cal_rel);
  if ((rc = check_launch())) return rc;
  return pairwise_medians(h->stream, h->cal_rel, n_edges, h->Fpad, h->cal_sel, transforms_out, counts_out);
}

// consensus_calib_poses (calibration.py:239-277): extrinsics (C, 6) world -> camera; poses_out (F, 6) = nan-median over the cameras of
// T_ext^-1 T_pose, NaN rows for frames no camera has a pose for
int mcba_calib_consensus(mcba_handle* h, const double* extrinsics, double* poses_out) {
  int rc = calib_ready(h, "mcba_calib_consensus");
  if (rc) return rc;
  if (!extrinsics || !poses_out) return fail(MCBA_ERR_ARG, "mcba_calib_consensus: bad argument");
  if (!h->have_cal_poses) return fail(MCBA_ERR_ARG, "mcba_calib_consensus: call mcba_calib_poses first");
  if (!h->cal_world && (rc = dalloc(h, &h->cal_world, (size_t)6 * h->C * h->Fpad, false))) return rc;
  if ((rc = dgrow(h, &h->cal_out, &h->cal_out_cap, (size_t)6 * h->F + (size_t)6 * h->C))) return rc;
  double* d_ext = h->cal_out + (size_t)6 * h->F;
  HIPCHK(hipMemcpyAsync(d_ext, extrinsics, (size_t)6 * h->C * sizeof(double), hipMemcpyHostToDevice, h->stream));
  mcba::launch_pose_consensus(h->stream, h->cal_poses_t, (size_t)6 * h->Fpad, 1, (size_t)h->Fpad, d_ext, h->C, h->F, h->Fpad, h->cal_world, h->cal_out);
  if ((rc = check_launch())) return rc;
  HIPCHK(hipMemcpyAsync(poses_out, h->cal_out, (size_t)6 * h->F * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return MCBA_OK;
}

// calibration.py:200-277 in ONE crossing on the poses mcba_calib_poses left on the device: the medians of the pairwise transforms of the spanning
// tree's edges (mcba_calib_pairwise's work), chained from `root` into the world -> camera extrinsics on the device (k_pose_chain; :226-235), and the
// consensus board poses with them (mcba_calib_consensus' work).  edges = n_edges x (c1, c2), ordered so that c1 is `root` or the c2 of an earlier
// edge, every camera reached exactly once (n_edges = C - 1; C = 1: none): the reference's tree, sorted by distance from the root.
// extrinsics_out (C, 6), the root's row exactly 0; poses_out (F, 6); transforms_out (n_edges, 6) and counts_out (n_edges) optional.
int mcba_calib_graph(mcba_handle* h, const int* edges, int n_edges, int root, double* extrinsics_out, double* poses_out, double* transforms_out, double* counts_out) {
  int rc = calib_ready(h, "mcba_calib_graph");
  if (rc) return rc;
  if (!extrinsics_out || !poses_out || n_edges < 0 || (n_edges > 0 && !edges) || root < 0 || root >= h->C) return fail(MCBA_ERR_ARG, "mcba_calib_graph: bad argument");
  if (!h->have_cal_poses) return fail(MCBA_ERR_ARG, "mcba_calib_graph: call mcba_calib_poses first");
  if (n_edges != h->C - 1) return fail(MCBA_ERR_ARG, "mcba_calib_graph: a spanning tree of C cameras has C - 1 edges");
  {
    std::vector<char> placed((size_t)h->C, 0);
    placed[(size_t)root] = 1;
    for (int e = 0; e < n_edges; ++e) {
      const int c1 = edges[2 * e], c2 = edges[2 * e + 1];
      if (c1 < 0 || c1 >= h->C || c2 < 0 || c2 >= h->C) return fail(MCBA_ERR_ARG, "mcba_calib_graph: camera index out of range");
      if (!placed[(size_t)c1] || placed[(size_t)c2]) return fail(MCBA_ERR_ARG, "mcba_calib_graph: edges must lead away from the root, every camera reached once");
      placed[(size_t)c2] = 1;
    }
  }
  const int E = n_edges;
  if (!h->cal_world && (rc = dalloc(h, &h->cal_world, (size_t)6 * h->C * h->Fpad, false))) return rc;
  if ((rc = dgrow(h, &h->cal_out, &h->cal_out_cap, (size_t)6 * h->F + (size_t)6 * h->C + (size_t)7 * E + 8))) return rc;
  double* d_ext = h->cal_out + (size_t)6 * h->F;
  double* d_tr = d_ext + (size_t)6 * h->C;
  double* d_cnt = d_tr + (size_t)6 * E;
  if (E > 0) {
    if ((rc = dgrow(h, &h->cal_views, &h->cal_views_cap, (size_t)2 * E))) return rc;
    if ((rc = dgrow(h, &h->cal_rel, &h->cal_rel_cap, (size_t)6 * E * h->Fpad))) return rc;
    if ((rc = dgrow(h, &h->cal_sel, &h->cal_sel_cap, mcba::select_state_bytes(12 * E)))) return rc;
    HIPCHK(hipMemcpyAsync(h->cal_views, edges, (size_t)2 * E * sizeof(int), hipMemcpyHostToDevice, h->stream));
    mcba::launch_pose_pairs(h->stream, h->cal_poses_t, (size_t)6 * h->Fpad, 1, (size_t)h->Fpad, h->cal_views, E, h->F, h->Fpad, h->cal_rel);
    mcba::launch_select(h->stream, h->cal_rel, nullptr, (size_t)h->Fpad, 6 * E, h->Fpad, h->cal_sel, 2, 1);
  }
  mcba::launch_pose_chain(h->stream, h->cal_sel, mcba::select_state_bytes(1), h->cal_views, E, root, h->C, d_ext, d_tr, d_cnt);
  mcba::launch_pose_consensus(h->stream, h->cal_poses_t, (size_t)6 * h->Fpad, 1, (size_t)h->Fpad, d_ext, h->C, h->F, h->Fpad, h->cal_world, h->cal_out);
  if ((rc = check_launch())) return rc;
  HIPCHK(hipMemcpyAsync(extrinsics_out, d_ext, (size_t)6 * h->C * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(poses_out, h->cal_out, (size_t)6 * h->F * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (transforms_out && E > 0) HIPCHK(hipMemcpyAsync(transforms_out, d_tr, (size_t)6 * E * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (counts_out && E > 0) HIPCHK(hipMemcpyAsync(counts_out, d_cnt, (size_t)E * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return MCBA_OK;
}

// The same two pose-graph steps for a caller's own (C, F, 6) pose array (NaN rows = no detection): what the reference's public
// estimate_pairwise_camera_transform / consensus_calib_poses take.  Stateless; host arrays in, host arrays out.
int mcba_pose_pairwise(int n_cameras, int n_frames, const double* poses, const int* edges, int n_edges, int device, double* transforms_out, double* counts_out) {
  if (n_cameras < 1 || n_frames < 1 || !poses || !edges || n_edges < 1 || !transforms_out) return fail(MCBA_ERR_ARG, "mcba_pose_pairwise: bad argument");
  for (int i = 0; i < 2 * n_edges; ++i)
    if (edges[i] < 0 || edges[i] >= n_cameras) return fail(MCBA_ERR_ARG, "mcba_pose_pairwise: camera index out of range");
  int rc = stateless_device(device, "mcba_pose_pairwise");
  if (rc) return rc;
  const int Fpad = (n_frames + 63) / 64 * 64;
  const size_t b_pose = up256((size_t)6 * n_cameras * n_frames * sizeof(double)), b_edge = up256((size_t)2 * n_edges * sizeof(int)), b_rel = up256((size_t)6 * n_edges * Fpad * sizeof(double)),
               b_sel = up256(mcba::select_state_bytes(12 * n_edges));
  TempDevice t;
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&t.p), b_pose + b_edge + b_rel + b_sel));
  double* d_pose = reinterpret_cast<double*>(t.p);
  int* d_edge = reinterpret_cast<int*>(t.p + b_pose);
  double* d_rel = reinterpret_cast<double*>(t.p + b_pose + b_edge);
  unsigned char* d_sel = t.p + b_pose + b_edge + b_rel;
  HIPCHK(hipMemcpyAsync(d_pose, poses, (size_t)6 * n_cameras * n_frames * sizeof(double), hipMemcpyHostToDevice, nullptr));
  HIPCHK(hipMemcpyAsync(d_edge, edges, (size_t)2 * n_edges * sizeof(int), hipMemcpyHostToDevice, nullptr));
  mcba::launch_pose_pairs(nullptr, d_pose, (size_t)6 * n_frames, 6, 1, d_edge, n_edges, n_frames, Fpad, d_rel);
  if ((rc = check_launch())) return rc;
  return pairwise_medians(nullptr, d_rel, n_edges, Fpad, d_sel, transforms_out, counts_out);
}

int mcba_pose_consensus(int n_cameras, int n_frames, const double* poses, const double* extrinsics, int device, double* poses_out) {
  if (n_cameras < 1 || n_cameras > 64 || n_frames < 1 || !poses || !extrinsics || !poses_out) return fail(MCBA_ERR_ARG, "mcba_pose_consensus: 1..64 cameras, non-NULL arrays required");
  int rc = stateless_device(device, "mcba_pose_consensus");
  if (rc) return rc;
  const int Fpad = (n_frames + 63) / 64 * 64;
  const size_t b_pose = up256((size_t)6 * n_cameras * n_frames * sizeof(double)), b_ext = up256((size_t)6 * n_cameras * sizeof(double)), b_world = up256((size_t)6 * n_cameras * Fpad * sizeof(double)),
               b_out = up256((size_t)6 * n_frames * sizeof(double));
  TempDevice t;
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&t.p), b_pose + b_ext + b_world + b_out));
  double* d_pose = reinterpret_cast<double*>(t.p);
  double* d_ext = reinterpret_cast<double*>(t.p + b_pose);
  double* d_world = reinterpret_cast<double*>(t.p + b_pose + b_ext);
  double* d_out = reinterpret_cast<double*>(t.p + b_pose + b_ext + b_world);
  HIPCHK(hipMemcpyAsync(d_pose, poses, (size_t)6 * n_cameras * n_frames * sizeof(double), hipMemcpyHostToDevice, nullptr));
  HIPCHK(hipMemcpyAsync(d_ext, extrinsics, (size_t)6 * n_cameras * sizeof(double), hipMemcpyHostToDevice, nullptr));
  mcba::launch_pose_consensus(nullptr, d_pose, (size_t)6 * n_frames, 6, 1, d_ext, n_cameras, n_frames, Fpad, d_world, d_out);
  if ((rc = check_launch())) return rc;
  HIPCHK(hipMemcpyAsync(poses_out, d_out, (size_t)6 * n_frames * sizeof(double), hipMemcpyDeviceToHost, nullptr));
  HIPCHK(hipStreamSynchronize(nullptr));
  return MCBA_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Reprojection diagnostics: the numeric core of plot_residuals (reference viz.py:166-186)
int mcba_reprojection_diagnostics(mcba_handle* h, int slot, const double* dist5, int undistort_iterations, double* median_error, double* reprojections, double* transformed) {
  if (!slot_ok(h, slot) || !median_error || undistort_iterations < 0) return fail(MCBA_ERR_ARG, "mcba_reprojection_diagnostics: bad argument");
  if (!h->have_obs || !h->obj_host) return fail(MCBA_ERR_ARG, "mcba_reprojection_diagnostics: upload observations first");
  HIPCHK(hipSetDevice(h->device));
  int rc = ensure_diag(h);
  if (rc) return rc;
  const size_t cnt = (size_t)2 * h->C * h->F * h->N;
  if (reprojections && !h->repro && (rc = dalloc(h, &h->repro, cnt))) return rc;
  if (transformed && !h->trans && (rc = dalloc(h, &h->trans, cnt))) return rc;
  if (!h->und && (rc = dalloc(h, &h->und, (size_t)2 * h->C * h->N * h->Fpad))) return rc;
  std::vector<double> d5((size_t)5 * h->C, 0.0), xc((size_t)12 * h->C);
  if (dist5) memcpy(d5.data(), dist5, d5.size() * sizeof(double));
  else {  // (k1, k2, 0, 0, 0) of the parameter vector
    HIPCHK(hipMemcpyAsync(xc.data(), h->x[slot], xc.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int c = 0; c < h->C; ++c) { d5[5 * c] = xc[12 * c + 4]; d5[5 * c + 1] = xc[12 * c + 5]; }
  }
  // Hartley normalisation of the board's XY: centroid and sqrt(2) / mean distance
  double bn[3] = {0.0, 0.0, 1.0};
  for (int p = 0; p < h->N; ++p) { bn[0] += h->obj_host[3 * p]; bn[1] += h->obj_host[3 * p + 1]; }
  bn[0] /= h->N; bn[1] /= h->N;
  double md = 0.0;
  for (int p = 0; p < h->N; ++p) md += hypot(h->obj_host[3 * p] - bn[0], h->obj_host[3 * p + 1] - bn[1]);
  bn[2] = md > 0.0 ? sqrt(2.0) * h->N / md : 1.0;
  double* d_bn = h->dmean;  // three doubles of scratch (the pre-filter's means are host-side by now)
  HIPCHK(hipMemcpyAsync(d_bn, bn, sizeof(bn), hipMemcpyHostToDevice, h->stream));
  mcba::launch_reproj_diag(h->stream, h->obs_t, h->obj, h->x[slot], d5.data(), d_bn, h->und, reprojections ? h->repro : nullptr, transformed ? h->trans : nullptr, h->err, h->C, h->F, h->N, h->Fpad,
                           undistort_iterations, 16);
  if ((rc = check_launch())) return rc;
  if ((rc = median_of_err(h, (size_t)h->N * h->Fpad, h->C, false, median_error, nullptr))) return rc;
  if (reprojections) HIPCHK(hipMemcpyAsync(reprojections, h->repro, cnt * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (transformed) HIPCHK(hipMemcpyAsync(transformed, h->trans, cnt * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return MCBA_OK;
}

// undistort_points (reference geometry.py:328-358): stateless; host arrays in, host array out
int mcba_undistort_points(size_t n_points, const double* uvs, const double* K4, const double* dist5, int iterations, int device, double* out) {
  if (!uvs || !K4 || !out || iterations < 0) return fail(MCBA_ERR_ARG, "mcba_undistort_points: bad argument");
  if (n_points == 0) return MCBA_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(MCBA_ERR_NODEVICE, "no HIP device visible");
  if (device < 0 || device >= ndev) return fail(MCBA_ERR_ARG, "device ordinal out of range");
  HIPCHK(hipSetDevice(device));
  double *d_in = nullptr, *d_out = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_in), 2 * n_points * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_out), 2 * n_points * sizeof(double));
  if (e == hipSuccess) e = hipMemcpy(d_in, uvs, 2 * n_points * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    mcba::launch_undistort(nullptr, d_in, d_out, n_points, K4, dist5, iterations);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(out, d_out, 2 * n_points * sizeof(double), hipMemcpyDeviceToHost);
  if (d_in) (void)hipFree(d_in);
  if (d_out) (void)hipFree(d_out);
  if (e != hipSuccess) { g_err = std::string("mcba_undistort_points: ") + hipGetErrorString(e); return MCBA_ERR_HIP; }
  return MCBA_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Robust triangulation (reference geometry.py:361-433): stateless; host arrays in, host array out.
int mcba_triangulate(int n_cameras, size_t n_points, const double* uvs, const double* cam12, const double* dist5, int iterations, int device, double* out, double* kernel_ms) {
  if (n_cameras < 2 || n_cameras > 64 || !uvs || !cam12 || !out || iterations < 0) return fail(MCBA_ERR_ARG, "mcba_triangulate: 2..64 cameras, non-NULL arrays, iterations >= 0 required");
  if (n_points == 0) return MCBA_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(MCBA_ERR_NODEVICE, "no HIP device visible");
  if (device < 0 || device >= ndev) return fail(MCBA_ERR_ARG, "device ordinal out of range");
  HIPCHK(hipSetDevice(device));
  // per camera {P = K [R | t] (12), K (4), dist (5)}: kernel arguments for the register path (<= 8 cameras), a device
  // array for the wavefront-per-point path
  std::vector<double> cam21((size_t)21 * n_cameras, 0.0);
  for (int c = 0; c < n_cameras; ++c) {
    const double* q = cam12 + 12 * c;
    double* P = cam21.data() + (size_t)21 * c;
    double R[9];
    mcba::rot_only(q + 6, R);
    const double fx = q[0], fy = q[1], cx = q[2], cy = q[3];
    for (int j = 0; j < 3; ++j) {
      P[j] = fx * R[j] + cx * R[6 + j];
      P[4 + j] = fy * R[3 + j] + cy * R[6 + j];
      P[8 + j] = R[6 + j];
    }
    P[3] = fx * q[9] + cx * q[11];
    P[7] = fy * q[10] + cy * q[11];
    P[11] = q[11];
    P[12] = fx; P[13] = fy; P[14] = cx; P[15] = cy;
    if (dist5) for (int k = 0; k < 5; ++k) P[16 + k] = dist5[5 * c + k];
    else { P[16] = q[4]; P[17] = q[5]; }
  }
  const bool reg_path = n_cameras <= 8;
  mcba::TriCams cams;
  memset(&cams, 0, sizeof(cams));
  if (reg_path)
    for (int c = 0; c < n_cameras; ++c) {
      memcpy(cams.P[c], cam21.data() + (size_t)21 * c, 12 * sizeof(double));
      memcpy(cams.K[c], cam21.data() + (size_t)21 * c + 12, 4 * sizeof(double));
      memcpy(cams.dist[c], cam21.data() + (size_t)21 * c + 16, 5 * sizeof(double));
    }
  double *d_uv = nullptr, *d_out = nullptr, *d_cams = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = MCBA_OK;
  auto cleanup = [&]() {
    if (d_uv) (void)hipFree(d_uv);
    if (d_out) (void)hipFree(d_out);
    if (d_cams) (void)hipFree(d_cams);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
  };
#define TRICHK(expr)                                                                                     \
  do {                                                                                                   \
    hipError_t e_ = (expr);                                                                              \
    if (e_ != hipSuccess) { g_err = std::string(#expr) + ": " + hipGetErrorString(e_); cleanup(); return MCBA_ERR_HIP; } \
  } while (0)
  const size_t nin = (size_t)2 * n_cameras * n_points;
  TRICHK(hipMalloc(reinterpret_cast<void**>(&d_uv), nin * sizeof(double)));
  TRICHK(hipMalloc(reinterpret_cast<void**>(&d_out), 3 * n_points * sizeof(double)));
  TRICHK(hipMemcpy(d_uv, uvs, nin * sizeof(double), hipMemcpyHostToDevice));
  TRICHK(hipEventCreate(&e0));
  TRICHK(hipEventCreate(&e1));
  if (!reg_path) {
    TRICHK(hipMalloc(reinterpret_cast<void**>(&d_cams), cam21.size() * sizeof(double)));
    TRICHK(hipMemcpy(d_cams, cam21.data(), cam21.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  TRICHK(hipEventRecord(e0, nullptr));
  const int lrc = reg_path ? mcba::launch_triangulate(nullptr, n_cameras, d_uv, cams, d_out, n_points, iterations)
                           : mcba::launch_triangulate_wave(nullptr, n_cameras, d_uv, d_cams, d_out, n_points, iterations);
  if (lrc != 0) { cleanup(); return fail(MCBA_ERR_ARG, "mcba_triangulate: unsupported camera count"); }
  if ((rc = check_launch())) { cleanup(); return rc; }
  TRICHK(hipEventRecord(e1, nullptr));
  TRICHK(hipMemcpy(out, d_out, 3 * n_points * sizeof(double), hipMemcpyDeviceToHost));
  if (kernel_ms) {
    float ms = 0.f;
    TRICHK(hipEventElapsedTime(&ms, e0, e1));
    *kernel_ms = ms;
  }
#undef TRICHK
  cleanup();
  return MCBA_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Single-camera calibration with OpenCV's five-coefficient model (reference calibration.py:11-71 -> cv2.calibrateCamera without
// CALIB_FIX_K3 / CALIB_ZERO_TANGENT_DIST; :74-113 -> cv2.solvePnP with such coefficients): stateless; per view the Gauss-Newton block of
// (fx fy cx cy k1 k2 p1 p2 k3 | w t), the gradient and the cost at the given parameters.  calibration.py drives the LM iteration.
int mcba_calib_normal_equations(int n_views, int n_points, const double* uvs, const double* objpoints, const double* intr9, const double* poses, int device, double* out) {
  if (n_views < 1 || n_points < 1 || !uvs || !objpoints || !intr9 || !poses || !out) return fail(MCBA_ERR_ARG, "mcba_calib_normal_equations: views >= 1, points >= 1, non-NULL arrays required");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(MCBA_ERR_NODEVICE, "no HIP device visible");
  if (device < 0 || device >= ndev) return fail(MCBA_ERR_ARG, "device ordinal out of range");
  HIPCHK(hipSetDevice(device));
  const size_t nuv = (size_t)2 * n_views * n_points, nobj = (size_t)3 * n_points, npose = (size_t)6 * n_views, nout = (size_t)136 * n_views;
  const size_t total = nuv + nobj + 9 + npose + nout;
  double* d = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&d), total * sizeof(double));
  if (e == hipSuccess) e = hipMemcpy(d, uvs, nuv * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d + nuv, objpoints, nobj * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d + nuv + nobj, intr9, 9 * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d + nuv + nobj + 9, poses, npose * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    mcba::launch_calib_views(nullptr, d, d + nuv, d + nuv + nobj, d + nuv + nobj + 9, n_views, n_points, d + nuv + nobj + 9 + npose);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(out, d + nuv + nobj + 9 + npose, nout * sizeof(double), hipMemcpyDeviceToHost);
  if (d) (void)hipFree(d);
  if (e != hipSuccess) { g_err = std::string("mcba_calib_normal_equations: ") + hipGetErrorString(e); return MCBA_ERR_HIP; }
  return MCBA_OK;
}

int mcba_comm_unique_id(unsigned char* out128) {
  if (!out128) return fail(MCBA_ERR_ARG, "mcba_comm_unique_id: NULL");
  int rc = load_rccl();
  if (rc) return rc;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  ncclResult_t r = g_rccl.GetUniqueId(&id);
  if (r != ncclSuccess) return rccl_fail("ncclGetUniqueId", r);
  memcpy(out128, &id, 128);
  return MCBA_OK;
}

int mcba_comm_init(mcba_handle* h, const unsigned char* id128, int rank, int world) {
  if (!h || !id128 || world < 1 || rank < 0 || rank >= world) return fail(MCBA_ERR_ARG, "mcba_comm_init: bad argument");
  int rc = load_rccl();
  if (rc) return rc;
  HIPCHK(hipSetDevice(h->device));
  if (h->comm) { g_rccl.CommDestroy(h->comm); h->comm = nullptr; }
  ncclUniqueId id;
  memcpy(&id, id128, 128);
  ncclResult_t r = g_rccl.CommInitRank(&h->comm, world, id, rank);
  if (r != ncclSuccess) { h->comm = nullptr; return rccl_fail("ncclCommInitRank", r); }
  return MCBA_OK;
}

int mcba_comm_allreduce(mcba_handle* h, size_t offset, size_t count) {
  if (!h || !h->comm) return fail(MCBA_ERR_ARG, "mcba_comm_allreduce: no communicator (call mcba_comm_init)");
  NEED_SOLVER(h);   // (the reduce buffer belongs to the lazily allocated solver set)
  if (offset + count > h->nsys + 8 + MCBA_LMS) return fail(MCBA_ERR_ARG, "mcba_comm_allreduce: range outside the reduce buffer");
  ncclResult_t r = g_rccl.AllReduce(h->red + offset, h->red + offset, count, ncclDouble, ncclSum, h->comm, h->stream);
  if (r != ncclSuccess) return rccl_fail("ncclAllReduce", r);
  return MCBA_OK;
}

int mcba_comm_count(mcba_handle* h, int* count) {
  if (!h || !count) return fail(MCBA_ERR_ARG, "mcba_comm_count: bad argument");
  *count = 0;
  if (!h->comm) return MCBA_OK;  // no direct communicator attached
  if (!g_rccl.CommCount) return fail(MCBA_ERR_ARG, "mcba_comm_count: ncclCommCount not available");
  ncclResult_t r = g_rccl.CommCount(h->comm, count);
  if (r != ncclSuccess) return rccl_fail("ncclCommCount", r);
  return MCBA_OK;
}

int mcba_comm_destroy(mcba_handle* h) {
  if (!h) return fail(MCBA_ERR_ARG, "NULL handle");
  if (h->comm && g_rccl.ok) { (void)hipStreamSynchronize(h->stream); g_rccl.CommDestroy(h->comm); }
  h->comm = nullptr;
  return MCBA_OK;
}


// ---------------------------------------------------------------------------------------------------------
// least_squares' numeric x_scale (the reference forwards **opt_kwargs verbatim: bundle_adjustment.py:301-313).  scipy's trust
// region lives in the variables x / x_scale; for Levenberg-Marquardt that is a FIXED damping matrix D = diag(1 / x_scale^2)
// in place of Marquardt's D = diag(J^T J) (= x_scale 'jac').  x_scale: 12C + 6F positive doubles in the layout of x, or NULL
// to return to 'jac'.
static int compose_dscale(mcba_handle* h) {
  // what the XS kernel instances read per parameter (csrc/mcba_kernels.hip: k_syrk's frame factor): > 0 the caller's fixed D = 1 / x_scale^2,
  // 0 Marquardt's diag(J^T J), < 0 frozen (frame coordinates only: frozen CAMERA parameters are flags of the camera system -- mcba_lm_auto_config,
  // or rows the host solve leaves out)
  const bool any_frozen = !h->frozen_host.empty();
  h->have_red = false;
  if (h->xs_host.empty() && !any_frozen) { h->have_xscale = false; return MCBA_OK; }
  std::vector<double> d(h->nx, 0.0);
  const size_t cnt = (size_t)12 * h->C + (size_t)6 * h->F, ncam = (size_t)12 * h->C;
  for (size_t i = 0; i < cnt; ++i) {
    d[i] = h->xs_host.empty() ? 0.0 : h->xs_host[i];
    if (any_frozen && i >= ncam && h->frozen_host[i]) d[i] = -1.0;
  }
  HIPCHK(hipMemcpyAsync(h->dscale, d.data(), h->nx * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->have_xscale = true;
  return MCBA_OK;
}

int mcba_set_x_scale(mcba_handle* h, const double* x_scale) {
  if (!h) return fail(MCBA_ERR_ARG, "NULL handle");
  HIPCHK(hipSetDevice(h->device));
  NEED_SOLVER(h);
  if (!x_scale) {
    if (h->xs_host.empty() && h->frozen_host.empty()) { h->have_red = false; h->have_xscale = false; return MCBA_OK; }   // (the usual call of every solve: nothing to upload)
    h->xs_host.clear();
    return compose_dscale(h);
  }
  const size_t cnt = (size_t)12 * h->C + (size_t)6 * h->F;
  std::vector<double> d(cnt);
  for (size_t i = 0; i < cnt; ++i) {
    if (!(x_scale[i] > 0.0) || !std::isfinite(x_scale[i])) return fail(MCBA_ERR_ARG, "`x_scale` must be 'jac' or array_like with positive numbers.");
    d[i] = 1.0 / (x_scale[i] * x_scale[i]);
  }
  h->xs_host.swap(d);
  return compose_dscale(h);
}

// Coordinates taken OUT of the system for the linear solves that follow (an active-set method's working set: solver.py, box constraints):
// mask = 12C + 6F bytes in the layout of x, non-zero = frozen, or NULL for none.  A frozen FRAME coordinate gets a step of exactly 0 and
// nothing couples to it (its row / column of V_f, its gradient entry, its column of every W block count as zero in k_syrk and k_backsub;
// the gradient mcba_get_frame_gradient reports stays the true one).  Camera entries of the mask are ignored here: freeze camera parameters
// with the flags of mcba_lm_auto_config, or leave their rows out of a host solve.  Takes effect with the next mcba_build_reduced.
int mcba_set_frozen(mcba_handle* h, const unsigned char* mask) {
  if (!h) return fail(MCBA_ERR_ARG, "NULL handle");
  HIPCHK(hipSetDevice(h->device));
  NEED_SOLVER(h);
  const size_t cnt = (size_t)12 * h->C + (size_t)6 * h->F, ncam = (size_t)12 * h->C;
  bool any = false;
  if (mask) for (size_t i = ncam; i < cnt; ++i) any = any || mask[i] != 0;
  if (!any) {
    if (h->frozen_host.empty()) return MCBA_OK;
    h->frozen_host.clear();
    return compose_dscale(h);
  }
  h->frozen_host.assign(mask, mask + cnt);
  return compose_dscale(h);
}

// Box constraints lo <= x <= hi (12C + 6F doubles each in the layout of x, -inf / +inf = none; both NULL: none at all): from now on the
// trial point of every mcba_step / mcba_step_linearize / mcba_step_fetch is PROJECTED onto the box before its cost is evaluated (k_clip).
// The working set is the caller's business (mcba_set_frozen + the camera rows it leaves out of the reduced solve): solver.py's bounded
// loop.  The device-resident loops (mcba_lm_iterate, mcba_lm_auto_*, mcba_lm_run) refuse to run while bounds are set.  Reference:
// bundle_adjustment.py:301-313 forwards `bounds` to scipy's least_squares (trf_bounds).
int mcba_set_bounds(mcba_handle* h, const double* lo, const double* hi) {
  if (!h || (lo == nullptr) != (hi == nullptr)) return fail(MCBA_ERR_ARG, "mcba_set_bounds: bad argument");
  HIPCHK(hipSetDevice(h->device));
  if (!lo) { h->have_bounds = false; return MCBA_OK; }
  const size_t cnt = (size_t)12 * h->C + (size_t)6 * h->F;
  for (size_t i = 0; i < cnt; ++i)
    if (!(lo[i] < hi[i])) return fail(MCBA_ERR_ARG, "Each lower bound must be strictly less than each upper bound.");
  int rc;
  if (!h->blo && (rc = dalloc(h, &h->blo, h->nx, false))) return rc;
  if (!h->bhi && (rc = dalloc(h, &h->bhi, h->nx, false))) return rc;
  std::vector<double> a(h->nx, -INFINITY), b(h->nx, INFINITY);
  memcpy(a.data(), lo, cnt * sizeof(double));
  memcpy(b.data(), hi, cnt * sizeof(double));
  HIPCHK(hipMemcpyAsync(h->blo, a.data(), h->nx * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->bhi, b.data(), h->nx * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->have_bounds = true;
  return MCBA_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Residual vector left ON THE DEVICE and handed to the caller as an object of its own: api.bundle_adjust attaches it to the
// OptimizeResult and downloads it when (if) `result.fun` is first read -- 52 MB of D2H at 6 x 10 000 x 54 that most callers
// never look at (the reference materialises it: scipy trf.py:557-560).  The buffer outlives the handle.

int mcba_residuals_detach(mcba_handle* h, int slot, mcba_buffer** out) {
  if (!slot_ok(h, slot) || !out) return fail(MCBA_ERR_ARG, "mcba_residuals_detach: bad argument");
  if (!h->have_obs) return fail(MCBA_ERR_ARG, "mcba_residuals_detach: upload observations first");
  HIPCHK(hipSetDevice(h->device));
  NEED_SOLVER(h);
  int rc = ensure_res(h);
  if (rc) return rc;
  {  // (the residual vector alone: the cost sums of run_cost are not wanted here)
    Scope sc(h, K_COST);
    mcba::launch_cost(h->stream, h->loss, h->f_scale, h->obs_t, h->obj, h->x[slot], h->cpart, h->res, h->C, h->F, h->N, h->Fpad, h->nch, NAN);   // NaN where a scalar is missing: the vector is its own row mask
  }
  if ((rc = check_launch())) return rc;
  mcba_buffer* b = new mcba_buffer{h->res, (size_t)2 * h->C * h->F * h->N, h->device, h->stream, h->res, (size_t)2 * h->C * h->F * h->N};
  for (size_t i = 0; i < h->bufs.size(); ++i)
    if (h->bufs[i].slot == reinterpret_cast<void**>(&h->res)) { h->bufs.erase(h->bufs.begin() + i); break; }
  h->res = nullptr;
  h->have_jac = false;
  *out = b;
  return MCBA_OK;
}
size_t mcba_buffer_count(const mcba_buffer* b) { return b ? b->count : 0; }
int mcba_buffer_download(mcba_buffer* b, double* host) {
  if (!b || !host) return fail(MCBA_ERR_ARG, "mcba_buffer_download: bad argument");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipMemcpyAsync(host, b->dev, b->count * sizeof(double), hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  return MCBA_OK;
}
int mcba_buffer_free(mcba_buffer* b) {
  if (!b) return MCBA_OK;
  (void)hipSetDevice(b->device);
  pool_free(b->base, b->base_count * sizeof(double), b->device, b->stream, true);   // (the kernel that fills it may still be running: parked as busy on its stream)
  delete b;
  return MCBA_OK;
}

// numpy.packbits(~numpy.isnan(uvs)) of the uploaded observations, taken from the device copy (k_seen_bits): 0.8 MB of D2H at
// 6 x 10 000 x 54 instead of 24 ms of numpy over the caller's 52 MB.
int mcba_seen_bits(mcba_handle* h, unsigned char* bits) {
  if (!h || !bits) return fail(MCBA_ERR_ARG, "mcba_seen_bits: bad argument");
  if (!h->have_obs) return fail(MCBA_ERR_ARG, "mcba_seen_bits: upload observations first");
  HIPCHK(hipSetDevice(h->device));
  const size_t count = (size_t)2 * h->C * h->F * h->N, words = (count + 63) / 64;
  unsigned long long* d = nullptr;
  HIPCHK(pool_malloc(reinterpret_cast<void**>(&d), words * 8, h->device, h->stream));
  mcba::launch_seen_bits(h->stream, h->obs_raw, count, d);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(bits, d, (count + 7) / 8, hipMemcpyDeviceToHost, h->stream);
  hipError_t e2 = hipStreamSynchronize(h->stream);
  pool_free(d, words * 8, h->device);
  if (e != hipSuccess || e2 != hipSuccess) { g_err = std::string("mcba_seen_bits: ") + hipGetErrorString(e != hipSuccess ? e : e2); return MCBA_ERR_HIP; }
  return MCBA_OK;
}

// ---------------------------------------------------------------------------------------------------------
// One pass of the radix select behind mcba_error_median, for callers that hold only a SHARD of the frames (frame-sharded
// bundle_adjust: every rank runs the pre-filter on its own slice): the 256-bin histogram of byte `pass` (0 = most significant)
// over this handle's per-point errors whose leading `pass` bytes equal `prefix`, restricted to frame_mask (F bytes, NULL = the mask
// of the previous call).  The caller sums the histograms over the ranks, picks the bin that holds the wanted rank and calls again
// with the longer prefix -- integer arithmetic only, so the order statistic is exact whatever the sharding.
int mcba_error_histogram(mcba_handle* h, const unsigned char* frame_mask, unsigned long long prefix, int pass, unsigned long long* hist256) {
  if (!h || !hist256 || pass < 0 || pass > 7) return fail(MCBA_ERR_ARG, "mcba_error_histogram: bad argument");
  if (!h->err) return fail(MCBA_ERR_ARG, "mcba_error_histogram: call mcba_frame_errors first");
  HIPCHK(hipSetDevice(h->device));
  if (frame_mask) {
    HIPCHK(hipMemsetAsync(h->fmask, 0, (size_t)h->Fpad, h->stream));
    HIPCHK(hipMemcpyAsync(h->fmask, frame_mask, (size_t)h->F, hipMemcpyHostToDevice, h->stream));
  }
  unsigned int hist[256];
  int rc = mcba::launch_select_hist(h->stream, h->err, h->fmask, (size_t)h->C * h->N * h->Fpad, h->Fpad, h->sel, prefix, pass, hist);
  if (rc) { g_err = "mcba_error_histogram: HIP error"; return MCBA_ERR_HIP; }
  for (int b = 0; b < 256; ++b) hist256[b] = hist[b];
  return MCBA_OK;
}

// device-resident loop: how often a back-substitution workgroup of k_solve_backsub gave up waiting for the solve (a bounded
// poll, ~0.5 s) since mcba_lm_auto_config, and whether the fused launch is still in use (the first such event switches the
// handle to the two-launch k_solve_cam + k_backsub path for good)
int mcba_lm_fuse_status(mcba_handle* h, double* timeouts, int* fused) {
  if (!h || !h->have_solver) return fail(MCBA_ERR_ARG, "mcba_lm_fuse_status: bad argument");
  if (timeouts) *timeouts = const_cast<volatile double*>(h->ring)[(size_t)kRing * MCBA_LMS];
  if (fused) *fused = h->fuse_backsub ? 1 : 0;
  return MCBA_OK;
}

int mcba_get_frame_gradient(mcba_handle* h, double* host) {
  if (!h || !host) return fail(MCBA_ERR_ARG, "mcba_get_frame_gradient: bad argument");
  if (!h->have_red) return fail(MCBA_ERR_ARG, "mcba_get_frame_gradient: call mcba_build_reduced first");
  HIPCHK(hipMemcpy2DAsync(host, 6 * sizeof(double), h->fbuf + 27, MCBA_FB * sizeof(double), 6 * sizeof(double), h->F, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return MCBA_OK;
}

// Measurement aid (bench.py): the FP64 vector rate this device sustains with one wavefront per SIMD on every compute unit issuing
// independent v_fma_f64 -- what k_gram's instruction stream can be priced against besides the datasheet peak.  ~0.3 ms of GPU time.
int mcba_fp64_issue_rate(int device, double* tflops) {
  if (!tflops) return fail(MCBA_ERR_ARG, "mcba_fp64_issue_rate: NULL");
  *tflops = 0.0;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(MCBA_ERR_NODEVICE, "no HIP device visible");
  if (device < 0 || device >= ndev) return fail(MCBA_ERR_ARG, "device ordinal out of range");
  HIPCHK(hipSetDevice(device));
  hipDeviceProp_t prop;
  int ncu = 256;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
  *tflops = mcba::measure_fp64_issue_rate(ncu);
  if (!(*tflops > 0.0)) return fail(MCBA_ERR_HIP, "mcba_fp64_issue_rate: the measurement kernel failed");
  return MCBA_OK;
}

int mcba_profile_enable(mcba_handle* h, int on) {
  if (!h) return fail(MCBA_ERR_ARG, "NULL handle");
  h->prof = on != 0;
  h->prof_mask = (on == 0 || on == 1) ? ~0u : ((unsigned)on >> 1);
  h->prof_stride = 1;
  h->prof_exact = false;
  return MCBA_OK;
}

// k_gram's launches (the fused kernel) are timed by events attached to the dispatch itself (hipExtLaunchKernelGGL): the kernel's own
// begin and end, what rocprofv3 reports -- an event pair recorded around a launch reads ~2.5 us more.  Reset by mcba_profile_enable.
int mcba_profile_exact(mcba_handle* h, int on) {
  if (!h) return fail(MCBA_ERR_ARG, "NULL handle");
  h->prof_exact = on != 0;
  return MCBA_OK;
}

int mcba_profile_stride(mcba_handle* h, int stride) {
  if (!h || stride < 1) return fail(MCBA_ERR_ARG, "mcba_profile_stride: stride >= 1 required");
  h->prof_stride = stride;
  return MCBA_OK;
}

// What a HIP-event bracket reads with NOTHING between its two records, on the handle's stream (mean of `pairs` back-to-back brackets, in
// microseconds): the part of a bracketed kernel's time that is the bracket's, not the kernel's.  bench.py subtracts it from the
// event-timed duration of the dominant kernel so that the figure agrees with rocprofv3's (50.4 us by events, 47.8 by rocprof).
int mcba_profile_bracket_overhead(mcba_handle* h, int pairs, double* us) {
  if (!h || !us || pairs < 1 || pairs > 4096) return fail(MCBA_ERR_ARG, "mcba_profile_bracket_overhead: bad argument");
  HIPCHK(hipSetDevice(h->device));
  std::vector<hipEvent_t> ev(2 * (size_t)pairs);
  for (auto& e : ev) HIPCHK(hipEventCreate(&e));
  HIPCHK(hipStreamSynchronize(h->stream));
  for (int i = 0; i < pairs; ++i) { HIPCHK(hipEventRecord(ev[2 * i], h->stream)); HIPCHK(hipEventRecord(ev[2 * i + 1], h->stream)); }
  HIPCHK(hipStreamSynchronize(h->stream));
  double tot = 0.0;
  for (int i = 0; i < pairs; ++i) { float ms = 0.f; HIPCHK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1])); tot += ms; }
  for (auto& e : ev) (void)hipEventDestroy(e);
  *us = 1e3 * tot / pairs;
  return MCBA_OK;
}

int mcba_profile_read(mcba_handle* h, double* ms_total, int* calls, int capacity, int* n_kernels) {
  if (!h || !ms_total || !calls || capacity < K_COUNT) return fail(MCBA_ERR_ARG, "mcba_profile_read: need capacity >= number of kernels");
  HIPCHK(hipStreamSynchronize(h->stream));
  for (int i = 0; i < K_COUNT; ++i) { ms_total[i] = 0.0; calls[i] = 0; }
  for (auto& e : h->evs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) { ms_total[e.kid] += ms; calls[e.kid] += 1; }
    h->pool.push_back(e.a);
    h->pool.push_back(e.b);
  }
  h->evs.clear();
  if (n_kernels) *n_kernels = K_COUNT;
  return MCBA_OK;
}

}  // extern "C"
