// Host-side internals of libmsm_hip.so shared by its translation units (round 5 split msm_api.hip into them):
//   msm_plan.hip      window size, plan, launch geometry, workspace budget model, error helpers
//   msm_sort.hip      digits + counting sort of one window group (driver of sort_kernels.h)
//   msm_tree.hip      accumulation tree of one window group (driver of k_batch_add / k_te_add) + k_bucket_finish
//   msm_reduce.hip    bucket reduction, window sums, host tail (Horner, to-affine, combine)
//   msm_upload.hip    staged and pipelined host -> device transfers
//   msm_pipeline.hip  window groups on two streams, point ranges, multi-device fan-out
//   msm_tables.hip    window tables (K resident tables 2^(c k) P: one set of buckets for all windows)
//   msm_abi.hip       the C ABI of include/msm_hip.h (contexts, points, msm_run, msm_window_sums, handles)
//   msm_test_abi.hip  the operator-level test entries (msm_test_*) and the input generators
// Kernels live in kernels_curve.hip (one TU per curve), sort_kernels.hip and te_kernels.hip; host TUs see declarations.
// Orchestration follows `createMsm().msm` (reference src/msm-batched-affine.ts:69-340); the per-thread SPMD phases separated
// by `barrier()` there become kernel launches on one HIP stream here.
#pragma once
#include "kernel_inst.h"   // curve-templated kernels: extern templates, defined in kernels_curve.hip per curve
#include "sort_kernels.h"
#include "tree_kernels.h"
#include "te_kernels.h"
#include "host_field.h"
#include "../../include/msm_hip.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

// Policy knobs -- the six thresholds that are re-measured when a plan changes (MSM_GROUPS, MSM_MAX_STEPS, MSM_PBL, MSM_TC,
// MSM_FINISH_MAX, MSM_TAIL_MIN) -- are environment variables ONLY in builds made with -DMSM_TUNING (make ab EXTRA=-DMSM_TUNING,
// tools/knob_sweep.sh); the product never reads the environment.  The knobs of closed experiments left in round 6.
#ifdef MSM_TUNING
#define MSM_KNOB(var, name, lo) do { if (const char* _e = getenv(name)) var = std::max<long long>((lo), atoll(_e)); } while (0)
#define MSM_KNOB_SET(name) (getenv(name) != nullptr)
#else
#define MSM_KNOB(var, name, lo) do { } while (0)
#define MSM_KNOB_SET(name) false
#endif

namespace msmi {

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

// a DevBuf local to one call: freed on every way out of the scope, exceptions included
struct ScopedDevBuf : DevBuf {
  ScopedDevBuf() = default;
  ScopedDevBuf(const ScopedDevBuf&) = delete;
  ScopedDevBuf& operator=(const ScopedDevBuf&) = delete;
  ~ScopedDevBuf() {
    if (p) (void)hipFree(p);
  }
};

struct HipFail {
  hipError_t e;
  const char* what;
  int line;
  const char* file;
};

// non-HIP failure raised inside the pipeline (mapped to its error code at the ABI boundary)
struct MsmFail {
  int code;
  std::string msg;
};


#define HIPCHK(x)                                   \
  do {                                              \
    hipError_t _e = (x);                            \
    if (_e != hipSuccess) throw msmi::HipFail{_e, #x, __LINE__, __FILE__}; \
  } while (0)

inline uint32_t ceil_log2_u64(uint64_t n) {
  uint32_t r = 0;
  while ((1ull << r) < n) r++;
  return r;
}

// host-side view of the per-curve constants (constants_gen.h)
struct CurveInfo {
  const uint32_t* pw;   // base field modulus, 12 words (zero-extended for the 8-word fields)
  const uint32_t* q;    // scalar field order, 8 words
  const uint32_t* gx;   // generator (Weierstrass curves), 12 words each, device Montgomery form
  const uint32_t* gy;
  int glv_max_bits;     // Scalar.maxBits after decomposition, src/wasm/glv.ts:216-226
  int q_bits;           // bit length of q
};

inline const CurveInfo& curve_info(int curve) {
  static const CurveInfo bls377 = {msm::Fp377::PW, msm::GlvBls377::Q, msm::Fp377::GXW, msm::Fp377::GYW, msm::GlvBls377::MAX_BITS, 253};
  static const CurveInfo bls381 = {msm::Fp381::PW, msm::GlvBls381::Q, msm::Fp381::GXW, msm::Fp381::GYW, msm::GlvBls381::MAX_BITS, 255};
  // Pallas lives on 8 packed words; the host side reads 12 (zero-extended copies)
  static uint32_t pal_p[12], pal_gx[12], pal_gy[12];
  static const bool pal_init = [] {
    for (int i = 0; i < 8; i++) { pal_p[i] = msm::FpPallas::PW[i]; pal_gx[i] = msm::FpPallas::GXW[i]; pal_gy[i] = msm::FpPallas::GYW[i]; }
    return true;
  }();
  (void)pal_init;
  static const CurveInfo pallas = {pal_p, msm::GlvPallas::Q, pal_gx, pal_gy, msm::GlvPallas::MAX_BITS, 255};
  static const CurveInfo ed377 = {msm::Fp253::PW, msm::FRED_Q, nullptr, nullptr, 251, 251};
  return curve == MSM_CURVE_BLS12_381_G1 ? bls381 : curve == MSM_CURVE_PALLAS ? pallas : curve == MSM_CURVE_ED_ON_BLS12_377 ? ed377 : bls377;
}

// One helper thread per context, started with it: the second window group of a big MSM runs here (the calling thread
// takes the first), so no thread is created per call.  run() hands over a job, wait() returns when it is done and
// re-raises whatever the job threw.
class HelperThread {
 public:
  HelperThread() : th_([this] { loop(); }) {}
  ~HelperThread() {
    {
      std::lock_guard<std::mutex> l(mu_);
      quit_ = true;
    }
    cv_.notify_all();
    th_.join();
  }
  void run(std::function<void()> job) {
    std::lock_guard<std::mutex> l(mu_);
    job_ = std::move(job);
    busy_ = true;
    err_ = nullptr;
    cv_.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> l(mu_);
    cv_.wait(l, [this] { return !busy_; });
    if (err_) {
      std::exception_ptr e = err_;
      err_ = nullptr;
      std::rethrow_exception(e);
    }
  }

 private:
  void loop() {
    std::unique_lock<std::mutex> l(mu_);
    for (;;) {
      cv_.wait(l, [this] { return quit_ || (busy_ && job_); });
      if (quit_) return;
      std::function<void()> job = std::move(job_);
      job_ = nullptr;
      l.unlock();
      std::exception_ptr e;
      try { job(); } catch (...) { e = std::current_exception(); }
      l.lock();
      err_ = e;
      busy_ = false;
      cv_.notify_all();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_;
  std::function<void()> job_;
  std::exception_ptr err_;
  bool busy_ = false, quit_ = false;
  std::thread th_;   // last member: the thread starts after everything it touches exists
};

}  // namespace msmi

struct msm_ctx {
  std::unique_ptr<msmi::HelperThread> helper;
  int curve = 0;
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[12] = {};
  hipEvent_t ev_dig[2] = {};   // around the digit kernel that serves both window groups of a call (GroupDigits)
  std::string err;
  int n_cu = 256;

  // resident points: `rows` / `n_points` are the CURRENT point set; the others wait in `sets` (msm_pointset_*)
  msmi::DevBuf rows;
  uint64_t n_points = 0;
  struct PointSet {
    msmi::DevBuf rows;
    uint64_t n = 0;
    bool live = false;
    int tab_c = 0, tab_K = 0;
    uint64_t tab_lo = 0, tab_n = 0;
    msmi::DevBuf tabs;
  };
  // Window tables of the CURRENT point set (msm_tables.hip): tab_K tables of tab_n rows each, table k = 2^(tab_c k) P over the
  // points [tab_lo, tab_lo + tab_n) (tab_K = 0: none).  Tables of the WHOLE set live in `rows` (which grows to hold them: table 0
  // is the plain row table); tables of a RANGE of the points -- the share of one rank of a points-split run -- in `tabs`, with
  // a copy of the range's rows as table 0.  Travel with the set through msm_pointset_select.
  int tab_c = 0, tab_K = 0;
  uint64_t tab_lo = 0, tab_n = 0;
  msmi::DevBuf tabs;
  // the range the last table-eligible call over a RANGE of the points asked for: range tables are built when a call comes back
  // for the same range (a rank of a sharded run does; a caller walking over the shards of one GPU does not and is spared a build per call)
  uint64_t cand_lo = 0, cand_n = 0;
  int cand_c = 0;
  const uint32_t* table_rows() const { return (const uint32_t*)(tabs.p ? tabs.p : rows.p); }
  void drop_tables() {
    tab_c = tab_K = 0;
    tab_lo = tab_n = 0;
    cand_n = 0;
    release(tabs);
  }
  uint64_t tables_limit = 0;   // bytes the tables of one point set may take (msm_set_tables_limit; default: 10 % of the device)
  std::vector<PointSet> sets = std::vector<PointSet>(1);   // slot 0 = the default set
  int cur_set = 0;
  std::vector<void*> allocs;        // device buffers handed out by msm_device_alloc
  // multi-device context (msm_ctx_create_multi): this context drives devices[0], one child context per further device
  std::vector<msm_ctx*> children;
  std::vector<std::unique_ptr<msmi::HelperThread>> fan;   // one host thread per child for the window-shard fan-out

  // staging / misc buffers shared by all window groups
  msmi::DevBuf scal, errflag, misc;
  uint32_t* h_info = nullptr;      // pinned
  // host -> device staging of big pageable buffers (upload_staged): pinned chunks, a copy stream and an event per chunk slot
  static constexpr int STAGE_THREADS = 4, STAGE_SLOTS = 2;
  static constexpr size_t STAGE_CHUNK = (size_t)16 << 20;
  char* stage_pin = nullptr;
  bool staging_ready = false;      // pinned slots, copy streams and events all exist (ensure_staging)
  hipStream_t stage_stream[STAGE_THREADS] = {};
  hipEvent_t stage_ev[STAGE_THREADS][STAGE_SLOTS + 1] = {};
  static constexpr int MAX_PIECES = 6;   // ranges of the points a host-scalar MSM is pipelined over (PieceUpload)
  hipEvent_t piece_ev[MAX_PIECES][STAGE_THREADS] = {};
  uint64_t ws_budget = 0;          // bytes the per-group workspaces may take in total
  uint64_t ws_limit = 0;           // msm_set_workspace_limit: the caller's cap on ws_budget (0 = automatic)

  // per-group workspace: two of them, each with its own stream, so that the memory-bound sort of one
  // window group runs under the ALU-bound accumulation of the other
  struct Workspace {
    msmi::DevBuf dig, counts, cursor, tail_off, info, slots, block_hist, scan_partial, desc, columns2, rows_sum, bucket_proj, bufA, bufB,
        scratch, columns, partials, part, dig2, idx2, rec, blk_tab2, slots2, dest, rows1, parts, sub;
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr;   // read-backs that the host needs while `stream` goes on (the sort's totals under its last pass)
    hipEvent_t ev[8] = {};        // (ev[7]: the scans of the bucket sizes are done)
    uint32_t* h_info = nullptr;   // pinned, 64 words
    uint32_t* h_part = nullptr;   // pinned, window sums read-back
    msmi::DevBuf* all[27] = {&dig, &counts, &cursor, &tail_off, &info, &slots, &block_hist, &scan_partial, &desc, &columns2, &rows_sum,
                       &bucket_proj, &bufA, &bufB, &scratch, &columns, &partials, &part, &dig2, &idx2, &rec, &blk_tab2, &slots2, &dest, &rows1,
                       &parts, &sub};
  };
  static constexpr int N_WS = 2;
  Workspace ws[N_WS];

  msm_host::Curve6 hc;
  msm_host::Fe6 k_dev_to_host;  // 2^(2 * 64 nl_host - 30 NL): device Montgomery (radix 2^(30 NL)) -> host Montgomery (2^384 or 2^256)
  msm_host::TeCurve6 hte;       // Ed-on-BLS12-377 over the 253-bit field (same 6-limb host field code)
  msm_host::Fe6 k_te_to_host;   // 2^(512 - 270): device Montgomery (2^270) -> host Montgomery (2^256: four active limbs)
  bool is_te() const { return curve == MSM_CURVE_ED_ON_BLS12_377; }
  // per-field sizes (the reference sizes limbs per field, src/parallel.ts:53-57): 30-bit limbs in registers, packed words
  // per coordinate in memory, and the coordinate bytes at the ABI (wire points, results, test operands)
  int nl() const { return (curve == MSM_CURVE_PALLAS || is_te()) ? 9 : 13; }
  int nw() const { return (curve == MSM_CURVE_PALLAS || is_te()) ? 8 : 12; }
  size_t coord_bytes() const { return (size_t)nw() * 4; }

  void ensure(msmi::DevBuf& b, size_t bytes) {
    if (bytes <= b.cap) return;
    if (b.p) HIPCHK(hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    size_t want = bytes + bytes / 16 + 256;
    const hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
      (void)hipGetLastError();   // the failure must not stay behind as this thread's "last error": the kernel-launch checks read it
      b.p = nullptr;
      throw msmi::HipFail{e, "hipMalloc(&b.p, want)", __LINE__, __FILE__};
    }
    b.cap = want;
  }
  void release(msmi::DevBuf& b) {
    if (b.p) (void)hipFree(b.p);   // teardown paths: nothing useful to do with an error here
    b.p = nullptr;
    b.cap = 0;
  }
};

// curve dispatch for the templated Weierstrass kernels
#define W_LAUNCH(ctx, KERNEL, ...)                                                         \
  do {                                                                                     \
    if ((ctx)->curve == MSM_CURVE_BLS12_381_G1) hipLaunchKernelGGL((KERNEL<msm::CvBls381>), __VA_ARGS__); \
    else if ((ctx)->curve == MSM_CURVE_PALLAS) hipLaunchKernelGGL((KERNEL<msm::CvPallas>), __VA_ARGS__);  \
    else hipLaunchKernelGGL((KERNEL<msm::CvBls377>), __VA_ARGS__);                          \
  } while (0)
#define W_LAUNCH_MODE(ctx, KERNEL, MODE, ...)                                              \
  do {                                                                                     \
    if ((ctx)->curve == MSM_CURVE_BLS12_381_G1) hipLaunchKernelGGL((KERNEL<msm::CvBls381, MODE>), __VA_ARGS__); \
    else if ((ctx)->curve == MSM_CURVE_PALLAS) hipLaunchKernelGGL((KERNEL<msm::CvPallas, MODE>), __VA_ARGS__);  \
    else hipLaunchKernelGGL((KERNEL<msm::CvBls377, MODE>), __VA_ARGS__);                    \
  } while (0)

// point rows -> tree planes (test ops), by the packed words of the curve's coordinates
#define ROWS_TO_PLANES(ctx, ...)                                                                       \
  do {                                                                                                 \
    if ((ctx)->nw() == 8) hipLaunchKernelGGL((k_test_rows_to_planes<8>), __VA_ARGS__);                 \
    else hipLaunchKernelGGL((k_test_rows_to_planes<12>), __VA_ARGS__);                                 \
  } while (0)

namespace msmi {
using namespace msm;

int fail(msm_ctx* ctx, int code, const char* fmt, ...);
int fail_hip(msm_ctx* ctx, const HipFail& f);

// every extern "C" entry point ends its try block with this: no C++ exception crosses the C ABI
#define MSM_CATCH_ALL(ctx)                                                                                  \
  catch (const HipFail& f) { return fail_hip(ctx, f); }                                                     \
  catch (const MsmFail& f) { return fail(ctx, f.code, "%s", f.msg.c_str()); }                               \
  catch (const std::bad_alloc&) { return fail(ctx, MSM_ERR_INTERNAL, "host memory allocation failed"); }    \
  catch (const std::exception& e) { return fail(ctx, MSM_ERR_INTERNAL, "unexpected exception: %s", e.what()); } \
  catch (...) { return fail(ctx, MSM_ERR_INTERNAL, "unexpected exception"); }

// ---- msm_plan.hip -----------------------------------------------------------------------------------------------
int pick_window(bool te, uint64_t n, int glv_max_bits);

struct Plan {
  int c, K, L_log;       // c: bits a window advances by (the weight of window k is 2^(c k)); L_log: bits of a bucket index
  int bits = 0;          // b + 1: scalar bits the windows cover (the top window holds bits - (K - 1) c of them)
  bool fold = false;     // the top window is c + 1 bits wide (see make_plan)
  uint32_t L;            // buckets per window = 2^L_log
  bool no_glv;
  bool strict = false;   // msm_opts.strict: scalars >= q fail the call instead of being reduced
  bool lone = false;   // one window, one group: nothing else shares the GPU (see round_geom)
  uint32_t b_lo = 0, b_n = 0xFFFFFFFFu;   // bucket-range shard (msm_opts.bucket_shard): bucket indices [b_lo, b_lo + b_n) only;
  uint32_t bt_lo = 0, bt_n = 0xFFFFFFFFu; // the top window's range (its digits cover another span than the recoded windows')
  bool tables = false; // the call runs on window tables (msm_tables.hip): the windows of a group share one set of buckets, a
                       // group hands back ONE sum that already carries the windows' weights
  const uint32_t* tab_rows = nullptr;   // tables: first row of table 0; its tables are tab_n rows each and cover the points
  uint64_t tab_lo = 0, tab_n = 0;       //         [tab_lo, tab_lo + tab_n)
  bool merged = false; // a full MSM (msm_run): a window group may hand back sum_k 2^(c (k - k_first)) P_k in the slot of its
                       // first window instead of one P_k per slot (reduce_buckets); msm_window_sums never sets it
};

// for_tables: the window a run on window tables wants (bucket work no longer grows with the number of windows)
int make_plan(const msm_ctx* ctx, uint64_t n, const msm_opts* opts, Plan& pl, bool for_tables = false);
// the plan of msm_run(n, opts) -- on window tables where the call is eligible and they exist or would be built -- and whether
// it is that plan (msm_tables.hip)
// note_range: the call is real (not msm_plan): a range of the points it asks for is remembered as the candidate for range tables
int make_run_plan(msm_ctx* ctx, uint64_t n, const msm_opts* opts, bool placed, Plan& pl, bool& tables_wanted, bool note_range = false);

struct GroupStats {
  uint64_t n_pairs = 0;
  uint64_t n_pairs_algo = 0;
  uint64_t max_bucket = 0;
  int rounds = 0;
  float ms_digits = 0, ms_sort = 0, ms_acc = 0, ms_red = 0, ms_r1 = 0;
};

// launch geometry of one tree round
struct RoundGeom {
  uint32_t steps, grid;
  uint64_t T;
};

RoundGeom round_geom(const msm_ctx* ctx, uint64_t n_out, bool gather = false, bool lone = false);
void release_workspaces(msm_ctx* ctx);   // drops every per-call buffer of both window-group workspaces (they only grow otherwise)
long double window_bytes(const msm_ctx* ctx, uint64_t n, const Plan& pl);
int windows_per_group(const msm_ctx* ctx, uint64_t n, const Plan& pl);
uint64_t point_pieces(const msm_ctx* ctx, uint64_t n, const Plan& pl);

// ---- msm_reduce.hip ---------------------------------------------------------------------------------------------
void words_to_fe6(msm_host::Fe6& r, const uint32_t* w, int nw = 12);
void fe6_to_bytes(uint8_t* out, const msm_host::Fe6& a);
void plane_element_to_wire(const msm_ctx* ctx, const uint32_t* planes, uint64_t cap, uint64_t e, uint8_t* out_xy);
msm_host::Proj6 partial_to_host(const msm_ctx* ctx, const uint32_t* w);
void host_to_partial(const msm_ctx*, const msm_host::Proj6& P, uint32_t* out36);
void reduce_buckets(msm_ctx* ctx, msm_ctx::Workspace& w, const uint4* fin, uint64_t fin_cap, const uint32_t* off_fin,
                    const uint32_t* bucket_proj, uint32_t L, int kc, uint32_t* h_partials_out, bool merged = false, int stride = 0);
msm_host::Proj6 horner_points(const msm_host::Curve6& C, const std::vector<msm_host::Proj6>& P, int c);
void proj_to_result(const msm_host::Curve6& C, const msm_host::Proj6& acc, msm_result* out);
void horner_to_affine(const msm_host::Curve6& C, const std::vector<msm_host::Proj6>& P, int c, msm_result* out);
void te_horner_points(const msm_host::TeCurve6& C, const std::vector<msm_host::Ext6>& P, int c, msm_result* out);
msm_host::Ext6 te_partial_to_host(const msm_ctx* ctx, const uint32_t* w);
void te_host_to_partial(const msm_ctx*, const msm_host::Ext6& P, uint32_t* out32);
void te_horner_to_affine(const msm_ctx* ctx, const std::vector<uint32_t>& words, int K, int c, msm_result* out);
const msm_host::Curve6* static_host_curve(int curve);
int combine_impl(msm_ctx* ctx, const msm_host::Curve6& C, const uint8_t* partials, int32_t K, int32_t c, msm_result* out, int32_t G);
int te_combine_impl(const uint8_t* partials, int32_t K, int32_t c, msm_result* out, int32_t G = 1);

// ---- msm_sort.hip / msm_tree.hip: the two halves of one window group --------------------------------------------
// what the sort of a window group leaves behind for its tree
struct SortOut {
  uint32_t logG = 1;            // buckets are padded to multiples of 2^logG slots
  int RT = 0;                   // tail rounds the largest bucket would need
  uint64_t total_slots = 0;
  uint32_t max_bucket = 0;
  const uint32_t* round1_slots = nullptr;   // pairs round 1 walks (bucket order, or the tile order of k_bin_pairs)
  const uint32_t* round1_dest = nullptr;    // tile order: the element index every pair's sum belongs to
  uint64_t rec_y_off = 0;       // 12-word fields: where the y records of round 1's results start inside w.rows1
  bool chunked = false;         // round 1 walks tile-ordered pairs and writes element records, round 2 reads them
};
// Digits and slice histograms of SEVERAL window groups from one launch of the digit kernel (the two groups of a call decompose
// the same scalars: one GLV decomposition instead of two).  sort_window_group(..., share) with share->produce set runs the digit
// kernel over [k_lo, k_hi) -- all groups' windows -- into w's buffers, fills the rest of `share`, records `ready` and returns;
// a group's own call then takes its part of the arrays (bin split only) instead of slicing the scalars again.
struct GroupDigits {
  bool produce = false;
  int k_lo = 0;                   // first window of the arrays
  const uint32_t* dig = nullptr;  // [windows][entries per window]
  uint32_t* hist = nullptr;       // [windows][sortB][hb]
  uint32_t hb = 0, sortB = 0;
  uint64_t pps = 0, chunk = 0;
  hipEvent_t ready = nullptr;     // behind the digit kernel, on the producing workspace's stream
  bool valid = false;             // the producer found the bin split applicable and has run
};
void sort_window_group(msm_ctx* ctx, msm_ctx::Workspace& w, const uint32_t* d_scalars, uint64_t n, const Plan& pl, int k_lo, int k_hi,
                       GroupStats& st, SortOut& so, GroupDigits* share = nullptr);
// what the tree leaves behind for the bucket reduction
struct TreeOut {
  const uint4* fin = nullptr;   // tree buffer holding what is left of every bucket
  uint64_t fin_cap = 0;
  const uint32_t* off_fin = nullptr;
  const uint32_t* bucket_proj = nullptr;   // bucket sums from k_bucket_finish (projective / extended)
};
// kc: windows of the group as the tree sees them (1 on window tables); row_off: first row of the point table the payloads count from
void accumulate_window_group(msm_ctx* ctx, msm_ctx::Workspace& w, const Plan& pl, int kc, uint64_t row_off, const SortOut& so,
                             GroupStats& st, TreeOut& to);
void sort_kernel_attributes();   // dynamic-LDS limits of the sort kernels (once per process and device)

// ---- msm_tables.hip ---------------------------------------------------------------------------------------------
// true if the MSM over the resident points [opts->point_lo, + n) under plan `pl` can run on window tables -- `pl` then knows
// where they are (tab_rows, tab_lo, tab_n); builds them when `may_build`
bool use_window_tables(msm_ctx* ctx, uint64_t n, const msm_opts* opts, Plan& pl, bool may_build);

// ---- msm_upload.hip ---------------------------------------------------------------------------------------------
void ensure_staging(msm_ctx* ctx);
void upload_staged(msm_ctx* ctx, void* dst, const void* src, size_t bytes);
int stage_scalars(msm_ctx* ctx, const void* scalars, uint64_t n, int on_device, const uint32_t** d_out);
// The same staged transfer running BEHIND the call that consumes it: a host scalar buffer of a big MSM crosses PCIe in the
// background while the MSM already runs over the ranges of the points ("pieces") whose scalars have arrived -- 2 GB take
// ~45 ms at the rate of the link, a 2^26 MSM ~150 ms, and the sort of a window needs every digit of its range, so the
// unit of overlap is a range of the points, not a chunk (window_sums_once picks growing ranges: the first one is small so the
// GPU starts early, the last one is half the input so most of the work runs at full-size efficiency).
// Chunks go out in address order over the staging threads as in upload_staged; when a thread has queued its last chunk of
// piece q it records piece_ev[q][t] on its copy stream, and wait_piece(q, stream) makes `stream` wait for all of them.
// The reference's counterpart is scalarsFromBytes into shared wasm memory before the call, src/parallel.ts:119-133.
class PieceUpload {
 public:
  static constexpr int T = msm_ctx::STAGE_THREADS, S = msm_ctx::STAGE_SLOTS;
  static constexpr size_t CH = msm_ctx::STAGE_CHUNK;
  PieceUpload(msm_ctx* ctx, void* dst, const void* src, size_t bytes, const std::vector<size_t>& piece_end_bytes)
      : ctx_(ctx), dst_((char*)dst), src_((const char*)src), bytes_(bytes), ends_(piece_end_bytes), enq_(piece_end_bytes.size(), 0) {
    ensure_staging(ctx);
    n_streams_ = std::max<long long>(1, std::min<long long>(n_streams_, T));
    HIPCHK(hipStreamSynchronize(ctx->stream));   // dst may still be in use by what the stream holds
    for (int t = 0; t < T; t++) HIPCHK(hipStreamSynchronize(ctx->stage_stream[t]));
    for (int t = 0; t < T; t++) rc_[t] = hipSuccess;
    t0_ = std::chrono::steady_clock::now();
    th_.reserve(T);
    try {
      for (int t = 0; t < T; t++) {
        // a thread that cannot be started (resource limits) must not leave the others unjoined: its share runs here
        try { th_.emplace_back([this, t] { run(t); }); } catch (const std::system_error&) { run(t); }
      }
    } catch (...) {   // anything else: the destructor of a half-built object does not run, so the started threads are joined here
      join();
      throw;
    }
  }
  ~PieceUpload() { join(); }
  // host: blocks until every staging thread has queued its part of piece q; device: `stream` then waits for those copies
  void wait_piece(int q, hipStream_t stream) {
    {
      std::unique_lock<std::mutex> l(mu_);
      cv_.wait(l, [&] { return enq_[q] == T; });
    }
    hipError_t rc[T];
    {
      std::lock_guard<std::mutex> l(mu_);   // the staging threads write rc_ under the mutex
      for (int t = 0; t < T; t++) rc[t] = rc_[t];
    }
    for (int t = 0; t < T; t++) {
      if (rc[t] != hipSuccess) throw HipFail{rc[t], "staged upload of the scalars", __LINE__, __FILE__};
      HIPCHK(hipStreamWaitEvent(stream, ctx_->piece_ev[q][t], 0));
    }
  }
  // joins the staging threads, waits for the last copy and returns the wall time of the whole transfer in ms
  float finish() {
    join();
    for (int t = 0; t < T; t++) HIPCHK(rc_[t]);
    for (int t = 0; t < T; t++) HIPCHK(hipStreamSynchronize(ctx_->stage_stream[t % n_streams_]));
    return ms_;
  }

 private:
  void join() {
    for (auto& x : th_) if (x.joinable()) x.join();
  }
  void run(int t) {
    hipError_t e = hipSetDevice(ctx_->device);
    const size_t n_chunks = (bytes_ + CH - 1) / CH;
    size_t turn = 0;
    int q = 0;
    auto mark = [&](int upto) {   // this thread has nothing more to send for the pieces below `upto`
      for (; q < upto; q++) {
        if (e == hipSuccess) e = hipEventRecord(ctx_->piece_ev[q][t], ctx_->stage_stream[t % n_streams_]);
        std::lock_guard<std::mutex> l(mu_);
        rc_[t] = e;
        enq_[q]++;
        cv_.notify_all();
      }
    };
    for (size_t i = t; i < n_chunks && e == hipSuccess; i += T, turn++) {
      const size_t off = i * CH, len = std::min(CH, bytes_ - off);
      int upto = q;
      while (upto < (int)ends_.size() && ends_[upto] <= off) upto++;   // pieces that end at or before this chunk
      mark(upto);
      const int slot = (int)(turn % S);
      char* pin = ctx_->stage_pin + ((size_t)t * S + slot) * CH;
      if (turn >= (size_t)S) e = hipEventSynchronize(ctx_->stage_ev[t][slot]);
      if (e != hipSuccess) break;
      memcpy(pin, src_ + off, len);
      e = hipMemcpyAsync(dst_ + off, pin, len, hipMemcpyHostToDevice, ctx_->stage_stream[t % n_streams_]);
      if (e == hipSuccess) e = hipEventRecord(ctx_->stage_ev[t][slot], ctx_->stage_stream[t % n_streams_]);
    }
    mark((int)ends_.size());   // on an error too: nobody may wait for ever
    if (e == hipSuccess) e = hipStreamSynchronize(ctx_->stage_stream[t % n_streams_]);
    std::lock_guard<std::mutex> l(mu_);
    rc_[t] = e;
    const float ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0_).count();
    ms_ = std::max(ms_, ms);
  }
  msm_ctx* ctx_;
  char* dst_;
  const char* src_;
  size_t bytes_;
  std::vector<size_t> ends_;   // byte offset where piece q ends (multiples of the chunk size, the last = bytes)
  std::vector<int> enq_;
  hipError_t rc_[T];
  std::vector<std::thread> th_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::chrono::steady_clock::time_point t0_;
  float ms_ = 0;
  long long n_streams_ = 2;   // copy streams the staging threads queue their chunks on (measured: 1, 2, 4 alike; 2 steadiest)
};

// ---- msm_pipeline.hip -------------------------------------------------------------------------------------------
int window_sums_impl(msm_ctx* ctx, const void* scalars, uint64_t n, int on_device, const msm_opts* opts, int k_lo, int k_hi,
                     const Plan& pl, std::vector<uint32_t>& words, msm_result* stats, uint64_t p_off = 0);
int any_window_sums(msm_ctx* ctx, const void* scalars, uint64_t n, int on_device, const msm_opts* opts, int k_lo, int k_hi,
                    const Plan& pl, std::vector<uint32_t>& words, msm_result* stats, const void* const* placed = nullptr);

// runs f(child) for every child of a multi-device context on the fan-out threads, and f(ctx) on the calling thread;
// returns the first error code
template <class F>
int on_all_devices(msm_ctx* ctx, F f) {
  const int nch = (int)ctx->children.size();
  std::vector<int> rc(nch + 1, MSM_OK);
  // The fan-out jobs write into this frame: whatever the caller's own leg or a wait() throws, every job is waited for
  // before the frame unwinds (the first exception is re-raised afterwards).
  std::exception_ptr err;
  for (int i = 0; i < nch; i++) ctx->fan[i]->run([&, i] { rc[i + 1] = f(ctx->children[i]); });
  try { rc[0] = f(ctx); } catch (...) { err = std::current_exception(); }
  for (int i = 0; i < nch; i++) {
    try { ctx->fan[i]->wait(); } catch (...) { if (!err) err = std::current_exception(); }
  }
  if (err) std::rethrow_exception(err);
  for (int i = 0; i <= nch; i++)
    if (rc[i] != MSM_OK) {
      if (i > 0) ctx->err = ctx->children[i - 1]->err;
      return rc[i];
    }
  return MSM_OK;
}

// ---- msm_gen.hip ------------------------------------------------------------------------------------------------
}  // namespace msmi
namespace msm_gen {
int generate_scalars(msm_ctx* ctx, uint64_t n, uint64_t seed, void* dev_dst, uint8_t* host_out);
int generate_points(msm_ctx* ctx, uint64_t n, uint64_t seed, uint8_t* a_out);
int generate_points_te(msm_ctx* ctx, uint64_t n, uint64_t seed, uint8_t* a_out);
}  // namespace msm_gen
