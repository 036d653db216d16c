// Host side of libmsm_hip.so: context, workspace, the kernel pipeline of one MSM and the C ABI
// declared in include/msm_hip.h.  Orchestration follows `createMsm().msm`
// (reference src/msm-batched-affine.ts:69-340); the per-thread SPMD phases separated by
// `barrier()` there become kernel launches on one HIP stream here.
#include "kernel_inst.h"   // curve-templated kernels: extern templates, defined in kernels_curve.hip per curve
#include "sort_kernels.h"
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

using namespace msm;

// Experiment knobs (window-group count, round geometry, ...) are environment variables ONLY in builds made with
// -DMSM_TUNING (tools/policy_sweep.sh, make ab EXTRA=-DMSM_TUNING); the product never reads the environment.
#ifdef MSM_TUNING
#define MSM_KNOB(var, name, lo) do { if (const char* _e = getenv(name)) var = std::max<long long>((lo), atoll(_e)); } while (0)
#define MSM_KNOB_SET(name) (getenv(name) != nullptr)
#else
#define MSM_KNOB(var, name, lo) do { } while (0)
#define MSM_KNOB_SET(name) false
#endif

namespace {

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
};

// non-HIP failure raised inside the pipeline (mapped to its error code at the ABI boundary)
struct MsmFail {
  int code;
  std::string msg;
};

#define HIPCHK(x)                                   \
  do {                                              \
    hipError_t _e = (x);                            \
    if (_e != hipSuccess) throw HipFail{_e, #x, __LINE__}; \
  } while (0)

inline uint32_t ceil_log2_u64(uint64_t n) {
  uint32_t r = 0;
  while ((1ull << r) < n) r++;
  return r;
}

}  // namespace

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

struct msm_ctx {
  std::unique_ptr<HelperThread> helper;
  int curve = 0;
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[12] = {};
  std::string err;
  int n_cu = 256;

  // resident points: `rows` / `n_points` are the CURRENT point set; the others wait in `sets` (msm_pointset_*)
  DevBuf rows;
  uint64_t n_points = 0;
  struct PointSet {
    DevBuf rows;
    uint64_t n = 0;
    bool live = false;
  };
  std::vector<PointSet> sets = std::vector<PointSet>(1);   // slot 0 = the default set
  int cur_set = 0;
  std::vector<void*> allocs;        // device buffers handed out by msm_device_alloc
  // multi-device context (msm_ctx_create_multi): this context drives devices[0], one child context per further device
  std::vector<msm_ctx*> children;
  std::vector<std::unique_ptr<HelperThread>> fan;   // one host thread per child for the window-shard fan-out

  // staging / misc buffers shared by all window groups
  DevBuf scal, errflag, misc;
  uint32_t* h_info = nullptr;      // pinned
  // host -> device staging of big pageable buffers (upload_staged): pinned chunks, a copy stream and an event per chunk slot
  static constexpr int STAGE_THREADS = 4, STAGE_SLOTS = 2;
  static constexpr size_t STAGE_CHUNK = (size_t)16 << 20;
  char* stage_pin = nullptr;
  bool staging_ready = false;      // pinned slots, copy streams and events all exist (ensure_staging)
  hipStream_t stage_stream[STAGE_THREADS] = {};
  hipEvent_t stage_ev[STAGE_THREADS][STAGE_SLOTS + 1] = {};
  static constexpr int MAX_PIECES = 4;   // ranges of the points a host-scalar MSM is pipelined over (PieceUpload)
  hipEvent_t piece_ev[MAX_PIECES][STAGE_THREADS] = {};
  uint64_t ws_budget = 0;          // bytes the per-group workspaces may take in total
  uint64_t ws_limit = 0;           // msm_set_workspace_limit: the caller's cap on ws_budget (0 = automatic)

  // per-group workspace: two of them, each with its own stream, so that the memory-bound sort of one
  // window group runs under the ALU-bound accumulation of the other
  struct Workspace {
    DevBuf dig, counts, cursor, tail_off, info, slots, block_hist, scan_partial, desc, columns2, rows_sum, bucket_proj, bufA, bufB,
        scratch, columns, partials, part, dig2, idx2, idx3, blk_tab2, slots2, oidx, rows1;
    hipStream_t stream = nullptr;
    hipEvent_t ev[8] = {};
    uint32_t* h_info = nullptr;   // pinned, 64 words
    uint32_t* h_part = nullptr;   // pinned, window sums read-back
    DevBuf* all[25] = {&dig, &counts, &cursor, &tail_off, &info, &slots, &block_hist, &scan_partial, &desc, &columns2, &rows_sum,
                       &bucket_proj, &bufA, &bufB, &scratch, &columns, &partials, &part, &dig2, &idx2, &idx3, &blk_tab2, &slots2, &oidx, &rows1};
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

  void ensure(DevBuf& b, size_t bytes) {
    if (bytes <= b.cap) return;
    if (b.p) HIPCHK(hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    size_t want = bytes + bytes / 16 + 256;
    const hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
      (void)hipGetLastError();   // the failure must not stay behind as this thread's "last error": the kernel-launch checks read it
      b.p = nullptr;
      throw HipFail{e, "hipMalloc(&b.p, want)", __LINE__};
    }
    b.cap = want;
  }
  void release(DevBuf& b) {
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

#include "msm_gen.h"

namespace {

int fail(msm_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  return code;
}

int fail_hip(msm_ctx* ctx, const HipFail& f) {
  return fail(ctx, MSM_ERR_HIP, "HIP error %d (%s) at msm_api.hip:%d: %s", (int)f.e, hipGetErrorString(f.e), f.line, f.what);
}

// every extern "C" entry point ends its try block with this: no C++ exception crosses the C ABI
#define MSM_CATCH_ALL(ctx)                                                                                  \
  catch (const HipFail& f) { return fail_hip(ctx, f); }                                                     \
  catch (const MsmFail& f) { return fail(ctx, f.code, "%s", f.msg.c_str()); }                               \
  catch (const std::bad_alloc&) { return fail(ctx, MSM_ERR_INTERNAL, "host memory allocation failed"); }    \
  catch (const std::exception& e) { return fail(ctx, MSM_ERR_INTERNAL, "unexpected exception: %s", e.what()); } \
  catch (...) { return fail(ctx, MSM_ERR_INTERNAL, "unexpected exception"); }

// GPU-tuned window size (the reference's table, src/msm-common.ts:25-41, was tuned for 16 CPU threads and copies
// points).  Weierstrass + GLV (b + 1 = 127 or 128 bits): measured over N = 2^4 .. 2^28 (tools/small_sizes.py,
// tools/knob_matrix.py), c = 16 (K = 8, no degenerate top window, one window's counters fit the LDS) wins from N = 2^12 to
// 2^25 -- by 20 % over c = 13 at 2^14 .. 2^18, where a smaller window mostly buys more rounds of fixed latency -- and c = 8
// below.  From 2^26 points the big windows (K = 7 / 6: a quarter fewer pair additions) win with the three-pass sort, the
// chunk-ordered round 1 and, since round 4, the last descriptor rounds left to k_bucket_finish: 2^26 c = 22 149.2 against
// 155.6 ms (2^25: 83.0 against 80.4, so c = 16 stays there), 2^27 c = 21 296 against 300 (c = 22) and 314 (c = 16), 2^28
// c = 22 584 against 643 (profiles/r03_experiments.txt items 5 and 12, profiles/r04_experiments.txt item 8).
// Twisted Edwards (b + 1 = 252, no inversion per round): a cost model over the window sizes whose top window is not
// degenerate, ~9 multiplications per pair addition against ~64 per bucket; its picks are within 3 % of the best
// measured ones.
int pick_window(bool te, uint64_t n, int glv_max_bits) {
  // measured (profiles/r03_experiments.txt item 5): the mean bucket of the big windows wants ~128 entries
  // b = 126 (BLS12-377 after GLV): 127 = 6 * 21 + 1, so 21-bit windows fold the carry bit into the sixth window (make_plan) --
  // K = 6 with half the buckets of the 22-bit plan: 2^24 43.6 against 44.3 (c = 16), 2^25 79.7 against 80.7, 2^26 148.9 against
  // 151.5 (c = 22) and 158.3 (c = 16), 2^27 302.8 against 311.5 (c = 22) on one box (profiles/r04_experiments.txt item 13).
  // b = 127 (BLS12-381, Pallas): 128 = 5 * 22 + 18, the 22-bit plan has six whole windows from 2^26 points up.
  // (127 = 7 * 18 + 1 folds too: seven 18-bit windows win between 2^22 and 2^23 -- 11.7 against 12.1 ms -- and are level at 2^23)
  if (!te && glv_max_bits == 126)
    return n >= (1ull << 24) ? 21 : (n >= (1ull << 22) && n < (1ull << 23)) ? 18 : n >= 4096 ? 16 : 8;
  if (!te) return n >= (1ull << 26) ? 22 : n >= 4096 ? 16 : 8;
  static const int cand_te[] = {4, 6, 7, 9, 12, 14, 16};
  const int b1 = 252;
  int best = 4;
  double best_cost = 1e300;
  for (int i = 0; i < 7; i++) {
    int c = cand_te[i];
    int K = (b1 + c - 1) / c;
    double cost = (double)n * K * 8.0 + (double)K * (double)(1u << (c - 1)) * 64.0;
    if (cost < best_cost) { best_cost = cost; best = c; }
  }
  return best;
}

struct Plan {
  int c, K, L_log;       // c: bits a window advances by (the weight of window k is 2^(c k)); L_log: bits of a bucket index
  int bits = 0;          // b + 1: scalar bits the windows cover (the top window holds bits - (K - 1) c of them)
  bool fold = false;     // the top window is c + 1 bits wide (see make_plan)
  uint32_t L;            // buckets per window = 2^L_log
  bool no_glv;
  bool strict = false;   // msm_opts.strict: scalars >= q fail the call instead of being reduced
  bool lone = false;   // one window, one group: nothing else shares the GPU (see round_geom)
  bool merged = false; // a full MSM (msm_run): a window group may hand back sum_k 2^(c (k - k_first)) P_k in the slot of its
                       // first window instead of one P_k per slot (reduce_buckets); msm_window_sums never sets it
};

int make_plan(const msm_ctx* ctx, uint64_t n, const msm_opts* opts, Plan& pl) {
  const bool te = ctx && ctx->is_te();
  const int glv_bits = curve_info(ctx ? ctx->curve : MSM_CURVE_BLS12_377_G1).glv_max_bits;
  int c = (opts && opts->c > 0) ? opts->c : pick_window(te, n, (opts && opts->no_glv) ? 0 : glv_bits);
  if (c < 2 || c > 24) return MSM_ERR_ARG;
  // b = Scalar.maxBits (126 after GLV, src/wasm/glv.ts:216-226) or Scalar.sizeInBits (251, src/msm-basic.ts:56)
  // b = Scalar.maxBits after GLV (src/wasm/glv.ts:216-226), or the bit length of q without it (src/msm-basic.ts:56)
  pl.no_glv = !te && opts && opts->no_glv;
  pl.strict = opts && opts->strict;
  const int b = te ? 251 : pl.no_glv ? curve_info(ctx ? ctx->curve : MSM_CURVE_BLS12_377_G1).q_bits : glv_bits;
  if (pl.no_glv && c < 4) return MSM_ERR_ARG;   // keeps K <= 64
  pl.c = c;
  pl.K = (b + 1 + c - 1) / c;  // K = ceil((b + 1) / c), src/msm-batched-affine.ts:90, src/msm-basic.ts:59
  pl.bits = b + 1;
  pl.L_log = c - 1;
  // K c >= b + 1 keeps the carry of the signed recoding inside the top window (src/msm-batched-affine.ts:183-193).  When
  // b + 1 = (K - 1) c + 1 -- BLS12-377 after GLV: 127 = 7 * 18 + 1 = 6 * 21 + 1 -- that top window holds the carry bit and
  // nothing else: every entry with a carry lands in its bucket 1, a full window's worth of tree work for one bit (c = 21 at
  // 2^26: seven windows in 155 ms, six of 22 bits in 149).  The big-window plans fold that bit into the window below instead:
  // K - 1 windows, the top one c + 1 bits wide and not recoded (its magnitude is at most 2^c, it cannot carry out), the
  // others as before.  Every window gets 2^c buckets (the lower ones fill the lower half); window k still weighs 2^(c k),
  // so sums, shards and msm_combine are unchanged.  Only for c >= 18: those windows sort with per-window effective bits
  // (WinSplit) already.
  pl.fold = !te && c >= 18 && pl.K > 1 && (b + 1) - (pl.K - 1) * c == 1;
  if (pl.fold) {
    pl.K -= 1;
    pl.L_log = c;
  }
  pl.L = 1u << pl.L_log;
  return MSM_OK;
}

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

// gather: round 1 (random row reads: wants two waves per SIMD to cover the latency).  The other rounds read
// coalesced, prefetched planes; a lone wave already gets ~88 % of a SIMD's issue rate, and every lane pays one field
// inversion (~19 pair additions' worth) per round, so small rounds run better on half as many lanes with twice the
// steps (2^18: 2.62 -> 2.45 ms; neutral at 2^20, 1 % at 2^22; round 1 at 2^22 would lose 60 %).
// lone: the launch has the GPU to itself (a window group of one window with no second group beside it -- the
// 8-GPU shard).  All waves of one resident batch then move through the memory-heavy forward sweep and the ALU-heavy
// backward sweep in step; four batches of 128 steps instead of one of 512 stagger the phases (2^26, one window:
// 31.5 -> 29.2 ms).  With two groups on two streams the other stream already fills the gaps and 512 is better.
RoundGeom round_geom(const msm_ctx* ctx, uint64_t n_out, bool gather = false, bool lone = false) {
  uint64_t target = (uint64_t)ctx->n_cu * 4 * MSM_BA_WAVES * 64;  // as many lanes as the kernel's launch bounds keep resident
  // steps at full width below which a non-gather round runs on half as many lanes: every lane pays one inversion per
  // round (~13 pair additions' worth), and one wave per SIMD already gets 89 % of the multiplier's two-wave rate
  // (tools/ubench_mul2.hip).  Measured with the round-2 kernel: 2^20 4.05 -> 3.91 ms, 2^22 12.9 -> 12.4, neutral elsewhere.
  uint64_t half_below = 128;
  MSM_KNOB(half_below, "MSM_HALF_BELOW", 0);
  if (!gather && n_out < target * half_below) target /= 2;
  uint32_t max_steps = (lone && n_out >= target * 512) ? 128 : 512;
  MSM_KNOB(max_steps, "MSM_MAX_STEPS", 1);
  {
    long long tw = 0;
    MSM_KNOB(tw, "MSM_TARGET_WAVES", 1);
    if (tw) target = (uint64_t)ctx->n_cu * 4 * 64 * (uint64_t)tw;
  }
  uint64_t steps = (n_out + target - 1) / target;
  steps = std::max<uint64_t>(1, std::min<uint64_t>(steps, max_steps));
  uint64_t threads = (n_out + steps - 1) / steps;
  uint64_t grid = std::max<uint64_t>(1, (threads + 255) / 256);
  return RoundGeom{(uint32_t)steps, (uint32_t)grid, grid * 256};
}

void words_to_fe6(msm_host::Fe6& r, const uint32_t* w, int nw = 12) {   // nw packed words, zero-extended
  for (int i = 0; i < 6; i++)
    r.v[i] = (2 * i < nw ? (uint64_t)w[2 * i] : 0) | ((2 * i + 1 < nw ? (uint64_t)w[2 * i + 1] : 0) << 32);
}

void fe6_to_bytes(uint8_t* out, const msm_host::Fe6& a);
// element e of a plane buffer read back to the host (uint4 piece c of element e at word (c * cap + e) * 4) -> x || y in wire
// form (coordinate bytes of the curve; the identity as zeros)
void plane_element_to_wire(const msm_ctx* ctx, const uint32_t* planes, uint64_t cap, uint64_t e, uint8_t* out_xy);

void fe6_to_bytes(uint8_t* out, const msm_host::Fe6& a) {
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 8; j++) out[8 * i + j] = (uint8_t)(a.v[i] >> (8 * j));
}

// Window sums travel as 36 words (X, Y, Z: 12 packed words each, device Montgomery form, below 2p).  A projective point is a
// class of triples, so the host neither converts them on the way in nor on the way out: read as host Montgomery values the
// three words of a device point carry one common factor (2^6 for 13 limbs), which is the same point, and so is every sum the
// host forms of such points.  Only affine values need the exact radix (horner_to_affine divides the factor out with Z).
msm_host::Proj6 partial_to_host(const msm_ctx* ctx, const uint32_t* w) {
  const auto& F = ctx->hc.F;
  msm_host::Proj6 P;
  msm_host::Fe6* co[3] = {&P.X, &P.Y, &P.Z};
  for (int j = 0; j < 3; j++) {
    words_to_fe6(*co[j], w + 12 * j);
    while (msm_host::Field6::ge(*co[j], F.p)) F.sub_raw(*co[j], *co[j], F.p);
  }
  return P;
}

// host projective point -> the packed form the pipeline carries (canonical values; see above for the radix)
void host_to_partial(const msm_ctx*, const msm_host::Proj6& P, uint32_t* out36) {
  const msm_host::Fe6* co[3] = {&P.X, &P.Y, &P.Z};
  for (int j = 0; j < 3; j++)
    for (int q = 0; q < 6; q++) {
      out36[12 * j + 2 * q] = (uint32_t)co[j]->v[q];
      out36[12 * j + 2 * q + 1] = (uint32_t)(co[j]->v[q] >> 32);
    }
}

// Bucket reduction of kc windows of L buckets each: P_k = sum_l l * B_(k,l) (reduceBucketsColumnProjective + the partition
// sums, src/msm-batched-affine.ts:556-583, :312-319) -> h_partials_out, kc x 36 (32 on the Edwards path) words.  The bucket
// sums come as projective points from k_bucket_finish (`bucket_proj`) or as the first element of every bucket in the tree
// buffer (`fin`, `off_fin`).  Runs on w.stream, records w.ev[4] behind its last kernel and returns when the sums are on the host.
// stride: bits a window advances by (Plan::c); 0 = log2(L) + 1, the plain plan's c.
void reduce_buckets(msm_ctx* ctx, msm_ctx::Workspace& w, const uint4* fin, uint64_t fin_cap, const uint32_t* off_fin,
                    const uint32_t* bucket_proj, uint32_t L, int kc, uint32_t* h_partials_out, bool merged = false, int stride = 0) {
  hipStream_t s = w.stream;
  const bool te = ctx->is_te();
  const uint64_t nb = (uint64_t)kc * L;
  const int part_words = te ? 32 : 36;
  // buckets per lane: enough lanes to fill the chip, but never more than 16 buckets deep (2 additions each)
  // (millions of buckets -- the big windows -- go 32 deep: 2^26 at c = 22, with the two policies below, 150.6 -> 149.3 ms)
  uint32_t TC = 2;
  const uint32_t tc_cap = nb >= (1ull << 22) ? 32 : 16;
  while (TC < tc_cap && nb / TC > 65536) TC *= 2;
  MSM_KNOB(TC, "MSM_TC", 1);
  TC = std::min<uint32_t>(TC, L);
  uint32_t nchunks = (L + TC - 1) / TC;
  // bit-sliced weighting (Weierstrass path, enough chunks to matter, TC a power of two)
  uint32_t nbits = 0;
  while ((1u << nbits) < nchunks) nbits++;
  const bool bit_sliced = !te && nchunks >= 64 && (TC & (TC - 1)) == 0;
  ctx->ensure(w.columns, (size_t)kc * nchunks * 4 * NL * 4);
  ctx->ensure(w.partials, (size_t)kc * 36 * 4);
  {
    uint32_t threads = nchunks * (uint32_t)kc;
    if (te) {
      hipLaunchKernelGGL(te::k_te_bucket_reduce, dim3((threads + 63) / 64), dim3(64), 0, s, (uint32_t*)w.columns.p, fin, fin_cap,
                         off_fin, bucket_proj, L, TC, nchunks, (uint32_t)kc);
      hipLaunchKernelGGL(te::k_te_window_sum, dim3(kc), dim3(te::TE_WS_THREADS), 0, s, (uint32_t*)w.partials.p,
                         (const uint32_t*)w.columns.p, nchunks);
    } else if (bit_sliced) {
      ctx->ensure(w.rows_sum, (size_t)kc * nchunks * 3 * NL * 4);
      W_LAUNCH(ctx, k_bucket_reduce, dim3((threads + 63) / 64), dim3(64), 0, s, (uint32_t*)w.columns.p,
                         (uint32_t*)w.rows_sum.p, fin, fin_cap, off_fin, bucket_proj, L, TC, nchunks, (uint32_t)kc);
      // first stage: one wave per 512 elements (8 per lane; at 2^20 that is about one wave per SIMD) -- the masked sums
      // have half as many elements as the triangle sum and get half as many blocks; second stage: one wave per
      // (window, bit) over the block sums (unused block slots stay zero = the identity)
      const uint32_t nblk = std::max<uint32_t>(2, nchunks / (8 * BT_THREADS));
      const size_t c2_bytes = (size_t)kc * (nbits + 1) * nblk * 3 * NL * 4;
      ctx->ensure(w.columns2, c2_bytes);
      ctx->ensure(w.partials, (size_t)kc * (nbits + 1) * 36 * 4);
      HIPCHK(hipMemsetAsync(w.columns2.p, 0, c2_bytes, s));
      W_LAUNCH(ctx, k_bit_tree, dim3(nbits * (nblk / 2) + nblk, 1, kc), dim3(BT_THREADS), 0, s, (uint32_t*)w.columns2.p,
                         (const uint32_t*)w.rows_sum.p, (const uint32_t*)w.columns.p, nchunks, nbits, 1, 0, nblk);
      W_LAUNCH(ctx, k_bit_tree, dim3(1, nbits + 1, kc), dim3(BT_THREADS), 0, s, (uint32_t*)w.partials.p,
                         (const uint32_t*)w.columns2.p, (const uint32_t*)nullptr, nblk, nbits, 0, 1, 1u);
    } else {
      W_LAUNCH(ctx, k_bucket_reduce, dim3((threads + 63) / 64), dim3(64), 0, s, (uint32_t*)w.columns.p, (uint32_t*)nullptr,
                         fin, fin_cap, off_fin, bucket_proj, L, TC, nchunks, (uint32_t)kc);
      if (nchunks > 2 * WS_THREADS) {
        // two-stage: blocks of 2 columns per lane, then one block per window over the block sums
        const uint32_t per_block = 2 * WS_THREADS;
        const uint32_t nblk = (nchunks + per_block - 1) / per_block;
        ctx->ensure(w.columns2, (size_t)kc * nblk * 3 * NL * 4);
        W_LAUNCH(ctx, k_column_tree, dim3(nblk, kc), dim3(WS_THREADS), 0, s, (uint32_t*)w.columns2.p,
                           (const uint32_t*)w.columns.p, nchunks, per_block);
        W_LAUNCH(ctx, k_window_sum, dim3(kc), dim3(WS_THREADS), 0, s, (uint32_t*)w.partials.p,
                           (const uint32_t*)w.columns2.p, nblk);
      } else {
        W_LAUNCH(ctx, k_window_sum, dim3(kc), dim3(WS_THREADS), 0, s, (uint32_t*)w.partials.p,
                           (const uint32_t*)w.columns.p, nchunks);
      }
    }
  }
  if (bit_sliced) {
    // read the (nbits + 1) sums per window back and finish P_k = tri + TC * sum_b 2^b S_b on the host
    HIPCHK(hipMemcpyAsync(w.h_part, w.partials.p, (size_t)kc * (nbits + 1) * 36 * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipEventRecord(w.ev[4], s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    const auto& C = ctx->hc;
    if (merged) {
      // The caller only wants S_g = sum_kk 2^(c kk) P_kk of the whole group (a full MSM on this device: the Horner step over
      // the windows follows anyway).  One double-and-add pass over the c kc bit positions then does both jobs -- inside window
      // kk the sum S_b sits at bit log2(TC) + b, the triangle sum at bit 0 -- with c kc doublings instead of (c - 1) kc for
      // the windows plus c (kc - 1) for their combination.  S_g goes into the slot of the group's first window, the identity
      // into the others: sum_k 2^(c k) (slot k) is the same group element as with one P_k per slot.
      // With a folded top window (Plan::fold) a window has one bit position more than it advances by: position `stride` of
      // window kk coincides with position 0 of window kk + 1, so the pass walks GLOBAL bit positions and adds what every
      // window has there.
      uint32_t lt = 0, cbits = 1;
      while ((1u << lt) < TC) lt++;
      while ((1u << (cbits - 1)) < L) cbits++;
      const int adv = stride ? stride : (int)cbits;
      msm_host::Proj6 acc = C.zero();
      for (int gpos = (kc - 1) * adv + (int)cbits - 1; gpos >= 0; gpos--) {
        acc = C.dbl(acc);
        for (int kk = kc - 1; kk >= 0; kk--) {
          const int pos = gpos - kk * adv;
          if (pos < 0 || pos >= (int)cbits) continue;
          const uint32_t* base = w.h_part + (size_t)kk * (nbits + 1) * 36;
          const int b = pos - (int)lt;
          if (b >= 0 && b < (int)nbits) acc = C.add(acc, partial_to_host(ctx, base + (size_t)b * 36));
          if (pos == 0) acc = C.add(acc, partial_to_host(ctx, base + (size_t)nbits * 36));
        }
      }
      memset(h_partials_out, 0, (size_t)kc * 36 * 4);
      host_to_partial(ctx, acc, h_partials_out);
      return;
    }
    for (int kk = 0; kk < kc; kk++) {
      const uint32_t* base = w.h_part + (size_t)kk * (nbits + 1) * 36;
      msm_host::Proj6 acc = C.zero();
      for (int b = (int)nbits - 1; b >= 0; b--) {
        acc = C.dbl(acc);
        acc = C.add(acc, partial_to_host(ctx, base + (size_t)b * 36));
      }
      for (uint32_t t = TC; t > 1; t >>= 1) acc = C.dbl(acc);   // TC is a power of two on this path
      acc = C.add(acc, partial_to_host(ctx, base + (size_t)nbits * 36));
      host_to_partial(ctx, acc, h_partials_out + (size_t)kk * 36);
    }
    return;
  }
  HIPCHK(hipMemcpyAsync(w.h_part, w.partials.p, (size_t)kc * part_words * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipEventRecord(w.ev[4], s));
  HIPCHK(hipStreamSynchronize(s));
  HIPCHK(hipGetLastError());
  memcpy(h_partials_out, w.h_part, (size_t)kc * part_words * 4);
}


// Two window groups that run side by side on the two streams meet here between their sorts and their trees: neither tree
// starts before BOTH sorts are done.  A tree kernel holds 2 x 226 of a SIMD's 512 VGPRs for the 512 steps of a workgroup, so a
// sort still running when the other group's tree arrives finds no room for its workgroups and takes three to four times as
// long (profiles/r04_experiments.txt items 1 and 8).  Groups with equal work (c = 16: 4 + 4 windows) reach this point together
// anyway; groups with unequal sorts (c = 22: the top window holds 17 bits) do not.
// Host side: both workers arrive after queueing their sorts and recording their event, then each makes its stream wait for the
// other's event.  A worker that fails releases its partner (abort).
class PairSync {
 public:
  // returns false if the partner will never arrive (it failed, or there is none)
  bool arrive_and_wait(int pair, int n_pairs_expected) {
    std::unique_lock<std::mutex> l(mu_);
    if ((int)count_.size() < n_pairs_expected) count_.resize(n_pairs_expected, 0);
    count_[pair]++;
    cv_.notify_all();
    cv_.wait(l, [&] { return count_[pair] >= 2 || aborted_; });
    return count_[pair] >= 2;
  }
  void abort() {
    std::lock_guard<std::mutex> l(mu_);
    aborted_ = true;
    cv_.notify_all();
  }

 private:
  std::mutex mu_;
  std::condition_variable cv_;
  std::vector<int> count_;
  bool aborted_ = false;
};

// Partition sums P_k for windows [k_lo, k_hi) over the points [p_lo, p_lo + n) -> w.h_part[(k - k_lo) * 36 ...]
// scalars: device pointer, n x 8 words.
// before_tree: called once with the group's stream when everything up to the scatter has been queued (see PairSync).
void run_window_group(msm_ctx* ctx, msm_ctx::Workspace& w, const uint32_t* d_scalars_all, uint64_t p_lo, uint64_t n, const Plan& pl,
                      int k_lo, int k_hi, uint32_t* h_partials_out, GroupStats& st, uint64_t p_off = 0,
                      const std::function<void(hipStream_t)>* before_tree = nullptr) {
  hipStream_t s = w.stream;
  const uint32_t* d_scalars = d_scalars_all + p_lo * 8;   // scalar i of the call <-> resident point p_off + i
  p_lo += p_off;
  const int kc = k_hi - k_lo;
  const bool lone = pl.lone;
  const uint32_t L = pl.L;
  const uint64_t nb = (uint64_t)kc * L;
  const bool te = ctx->is_te();
  const uint64_t two_n = te ? n : 2 * n;   // entries per window: both GLV halves, or the plain scalar
  const uint64_t n_entries = (uint64_t)kc * two_n;
  const size_t elem_bytes = te ? 128 : 96;  // tree node: extended (X, Y, Z, T) x 32 B, or affine (x, y) x 48 B
  const int part_words = te ? 32 : 36;

  // padding granule G = 2^g: about 1/8 of the mean bucket population (pads cost G/2 slots per bucket; what is
  // left after g regular rounds, ~8 elements per bucket, is finished without inversions by k_bucket_finish)
  // Bigger buckets leave more: pads are identity pairs that occupy a lane for nothing (mean / (2 * left) of all slots), and
  // k_bucket_finish is cheap next to them.  Measured (profiles/r04_experiments.txt items 6 and 14, best `left` per mean bucket):
  // 64 entries 8 with 16-bit windows (2^20: 3.89 against 3.93 ms) but 16 with the big ones (2^25: 81.5 -> 79.9);
  // 128 .. 256: 16 (2^21 6.92 -> 6.88, 2^22 12.45 -> 12.17, Ed-377 2^20 2.64 -> 2.59, 2^26 145.7 -> 142.5, 2^27 304.9 -> 301.8);
  // from 512: 32 (2^23 22.65 -> 22.18, 2^26 at c = 16 152.7 -> 152.1); 32 entries: 8 (2^24: 16 costs 8 %).
  uint64_t mean = std::max<uint64_t>(1, two_n / (pl.fold ? L / 2 : L));   // (a folded plan's lower windows fill half their buckets)
  uint64_t per_bucket_left = mean >= 512 ? 32 : (mean >= 128 || (mean >= 64 && pl.c >= 18)) ? 16 : 8;
  MSM_KNOB(per_bucket_left, "MSM_PBL", 1);
  uint32_t logG = 1;
  while (logG < 10 && (1ull << (logG + 1)) * per_bucket_left <= mean) logG++;

  ctx->ensure(w.dig, n_entries * 4);
  ctx->ensure(w.counts, nb * 4);
  ctx->ensure(w.cursor, nb * 4);
  ctx->ensure(w.tail_off, (size_t)34 * (nb + 1) * 4);
  ctx->ensure(w.info, 64 * 4);

  // Sort path: LDS-privatised histogram / ranking, every pass staged through the LDS so that a wave store is a full segment
  // (sort_kernels.h; a direct scatter sends each 4-byte payload to a line of its own: round 1 wrote 7.7x the algorithmic bytes).
  //   one level  : a window's counters fit the LDS (c <= 16) and the input is small
  //   two passes : c <= 16, big inputs -- 2^(c-8) coarse bins x 128 buckets
  //   three passes: c > 16 (up to c = 24, the largest window make_plan accepts) -- coarse bins x mid bins x 128 (256) buckets
  const int cbits = pl.L_log;   // bits of a bucket index
  const bool fits_lds = (size_t)L * 4 <= 128 * 1024;
  long long want_radix = (fits_lds && cbits > (int)RX_FINE_BITS && two_n >= (1ull << 22)) ? 1 : 0;   // measured: wins from N = 2^21 up
  MSM_KNOB(want_radix, "MSM_RADIX", 0);
  const bool radix = fits_lds && want_radix && cbits > (int)RX_FINE_BITS && cbits - (int)RX_FINE_BITS <= 8;
  const bool one_level = fits_lds && !radix;
  const bool three_pass = !fits_lds;
  const uint32_t fb = three_pass ? (cbits >= 23 ? 8u : 7u) : RX_FINE_BITS;   // bucket bits sorted by the last pass (full windows)
  const uint32_t shift = radix ? (uint32_t)cbits - fb : 0;                  // two passes: log2 of the coarse bins
  const uint32_t Hn = 1u << shift;
  const uint32_t Lp = three_pass ? 1u << ((uint32_t)cbits - fb) : Hn;        // fine windows (blocks of the last pass) per window
  const uint32_t V = (uint32_t)kc * Lp;
  WinSplit ws{};
  if (three_pass) {
    if (kc > 16) throw MsmFail{MSM_ERR_INTERNAL, "more than 16 windows in a group of a window size above 16"};
    const int lp_log = cbits - (int)fb;
    for (int kk = 0; kk < kc; kk++) {
      // bits the digits of this window really have: the top window of a scalar is usually short (sort_kernels.h, WinSplit)
      // (a window below the top one holds signed digits of magnitude <= 2^(c - 1); the top one what is left of the scalar)
      const bool top = k_lo + kk == pl.K - 1;
      const int eff = std::max(1, top ? std::min(cbits, pl.bits - (k_lo + kk) * pl.c) : std::min(cbits, pl.c - 1));
      const int fbk = std::max(0, eff - lp_log);
      const int hi = eff - fbk;
      ws.fb[kk] = (uint8_t)fbk;
      ws.ab[kk] = (uint8_t)std::min(8, hi);
      ws.mb[kk] = (uint8_t)(hi - ws.ab[kk]);
    }
  } else if (radix) {
    if (kc > 16) throw MsmFail{MSM_ERR_INTERNAL, "more than 16 windows in a radix-split group"};
    for (int kk = 0; kk < kc; kk++) { ws.ab[kk] = (uint8_t)shift; ws.fb[kk] = (uint8_t)fb; }
  }
  uint32_t sortB = 1;
  uint64_t chunk = two_n;
  {
    // big inputs: finer slices also keep the round-1 gathers of neighbouring lanes inside one Infinity-Cache-sized
    // range of point rows (measured: 193 -> 183 ms at 2^26); small inputs: fewer, larger blocks (less fixed cost)
    uint64_t mult = two_n >= (1ull << 27) ? 8 : two_n >= (1ull << 24) ? 4 : 2;   // measured 2^21 .. 2^26
    if (three_pass) mult = std::min<uint64_t>(mult, 4);   // the chunk-ordered round 1 makes its own locality: fewer, larger slices
    MSM_KNOB(mult, "MSM_SORTB_MULT", 1);
    uint64_t want = std::max<uint64_t>(1, (mult * ctx->n_cu + kc - 1) / kc);
    uint64_t maxb = std::max<uint64_t>(1, two_n / 8192);
    sortB = (uint32_t)std::min<uint64_t>(want, maxb);
    chunk = (two_n + sortB - 1) / sortB;
    ctx->ensure(w.block_hist, (size_t)kc * sortB * (three_pass ? Lp : L) * 4 + 64);
  }
  const uint32_t* d_v2start = nullptr;   // three passes: starts of the fine windows in the record arrays
  const uint32_t* d_rec_dig = nullptr;   // records read by the last pass
  const uint32_t* d_rec_idx = nullptr;

  HIPCHK(hipEventRecord(w.ev[0], s));
  {
    uint32_t grid = (uint32_t)((n + 255) / 256);
    if (te)
      hipLaunchKernelGGL(te::k_te_digits, dim3(grid), dim3(256), 0, s, (uint32_t*)w.dig.p, d_scalars, (uint32_t)n, pl.c, pl.K,
                         k_lo, kc, pl.strict ? 1 : 0, (uint32_t*)ctx->errflag.p);
    else
      W_LAUNCH(ctx, k_digits, dim3(grid), dim3(256), 0, s, (uint32_t*)w.dig.p, d_scalars, (uint32_t)n, pl.c, pl.K, k_lo, kc,
                         (pl.no_glv ? 0 : 1) | (pl.fold ? 2 : 0), pl.strict ? 1 : 0, (uint32_t*)ctx->errflag.p);
  }
  HIPCHK(hipEventRecord(w.ev[1], s));
  if (!three_pass) {
    hipLaunchKernelGGL(k_hist, dim3(sortB, kc), dim3(SORT_THREADS), (size_t)L * 4, s, (uint32_t*)w.block_hist.p,
                       (const uint32_t*)w.dig.p, two_n, chunk, L, ws, 0u, 0u);
    hipLaunchKernelGGL(k_colscan, dim3((uint32_t)((nb + 255) / 256)), dim3(256), 0, s, (uint32_t*)w.block_hist.p,
                       (uint32_t*)w.counts.p, sortB, L, (uint32_t)kc);
  } else {
    // totals of the fine windows -> their starts (= the starts of the mid and coarse bins too); pass A; pass M; bucket sizes
    const size_t n_off = (size_t)kc * sortB * 256;
    ctx->ensure(w.part, ((size_t)2 * V + 2 + n_off) * 4);
    uint32_t* d_v2tot = (uint32_t*)w.part.p;
    uint32_t* d_vs = d_v2tot + V;
    uint32_t* d_blk_off = d_vs + V + 1;
    ctx->ensure(w.dig2, n_entries * 4);
    ctx->ensure(w.idx2, n_entries * 4);
    ctx->ensure(w.idx3, n_entries * 4);
    hipLaunchKernelGGL(k_hist, dim3(sortB, kc), dim3(SORT_THREADS), (size_t)Lp * 4, s, (uint32_t*)w.block_hist.p,
                       (const uint32_t*)w.dig.p, two_n, chunk, Lp, ws, 1u, 0u);
    hipLaunchKernelGGL(k_colscan, dim3((V + 255) / 256), dim3(256), 0, s, (uint32_t*)w.block_hist.p, d_v2tot, sortB, Lp,
                       (uint32_t)kc);
    hipLaunchKernelGGL(k_coarse_offsets3, dim3((uint32_t)((n_off + 255) / 256)), dim3(256), 0, s, d_blk_off,
                       (const uint32_t*)w.block_hist.p, sortB, Lp, (uint32_t)kc, ws);
    hipLaunchKernelGGL(k_vscan, dim3(1), dim3(SCAN_THREADS), 0, s, d_vs, (const uint32_t*)d_v2tot, V);
    hipLaunchKernelGGL(k_radix_coarse, dim3(sortB, kc), dim3(RX_THREADS), 0, s, (uint32_t*)w.dig2.p, (uint32_t*)w.idx2.p,
                       (const uint32_t*)d_vs, (const uint32_t*)d_blk_off, (const uint32_t*)w.dig.p, two_n, chunk, Lp, 256u, ws);
    // the digits are dead now: the second record array reuses their buffer
    hipLaunchKernelGGL(k_radix_mid, dim3(256, kc), dim3(RXB_THREADS), 0, s, (uint32_t*)w.dig.p, (uint32_t*)w.idx3.p,
                       (const uint32_t*)d_vs, (const uint32_t*)w.dig2.p, (const uint32_t*)w.idx2.p, Lp, ws);
    HIPCHK(hipMemsetAsync(w.counts.p, 0, nb * 4, s));
    hipLaunchKernelGGL(k_fine_hist, dim3(V), dim3(256), 0, s, (uint32_t*)w.counts.p, (const uint32_t*)d_vs,
                       (const uint32_t*)w.dig.p, Lp, L, ws);
    d_v2start = d_vs;
    d_rec_dig = (const uint32_t*)w.dig.p;
    d_rec_idx = (const uint32_t*)w.idx3.p;
  }
  int RT = 0;
  uint64_t total_slots = 0;
  uint32_t max_bucket = 0;
  {
    // largest bucket -> number of tail rounds RT, then the multi-block scan of RT + 2 quantities.  The scan kernels take RT
    // from the device (pscan_nq), so ONE read-back behind them brings the largest bucket and the totals together.
    HIPCHK(hipMemsetAsync(w.info.p, 0, 64 * 4, s));
    hipLaunchKernelGGL(k_bucket_max, dim3((uint32_t)std::min<uint64_t>(1024, (nb + 255) / 256)), dim3(256), 0, s,
                       (const uint32_t*)w.counts.p, (uint32_t)nb, (uint32_t*)w.info.p);
    const uint32_t nblocks = (uint32_t)((nb + PS_SPAN - 1) / PS_SPAN);
    ctx->ensure(w.scan_partial, (size_t)PS_MAX_NQ * nblocks * 4);
    hipLaunchKernelGGL(k_pscan_partial, dim3(nblocks), dim3(PS_BLOCK), 0, s, (const uint32_t*)w.counts.p, (uint32_t)nb, logG,
                       (const uint32_t*)w.info.p, (uint32_t*)w.scan_partial.p, nblocks);
    hipLaunchKernelGGL(k_pscan_top, dim3(1), dim3(SCAN_THREADS), 0, s, (uint32_t*)w.scan_partial.p, nblocks, logG,
                       (uint32_t*)w.info.p);
    hipLaunchKernelGGL(k_pscan_final, dim3(nblocks), dim3(PS_BLOCK), 0, s, (const uint32_t*)w.counts.p, (uint32_t)nb, logG,
                       (const uint32_t*)w.scan_partial.p, nblocks, (uint32_t*)w.cursor.p, (uint32_t*)w.tail_off.p,
                       (const uint32_t*)w.info.p);
    HIPCHK(hipMemcpyAsync(w.h_info, w.info.p, 64 * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    total_slots = w.h_info[0];
    max_bucket = w.h_info[1];
    st.n_pairs_algo += (uint64_t)w.h_info[INFO_ALGO_PAIRS] | ((uint64_t)w.h_info[INFO_ALGO_PAIRS + 1] << 32);
    const uint32_t capmax = (max_bucket + (1u << logG) - 1) >> logG;
    while (RT < 32 && (1u << RT) < capmax) RT++;
  }
  st.max_bucket = std::max<uint64_t>(st.max_bucket, max_bucket);

  // scatter
  ctx->ensure(w.slots, std::max<uint64_t>(total_slots, 2) * 4);
  HIPCHK(hipMemsetAsync(w.slots.p, 0xFF, std::max<uint64_t>(total_slots, 2) * 4, s));
  if (radix) {
    // pass A: coarse split into dig2 / idx2; pass B: one block per virtual window, payloads to their padded slots
    ctx->ensure(w.part, ((size_t)kc * sortB * Hn + 2 * (size_t)V + 2) * 4);
    uint32_t* d_blk_off = (uint32_t*)w.part.p;
    uint32_t* d_vtot = d_blk_off + (size_t)kc * sortB * Hn;
    uint32_t* d_vstart = d_vtot + V;
    ctx->ensure(w.dig2, n_entries * 4);
    ctx->ensure(w.idx2, n_entries * 4);
    const uint64_t co = (uint64_t)kc * (sortB + 1) * Hn;
    hipLaunchKernelGGL(k_coarse_offsets, dim3((uint32_t)((co + 255) / 256)), dim3(256), 0, s, d_blk_off, d_vtot,
                       (const uint32_t*)w.block_hist.p, (const uint32_t*)w.counts.p, sortB, L, Hn, (uint32_t)kc);
    hipLaunchKernelGGL(k_vscan, dim3(1), dim3(SCAN_THREADS), 0, s, d_vstart, (const uint32_t*)d_vtot, V);
    hipLaunchKernelGGL(k_radix_coarse, dim3(sortB, kc), dim3(RX_THREADS), 0, s, (uint32_t*)w.dig2.p, (uint32_t*)w.idx2.p,
                       (const uint32_t*)d_vstart, (const uint32_t*)d_blk_off, (const uint32_t*)w.dig.p, two_n, chunk, Hn, Hn, ws);
    hipLaunchKernelGGL(k_radix_fine, dim3(V), dim3(RXB_THREADS), 0, s, (uint32_t*)w.slots.p, (const uint32_t*)w.cursor.p,
                       (const uint32_t*)d_vstart, (const uint32_t*)w.dig2.p, (const uint32_t*)w.idx2.p, Lp, L, ws);
  } else if (one_level) {
    hipLaunchKernelGGL(k_scatter_lds, dim3(sortB, kc), dim3(SORT_THREADS), (size_t)L * 4, s, (uint32_t*)w.slots.p,
                       (const uint32_t*)w.cursor.p, (const uint32_t*)w.block_hist.p, (const uint32_t*)w.dig.p, two_n,
                       chunk, L, 0u);
  } else {
    hipLaunchKernelGGL(k_radix_fine, dim3(V), dim3(RXB_THREADS), 0, s, (uint32_t*)w.slots.p, (const uint32_t*)w.cursor.p,
                       d_v2start, d_rec_dig, d_rec_idx, Lp, L, ws);
  }
  // Big windows over a big table: walk round 1 chunk by chunk of the point rows (k_chunk_order, sort_kernels.h).  Needed once
  // the 128 slots of a wave span more than ~1 GB of rows: 64 L rows of 256 bytes, i.e. from c = 18 with more than 2^22 points.
  const uint32_t* round1_slots = (const uint32_t*)w.slots.p;
  const uint16_t* round1_oidx = nullptr;
  uint64_t rec_y_off = 0;   // 12-word fields: where the y records of round 1's results start inside w.rows1
  bool chunked = false;   // round 1 walks chunk-ordered pairs and writes element records, round 2 reads them
  {
    long long chunk_rows_log = 22;   // 2^22 rows of 256 bytes = 1 GB
    MSM_KNOB(chunk_rows_log, "MSM_CHUNK_LOG", 10);
    long long want_chunks = (!te && pl.c >= 18 && n > (1ull << chunk_rows_log)) ? 1 : 0;
    MSM_KNOB(want_chunks, "MSM_CHUNKED", 0);
    const uint64_t nch = (n + (1ull << chunk_rows_log) - 1) >> chunk_rows_log;
    // (round 2 must be an index-free round to read the element records round 1 then writes: logG >= 2)
    if (want_chunks && !te && logG >= 2 && total_slots >= 2 && nch >= 2 && nch + 1 <= (uint64_t)CO_MAX_KEYS) {
      const uint64_t n_pairs = total_slots / 2;
      ctx->ensure(w.slots2, total_slots * 4);
      ctx->ensure(w.oidx, n_pairs * 2);
      rec_y_off = (n_pairs * 64 + 255) & ~(uint64_t)255;
      ctx->ensure(w.rows1, 2 * rec_y_off + 256);
      hipLaunchKernelGGL(k_chunk_order, dim3((uint32_t)((n_pairs + CO_PAIRS - 1) / CO_PAIRS)), dim3(CO_THREADS), 0, s,
                         (uint2*)w.slots2.p, (uint16_t*)w.oidx.p, (const uint2*)w.slots.p, n_pairs, (uint32_t)chunk_rows_log,
                         (uint32_t)nch + 1);
      round1_slots = (const uint32_t*)w.slots2.p;
      round1_oidx = (const uint16_t*)w.oidx.p;
      if (MSM_KNOB_SET("MSM_CHUNK_NOSTORE")) round1_oidx = nullptr;   // experiment (wrong sums): chunk-ordered loads, natural stores
      chunked = true;
    }
  }
  HIPCHK(hipEventRecord(w.ev[2], s));
  if (before_tree) (*before_tree)(s);
  HIPCHK(hipEventRecord(w.ev[5], s));   // the tree starts here (behind the partner group's sort, if there is one)

  // accumulation tree
  // Weierstrass: tail rounds run only until no bucket holds more than FINISH_MAX elements; k_bucket_finish ends it
  uint32_t FINISH_MAX = pl.c >= 18 ? 64 : 32;
  MSM_KNOB(FINISH_MAX, "MSM_FINISH_MAX", 1);
  // big windows (millions of small buckets per group): the last descriptor rounds pay a binary search over all buckets per
  // pair and a launch each for a few million pair additions -- k_bucket_finish takes the last two or four elements of every
  // bucket cheaper (2^26, c = 22: 154.3 -> 150.8 ms; profiles/r04_experiments.txt item 8)
  uint32_t tail_min_pairs = pl.c >= 18 ? 1u << 23 : 1u << 21;
  MSM_KNOB(tail_min_pairs, "MSM_TAIL_MIN", 1);
  const bool use_finish = true;
  int r_stop = RT;
  if (use_finish) {
    // a tail round is worth its launch + inversion latency (~0.25 ms) only while it still has a few million pairs;
    // below that, and once no bucket holds more than FINISH_MAX elements, k_bucket_finish takes over
    uint32_t cap_elems = (max_bucket + (1u << logG) - 1) >> logG;   // largest bucket after the regular rounds
    r_stop = 0;
    while (r_stop < RT &&
           (((cap_elems + (1u << r_stop) - 1) >> r_stop) > FINISH_MAX || w.h_info[3 + r_stop + 1] >= tail_min_pairs))
      r_stop++;
  }
  // outputs alternate between two buffers: size each for the largest round it receives
  long long scratch_pad = 0;   // lanes added to the scratch plane stride (experiment: power-of-two strides vs HBM channels)
  MSM_KNOB(scratch_pad, "MSM_SCRATCH_PAD", 0);
  uint64_t capA = 1, capB = 1;
  {
    int which = 0;
    uint64_t cnt = total_slots;
    for (uint32_t r = 1; r <= logG; r++) {
      cnt /= 2;
      if (r == 1 && chunked) continue;   // element records (w.rows1), not planes
      (which ? capB : capA) = std::max<uint64_t>(which ? capB : capA, cnt);
      which ^= 1;
    }
    for (int r = 1; r <= r_stop; r++) {
      cnt = w.h_info[3 + r];
      (which ? capB : capA) = std::max<uint64_t>(which ? capB : capA, cnt);
      which ^= 1;
    }
  }
  // Idle lanes of a plane-reading round read (and ignore) elements past the end of its input: a round of n_out pairs runs
  // steps * T lanes-steps with T = 256 ceil(ceil(n_out / steps) / 256), so fewer than 257 * steps pairs -- 2 elements of 16
  // bytes per plane each -- lie beyond n_out, whatever the grid (hence the CU count) is.  steps <= 512.
  const size_t tree_slack = (size_t)2 * 257 * 512 * 16 + 4096;
  ctx->ensure(w.bufA, capA * elem_bytes + tree_slack);
  ctx->ensure(w.bufB, capB * elem_bytes + tree_slack);
  uint4* buf[2] = {(uint4*)w.bufA.p, (uint4*)w.bufB.p};
  uint64_t cap[2] = {capA, capB};
  int cur = 0;  // buffer that receives the next round's output
  uint64_t n_in = total_slots;
  int round = 0;
  const uint4* fin = buf[0];
  uint64_t fin_cap = cap[0];
  const uint32_t* off_fin = (const uint32_t*)w.tail_off.p;
  if (total_slots > 0) {
    for (uint32_t r = 1; r <= logG; r++) {
      uint64_t n_out = n_in / 2;
      RoundGeom g = round_geom(ctx, n_out, r == 1 || (r == 2 && chunked) || te, lone);   // no inversion on the Edwards path: always two waves
      const uint64_t sstride = g.T + scratch_pad;
      if (!te) ctx->ensure(w.scratch, (size_t)g.steps * NL * sstride * 4);
      BatchArgs a{};
      a.points = (const uint32_t*)ctx->rows.p + p_lo * (te ? (uint64_t)te::TE_ROW_WORDS : (uint64_t)ROW_WORDS);
      a.slots = r == 1 ? round1_slots : (const uint32_t*)w.slots.p;
      a.oidx = r == 1 ? round1_oidx : nullptr;
      a.in = buf[cur ^ 1];
      a.in_cap = cap[cur ^ 1];
      a.out = buf[cur];
      a.out_cap = cap[cur];
      a.scratch = (uint32_t*)w.scratch.p;
      a.sstride = sstride;
      a.n_out = n_out;
      a.steps = g.steps;
      const bool rows_out = r == 1 && chunked, rows_in = r == 2 && chunked;
      a.y_off = 4 * ctx->nw();
      if (rows_out) a.out_rows = (uint32_t*)w.rows1.p;
      if (rows_in) { a.points = (const uint32_t*)w.rows1.p; a.slots = nullptr; }
      if (rows_out) a.out_y_off = rec_y_off;                      // 12-word fields: x records, then y records (batch_add.h)
      if (rows_in && ctx->nw() == 12) a.y_off = rec_y_off;
      if (r == 1 || rows_in) {
        if (te) hipLaunchKernelGGL(te::k_te_add<MODE_GATHER>, dim3(g.grid), dim3(256), 0, s, a);
        else W_LAUNCH_MODE(ctx, k_batch_add, MODE_GATHER, dim3(g.grid), dim3(256), 0, s, a);
        if (r == 1) HIPCHK(hipEventRecord(w.ev[6], s));
      } else {
        if (te) hipLaunchKernelGGL(te::k_te_add<MODE_REGULAR>, dim3(g.grid), dim3(256), 0, s, a);
        else W_LAUNCH_MODE(ctx, k_batch_add, MODE_REGULAR, dim3(g.grid), dim3(256), 0, s, a);
      }
      st.n_pairs += n_out;
      n_in = n_out;
      round++;
      if (rows_out) continue;   // the plane buffers have not been touched yet
      fin = buf[cur];
      fin_cap = cap[cur];
      cur ^= 1;
    }
    for (int r = 1; r <= r_stop; r++) {
      uint64_t n_out = w.h_info[3 + r];
      RoundGeom g = round_geom(ctx, n_out, te);
      const uint64_t sstride = g.T + scratch_pad;
      if (!te) ctx->ensure(w.scratch, (size_t)g.steps * NL * sstride * 4);
      BatchArgs a{};
      a.in = buf[cur ^ 1];
      a.in_cap = cap[cur ^ 1];
      a.out = buf[cur];
      a.out_cap = cap[cur];
      a.scratch = (uint32_t*)w.scratch.p;
      a.sstride = sstride;
      a.n_out = n_out;
      a.steps = g.steps;
      if (n_out) {
        const uint32_t* off_in = (const uint32_t*)w.tail_off.p + (uint64_t)(r - 1) * (nb + 1);
        const uint32_t* off_out = (const uint32_t*)w.tail_off.p + (uint64_t)r * (nb + 1);
        ctx->ensure(w.desc, n_out * 4);
        hipLaunchKernelGGL(k_tail_desc, dim3((uint32_t)((n_out + 255) / 256)), dim3(256), 0, s, (uint32_t*)w.desc.p, off_in,
                           off_out, (uint32_t)nb, (uint32_t)n_out);
        a.desc = (const uint32_t*)w.desc.p;
        if (te) hipLaunchKernelGGL(te::k_te_add<MODE_SEARCH>, dim3(g.grid), dim3(256), 0, s, a);
        else W_LAUNCH_MODE(ctx, k_batch_add, MODE_SEARCH, dim3(g.grid), dim3(256), 0, s, a);
      }
      st.n_pairs += n_out;
      fin = buf[cur];
      fin_cap = cap[cur];
      cur ^= 1;
      round++;
    }
    off_fin = (const uint32_t*)w.tail_off.p + (uint64_t)r_stop * (nb + 1);
  }
  st.rounds += round;
  const uint32_t* bucket_proj = nullptr;
  if (use_finish && total_slots > 0) {
    ctx->ensure(w.bucket_proj, nb * (te ? 4 * te::TL : 3 * NL) * 4);
    // lanes of a wave should have equal trip counts: order the buckets by what they still hold
    const uint32_t* perm = nullptr;
    if (nb >= 4096) {
      ctx->ensure(w.blk_tab2, (nb + 2 * FINISH_BINS) * 4);
      uint32_t* hist = (uint32_t*)w.blk_tab2.p;
      HIPCHK(hipMemsetAsync(hist, 0, 2 * FINISH_BINS * 4, s));
      const uint32_t fgrid = (uint32_t)((nb + FINISH_THREADS - 1) / FINISH_THREADS);
      hipLaunchKernelGGL(k_finish_hist, dim3(fgrid), dim3(FINISH_THREADS), 0, s, off_fin, (uint32_t)nb, hist);
      hipLaunchKernelGGL(k_finish_perm, dim3(fgrid), dim3(FINISH_THREADS), 0, s, off_fin, (uint32_t)nb,
                         (const uint32_t*)hist, hist + FINISH_BINS, hist + 2 * FINISH_BINS);
      perm = hist + 2 * FINISH_BINS;
    }
    if (te)
      hipLaunchKernelGGL(te::k_te_bucket_finish, dim3((uint32_t)((nb + 255) / 256)), dim3(256), 0, s, (uint32_t*)w.bucket_proj.p,
                         fin, fin_cap, off_fin, (uint32_t)nb, perm);
    else
      W_LAUNCH(ctx, k_bucket_finish, dim3((uint32_t)((nb + 255) / 256)), dim3(256), 0, s, (uint32_t*)w.bucket_proj.p, fin,
                         fin_cap, off_fin, (uint32_t)nb, perm);
    bucket_proj = (const uint32_t*)w.bucket_proj.p;
  }
  if (total_slots == 0) HIPCHK(hipEventRecord(w.ev[6], s));
  HIPCHK(hipEventRecord(w.ev[3], s));

  reduce_buckets(ctx, w, fin, fin_cap, off_fin, bucket_proj, L, kc, h_partials_out, pl.merged, pl.c);
  float ms;
  HIPCHK(hipEventElapsedTime(&ms, w.ev[0], w.ev[1])); st.ms_digits += ms;
  HIPCHK(hipEventElapsedTime(&ms, w.ev[1], w.ev[2])); st.ms_sort += ms;
  HIPCHK(hipEventElapsedTime(&ms, w.ev[5], w.ev[3])); st.ms_acc += ms;
  HIPCHK(hipEventElapsedTime(&ms, w.ev[5], w.ev[6])); st.ms_r1 += ms;
  HIPCHK(hipEventElapsedTime(&ms, w.ev[3], w.ev[4])); st.ms_red += ms;
}

void plane_element_to_wire(const msm_ctx* ctx, const uint32_t* planes, uint64_t cap, uint64_t e, uint8_t* out_xy) {
  const int nw = ctx->nw(), np = nw / 4;
  const size_t cb = ctx->coord_bytes();
  uint32_t w[24];
  for (int cpl = 0; cpl < 2 * np; cpl++)
    for (int q = 0; q < 4; q++) w[4 * cpl + q] = planes[((uint64_t)cpl * cap + e) * 4 + q];
  memset(out_xy, 0, 2 * cb);
  if (w[nw - 1] == INF_WORD) return;
  msm_host::Fe6 one = {{1, 0, 0, 0, 0, 0}};
  for (int j = 0; j < 2; j++) {
    msm_host::Fe6 t;
    words_to_fe6(t, w + nw * j, nw);
    ctx->hc.F.mul(t, t, ctx->k_dev_to_host);
    ctx->hc.F.mul(t, t, one);
    uint8_t b48[48];
    fe6_to_bytes(b48, t);
    memcpy(out_xy + cb * j, b48, cb);
  }
}

// how many windows fit one group under the workspace budget
long double window_bytes(const msm_ctx* ctx, uint64_t n, const Plan& pl) {
  // bytes per window and point (Weierstrass: 2 entries per point): digits 8, record arrays of the radix passes 16 (+ 8 for the
  // third pass of windows above 2^15 buckets), slots ~9, tree buffers 96 + 48, prefix scratch 56; a chunk-ordered round 1
  // (c >= 18) adds its reordered slots, the index table and the element records (128 bytes per pair), and its plane buffers start one round
  // later.  Per window and bucket: counters, cursors, up to 34 offset tables, and the block histograms of the sort.
  const bool te = ctx->is_te();
  const bool big = pl.c > 16;
  long double per_point = te ? (4 + 8 + 5 + 64 + 32) : (8 + 16 + 9 + 96 + 48 + 56);
  if (big && !te) per_point += 8 + 9 + 2 + 128 - 72;
  const long double hist_bins = big ? (long double)(pl.L >> 7) : (long double)pl.L;
  return (long double)n * per_point + (long double)pl.L * 4 * 40 + hist_bins * 4 * (2.0L * ctx->n_cu);
}
int windows_per_group(const msm_ctx* ctx, uint64_t n, const Plan& pl) {
  int w = (int)std::max<long double>(1, (long double)ctx->ws_budget / msm_ctx::N_WS / window_bytes(ctx, n, pl));
  return std::min(w, pl.K);
}
// how many ranges of the points ONE window has to be cut into for its workspace to fit (1: it fits as a whole)
uint64_t point_pieces(const msm_ctx* ctx, uint64_t n, const Plan& pl) {
  const long double room = (long double)ctx->ws_budget / msm_ctx::N_WS;
  uint64_t pieces = 1;
  while (pieces < 256 && n / pieces > 4096 && window_bytes(ctx, (n + pieces - 1) / pieces, pl) > room) pieces++;
  return pieces;
}

// S = sum_k 2^(ck) P_k, then affine (src/msm-batched-affine.ts:322-333, src/curve-projective.ts:335-349)
void horner_to_affine(const msm_host::Curve6& C, const std::vector<msm_host::Proj6>& P, int c, msm_result* out) {
  int K = (int)P.size();
  msm_host::Proj6 acc = P[K - 1];
  for (int k = K - 2; k >= 0; k--) {
    for (int j = 0; j < c; j++) acc = C.dbl(acc);
    acc = C.add(acc, P[k]);
  }
  memset(out->x, 0, 48);
  memset(out->y, 0, 48);
  if (C.is_zero(acc)) {
    out->is_infinity = 1;
    return;
  }
  out->is_infinity = 0;
  msm_host::Fe6 zi, x, y, one = {{1, 0, 0, 0, 0, 0}};
  C.F.inv(zi, acc.Z);
  C.F.mul(x, acc.X, zi);
  C.F.mul(y, acc.Y, zi);
  C.F.mul(x, x, one);  // leave Montgomery form
  C.F.mul(y, y, one);
  fe6_to_bytes(out->x, x);
  fe6_to_bytes(out->y, y);
}

// twisted Edwards tail: S = sum_k 2^(ck) P_k with unified additions (src/msm-basic.ts:142-158), then x = X/Z, y = Y/Z
void te_horner_points(const msm_host::TeCurve6& C, const std::vector<msm_host::Ext6>& P, int c, msm_result* out) {
  const int K = (int)P.size();
  msm_host::Ext6 acc = P[K - 1];
  for (int k = K - 2; k >= 0; k--) {
    for (int j = 0; j < c; j++) acc = C.add(acc, acc);
    acc = C.add(acc, P[k]);
  }
  msm_host::Fe6 zi, x, y, one = {{1, 0, 0, 0, 0, 0}};
  C.F.inv(zi, acc.Z);
  C.F.mul(x, acc.X, zi);
  C.F.mul(y, acc.Y, zi);
  C.F.mul(x, x, one);
  C.F.mul(y, y, one);
  memset(out->x, 0, 48);
  memset(out->y, 0, 48);
  fe6_to_bytes(out->x, x);
  fe6_to_bytes(out->y, y);
  out->is_infinity = 0;
}

// device window sum (X, Y, Z, T: 8 words each, below 2p) -> host extended point: as for the Weierstrass sums no change of
// radix -- (X, Y, Z, T) with T = X Y / Z stays a valid extended point when all four carry one common factor
msm_host::Ext6 te_partial_to_host(const msm_ctx* ctx, const uint32_t* w) {
  const auto& C = ctx->hte;
  msm_host::Ext6 P;
  msm_host::Fe6* dst[4] = {&P.X, &P.Y, &P.Z, &P.T};
  for (int j = 0; j < 4; j++) {
    msm_host::Fe6 t = {{0, 0, 0, 0, 0, 0}};
    for (int i = 0; i < 4; i++) t.v[i] = (uint64_t)w[8 * j + 2 * i] | ((uint64_t)w[8 * j + 2 * i + 1] << 32);
    while (msm_host::Field6::ge(t, C.F.p)) C.F.sub_raw(t, t, C.F.p);
    *dst[j] = t;
  }
  return P;
}

// host extended point -> the window-sum form (X, Y, Z, T: 8 words each)
void te_host_to_partial(const msm_ctx*, const msm_host::Ext6& P, uint32_t* out32) {
  const msm_host::Fe6* co[4] = {&P.X, &P.Y, &P.Z, &P.T};
  for (int j = 0; j < 4; j++)
    for (int i = 0; i < 4; i++) {
      out32[8 * j + 2 * i] = (uint32_t)co[j]->v[i];
      out32[8 * j + 2 * i + 1] = (uint32_t)(co[j]->v[i] >> 32);
    }
}

void te_horner_to_affine(const msm_ctx* ctx, const std::vector<uint32_t>& words, int K, int c, msm_result* out) {
  std::vector<msm_host::Ext6> P(K);
  for (int k = 0; k < K; k++) P[k] = te_partial_to_host(ctx, &words[(size_t)k * 32]);
  te_horner_points(ctx->hte, P, c, out);
}

// A big buffer in pageable host memory (what a caller of msm_run normally holds: 2 GB of scalars at 2^26) crosses PCIe at
// ~20 GB/s through one hipMemcpy, which stages it through pinned memory on one thread.  Here a few host threads copy 16 MB
// chunks into pinned slots of their own and queue each chunk's transfer behind it, so the host copies and the DMA overlap.
// Ordered into ctx->stream: work queued there afterwards sees the whole buffer.
void ensure_staging(msm_ctx* ctx) {
  constexpr int T = msm_ctx::STAGE_THREADS, S = msm_ctx::STAGE_SLOTS;
  if (ctx->staging_ready) return;
  // (a failure half way leaves what exists in place: the next call creates only what is still missing)
  if (!ctx->stage_pin) HIPCHK(hipHostMalloc((void**)&ctx->stage_pin, (size_t)T * S * msm_ctx::STAGE_CHUNK, hipHostMallocDefault));
  int prio_lo = 0, prio_hi = 0;   // least and greatest priority (numerically lower = higher)
  HIPCHK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
  for (int t = 0; t < T; t++) {
    // copies queued while kernels hold the chip must not wait behind them: highest priority the device offers
    if (!ctx->stage_stream[t]) HIPCHK(hipStreamCreateWithPriority(&ctx->stage_stream[t], hipStreamNonBlocking, prio_hi));
    for (int q = 0; q <= S; q++)
      if (!ctx->stage_ev[t][q]) HIPCHK(hipEventCreateWithFlags(&ctx->stage_ev[t][q], hipEventDisableTiming));
    for (int q = 0; q < msm_ctx::MAX_PIECES; q++)
      if (!ctx->piece_ev[q][t]) HIPCHK(hipEventCreateWithFlags(&ctx->piece_ev[q][t], hipEventDisableTiming));
  }
  ctx->staging_ready = true;
}

void upload_staged(msm_ctx* ctx, void* dst, const void* src, size_t bytes) {
  constexpr int T = msm_ctx::STAGE_THREADS, S = msm_ctx::STAGE_SLOTS;
  constexpr size_t CH = msm_ctx::STAGE_CHUNK;
  if (bytes < 4 * CH) {
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return;
  }
  ensure_staging(ctx);
  HIPCHK(hipStreamSynchronize(ctx->stream));   // dst may still be in use by what the stream holds
  for (int t = 0; t < T; t++) HIPCHK(hipStreamSynchronize(ctx->stage_stream[t]));   // and the slots by an earlier upload
  const size_t n_chunks = (bytes + CH - 1) / CH;
  hipError_t rc[T];
  std::vector<std::thread> th;
  th.reserve(T);
  for (int t = 0; t < T; t++) rc[t] = hipSuccess;
  for (int t = 0; t < T; t++) {
    auto job = [&, t] {
      hipError_t e = hipSetDevice(ctx->device);
      size_t turn = 0;
      for (size_t i = t; i < n_chunks && e == hipSuccess; i += T, turn++) {
        const int q = (int)(turn % S);
        char* pin = ctx->stage_pin + ((size_t)t * S + q) * CH;
        if (turn >= (size_t)S) e = hipEventSynchronize(ctx->stage_ev[t][q]);   // the slot's previous transfer has left it
        if (e != hipSuccess) break;
        const size_t off = i * CH, len = std::min(CH, bytes - off);
        memcpy(pin, (const char*)src + off, len);
        e = hipMemcpyAsync((char*)dst + off, pin, len, hipMemcpyHostToDevice, ctx->stage_stream[t]);
        if (e == hipSuccess) e = hipEventRecord(ctx->stage_ev[t][q], ctx->stage_stream[t]);
      }
      if (e == hipSuccess) e = hipEventRecord(ctx->stage_ev[t][S], ctx->stage_stream[t]);
      rc[t] = e;
    };
    // a thread that cannot be started (resource limits) must not leave joinable threads behind: its share runs here
    try { th.emplace_back(job); } catch (const std::system_error&) { job(); }
  }
  for (auto& x : th) x.join();
  for (int t = 0; t < T; t++) HIPCHK(rc[t]);
  for (int t = 0; t < T; t++) HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->stage_ev[t][S], 0));
}

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
    MSM_KNOB(n_streams_, "MSM_UPLOAD_STREAMS", 1);
    n_streams_ = std::min<long long>(n_streams_, T);
    HIPCHK(hipStreamSynchronize(ctx->stream));   // dst may still be in use by what the stream holds
    for (int t = 0; t < T; t++) HIPCHK(hipStreamSynchronize(ctx->stage_stream[t]));
    for (int t = 0; t < T; t++) rc_[t] = hipSuccess;
    t0_ = std::chrono::steady_clock::now();
    for (int t = 0; t < T; t++) {
      try { th_.emplace_back([this, t] { run(t); }); } catch (const std::system_error&) { run(t); }
    }
  }
  ~PieceUpload() { join(); }
  // host: blocks until every staging thread has queued its part of piece q; device: `stream` then waits for those copies
  void wait_piece(int q, hipStream_t stream) {
    {
      std::unique_lock<std::mutex> l(mu_);
      cv_.wait(l, [&] { return enq_[q] == T; });
    }
    for (int t = 0; t < T; t++) {
      if (rc_[t] != hipSuccess) throw HipFail{rc_[t], "staged upload of the scalars", __LINE__};
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
    double t_wait = 0, t_copy = 0, t_enq = 0;   // tuning builds: where the host side of the transfer spends its time
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(now() - a).count(); };
    for (size_t i = t; i < n_chunks && e == hipSuccess; i += T, turn++) {
      const size_t off = i * CH, len = std::min(CH, bytes_ - off);
      int upto = q;
      while (upto < (int)ends_.size() && ends_[upto] <= off) upto++;   // pieces that end at or before this chunk
      mark(upto);
      const int slot = (int)(turn % S);
      char* pin = ctx_->stage_pin + ((size_t)t * S + slot) * CH;
      auto a = now();
      if (turn >= (size_t)S) e = hipEventSynchronize(ctx_->stage_ev[t][slot]);
      t_wait += since(a);
      if (e != hipSuccess) break;
      a = now();
      memcpy(pin, src_ + off, len);
      t_copy += since(a);
      a = now();
      e = hipMemcpyAsync(dst_ + off, pin, len, hipMemcpyHostToDevice, ctx_->stage_stream[t % n_streams_]);
      if (e == hipSuccess) e = hipEventRecord(ctx_->stage_ev[t][slot], ctx_->stage_stream[t % n_streams_]);
      t_enq += since(a);
    }
    if (MSM_KNOB_SET("MSM_UPLOAD_TRACE"))
      fprintf(stderr, "upload thread %d: slot wait %.1f ms, host copy %.1f ms, enqueue %.1f ms, total %.1f ms\n", t, t_wait, t_copy, t_enq,
              since(t0_));
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

int stage_scalars(msm_ctx* ctx, const void* scalars, uint64_t n, int on_device, const uint32_t** d_out) {
  if (on_device) {
    *d_out = (const uint32_t*)scalars;
    return MSM_OK;
  }
  ctx->ensure(ctx->scal, n * 32);
  upload_staged(ctx, ctx->scal.p, scalars, n * 32);
  *d_out = (const uint32_t*)ctx->scal.p;
  return MSM_OK;
}

// windows [k_lo, k_hi) over the resident points [p_off, p_off + n); scalars[i] belongs to point p_off + i
int window_sums_once(msm_ctx* ctx, const void* scalars, uint64_t n, int on_device, const msm_opts* opts, int k_lo, int k_hi,
                     const Plan& pl, std::vector<uint32_t>& words, msm_result* stats, uint64_t p_off) {
  const uint32_t* d_scal = nullptr;
  HIPCHK(hipEventRecord(ctx->ev[8], ctx->stream));
  // Host scalars of a big call cross PCIe BEHIND the computation, range by range of the points (PieceUpload); everything
  // else is staged before the window groups start.
  std::vector<uint64_t> piece_end;   // pipelined upload: point index where piece q ends (the last = n)
  if (!on_device && n >= (1ull << 24)) {
    // The link moves scalars ~4x as fast as the GPU consumes them (2 GB in ~40 ms against ~154 ms of MSM at 2^26), so
    // every range may be ~4x its predecessor and still arrive before the GPU is done with the one before: 1/16, 3/16, the
    // rest from 2^25 points; 1/8, 3/8, the rest below.  The first range is what the GPU waits for (2-3 ms); few ranges keep
    // the sub-MSMs near full-size efficiency.
    const uint64_t gran = msm_ctx::STAGE_CHUNK / 32;   // scalars per staging chunk
    const int big = n >= (1ull << 25);
    for (int sh : {big ? 4 : 3, big ? 2 : 1}) piece_end.push_back(((n >> sh) / gran) * gran);
    piece_end.push_back(n);
  }
  std::unique_ptr<PieceUpload> pipe;
  if (piece_end.empty()) stage_scalars(ctx, scalars, n, on_device, &d_scal);
  HIPCHK(hipEventRecord(ctx->ev[9], ctx->stream));
  GroupStats st;
  const int pw = ctx->is_te() ? 32 : 36;
  words.assign((size_t)(k_hi - k_lo) * pw, 0);
  // window groups: as large as the workspace budget allows; for big inputs two of them on two streams.  The streams
  // run in step (both sort, both gather, ...): what the second one buys is two tree kernels sharing the chip -- forward
  // (memory-heavy) and backward (issue-heavy) sweeps of different waves mix, the small last rounds fill each other's
  // idle CUs -- not a sort hidden under an accumulation (a sort started under the other group's tree finds no free
  // registers on any CU and takes four times as long: profiles/r04_experiments.txt item 1)
  if (ctx->ws_limit) {
    ctx->ws_budget = ctx->ws_limit;
  } else if (n >= (1ull << 22)) {
    // big inputs: the budget is what the device has free NOW (point sets, scalar buffers and other contexts have come and
    // gone since the context was made) plus what the workspaces already hold
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    uint64_t held = 0;
    for (auto& w : ctx->ws)
      for (DevBuf* b : w.all) held += b->cap;
    ctx->ws_budget = (uint64_t)((free_b + held) * 0.85L);
  }
  int wpg = std::min(windows_per_group(ctx, n, pl), 128);
  // the radix-split and three-pass sorts describe their windows in a WinSplit of 16 entries (sort_kernels.h): a group that
  // may take one of them holds at most 16 windows (msmProjective with a small explicit window: K = 17 .. 29 at c = 15 .. 9)
  // (the one-level sort of small inputs -- a window's counters fit the LDS and fewer than 2^22 entries per window -- has no
  // such table: Ed-on-BLS12-377 at 2^20 keeps its 18 windows in one group)
  {
    const uint64_t entries = ctx->is_te() ? n : 2 * n;
    const bool fits_lds = ((size_t)pl.L * 4 <= 128 * 1024);
    if (pl.c - 1 > (int)RX_FINE_BITS && (!fits_lds || entries >= (1ull << 22))) wpg = std::min(wpg, 16);
  }
  const int nwin = k_hi - k_lo;
  // measured on MI355X: two groups win 14 % at 2^23 / 2^24, 3 % at 2^22, nothing at 2^21 -- below that the fixed
  // per-group latencies (read-backs, bucket reduction depth) cost more than the overlap returns
  int want_groups = (nwin >= 2 && n >= (1ull << 22)) ? 2 : 1;
  MSM_KNOB(want_groups, "MSM_GROUPS", 1);
  wpg = std::max(1, std::min(wpg, (nwin + want_groups - 1) / want_groups));
  struct Group {
    int ka, kb;
    uint64_t p_lo, p_n;
    int piece;   // pipelined upload: the piece whose arrival the group waits for (-1: the scalars are in place)
    int pair;    // the two groups with the same pair id run side by side and start their trees together (PairSync); -1: none
  };
  std::vector<Group> groups;
  // A single window (the 8-GPU shard) has no second window group to hide its sort and tails under: split it by
  // points instead -- two half-size sub-MSMs of the same window on the two streams, their sums added on the host.
  // The same split serves inputs whose single window no longer fits the workspace budget (2^29 points: 165 GB per window
  // at c = 22 next to a 137 GB row table): every window runs over as many ranges of the points as it takes, one after the
  // other on the two streams, and the sums of its ranges are added on the host.
  uint64_t pieces = 1;
  if (nwin == 1 && want_groups == 1 && !ctx->is_te() && n >= (1ull << 24) && !MSM_KNOB_SET("MSM_GROUPS")) pieces = 2;
  pieces = std::max(pieces, point_pieces(ctx, n, pl));
  MSM_KNOB(pieces, "MSM_PIECES", 1);
  if (!piece_end.empty() && (point_pieces(ctx, n, pl) > 1 || MSM_KNOB_SET("MSM_PIECES"))) {
    // the workspace forces its own ranges: plain staged upload first (rare: 2^29 points, or a tight msm_set_workspace_limit)
    piece_end.clear();
    stage_scalars(ctx, scalars, n, on_device, &d_scal);
  }
  if (!piece_end.empty()) {
    // pipelined host scalars: per arriving range of the points the usual window groups (two above 2^22 points), in order
    ctx->ensure(ctx->scal, n * 32);
    d_scal = (const uint32_t*)ctx->scal.p;
    uint64_t lo = 0;
    for (size_t q = 0; q < piece_end.size(); q++) {
      const uint64_t cnt = piece_end[q] - lo;
      const int g = (nwin >= 2 && cnt >= (1ull << 22)) ? 2 : 1;
      const int per = std::max(1, std::min(wpg, (nwin + g - 1) / g));
      const size_t first = groups.size();
      for (int k = k_lo; k < k_hi; k += per) groups.push_back({k, std::min(k_hi, k + per), lo, cnt, (int)q, -1});
      if (groups.size() - first == 2) groups[first].pair = groups[first + 1].pair = (int)q;
      lo = piece_end[q];
    }
  } else if (pieces > 1) {
    for (int k = k_lo; k < k_hi; k++)
      for (uint64_t q = 0; q < pieces; q++) {
        const uint64_t lo = n * q / pieces, hi = n * (q + 1) / pieces;
        groups.push_back({k, k + 1, lo, hi - lo, -1, -1});
      }
  } else {
    long long first_group = 0;   // experiment: windows in the first of two uneven groups
    MSM_KNOB(first_group, "MSM_WPG_A", 1);
    if (first_group > 0 && first_group < nwin) {
      groups.push_back({k_lo, k_lo + (int)first_group, 0, n, -1, -1});
      groups.push_back({k_lo + (int)first_group, k_hi, 0, n, -1, -1});
    } else {
      for (int k = k_lo; k < k_hi; k += wpg) groups.push_back({k, std::min(k_hi, k + wpg), 0, n, -1, -1});
    }
    if (groups.size() == 2) groups[0].pair = groups[1].pair = 0;
  }
  // does more than one group contribute to a window?  Then the sums of its ranges are added on the host below.
  bool split_points = false;
  for (const Group& g : groups) split_points |= g.p_n != n;
  std::vector<std::vector<uint32_t>> split_part(split_points ? groups.size() : 0);
  HIPCHK(hipMemsetAsync(ctx->errflag.p, 0, 4, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));   // staged scalars are in place before the group streams start
  if (!piece_end.empty()) {
    std::vector<size_t> ends;
    for (uint64_t e : piece_end) ends.push_back((size_t)e * 32);
    pipe.reset(new PieceUpload(ctx, ctx->scal.p, scalars, n * 32, ends));
  }
  std::atomic<int> next{0};
  GroupStats sts[msm_ctx::N_WS];
  const int nthreads = (opts && opts->serial) ? 1 : std::min<int>(msm_ctx::N_WS, (int)groups.size());
  PairSync psync;
  // Off: measured neutral where the groups' sorts differ (c = 22: 153.5 against 154.4 ms -- the stretched sort of one group
  // was time the other group's tree had the chip to itself) and harmful where they are equal (c = 16: 159.9 against 157.3 --
  // trees that start at the same instant walk their sweeps in step).  Kept as a knob of the tuning build.
  long long want_pair_sync = 0;
  MSM_KNOB(want_pair_sync, "MSM_PAIR_SYNC", 0);
  auto worker = [&](int slot) {
    HIPCHK(hipSetDevice(ctx->device));
    for (;;) {
      int gi = next.fetch_add(1);
      if (gi >= (int)groups.size()) break;
      const int ka = groups[gi].ka, kb = groups[gi].kb;
      std::vector<uint32_t> part((size_t)(kb - ka) * pw);
      Plan pg = pl;
      // a launch that has the chip to itself -- the one-window shard, or every launch of a serialised call (msm_opts.serial,
      // the exclusive timing of the roofline) -- walks its pairs in four short batches instead of one long one (round_geom)
      pg.lone = (groups.size() == 1 && kb - ka == 1) || (opts && opts->serial);
      if (groups[gi].piece >= 0) pipe->wait_piece(groups[gi].piece, ctx->ws[slot].stream);
      // the partner group runs on the other workspace; its ev[2] closes its sort
      const int pair = (nthreads == 2 && want_pair_sync) ? groups[gi].pair : -1;
      const std::function<void(hipStream_t)> meet = [&, slot, pair](hipStream_t s) {
        if (psync.arrive_and_wait(pair, (int)groups.size())) HIPCHK(hipStreamWaitEvent(s, ctx->ws[1 - slot].ev[2], 0));
        long long tree_delay_us = 0;   // experiment: the second group's tree starts this much after the first one's
        MSM_KNOB(tree_delay_us, "MSM_TREE_DELAY_US", 0);
        if (tree_delay_us && slot == 1) {
          HIPCHK(hipStreamSynchronize(s));
          std::this_thread::sleep_for(std::chrono::microseconds(tree_delay_us));
        }
      };
      try {
        run_window_group(ctx, ctx->ws[slot], d_scal, groups[gi].p_lo, groups[gi].p_n, pg, ka, kb, part.data(), sts[slot], p_off,
                         pair >= 0 ? &meet : nullptr);
      } catch (...) {
        psync.abort();   // the partner must not wait for a group that will not arrive
        throw;
      }
      if (split_points) split_part[gi] = part;
      else memcpy(&words[(size_t)(ka - k_lo) * pw], part.data(), part.size() * 4);
    }
  };
  {
    // Whatever either worker throws (HIP failure, bad_alloc, ...) is re-raised here only after BOTH have stopped and both
    // group streams are idle: no queued kernel of a failed call may still run when the context is used again.
    std::exception_ptr err;
    long long stagger_us = 0;
    MSM_KNOB(stagger_us, "MSM_STAGGER_US", 0);
    if (nthreads > 1) ctx->helper->run([&, stagger_us] {
      if (stagger_us) std::this_thread::sleep_for(std::chrono::microseconds(stagger_us));
      worker(1);
    });
    try { worker(0); } catch (...) { err = std::current_exception(); }
    if (nthreads > 1) {
      try { ctx->helper->wait(); } catch (...) { if (!err) err = std::current_exception(); }
    }
    if (err) {
      next.store((int)groups.size());
      for (auto& w : ctx->ws) (void)hipStreamSynchronize(w.stream);
      std::rethrow_exception(err);
    }
  }
  {
    // scalars >= q seen by k_digits: refused under msm_opts.strict (otherwise they were reduced mod q)
    HIPCHK(hipMemcpyAsync(ctx->h_info, ctx->errflag.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (pl.strict && (ctx->h_info[0] & 4u)) throw MsmFail{MSM_ERR_SCALAR, "a scalar is >= the group order q (msm_opts.strict)"};
  }
  float upload_ms = -1;
  if (pipe) upload_ms = pipe->finish();   // joins the staging threads; their last copy is done
  if (split_points) {
    // P_k = sum over the ranges of the points (groups of one or several windows each); an all-zero partial (Z = 0) is the
    // identity.  (Plan.merged: a group then carries sum_kk 2^(c kk) P_kk in its first slot and identities in the others --
    // slot-wise sums of such groups are still a valid set of slots for the Horner step.)
    for (int k = k_lo; k < k_hi; k++) {
      uint32_t* out = &words[(size_t)(k - k_lo) * pw];
      if (ctx->is_te()) {
        msm_host::Ext6 acc = ctx->hte.zero();
        for (size_t gi = 0; gi < groups.size(); gi++)
          if (groups[gi].ka <= k && k < groups[gi].kb && !split_part[gi].empty())
            acc = ctx->hte.add(acc, te_partial_to_host(ctx, split_part[gi].data() + (size_t)(k - groups[gi].ka) * pw));
        te_host_to_partial(ctx, acc, out);
      } else {
        msm_host::Proj6 acc = ctx->hc.zero();
        for (size_t gi = 0; gi < groups.size(); gi++)
          if (groups[gi].ka <= k && k < groups[gi].kb && !split_part[gi].empty())
            acc = ctx->hc.add(acc, partial_to_host(ctx, split_part[gi].data() + (size_t)(k - groups[gi].ka) * pw));
        host_to_partial(ctx, acc, out);
      }
    }
  }
  for (int i = 0; i < msm_ctx::N_WS; i++) {
    st.n_pairs += sts[i].n_pairs;
    st.n_pairs_algo += sts[i].n_pairs_algo;
    st.max_bucket = std::max(st.max_bucket, sts[i].max_bucket);
    st.rounds += sts[i].rounds;   // tree rounds (k_batch_add launches) of ALL window groups, like n_pairs and ms_acc
    st.ms_digits += sts[i].ms_digits; st.ms_sort += sts[i].ms_sort; st.ms_acc += sts[i].ms_acc;
    st.ms_red += sts[i].ms_red; st.ms_r1 += sts[i].ms_r1;
  }
  HIPCHK(hipEventRecord(ctx->ev[10], ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (stats) {
    float ms;
    HIPCHK(hipEventElapsedTime(&ms, ctx->ev[8], ctx->ev[9]));
    stats->phase_ms[MSM_T_UPLOAD] = upload_ms >= 0 ? upload_ms : ms;   // pipelined: host clock of the background transfer
    HIPCHK(hipEventElapsedTime(&ms, ctx->ev[8], ctx->ev[10]));
    stats->phase_ms[MSM_T_TOTAL] = ms;
    stats->phase_ms[MSM_T_DIGITS] = st.ms_digits;
    stats->phase_ms[MSM_T_SORT] = st.ms_sort;
    stats->phase_ms[MSM_T_ACCUMULATE] = st.ms_acc;
    stats->phase_ms[MSM_T_ACC_ROUND1] = st.ms_r1;
    stats->phase_ms[MSM_T_REDUCE] = st.ms_red;
    stats->n_pairs = st.n_pairs;
    stats->n_pairs_algo = st.n_pairs_algo;
    stats->max_bucket = st.max_bucket;
    stats->rounds = st.rounds;
    stats->c = pl.c;
    stats->K = pl.K;
  }
  return MSM_OK;
}


// Workspace buffers only grow, and a call with another shape (window size, curve of the point set, sort path) leaves buffers
// behind that the next shape does not use: if the device runs out of memory the workspaces are dropped and the call runs
// once more from a clean slate, where the budget model of window_sums_once holds again.
int window_sums_impl(msm_ctx* ctx, const void* scalars, uint64_t n, int on_device, const msm_opts* opts, int k_lo, int k_hi,
                     const Plan& pl, std::vector<uint32_t>& words, msm_result* stats, uint64_t p_off = 0) {
  // The budget model is an estimate and other contexts may take memory while the call runs, so one clean-slate retry is not a
  // guarantee: every further attempt also halves what the workspaces may take (more window groups, then ranges of the points),
  // which trades time for memory as include/msm_hip.h promises.  The caller's own limit is restored afterwards.
  const uint64_t limit0 = ctx->ws_limit;
  for (int attempt = 0;; attempt++) {
    try {
      const int rc = window_sums_once(ctx, scalars, n, on_device, opts, k_lo, k_hi, pl, words, stats, p_off);
      ctx->ws_limit = limit0;
      return rc;
    } catch (const HipFail& f) {
      if (f.e != hipErrorOutOfMemory || attempt >= 4) {
        ctx->ws_limit = limit0;
        throw;
      }
    } catch (...) {
      ctx->ws_limit = limit0;
      throw;
    }
    (void)hipGetLastError();
    for (auto& w : ctx->ws) {
      (void)hipStreamSynchronize(w.stream);
      for (DevBuf* b : w.all) ctx->release(*b);
    }
    if (attempt >= 1) ctx->ws_limit = std::max<uint64_t>(ctx->ws_budget / 2, (uint64_t)64 << 20);
  }
}

// Multi-device context: one MSM over the devices of the list, each from its own host thread on its own context.
//   by points (default): device d runs ALL windows [k_lo, k_hi) on its share [n d / G, n (d + 1) / G) of the points and needs
//     only that share of the scalars; the G sums of every window are added on the host (G - 1 projective additions each);
//   by window (msm_opts.by_window): the window range is cut into contiguous shards (windows are independent until the Horner
//     step, src/msm-batched-affine.ts:312-333), every device needs all n scalars.
// Scalars: `placed` != nullptr -- one device pointer per device, already on that device (by points: the device's share);
// else a host buffer, of which every device uploads what it needs, or a device buffer on devices[0], of which the other
// devices first copy their part peer-to-peer -- all devices at once, each from its own thread, under device 0's shard.
int multi_window_sums(msm_ctx* ctx, const void* scalars, const void* const* placed, uint64_t n, int on_device, const msm_opts* opts,
                      int k_lo, int k_hi, const Plan& pl, std::vector<uint32_t>& words, msm_result* stats, uint64_t p_off) {
  const int ndev = 1 + (int)ctx->children.size();
  const int nwin = k_hi - k_lo, pw = ctx->is_te() ? 32 : 36;
  const bool by_window = opts && opts->by_window;
  words.assign((size_t)nwin * pw, 0);
  std::vector<int> lo(ndev, k_lo), hi(ndev, k_hi);
  std::vector<uint64_t> p0(ndev, 0), pn(ndev, n);
  for (int d = 0, k = k_lo; d < ndev; d++) {
    if (by_window) {
      const int cnt = nwin / ndev + (d < nwin % ndev ? 1 : 0);
      lo[d] = k;
      hi[d] = k + cnt;
      k += cnt;
    } else {
      p0[d] = n * (uint64_t)d / ndev;
      pn[d] = n * (uint64_t)(d + 1) / ndev - p0[d];
    }
  }
  std::vector<std::vector<uint32_t>> part(ndev);
  std::vector<msm_result> st(ndev);
  for (auto& r : st) memset(&r, 0, sizeof r);
  auto shard = [&](int d) {
    if (hi[d] <= lo[d] || pn[d] == 0) return;
    msm_ctx* c = d == 0 ? ctx : ctx->children[d - 1];
    HIPCHK(hipSetDevice(c->device));
    const void* sc;
    int dev_side = on_device;
    if (placed) {
      sc = placed[d];
      dev_side = 1;
    } else {
      sc = (const uint8_t*)scalars + p0[d] * 32;
      if (on_device && d > 0) {
        c->ensure(c->scal, pn[d] * 32);
        HIPCHK(hipMemcpyPeerAsync(c->scal.p, c->device, sc, ctx->device, pn[d] * 32, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        sc = c->scal.p;
      }
    }
    window_sums_impl(c, sc, pn[d], dev_side, opts, lo[d], hi[d], pl, part[d], &st[d], p_off + p0[d]);
  };
  std::exception_ptr err;
  for (int d = 1; d < ndev; d++) ctx->fan[d - 1]->run([&, d] { shard(d); });
  try { shard(0); } catch (...) { err = std::current_exception(); }
  for (int d = 1; d < ndev; d++) {
    try { ctx->fan[d - 1]->wait(); } catch (...) { if (!err) err = std::current_exception(); }
  }
  HIPCHK(hipSetDevice(ctx->device));
  if (err) std::rethrow_exception(err);
  if (by_window) {
    for (int d = 0; d < ndev; d++)
      if (hi[d] > lo[d]) memcpy(&words[(size_t)(lo[d] - k_lo) * pw], part[d].data(), part[d].size() * 4);
  } else {
    // P_k = sum over the devices; an all-zero partial (Z = 0) is the identity, a device without points has none at all
    for (int k = 0; k < nwin; k++) {
      if (ctx->is_te()) {
        msm_host::Ext6 acc = ctx->hte.zero();
        for (int d = 0; d < ndev; d++)
          if (!part[d].empty()) acc = ctx->hte.add(acc, te_partial_to_host(ctx, &part[d][(size_t)k * pw]));
        te_host_to_partial(ctx, acc, &words[(size_t)k * pw]);
      } else {
        msm_host::Proj6 acc = ctx->hc.zero();
        for (int d = 0; d < ndev; d++)
          if (!part[d].empty()) acc = ctx->hc.add(acc, partial_to_host(ctx, &part[d][(size_t)k * pw]));
        host_to_partial(ctx, acc, &words[(size_t)k * pw]);
      }
    }
  }
  if (stats) {
    for (int d = 0; d < ndev; d++) {
      stats->n_pairs += st[d].n_pairs;
      stats->n_pairs_algo += st[d].n_pairs_algo;
      stats->rounds += st[d].rounds;
      stats->max_bucket = std::max(stats->max_bucket, st[d].max_bucket);
      for (int j = 0; j < MSM_N_PHASES; j++) stats->phase_ms[j] = std::max(stats->phase_ms[j], st[d].phase_ms[j]);
    }
    stats->c = pl.c;
    stats->K = pl.K;
  }
  return MSM_OK;
}

// the one entry the ABI functions use: single- or multi-device
int any_window_sums(msm_ctx* ctx, const void* scalars, uint64_t n, int on_device, const msm_opts* opts, int k_lo, int k_hi,
                    const Plan& pl, std::vector<uint32_t>& words, msm_result* stats, const void* const* placed = nullptr) {
  const uint64_t p_off = opts ? opts->point_lo : 0;
  if (ctx->children.empty())
    return window_sums_impl(ctx, placed ? placed[0] : scalars, n, placed ? 1 : on_device, opts, k_lo, k_hi, pl, words, stats, p_off);
  return multi_window_sums(ctx, scalars, placed, n, on_device, opts, k_lo, k_hi, pl, words, stats, p_off);
}

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

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================

extern "C" {

uint32_t msm_abi_version(void) { return MSM_ABI_VERSION; }
uint32_t msm_abi_struct_bytes(int which) { return which == 0 ? (uint32_t)sizeof(msm_opts) : which == 1 ? (uint32_t)sizeof(msm_result) : 0u; }

int msm_ctx_create(msm_ctx** out, int curve, int device) {
  if (!out) return MSM_ERR_ARG;
  *out = nullptr;
  if (curve != MSM_CURVE_BLS12_377_G1 && curve != MSM_CURVE_ED_ON_BLS12_377 && curve != MSM_CURVE_BLS12_381_G1 &&
      curve != MSM_CURVE_PALLAS)
    return MSM_ERR_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return MSM_ERR_NO_DEVICE;
  msm_ctx* ctx = new (std::nothrow) msm_ctx();
  if (!ctx) return MSM_ERR_INTERNAL;
  ctx->curve = curve;
  ctx->device = device;
  try {
    ctx->helper.reset(new HelperThread());
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIPCHK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    for (auto& e : ctx->ev) HIPCHK(hipEventCreate(&e));
    HIPCHK(hipHostMalloc((void**)&ctx->h_info, 64 * 4, hipHostMallocDefault));
    for (auto& w : ctx->ws) {
      HIPCHK(hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
      for (auto& e : w.ev) HIPCHK(hipEventCreate(&e));
      HIPCHK(hipHostMalloc((void**)&w.h_info, 64 * 4, hipHostMallocDefault));
      HIPCHK(hipHostMalloc((void**)&w.h_part, 128 * 20 * 36 * 4, hipHostMallocDefault));
    }
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    // leave room for the resident points (144 B/point at 2^26 = 9.7 GB) and fragmentation
    ctx->ws_budget = (uint64_t)(free_b * 0.55);
    ctx->ensure(ctx->errflag, 16);
    HIPCHK(hipFuncSetAttribute((const void*)k_hist, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)k_scatter_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  } catch (const HipFail& f) {
    fprintf(stderr, "msm_ctx_create: HIP error %s at line %d\n", hipGetErrorString(f.e), f.line);
    msm_ctx_destroy(ctx);
    return MSM_ERR_HIP;
  } catch (...) {
    msm_ctx_destroy(ctx);
    return MSM_ERR_INTERNAL;
  }
  ctx->hc.F.init(curve == MSM_CURVE_ED_ON_BLS12_377 ? Fp377::PW : curve_info(curve).pw);   // (the Edwards context uses hte)
  ctx->k_dev_to_host = ctx->hc.F.pow2(2 * ctx->hc.F.radix_bits() - 30 * ctx->nl());   // device radix 2^(30 NL); host radix 2^384 or 2^256
  {
    uint32_t pw[12] = {0};
    for (int i = 0; i < 8; i++) pw[i] = Fp253::PW[i];
    ctx->hte.init(pw, 3021);
    ctx->k_te_to_host = ctx->hte.F.pow2(2 * ctx->hte.F.radix_bits() - 270);
  }
  *out = ctx;
  return MSM_OK;
}

void msm_ctx_destroy(msm_ctx* ctx) {
  if (!ctx) return;
  ctx->fan.clear();
  for (msm_ctx* c : ctx->children) msm_ctx_destroy(c);
  ctx->children.clear();
  (void)hipSetDevice(ctx->device);
  for (size_t i = 0; i < ctx->sets.size(); i++)
    if ((int)i != ctx->cur_set) ctx->release(ctx->sets[i].rows);
  for (void* p : ctx->allocs) (void)hipFree(p);
  ctx->allocs.clear();
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  for (DevBuf* b : {&ctx->rows, &ctx->scal, &ctx->errflag, &ctx->misc}) ctx->release(*b);
  for (auto& w : ctx->ws) {
    if (w.stream) (void)hipStreamSynchronize(w.stream);
    for (DevBuf* b : w.all) ctx->release(*b);
    if (w.h_info) (void)hipHostFree(w.h_info);
    if (w.h_part) (void)hipHostFree(w.h_part);
    for (auto& e : w.ev) if (e) (void)hipEventDestroy(e);
    if (w.stream) (void)hipStreamDestroy(w.stream);
  }
  if (ctx->h_info) (void)hipHostFree(ctx->h_info);
  if (ctx->stage_pin) (void)hipHostFree(ctx->stage_pin);
  for (auto& st : ctx->stage_stream) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
  for (auto& row : ctx->stage_ev) for (auto& e : row) if (e) (void)hipEventDestroy(e);
  for (auto& row : ctx->piece_ev) for (auto& e : row) if (e) (void)hipEventDestroy(e);
  for (auto& e : ctx->ev) if (e) (void)hipEventDestroy(e);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char* msm_last_error(const msm_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

static int set_points_one(msm_ctx* ctx, const void* points, uint64_t n, int on_device, int check_curve) {
  if (!ctx || (!points && n)) return fail(ctx, MSM_ERR_ARG, "msm_set_points: null argument");
  if (n >= (1ull << 30)) return fail(ctx, MSM_ERR_ARG, "msm_set_points: n must be < 2^30");
  const bool te = ctx->is_te();
  const size_t wire_bytes = 2 * ctx->coord_bytes();   // x || y, little-endian
  const size_t row_words = te ? te::TE_ROW_WORDS : ROW_WORDS;
  try {
    HIPCHK(hipSetDevice(ctx->device));
    ctx->n_points = 0;
    ctx->ensure(ctx->rows, std::max<uint64_t>(n, 1) * row_words * 4);
    const uint32_t* d_wire = (const uint32_t*)points;
    if (!on_device && n) {
      ctx->ensure(ctx->misc, n * wire_bytes);
      upload_staged(ctx, ctx->misc.p, points, n * wire_bytes);
      d_wire = (const uint32_t*)ctx->misc.p;
    }
    HIPCHK(hipMemsetAsync(ctx->errflag.p, 0, 4, ctx->stream));
    if (n) {
      uint64_t grid = (n + 255) / 256;
      if (te)
        hipLaunchKernelGGL(te::k_te_points_from_wire, dim3((uint32_t)grid), dim3(256), 0, ctx->stream, (uint32_t*)ctx->rows.p, d_wire,
                           n, check_curve, (uint32_t*)ctx->errflag.p);
      else
        W_LAUNCH(ctx, k_points_from_wire, dim3((uint32_t)grid), dim3(256), 0, ctx->stream, (uint32_t*)ctx->rows.p, d_wire, n,
                           check_curve, (uint32_t*)ctx->errflag.p);
    }
    HIPCHK(hipMemcpyAsync(ctx->h_info, ctx->errflag.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipGetLastError());
    if (!on_device) ctx->release(ctx->misc);
    if (ctx->h_info[0] & 1) return fail(ctx, MSM_ERR_POINT, "msm_set_points: coordinate >= p");
    if (ctx->h_info[0] & 2) return fail(ctx, MSM_ERR_POINT, "msm_set_points: point not on curve");
    ctx->n_points = n;
  } MSM_CATCH_ALL(ctx)
  return MSM_OK;
}

int msm_set_points(msm_ctx* ctx, const void* points, uint64_t n, int on_device, int check_curve) {
  if (!ctx || ctx->children.empty()) return set_points_one(ctx, points, n, on_device, check_curve);
  // multi-device context: every device keeps the whole point set (an MSM then runs over a share of the points per device by
  // default, or over a share of the windows with msm_opts.by_window: either way without moving points)
  try {
    std::vector<uint8_t> host;
    const void* src = points;
    if (on_device && n) {   // the buffer lives on devices[0]: the other devices take it through the host
      host.resize((size_t)n * 2 * ctx->coord_bytes());
      HIPCHK(hipSetDevice(ctx->device));
      HIPCHK(hipMemcpy(host.data(), points, host.size(), hipMemcpyDeviceToHost));
      src = host.data();
    }
    return on_all_devices(ctx, [&](msm_ctx* c) {
      return (c == ctx) ? set_points_one(c, points, n, on_device, check_curve) : set_points_one(c, src, n, 0, check_curve);
    });
  } MSM_CATCH_ALL(ctx)
}

int msm_plan(const msm_ctx* ctx, uint64_t n, const msm_opts* opts, int32_t* c_out, int32_t* K_out) {
  Plan pl;   // plain arithmetic: nothing here can throw
  int rc = make_plan(ctx, n, opts, pl);
  if (rc) return rc;
  if (c_out) *c_out = pl.c;
  if (K_out) *K_out = pl.K;
  return MSM_OK;
}

int msm_window_sums(msm_ctx* ctx, const void* scalars, uint64_t n, int on_device, const msm_opts* opts, uint8_t* partials_out,
                    msm_result* stats) {
  if (!ctx || !partials_out || (!scalars && n)) return fail(ctx, MSM_ERR_ARG, "msm_window_sums: null argument");
  if ((opts ? opts->point_lo : 0) + n > ctx->n_points)
    return fail(ctx, MSM_ERR_NO_POINTS, "msm_window_sums: points [%llu, +%llu) but %llu resident points",
                (unsigned long long)(opts ? opts->point_lo : 0), (unsigned long long)n, (unsigned long long)ctx->n_points);
  Plan pl;
  if (make_plan(ctx, n, opts, pl)) return fail(ctx, MSM_ERR_ARG, "msm_window_sums: bad window size");
  int k_lo = opts ? opts->k_lo : 0, k_hi = opts ? opts->k_hi : 0;
  if (k_lo == 0 && k_hi == 0) k_hi = pl.K;
  if (k_lo < 0 || k_hi > pl.K || k_lo >= k_hi) return fail(ctx, MSM_ERR_ARG, "msm_window_sums: bad window shard [%d, %d) of %d", k_lo, k_hi, pl.K);
  if (stats) memset(stats, 0, sizeof(*stats));
  try {
    HIPCHK(hipSetDevice(ctx->device));
    std::vector<uint32_t> words;
    if (ctx->is_te()) {
      // extended point (X : Y : Z : T) sent as X || Y || Z; the receiver rebuilds T (msm_combine: T Z = X Y)
      if (n) any_window_sums(ctx, scalars, n, on_device, opts, k_lo, k_hi, pl, words, stats);
      const auto& C = ctx->hte;
      msm_host::Fe6 one = {{1, 0, 0, 0, 0, 0}}, t;
      for (int k = 0; k < k_hi - k_lo; k++) {
        const msm_host::Ext6 P = n ? te_partial_to_host(ctx, &words[(size_t)k * 32]) : C.zero();
        C.F.mul(t, P.X, one); fe6_to_bytes(partials_out + (size_t)k * 144, t);
        C.F.mul(t, P.Y, one); fe6_to_bytes(partials_out + (size_t)k * 144 + 48, t);
        C.F.mul(t, P.Z, one); fe6_to_bytes(partials_out + (size_t)k * 144 + 96, t);
      }
      if (stats) { stats->c = pl.c; stats->K = pl.K; }
      return MSM_OK;
    }
    if (n == 0) {
      words.assign((size_t)(k_hi - k_lo) * 36, 0);
    } else {
      any_window_sums(ctx, scalars, n, on_device, opts, k_lo, k_hi, pl, words, stats);
    }
    // to 48-byte canonical integers (leave device Montgomery form on the host)
    for (int k = 0; k < k_hi - k_lo; k++) {
      const uint32_t* w = &words[(size_t)k * 36];
      bool zero_z = true;
      for (int j = 0; j < 12; j++) zero_z &= w[24 + j] == 0;
      msm_host::Proj6 P;
      if (n == 0 || zero_z) P = ctx->hc.zero();
      else P = partial_to_host(ctx, w);
      msm_host::Fe6 one = {{1, 0, 0, 0, 0, 0}}, t;
      ctx->hc.F.mul(t, P.X, one); fe6_to_bytes(partials_out + (size_t)k * 144, t);
      ctx->hc.F.mul(t, P.Y, one); fe6_to_bytes(partials_out + (size_t)k * 144 + 48, t);
      ctx->hc.F.mul(t, P.Z, one); fe6_to_bytes(partials_out + (size_t)k * 144 + 96, t);
    }
    if (stats) { stats->c = pl.c; stats->K = pl.K; }
  } MSM_CATCH_ALL(ctx)
  return MSM_OK;
}

namespace {
// host curve constants without a context (rank 0 of a sharded run may combine without touching a GPU)
const msm_host::Curve6* static_host_curve(int curve) {
  static msm_host::Curve6 hc[4];
  static std::atomic<int> ready[4];
  if (curve < 0 || curve > 3 || curve == MSM_CURVE_ED_ON_BLS12_377) return nullptr;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  if (!ready[curve].load()) {
    hc[curve].F.init(curve_info(curve).pw);
    ready[curve].store(1);
  }
  return &hc[curve];
}
int combine_impl(msm_ctx* ctx, const msm_host::Curve6& C, const uint8_t* partials, int32_t K, int32_t c, msm_result* out, int32_t G);
}  // namespace

namespace {
// twisted Edwards: (X : Y : Z) in, T rebuilt as (X Z : Y Z : Z^2 : X Y), then the unified-addition Horner
int te_combine_impl(const uint8_t* partials, int32_t K, int32_t c, msm_result* out, int32_t G = 1) {
  static msm_host::TeCurve6 C;
  static std::once_flag once;
  std::call_once(once, [] {
    uint32_t pw[12] = {0};
    for (int i = 0; i < 8; i++) pw[i] = Fp253::PW[i];
    C.init(pw, 3021);
  });
  std::vector<msm_host::Ext6> P(K);
  for (int k = 0; k < K; k++) {
    P[k] = C.zero();
    for (int g = 0; g < G; g++) {   // group g's sum of window k
      msm_host::Fe6 t[3];
      for (int j = 0; j < 3; j++) {
        const uint8_t* b = partials + ((size_t)g * K + k) * 144 + 48 * j;
        for (int i = 0; i < 6; i++) {
          uint64_t v = 0;
          for (int q = 0; q < 8; q++) v |= (uint64_t)b[8 * i + q] << (8 * q);
          t[j].v[i] = v;
        }
        if (msm_host::Field6::ge(t[j], C.F.p)) return MSM_ERR_ARG;
        C.F.mul(t[j], t[j], C.F.r2);
      }
      msm_host::Ext6 Q;
      C.F.mul(Q.X, t[0], t[2]);
      C.F.mul(Q.Y, t[1], t[2]);
      C.F.mul(Q.Z, t[2], t[2]);
      C.F.mul(Q.T, t[0], t[1]);
      P[k] = G == 1 ? Q : C.add(P[k], Q);
    }
  }
  memset(out, 0, sizeof(*out));
  te_horner_points(C, P, c, out);
  out->c = c;
  out->K = K;
  return MSM_OK;
}
}  // namespace

int msm_combine(msm_ctx* ctx, const uint8_t* partials, int32_t K, int32_t c, msm_result* out) {
  // pure host arithmetic: ctx may be NULL (then BLS12-377 G1; msm_combine_curve names the curve without a context)
  if (!partials || !out || K <= 0 || c <= 0) return fail(ctx, MSM_ERR_ARG, "msm_combine: bad argument");
  try {
    if (ctx && ctx->is_te()) {
      int rc = te_combine_impl(partials, K, c, out);
      return rc ? fail(ctx, rc, "msm_combine: coordinate >= p") : MSM_OK;
    }
    return combine_impl(ctx, ctx ? ctx->hc : *static_host_curve(MSM_CURVE_BLS12_377_G1), partials, K, c, out, 1);
  } MSM_CATCH_ALL(ctx)
}

int msm_combine_groups(int curve, const uint8_t* partials, int32_t G, int32_t K, int32_t c, msm_result* out) {
  if (!partials || !out || K <= 0 || c <= 0 || G <= 0) return MSM_ERR_ARG;
  msm_ctx* const no_ctx = nullptr;
  try {
    if (curve == MSM_CURVE_ED_ON_BLS12_377) return te_combine_impl(partials, K, c, out, G);
    const msm_host::Curve6* C = static_host_curve(curve);
    if (!C) return MSM_ERR_ARG;
    return combine_impl(nullptr, *C, partials, K, c, out, G);
  } MSM_CATCH_ALL(no_ctx)
}

int msm_combine_curve(int curve, const uint8_t* partials, int32_t K, int32_t c, msm_result* out) {
  return msm_combine_groups(curve, partials, 1, K, c, out);
}

namespace {
int combine_impl(msm_ctx* ctx, const msm_host::Curve6& C, const uint8_t* partials, int32_t K, int32_t c, msm_result* out, int32_t G) {
  std::vector<msm_host::Proj6> P(K);
  for (int k = 0; k < K; k++) {
    P[k] = C.zero();
    for (int g = 0; g < G; g++) {   // group g's sum of window k
      msm_host::Fe6 t[3];
      for (int j = 0; j < 3; j++) {
        const uint8_t* b = partials + ((size_t)g * K + k) * 144 + 48 * j;
        for (int i = 0; i < 6; i++) {
          uint64_t v = 0;
          for (int q = 0; q < 8; q++) v |= (uint64_t)b[8 * i + q] << (8 * q);
          t[j].v[i] = v;
        }
        if (msm_host::Field6::ge(t[j], C.F.p)) return fail(ctx, MSM_ERR_ARG, "msm_combine: coordinate >= p");
        C.F.mul(t[j], t[j], C.F.r2);  // to host Montgomery form
      }
      msm_host::Proj6 Q;
      Q.X = t[0]; Q.Y = t[1]; Q.Z = t[2];
      P[k] = G == 1 ? Q : C.add(P[k], Q);
    }
  }
  memset(out, 0, sizeof(*out));
  horner_to_affine(C, P, c, out);
  out->c = c;
  out->K = K;
  return MSM_OK;
}
}  // namespace

static int run_impl(msm_ctx* ctx, const void* scalars, const void* const* placed, uint64_t n, int on_device, const msm_opts* opts,
                    msm_result* out, const char* who) {
  if ((opts ? opts->point_lo : 0) + n > ctx->n_points)
    return fail(ctx, MSM_ERR_NO_POINTS, "%s: points [%llu, +%llu) but %llu resident points", who,
                (unsigned long long)(opts ? opts->point_lo : 0), (unsigned long long)n, (unsigned long long)ctx->n_points);
  // Host scalars of a big call arrive range by range of the points (PieceUpload) and every range runs with the call's window:
  // the first ranges are a sixteenth and three sixteenths of the input, so the window is picked for an eighth of the input
  // rather than for all of it (2^26: c = 16 for every range 164.8 ms, c = 22 180.6 -- a 2^22-point range under 2^21 buckets
  // per window)
  msm_opts piped;
  if (!placed && !on_device && n >= (1ull << 24) && !(opts && opts->c) && !ctx->is_te()) {
    if (opts) piped = *opts; else memset(&piped, 0, sizeof piped);
    piped.c = pick_window(false, n / 8, (opts && opts->no_glv) ? 0 : curve_info(ctx->curve).glv_max_bits);
    opts = &piped;
  }
  Plan pl;
  if (make_plan(ctx, n, opts, pl)) return fail(ctx, MSM_ERR_ARG, "%s: bad window size", who);
  pl.merged = true;
  memset(out, 0, sizeof(*out));
  out->c = pl.c;
  out->K = pl.K;
  if (n == 0) {
    if (ctx->is_te()) out->y[0] = 1;   // identity (0, 1)
    else out->is_infinity = 1;
    return MSM_OK;
  }
  try {
    HIPCHK(hipSetDevice(ctx->device));
    std::vector<uint32_t> words;
    any_window_sums(ctx, scalars, n, on_device, opts, 0, pl.K, pl, words, out, placed);
    HIPCHK(hipEventRecord(ctx->ev[10], ctx->stream));
    if (ctx->is_te()) {
      te_horner_to_affine(ctx, words, pl.K, pl.c, out);
    } else {
      std::vector<msm_host::Proj6> P(pl.K);
      for (int k = 0; k < pl.K; k++) P[k] = partial_to_host(ctx, &words[(size_t)k * 36]);
      horner_to_affine(ctx->hc, P, pl.c, out);
    }
    HIPCHK(hipEventRecord(ctx->ev[11], ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    float ms;
    HIPCHK(hipEventElapsedTime(&ms, ctx->ev[10], ctx->ev[11]));
    out->phase_ms[MSM_T_FINAL] = ms;
    out->phase_ms[MSM_T_TOTAL] += ms;
  } MSM_CATCH_ALL(ctx)
  return MSM_OK;
}

int msm_run(msm_ctx* ctx, const void* scalars, uint64_t n, int on_device, const msm_opts* opts, msm_result* out) {
  if (!ctx || !out || (!scalars && n)) return fail(ctx, MSM_ERR_ARG, "msm_run: null argument");
  return run_impl(ctx, scalars, nullptr, n, on_device, opts, out, "msm_run");
}

int msm_run_placed(msm_ctx* ctx, const void* const* dev_scalars, uint64_t n, const msm_opts* opts, msm_result* out) {
  if (!ctx || !out || !dev_scalars) return fail(ctx, MSM_ERR_ARG, "msm_run_placed: null argument");
  if (opts && opts->by_window && !ctx->children.empty())
    return fail(ctx, MSM_ERR_ARG, "msm_run_placed: placed scalars are the shares of a points split (by_window must be 0)");
  const int ndev = 1 + (int)ctx->children.size();
  for (int d = 0; d < ndev; d++)
    if (!dev_scalars[d] && n * (uint64_t)(d + 1) / ndev > n * (uint64_t)d / ndev)
      return fail(ctx, MSM_ERR_ARG, "msm_run_placed: no scalars for device %d", d);
  return run_impl(ctx, nullptr, dev_scalars, n, 1, opts, out, "msm_run_placed");
}

int msm_get_points(msm_ctx* ctx, uint64_t first, uint64_t count, uint8_t* out_xy) {
  if (!ctx || !out_xy || first + count > ctx->n_points) return fail(ctx, MSM_ERR_ARG, "msm_get_points: bad argument");
  if (ctx->is_te()) {
    try {
      HIPCHK(hipSetDevice(ctx->device));
      std::vector<uint32_t> rows((size_t)count * te::TE_ROW_WORDS);
      if (count)
        HIPCHK(hipMemcpy(rows.data(), (const uint32_t*)ctx->rows.p + first * te::TE_ROW_WORDS, rows.size() * 4, hipMemcpyDeviceToHost));
      msm_host::Fe6 one = {{1, 0, 0, 0, 0, 0}};
      for (uint64_t i = 0; i < count; i++)
        for (int j = 0; j < 2; j++) {
          msm_host::Fe6 t = {{0, 0, 0, 0, 0, 0}};
          const uint32_t* w = &rows[(size_t)i * te::TE_ROW_WORDS + 8 * j];
          for (int q = 0; q < 4; q++) t.v[q] = (uint64_t)w[2 * q] | ((uint64_t)w[2 * q + 1] << 32);
          ctx->hte.F.mul(t, t, ctx->k_te_to_host);
          ctx->hte.F.mul(t, t, one);
          for (int q = 0; q < 4; q++)
            for (int b = 0; b < 8; b++) out_xy[i * 64 + 32 * j + 8 * q + b] = (uint8_t)(t.v[q] >> (8 * b));
        }
    } MSM_CATCH_ALL(ctx)
    return MSM_OK;
  }
  try {
    HIPCHK(hipSetDevice(ctx->device));
    std::vector<uint32_t> rows((size_t)count * ROW_WORDS);
    if (count)
      HIPCHK(hipMemcpy(rows.data(), (const uint32_t*)ctx->rows.p + first * ROW_WORDS, rows.size() * 4, hipMemcpyDeviceToHost));
    const int nw = ctx->nw();
    const size_t cb = ctx->coord_bytes();
    memset(out_xy, 0, (size_t)count * 2 * cb);
    msm_host::Fe6 one = {{1, 0, 0, 0, 0, 0}};
    for (uint64_t i = 0; i < count; i++) {
      const uint32_t* row = &rows[(size_t)i * ROW_WORDS];
      if (row[nw - 1] == INF_WORD) continue;
      for (int j = 0; j < 2; j++) {
        msm_host::Fe6 t;
        words_to_fe6(t, row + nw * j, nw);
        ctx->hc.F.mul(t, t, ctx->k_dev_to_host);  // host Montgomery
        ctx->hc.F.mul(t, t, one);                 // plain
        uint8_t b48[48];
        fe6_to_bytes(b48, t);
        memcpy(out_xy + i * 2 * cb + cb * j, b48, cb);
      }
    }
  } MSM_CATCH_ALL(ctx)
  return MSM_OK;
}

int msm_test_fp(msm_ctx* ctx, int op, const uint8_t* a, const uint8_t* b, uint8_t* out, uint64_t n) {
  if (!ctx || !a || !b || !out) return fail(ctx, MSM_ERR_ARG, "msm_test_fp: null argument");
  const size_t nb = ctx->coord_bytes();
  try {
    HIPCHK(hipSetDevice(ctx->device));
    ctx->ensure(ctx->misc, n * nb * 3 + 64);
    uint8_t* d = (uint8_t*)ctx->misc.p;
    HIPCHK(hipMemcpyAsync(d, a, n * nb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d + n * nb, b, n * nb, hipMemcpyHostToDevice, ctx->stream));
    if (ctx->is_te())
      hipLaunchKernelGGL(te::k_te_test_fp, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, (uint32_t*)(d + 2 * n * nb),
                         (const uint32_t*)d, (const uint32_t*)(d + n * nb), (uint32_t)n, op);
    else
      W_LAUNCH(ctx, k_test_fp, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, (uint32_t*)(d + 2 * n * nb),
                         (const uint32_t*)d, (const uint32_t*)(d + n * nb), (uint32_t)n, op);
    HIPCHK(hipMemcpyAsync(out, d + 2 * n * nb, n * nb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipGetLastError());
  } MSM_CATCH_ALL(ctx)
  return MSM_OK;
}

int msm_test_batch_inverse(msm_ctx* ctx, const uint8_t* xs, uint8_t* out, uint64_t n, uint32_t per_lane) {
  if (!ctx || !xs || !out || per_lane == 0) return fail(ctx, MSM_ERR_ARG, "msm_test_batch_inverse: bad argument");
  try {
    HIPCHK(hipSetDevice(ctx->device));
    const size_t nb = ctx->coord_bytes();
    ctx->ensure(ctx->misc, n * 2 * nb + 64);
    uint8_t* d = (uint8_t*)ctx->misc.p;
    HIPCHK(hipMemcpyAsync(d, xs, n * nb, hipMemcpyHostToDevice, ctx->stream));
    uint64_t lanes = (n + per_lane - 1) / per_lane;
    if (ctx->is_te())
      hipLaunchKernelGGL((k_test_batch_inverse<te::CvEdField>), dim3((uint32_t)((lanes + 255) / 256)), dim3(256), 0, ctx->stream,
                         (uint32_t*)(d + n * nb), (const uint32_t*)d, (uint32_t)n, per_lane);
    else
      W_LAUNCH(ctx, k_test_batch_inverse, dim3((uint32_t)((lanes + 255) / 256)), dim3(256), 0, ctx->stream,
                         (uint32_t*)(d + n * nb), (const uint32_t*)d, (uint32_t)n, per_lane);
    HIPCHK(hipMemcpyAsync(out, d + n * nb, n * nb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipGetLastError());
  } MSM_CATCH_ALL(ctx)
  return MSM_OK;
}

int msm_test_glv(msm_ctx* ctx, const uint8_t* scalars, uint8_t* out, uint64_t n) {
  if (!ctx || !scalars || !out) return fail(ctx, MSM_ERR_ARG, "msm_test_glv: null argument");
  if (ctx->is_te()) return fail(ctx, MSM_ERR_ARG, "msm_test_glv: the twisted Edwards path has no GLV step");
  try {
    HIPCHK(hipSetDevice(ctx->device));
    ctx->ensure(ctx->misc, n * 72 + 64);
    uint8_t* d = (uint8_t*)ctx->misc.p;
    HIPCHK(hipMemcpyAsync(d, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream));
    W_LAUNCH(ctx, k_test_glv, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, (uint32_t*)(d + n * 32),
                       (const uint32_t*)d, (uint32_t)n);
    HIPCHK(hipMemcpyAsync(out, d + n * 32, n * 40, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipGetLastError());
  } MSM_CATCH_ALL(ctx)
  return MSM_OK;
}

int msm_test_batch_add(msm_ctx* ctx, const uint8_t* g, const uint8_t* h, uint8_t* out, uint64_t n) {
  if (!ctx || !g || !h || !out || n == 0) return fail(ctx, MSM_ERR_ARG, "msm_test_batch_add: bad argument");
  if (ctx->is_te()) {
    // unified extended addition of the gather round (te_add_rows, src/curve-twisted-edwards.ts:84-165): n pairs of
    // 64-byte affine points in, n affine sums out
    try {
      HIPCHK(hipSetDevice(ctx->device));
      DevBuf rows, wire, slots, outb;
      ctx->ensure(wire, 2 * n * 64);
      ctx->ensure(rows, 2 * n * te::TE_ROW_WORDS * 4);
      ctx->ensure(slots, 2 * n * 4);
      ctx->ensure(outb, n * 128);
      std::vector<uint8_t> inter(2 * n * 64);
      std::vector<uint32_t> sl(2 * n);
      for (uint64_t i = 0; i < n; i++) {
        memcpy(&inter[(2 * i) * 64], g + i * 64, 64);
        memcpy(&inter[(2 * i + 1) * 64], h + i * 64, 64);
        sl[2 * i] = (uint32_t)((2 * i) << 1);
        sl[2 * i + 1] = (uint32_t)((2 * i + 1) << 1);
      }
      HIPCHK(hipMemcpyAsync(wire.p, inter.data(), inter.size(), hipMemcpyHostToDevice, ctx->stream));
      HIPCHK(hipMemcpyAsync(slots.p, sl.data(), sl.size() * 4, hipMemcpyHostToDevice, ctx->stream));
      HIPCHK(hipMemsetAsync(ctx->errflag.p, 0, 4, ctx->stream));
      hipLaunchKernelGGL(te::k_te_points_from_wire, dim3((uint32_t)((2 * n + 255) / 256)), dim3(256), 0, ctx->stream,
                         (uint32_t*)rows.p, (const uint32_t*)wire.p, 2 * n, 0, (uint32_t*)ctx->errflag.p);
      BatchArgs a{};
      a.points = (const uint32_t*)rows.p;
      a.slots = (const uint32_t*)slots.p;
      a.out = (uint4*)outb.p;
      a.out_cap = n;
      a.n_out = n;
      a.steps = 1;
      hipLaunchKernelGGL(te::k_te_add<MODE_GATHER>, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, a);
      std::vector<uint32_t> planes(n * 32);
      HIPCHK(hipMemcpyAsync(planes.data(), outb.p, n * 128, hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(hipStreamSynchronize(ctx->stream));
      HIPCHK(hipGetLastError());
      const auto& C = ctx->hte;
      msm_host::Fe6 one = {{1, 0, 0, 0, 0, 0}};
      for (uint64_t e = 0; e < n; e++) {
        msm_host::Fe6 co[3];   // X, Y, Z
        for (int j = 0; j < 3; j++) {
          msm_host::Fe6 t = {{0, 0, 0, 0, 0, 0}};
          for (int pl = 0; pl < 2; pl++)
            for (int q = 0; q < 2; q++) {
              const uint32_t* w = &planes[((uint64_t)(2 * j + pl) * n + e) * 4 + 2 * q];
              t.v[2 * pl + q] = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
            }
          if (msm_host::Field6::ge(t, C.F.p)) C.F.sub_raw(t, t, C.F.p);
          C.F.mul(co[j], t, ctx->k_te_to_host);
        }
        msm_host::Fe6 zi, x, y;
        C.F.inv(zi, co[2]);
        C.F.mul(x, co[0], zi);
        C.F.mul(y, co[1], zi);
        C.F.mul(x, x, one);
        C.F.mul(y, y, one);
        uint8_t xb[48], yb[48];
        fe6_to_bytes(xb, x);
        fe6_to_bytes(yb, y);
        memcpy(out + e * 64, xb, 32);
        memcpy(out + e * 64 + 32, yb, 32);
      }
      for (DevBuf* b : {&rows, &wire, &slots, &outb}) ctx->release(*b);
    } MSM_CATCH_ALL(ctx)
    return MSM_OK;
  }
  try {
    HIPCHK(hipSetDevice(ctx->device));
    // rows for 2n points: pair e = (row 2e, row 2e + 1), gathered through identity payload slots
    DevBuf rows, wire, slots, outb, scr;
    const size_t pb = 2 * ctx->coord_bytes();   // wire point; a tree node has the same size
    ctx->ensure(wire, 2 * n * pb);
    ctx->ensure(rows, 2 * n * ROW_WORDS * 4);
    ctx->ensure(slots, 2 * n * 4);
    ctx->ensure(outb, n * pb);
    std::vector<uint8_t> inter(2 * n * pb);
    std::vector<uint32_t> sl(2 * n);
    for (uint64_t i = 0; i < n; i++) {
      memcpy(&inter[(2 * i) * pb], g + i * pb, pb);
      memcpy(&inter[(2 * i + 1) * pb], h + i * pb, pb);
      sl[2 * i] = (uint32_t)((2 * i) << 2);
      sl[2 * i + 1] = (uint32_t)((2 * i + 1) << 2);
    }
    HIPCHK(hipMemcpyAsync(wire.p, inter.data(), 2 * n * pb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(slots.p, sl.data(), 2 * n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->errflag.p, 0, 4, ctx->stream));
    W_LAUNCH(ctx, k_points_from_wire, dim3((uint32_t)((2 * n + 255) / 256)), dim3(256), 0, ctx->stream, (uint32_t*)rows.p,
                       (const uint32_t*)wire.p, 2 * n, 0, (uint32_t*)ctx->errflag.p);
    RoundGeom gm = round_geom(ctx, n);
    gm.steps = (uint32_t)std::min<uint64_t>(n, 3);  // exercise the shared inversion with a few pairs per lane
    uint64_t threads = (n + gm.steps - 1) / gm.steps;
    gm.grid = (uint32_t)((threads + 255) / 256);
    gm.T = (uint64_t)gm.grid * 256;
    ctx->ensure(scr, (size_t)gm.steps * NL * gm.T * 4);
    BatchArgs a{};
    a.points = (const uint32_t*)rows.p;
    a.slots = (const uint32_t*)slots.p;
    a.y_off = 4 * ctx->nw();
    a.out = (uint4*)outb.p;
    a.out_cap = n;
    a.scratch = (uint32_t*)scr.p;
    a.sstride = gm.T;
    a.n_out = n;
    a.steps = gm.steps;
    W_LAUNCH_MODE(ctx, k_batch_add, MODE_GATHER, dim3(gm.grid), dim3(256), 0, ctx->stream, a);
    std::vector<uint32_t> planes(n * pb / 4);
    HIPCHK(hipMemcpyAsync(planes.data(), outb.p, n * pb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipGetLastError());
    for (uint64_t e = 0; e < n; e++) plane_element_to_wire(ctx, planes.data(), n, e, out + e * pb);
    for (DevBuf* b : {&rows, &wire, &slots, &outb, &scr}) ctx->release(*b);
  } MSM_CATCH_ALL(ctx)
  return MSM_OK;
}

static int generate_points_one(msm_ctx* ctx, uint64_t n, uint64_t seed, uint8_t* a_out) {
  try {
    HIPCHK(hipSetDevice(ctx->device));
    if (ctx->is_te()) return msm_gen::generate_points_te(ctx, n, seed, a_out);
    return msm_gen::generate_points(ctx, n, seed, a_out);
  } MSM_CATCH_ALL(ctx)
}

int msm_test_fp_raw(msm_ctx* ctx, int op, const uint32_t* a, const uint32_t* b, uint32_t* out, uint64_t n) {
  if (!ctx || !a || !b || !out) return fail(ctx, MSM_ERR_ARG, "msm_test_fp_raw: null argument");
  if (op != MSM_OP_MUL && op != MSM_OP_SQR) return fail(ctx, MSM_ERR_ARG, "msm_test_fp_raw: op must be MSM_OP_MUL or MSM_OP_SQR");
  const size_t nb = (size_t)ctx->nl() * 4;
  try {
    HIPCHK(hipSetDevice(ctx->device));
    ctx->ensure(ctx->misc, n * nb * 3 + 64);
    uint8_t* d = (uint8_t*)ctx->misc.p;
    HIPCHK(hipMemcpyAsync(d, a, n * nb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d + n * nb, b, n * nb, hipMemcpyHostToDevice, ctx->stream));
    const dim3 grid((uint32_t)((n + 255) / 256));
    if (ctx->is_te())
      hipLaunchKernelGGL(te::k_te_test_fp_raw, grid, dim3(256), 0, ctx->stream, (uint32_t*)(d + 2 * n * nb), (const uint32_t*)d,
                         (const uint32_t*)(d + n * nb), (uint32_t)n, op);
    else
      W_LAUNCH(ctx, k_test_fp_raw, grid, dim3(256), 0, ctx->stream, (uint32_t*)(d + 2 * n * nb), (const uint32_t*)d,
                         (const uint32_t*)(d + n * nb), (uint32_t)n, op);
    HIPCHK(hipMemcpyAsync(out, d + 2 * n * nb, n * nb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipGetLastError());
    return MSM_OK;
  } MSM_CATCH_ALL(ctx)
}

int msm_test_curve_op(msm_ctx* ctx, int op, const uint8_t* p, const uint8_t* q, uint8_t* out, uint64_t n) {
  if (!ctx || !p || !q || !out) return fail(ctx, MSM_ERR_ARG, "msm_test_curve_op: null argument");
  if (op < 0 || op > 2) return fail(ctx, MSM_ERR_ARG, "msm_test_curve_op: unknown operator");
  const size_t nb = ctx->is_te() ? 128 : 3 * ctx->coord_bytes();   // extended (X, Y, Z, T) or projective (X, Y, Z)
  try {
    HIPCHK(hipSetDevice(ctx->device));
    ctx->ensure(ctx->misc, n * nb * 3 + 64);
    uint8_t* d = (uint8_t*)ctx->misc.p;
    HIPCHK(hipMemcpyAsync(d, p, n * nb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d + n * nb, q, n * nb, hipMemcpyHostToDevice, ctx->stream));
    const dim3 grid((uint32_t)((n + 63) / 64));
    if (ctx->is_te())
      hipLaunchKernelGGL(te::k_te_test_curve_op, grid, dim3(64), 0, ctx->stream, (uint32_t*)(d + 2 * n * nb), (const uint32_t*)d,
                         (const uint32_t*)(d + n * nb), (uint32_t)n, op);
    else
      W_LAUNCH(ctx, k_test_curve_op, grid, dim3(64), 0, ctx->stream, (uint32_t*)(d + 2 * n * nb), (const uint32_t*)d,
                         (const uint32_t*)(d + n * nb), (uint32_t)n, op);
    HIPCHK(hipMemcpyAsync(out, d + 2 * n * nb, n * nb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipGetLastError());
    return MSM_OK;
  } MSM_CATCH_ALL(ctx)
}

int msm_test_bucket_reduce(msm_ctx* ctx, const uint8_t* buckets, int32_t K, uint32_t L, int mode, int c0, uint8_t* partials_out,
                           float* ms_out) {
  if (!ctx || !buckets || !partials_out || K <= 0 || L == 0 || (L & (L - 1)) || (mode != 0 && mode != 1))
    return fail(ctx, MSM_ERR_ARG, "msm_test_bucket_reduce: bad argument");
  if (ctx->is_te()) return fail(ctx, MSM_ERR_ARG, "msm_test_bucket_reduce: Weierstrass curves only");
  int cl = 0;
  while ((1u << cl) < L) cl++;
  if (c0 < 0 || c0 > cl) return fail(ctx, MSM_ERR_ARG, "msm_test_bucket_reduce: c0 must be in [0, log2 L]");
  try {
    HIPCHK(hipSetDevice(ctx->device));
    msm_ctx::Workspace& w = ctx->ws[0];
    hipStream_t s = w.stream;
    const uint64_t nb = (uint64_t)K * L;
    const uint64_t cap = nb + 2 * 257 * 512 + 256;   // plane capacity: idle lanes read (and ignore) past the end
    ScopedDevBuf wire, rows, planes, desc, scr;   // released on every path, a HIPCHK / MsmFail thrown in between included
    const size_t pb = 2 * ctx->coord_bytes();
    const int nw = ctx->nw(), np = nw / 4;
    ctx->ensure(wire, nb * pb);
    ctx->ensure(rows, nb * ROW_WORDS * 4);
    ctx->ensure(planes, cap * pb);
    HIPCHK(hipMemcpyAsync(wire.p, buckets, nb * pb, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(ctx->errflag.p, 0, 4, s));
    HIPCHK(hipMemsetAsync(planes.p, 0, cap * pb, s));
    W_LAUNCH(ctx, k_points_from_wire, dim3((uint32_t)((nb + 255) / 256)), dim3(256), 0, s, (uint32_t*)rows.p, (const uint32_t*)wire.p,
             nb, 0, (uint32_t*)ctx->errflag.p);
    ROWS_TO_PLANES(ctx, dim3((uint32_t)((nb + 255) / 256)), dim3(256), 0, s, (uint4*)planes.p, cap, (const uint32_t*)rows.p,
                   (uint32_t)nb);
    std::vector<uint32_t> parts((size_t)K * 36, 0);
    float ms = 0;
    if (mode == 0) {
      // every bucket holds exactly one element of the tree buffer: offsets 0, 1, 2, ...
      std::vector<uint32_t> off(nb + 1);
      for (uint64_t b = 0; b <= nb; b++) off[b] = (uint32_t)b;
      ctx->ensure(desc, (nb + 1) * 4);
      HIPCHK(hipMemcpyAsync(desc.p, off.data(), (nb + 1) * 4, hipMemcpyHostToDevice, s));
      HIPCHK(hipEventRecord(w.ev[3], s));
      reduce_buckets(ctx, w, (const uint4*)planes.p, cap, (const uint32_t*)desc.p, nullptr, L, K, parts.data());
      HIPCHK(hipEventElapsedTime(&ms, w.ev[3], w.ev[4]));
    } else {
      // The rounds of reduceBucketsAffine as (first operand, second operand) element lists; the sum replaces the first.
      // e(k, l) = k L + l - 1 for the 1-based bucket index l of the reference.
      const uint32_t L0 = 1u << c0, D = L / L0;
      std::vector<std::vector<uint32_t>> ga, gb;
      auto e = [&](int k, uint64_t l) { return (uint32_t)((uint64_t)k * L + l - 1); };
      auto round = [&](const std::function<void(int, std::vector<uint32_t>&, std::vector<uint32_t>&)>& fill) {
        std::vector<uint32_t> A, B;
        for (int k = 0; k < K; k++) fill(k, A, B);
        if (!A.empty()) { ga.push_back(std::move(A)); gb.push_back(std::move(B)); }
      };
      // linear part: suffix sums inside every chunk of L0 buckets (:563-588)
      for (uint32_t l = L0 - 1; l >= 1; l--)
        round([&](int k, std::vector<uint32_t>& A, std::vector<uint32_t>& B) {
          for (uint32_t d = 0; d < D; d++) { A.push_back(e(k, (uint64_t)d * L0 + l)); B.push_back(e(k, (uint64_t)d * L0 + l + 1)); }
        });
      // logarithmic part: chunk heads collect the chunks to their right, power-of-two spans (:590-615)
      for (uint64_t L1 = L0, D1 = D >> 1; D1 > 0; L1 <<= 1, D1 >>= 1)
        round([&](int k, std::vector<uint32_t>& A, std::vector<uint32_t>& B) {
          for (uint64_t d = 0; d < D1; d++) { A.push_back(e(k, d * 2 * L1 + 1)); B.push_back(e(k, (d * 2 + 1) * L1 + 1)); }
        });
      // doublings: every head is weighted with the number of buckets it stands for (:616-641)
      if (D > 1)
        for (int j = 0; j < c0; j++)
          round([&](int k, std::vector<uint32_t>& A, std::vector<uint32_t>& B) {
            for (uint32_t d = 1; d < D; d++) { A.push_back(e(k, (uint64_t)d * L0 + 1)); B.push_back(e(k, (uint64_t)d * L0 + 1)); }
          });
      for (uint64_t L1 = 2ull * L0, D1 = D >> 1; D1 > 1; L1 <<= 1, D1 >>= 1)
        round([&](int k, std::vector<uint32_t>& A, std::vector<uint32_t>& B) {
          for (uint64_t d = 1; d < D1; d++) { A.push_back(e(k, d * L1 + 1)); B.push_back(e(k, d * L1 + 1)); }
        });
      // the buckets now fill the triangle: one addition tree over all of them (:643-662)
      for (uint64_t m = 1; m < L; m *= 2)
        round([&](int k, std::vector<uint32_t>& A, std::vector<uint32_t>& B) {
          for (uint64_t l = 1; l < L; l += 2 * m) { A.push_back(e(k, l)); B.push_back(e(k, l + m)); }
        });
      size_t total = 0, biggest = 0;
      for (auto& v : ga) { total += v.size(); biggest = std::max(biggest, v.size()); }
      ctx->ensure(desc, std::max<size_t>(total, 1) * 8);
      std::vector<uint32_t> flat(2 * total);
      {
        size_t o = 0;
        for (size_t r = 0; r < ga.size(); r++) {
          for (size_t i = 0; i < ga[r].size(); i++) { flat[o + i] = (ga[r][i] << 1) | 1u; flat[total + o + i] = gb[r][i]; }
          o += ga[r].size();
        }
      }
      HIPCHK(hipMemcpyAsync(desc.p, flat.data(), flat.size() * 4, hipMemcpyHostToDevice, s));
      {
        const RoundGeom g = round_geom(ctx, std::max<uint64_t>(biggest, 1));
        ctx->ensure(scr, (size_t)g.steps * NL * g.T * 4);
      }
      HIPCHK(hipEventRecord(w.ev[3], s));
      size_t o = 0;
      for (size_t r = 0; r < ga.size(); r++) {
        const uint64_t np = ga[r].size();
        const RoundGeom g = round_geom(ctx, np);
        BatchArgs a{};
        a.in = (const uint4*)planes.p;
        a.in_cap = cap;
        a.out = (uint4*)planes.p;
        a.out_cap = cap;
        a.scratch = (uint32_t*)scr.p;
        a.sstride = g.T;
        a.n_out = np;
        a.steps = g.steps;
        a.desc = (const uint32_t*)desc.p + o;
        a.desc_b = (const uint32_t*)desc.p + total + o;
        a.inplace = 1;
        W_LAUNCH_MODE(ctx, k_batch_add, MODE_SEARCH, dim3(g.grid), dim3(256), 0, s, a);
        o += np;
      }
      HIPCHK(hipEventRecord(w.ev[4], s));
      // element e(k, 1) -> partial (X, Y, Z = 1 in device Montgomery form; all-zero = the identity)
      std::vector<uint32_t> el((size_t)K * 24);
      for (int k = 0; k < K; k++)
        for (int cpl = 0; cpl < 2 * np; cpl++)
          HIPCHK(hipMemcpyAsync(&el[(size_t)k * 24 + 4 * cpl], (const uint4*)planes.p + (uint64_t)cpl * cap + (uint64_t)k * L, 16,
                                hipMemcpyDeviceToHost, s));
      HIPCHK(hipStreamSynchronize(s));
      HIPCHK(hipGetLastError());
      HIPCHK(hipEventElapsedTime(&ms, w.ev[3], w.ev[4]));
      for (int k = 0; k < K; k++) {
        const uint32_t* q = &el[(size_t)k * 24];
        if (q[nw - 1] == INF_WORD) continue;   // identity: the partial stays all-zero
        memcpy(&parts[(size_t)k * 36], q, nw * 4);            // X, Y at words 0 and 12 of the partial (upper words zero)
        memcpy(&parts[(size_t)k * 36 + 12], q + nw, nw * 4);
        const msm_host::Fe6 one_dev = ctx->hc.F.pow2(30 * ctx->nl());   // Z = 1 in the form x and y are in: device Montgomery
        for (int q2 = 0; q2 < 6; q2++) {
          parts[(size_t)k * 36 + 24 + 2 * q2] = (uint32_t)one_dev.v[q2];
          parts[(size_t)k * 36 + 24 + 2 * q2 + 1] = (uint32_t)(one_dev.v[q2] >> 32);
        }
      }
    }
    for (int k = 0; k < K; k++) {
      const uint32_t* q = &parts[(size_t)k * 36];
      bool zero_z = true;
      for (int j = 0; j < 12; j++) zero_z &= q[24 + j] == 0;
      const msm_host::Proj6 P = zero_z ? ctx->hc.zero() : partial_to_host(ctx, q);
      msm_host::Fe6 one = {{1, 0, 0, 0, 0, 0}}, t;
      ctx->hc.F.mul(t, P.X, one); fe6_to_bytes(partials_out + (size_t)k * 144, t);
      ctx->hc.F.mul(t, P.Y, one); fe6_to_bytes(partials_out + (size_t)k * 144 + 48, t);
      ctx->hc.F.mul(t, P.Z, one); fe6_to_bytes(partials_out + (size_t)k * 144 + 96, t);
    }
    if (ms_out) *ms_out = ms;
    for (DevBuf* b : {&wire, &rows, &planes, &desc, &scr}) ctx->release(*b);
    return MSM_OK;
  } MSM_CATCH_ALL(ctx)
}

int msm_test_batch_add_mode(msm_ctx* ctx, const uint8_t* g, const uint8_t* h, uint8_t* out, uint64_t n, int mode, uint32_t steps) {
  if (!ctx || !g || !h || !out || n == 0 || steps == 0) return fail(ctx, MSM_ERR_ARG, "msm_test_batch_add_mode: bad argument");
  if (ctx->is_te() || (mode != MODE_REGULAR && mode != MODE_SEARCH))
    return fail(ctx, MSM_ERR_ARG, "msm_test_batch_add_mode: Weierstrass curves, mode 1 (regular) or 2 (search)");
  try {
    HIPCHK(hipSetDevice(ctx->device));
    // element 2e = G_e, 2e + 1 = H_e in plane layout; search mode: an all-zero H_e is passed as "no second operand"
    const uint64_t T = ((n + steps - 1) / steps + 255) / 256 * 256;
    const uint64_t in_cap = 2 * (uint64_t)steps * T;     // idle lanes of the last step read (and ignore) up to here
    const size_t pb = 2 * ctx->coord_bytes();            // wire point = tree node
    DevBuf rows, wire, planes, outb, scr, desc;
    ctx->ensure(wire, 2 * n * pb);
    ctx->ensure(rows, 2 * n * ROW_WORDS * 4);
    ctx->ensure(planes, in_cap * pb);
    ctx->ensure(outb, (uint64_t)steps * T * pb);
    ctx->ensure(scr, (size_t)steps * NL * T * 4);
    std::vector<uint8_t> inter(2 * n * pb);
    std::vector<uint32_t> hd(n);
    for (uint64_t i = 0; i < n; i++) {
      memcpy(&inter[(2 * i) * pb], g + i * pb, pb);
      memcpy(&inter[(2 * i + 1) * pb], h + i * pb, pb);
      bool hz = true;
      for (size_t j = 0; j < pb; j++) hz = hz && h[i * pb + j] == 0;
      hd[i] = (uint32_t)((2 * i) << 1) | (hz ? 0u : 1u);
    }
    HIPCHK(hipMemcpyAsync(wire.p, inter.data(), inter.size(), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->errflag.p, 0, 4, ctx->stream));
    HIPCHK(hipMemsetAsync(planes.p, 0, in_cap * pb, ctx->stream));
    W_LAUNCH(ctx, k_points_from_wire, dim3((uint32_t)((2 * n + 255) / 256)), dim3(256), 0, ctx->stream, (uint32_t*)rows.p,
                       (const uint32_t*)wire.p, 2 * n, 0, (uint32_t*)ctx->errflag.p);
    ROWS_TO_PLANES(ctx, dim3((uint32_t)((2 * n + 255) / 256)), dim3(256), 0, ctx->stream, (uint4*)planes.p, in_cap,
                   (const uint32_t*)rows.p, (uint32_t)(2 * n));
    BatchArgs a{};
    a.in = (const uint4*)planes.p;
    a.in_cap = in_cap;
    a.out = (uint4*)outb.p;
    a.out_cap = (uint64_t)steps * T;
    a.scratch = (uint32_t*)scr.p;
    a.sstride = T;
    a.n_out = n;
    a.steps = steps;
    if (mode == MODE_SEARCH) {
      ctx->ensure(desc, n * 4);
      HIPCHK(hipMemcpyAsync(desc.p, hd.data(), n * 4, hipMemcpyHostToDevice, ctx->stream));
      a.desc = (const uint32_t*)desc.p;
      W_LAUNCH_MODE(ctx, k_batch_add, MODE_SEARCH, dim3((uint32_t)(T / 256)), dim3(256), 0, ctx->stream, a);
    } else {
      W_LAUNCH_MODE(ctx, k_batch_add, MODE_REGULAR, dim3((uint32_t)(T / 256)), dim3(256), 0, ctx->stream, a);
    }
    std::vector<uint32_t> pl(a.out_cap * pb / 4);
    HIPCHK(hipMemcpyAsync(pl.data(), outb.p, pl.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipGetLastError());
    for (uint64_t e = 0; e < n; e++) plane_element_to_wire(ctx, pl.data(), a.out_cap, e, out + e * pb);
    for (DevBuf* b : {&rows, &wire, &planes, &outb, &scr, &desc}) ctx->release(*b);
    return MSM_OK;
  } MSM_CATCH_ALL(ctx)
}

int msm_generate_points(msm_ctx* ctx, uint64_t n, uint64_t seed, uint8_t* a_out) {
  if (!ctx) return MSM_ERR_ARG;
  if (ctx->children.empty()) return generate_points_one(ctx, n, seed, a_out);
  try {   // the generator is deterministic in (seed, index): every device builds the identical set
    return on_all_devices(ctx, [&](msm_ctx* c) { return generate_points_one(c, n, seed, c == ctx ? a_out : nullptr); });
  } MSM_CATCH_ALL(ctx)
}

int msm_generate_scalars(msm_ctx* ctx, uint64_t n, uint64_t seed, void* dev_dst, uint8_t* host_out) {
  if (!ctx) return MSM_ERR_ARG;
  if (!dev_dst && !host_out) return fail(ctx, MSM_ERR_ARG, "msm_generate_scalars: neither a device nor a host destination");
  try {
    HIPCHK(hipSetDevice(ctx->device));
    return msm_gen::generate_scalars(ctx, n, seed, dev_dst, host_out);
  } MSM_CATCH_ALL(ctx)
}

// ---- point-set handles: several resident point sets per context, one of them current ------------------------

static int pointset_select_one(msm_ctx* ctx, int32_t id) {
  if (id < 0 || id >= (int)ctx->sets.size() || (id != 0 && !ctx->sets[id].live))
    return fail(ctx, MSM_ERR_ARG, "msm_pointset_select: no point set %d", (int)id);
  if (id == ctx->cur_set) return MSM_OK;
  ctx->sets[ctx->cur_set].rows = ctx->rows;
  ctx->sets[ctx->cur_set].n = ctx->n_points;
  ctx->rows = ctx->sets[id].rows;
  ctx->n_points = ctx->sets[id].n;
  ctx->cur_set = id;
  return MSM_OK;
}

int msm_pointset_create(msm_ctx* ctx, int32_t* id_out) {
  if (!ctx || !id_out) return fail(ctx, MSM_ERR_ARG, "msm_pointset_create: null argument");
  try {
    return on_all_devices(ctx, [&](msm_ctx* c) {
      int id = -1;
      for (size_t i = 1; i < c->sets.size(); i++)
        if (!c->sets[i].live) { id = (int)i; break; }
      if (id < 0) { c->sets.emplace_back(); id = (int)c->sets.size() - 1; }   // children stay in lockstep: same ids
      c->sets[id] = msm_ctx::PointSet();
      c->sets[id].live = true;
      if (c == ctx) *id_out = id;
      return pointset_select_one(c, id);
    });
  } MSM_CATCH_ALL(ctx)
}

int msm_pointset_select(msm_ctx* ctx, int32_t id) {
  if (!ctx) return MSM_ERR_ARG;
  try {
    return on_all_devices(ctx, [&](msm_ctx* c) { return pointset_select_one(c, id); });
  } MSM_CATCH_ALL(ctx)
}

int msm_pointset_destroy(msm_ctx* ctx, int32_t id) {
  if (!ctx) return MSM_ERR_ARG;
  if (id <= 0 || id >= (int)ctx->sets.size() || !ctx->sets[id].live)
    return fail(ctx, MSM_ERR_ARG, "msm_pointset_destroy: no such point set %d (the default set 0 stays)", (int)id);
  try {
    return on_all_devices(ctx, [&](msm_ctx* c) {
      if (c->cur_set == id) pointset_select_one(c, 0);
      (void)hipSetDevice(c->device);
      c->release(c->sets[id].rows);
      c->sets[id] = msm_ctx::PointSet();
      return (int)MSM_OK;
    });
  } MSM_CATCH_ALL(ctx)
}

// ---- device buffers for scalar handles ------------------------------------------------------------------------

int msm_device_alloc(msm_ctx* ctx, uint64_t bytes, void** dev_ptr_out) {
  if (!ctx || !dev_ptr_out) return fail(ctx, MSM_ERR_ARG, "msm_device_alloc: null argument");
  try {
    HIPCHK(hipSetDevice(ctx->device));
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, std::max<uint64_t>(bytes, 32)));
    ctx->allocs.push_back(p);
    *dev_ptr_out = p;
    return MSM_OK;
  } MSM_CATCH_ALL(ctx)
}

int msm_device_free(msm_ctx* ctx, void* dev_ptr) {
  if (!ctx) return MSM_ERR_ARG;
  auto it = std::find(ctx->allocs.begin(), ctx->allocs.end(), dev_ptr);
  if (it == ctx->allocs.end()) return fail(ctx, MSM_ERR_ARG, "msm_device_free: not a buffer of this context");
  try {
    HIPCHK(hipSetDevice(ctx->device));
    ctx->allocs.erase(it);
    HIPCHK(hipFree(dev_ptr));
    return MSM_OK;
  } MSM_CATCH_ALL(ctx)
}

int msm_device_upload(msm_ctx* ctx, void* dev_ptr, const void* host, uint64_t bytes) {
  if (!ctx || !dev_ptr || (!host && bytes)) return fail(ctx, MSM_ERR_ARG, "msm_device_upload: null argument");
  try {
    HIPCHK(hipSetDevice(ctx->device));
    if (bytes) {
      upload_staged(ctx, dev_ptr, host, bytes);
      HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    return MSM_OK;
  } MSM_CATCH_ALL(ctx)
}

int msm_set_workspace_limit(msm_ctx* ctx, uint64_t bytes) {
  if (!ctx) return MSM_ERR_ARG;
  ctx->ws_limit = bytes;
  if (!bytes) {   // back to automatic: small calls use the creation-time rule again
    try {
      HIPCHK(hipSetDevice(ctx->device));
      size_t free_b = 0, total_b = 0;
      HIPCHK(hipMemGetInfo(&free_b, &total_b));
      uint64_t held = 0;
      for (auto& w : ctx->ws)
        for (DevBuf* b : w.all) held += b->cap;
      ctx->ws_budget = (uint64_t)((free_b + held) * 0.55L);
    } MSM_CATCH_ALL(ctx)
  }
  for (msm_ctx* c : ctx->children) {
    int rc = msm_set_workspace_limit(c, bytes);
    if (rc != MSM_OK) return rc;
  }
  return MSM_OK;
}

// ---- multi-device context ---------------------------------------------------------------------------------------

int msm_ctx_create_multi(msm_ctx** out, int curve, const int32_t* devices, int32_t n_devices) {
  if (!out || !devices || n_devices < 1) return MSM_ERR_ARG;
  *out = nullptr;
  msm_ctx* ctx = nullptr;
  int rc = msm_ctx_create(&ctx, curve, devices[0]);
  if (rc != MSM_OK) return rc;
  try {
    for (int i = 1; i < n_devices; i++) {
      msm_ctx* c = nullptr;
      rc = msm_ctx_create(&c, curve, devices[i]);
      if (rc != MSM_OK) {
        msm_ctx_destroy(ctx);
        return rc;
      }
      ctx->children.push_back(c);
      ctx->fan.emplace_back(new HelperThread());
    }
  } catch (...) {
    msm_ctx_destroy(ctx);
    return MSM_ERR_INTERNAL;
  }
  *out = ctx;
  return MSM_OK;
}

int msm_ctx_device_count(const msm_ctx* ctx) { return ctx ? 1 + (int)ctx->children.size() : 0; }

}  // extern "C"
