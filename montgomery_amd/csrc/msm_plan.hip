// Plan of one MSM: window size, number of windows, launch geometry of the tree rounds, workspace budget model; and the
// error helpers every ABI entry ends in.  (reference: windowSize table src/msm-common.ts:25-41, K = ceil((b + 1) / c)
// src/msm-batched-affine.ts:90)
#include "msm_internal.h"

using namespace msm;
using namespace msmi;

namespace msmi {

int fail(msm_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  return code;
}

static const char* base_name(const char* p) {
  const char* s = strrchr(p ? p : "", '/');
  return s ? s + 1 : (p ? p : "?");
}

int fail_hip(msm_ctx* ctx, const HipFail& f) {
  return fail(ctx, MSM_ERR_HIP, "HIP error %d (%s) at %s:%d: %s", (int)f.e, hipGetErrorString(f.e), base_name(f.file), f.line, f.what);
}

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
  // b = 127 (BLS12-381, Pallas): 128 = 5 * 22 + 18, the 22-bit plan has six whole windows (from 2^25 points, see below).
  // (127 = 7 * 18 + 1 folds too: seven 18-bit windows win from 2^22 points to 2^24 -- with the round-5 sort of the big windows
  // 2^22 11.4 against 12.1 ms, 2^23 20.6 / 21.8, level at 2^21 (6.8), behind the 21-bit plan at 2^24 (41.3 / 39.5): tools/plan_sweep.py)
  if (!te && glv_max_bits == 126)
    return n >= (1ull << 24) ? 21 : n >= (1ull << 22) ? 18 : n >= 4096 ? 16 : 8;
  // b = 127 (BLS12-381, Pallas), with the round-5 sort of the big windows (tools/plan_sweep_curve.py, c = 16 / 19 / 22):
  // BLS12-381 2^23 22.6 / 22.1 / 24.7 ms, 2^24 44.0 / 43.3 / 44.0, 2^25 81.6 / 80.5 / 75.0; Pallas 2^23 15.6 / 14.5 / 16.3,
  // 2^24 32.0 / 30.6 / 29.7, 2^25 58.6 / 59.0 / 55.8; level at 2^22
  if (!te) return n >= (1ull << 25) ? 22 : n >= (1ull << 23) ? 19 : n >= 4096 ? 16 : 8;
  // Edwards plain path with the bin split of round 5 (tools/plan_sweep_curve.py 1, ms at c = 16 / 18): 2^22 8.6 / 7.9, 2^23 15.0 / 14.2,
  // 2^24 28.0 / 26.3, 2^25 53.4 / 54.4, 2^26 101.8 / 105.7 (its gather round is not tile-ordered: beyond 2^25 the big windows lose)
  if (te && n >= (1ull << 22) && n < (1ull << 25)) return 18;
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

// On window tables all windows of a group share one set of buckets, so a wider window costs its 2^(c-1) buckets once instead
// of K times and the optimum moves up.  BLS12-377 after GLV (127 bits: 18- and 21-bit windows fold the carry bit, no short top
// window to skew the merged buckets), measured with tools/tables_csweep.py (profiles/r05_experiments.txt item 5): 16 bits below
// 2^16 points, 18 from there (2^18 1.58 against 1.76 ms plain, 2^20 3.65 / 3.90, 2^22 11.0 / 11.9, 2^23 20.3 / 21.7), 21 from
// 2^24 (36.7 / 40.1).  BLS12-381 and Pallas keep the plain choice (their 18-bit plan would end in a two-bit top window).
// Ed-on-BLS12-377 (252 bits, no endomorphism: K = 13 .. 28 windows on the plain path, whose reduction and host tail grow with K):
// 14 bits below 2^17 points, 17 from there (K = 15; 2^18 0.98 against 1.09 ms plain, 2^20 2.11 / 2.63, 2^22 7.3 / 8.6; 16 bits
// 2.28, 18 bits 2.17 at 2^20) -- with the merged window's sums finished bit-sliced (reduce_buckets).
static int pick_window_tables(bool te, uint64_t n, int glv_max_bits) {
  if (te) return n >= (1ull << 17) ? 17 : 14;
  if (glv_max_bits == 126) return n >= (1ull << 24) ? 21 : n >= (1ull << 16) ? 18 : 16;   // (2^15: 0.86 ms at 16 bits, 0.90 at 18; 2^16: 1.03 / 1.00)
  return pick_window(te, n, glv_max_bits);
}

int make_plan(const msm_ctx* ctx, uint64_t n, const msm_opts* opts, Plan& pl, bool for_tables) {
  const bool te = ctx && ctx->is_te();
  const int glv_bits = curve_info(ctx ? ctx->curve : MSM_CURVE_BLS12_377_G1).glv_max_bits;
  const int glv_arg = (opts && opts->no_glv) ? 0 : glv_bits;
  int c = (opts && opts->c > 0) ? opts->c : for_tables ? pick_window_tables(te, n, glv_arg) : pick_window(te, n, glv_arg);
  if (c < 2 || c > 24) return MSM_ERR_ARG;
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
  pl.b_lo = pl.bt_lo = 0;
  pl.b_n = pl.bt_n = 0xFFFFFFFFu;
  if (opts && opts->bucket_shards > 1) {
    // Rank g of G takes the g-th of G equal parts of the bucket range every window's digits really cover -- 2^(c-1) for a
    // recoded window, the whole 2^c of a folded top window, 2^(bits left) for a short one -- so that all ranks sort and add the
    // same number of entries.  The last part runs to the end of the window's buckets whatever the digits do.
    if (opts->bucket_shard < 0 || opts->bucket_shard >= opts->bucket_shards) return MSM_ERR_ARG;
    const uint32_t g = (uint32_t)opts->bucket_shard, G = (uint32_t)opts->bucket_shards;
    auto cut = [&](uint64_t span, uint32_t& lo, uint32_t& cnt) {
      lo = (uint32_t)(span * g / G);
      const uint64_t hi = g + 1 == G ? (uint64_t)pl.L : span * (g + 1) / G;
      cnt = (uint32_t)(hi - lo);
    };
    const int top_bits = pl.fold ? c : std::min(c - 1, pl.bits - (pl.K - 1) * c);
    cut(pl.K > 1 ? 1ull << (c - 1) : 1ull << std::max(0, top_bits), pl.b_lo, pl.b_n);
    cut(1ull << std::max(0, top_bits), pl.bt_lo, pl.bt_n);
  }
  return MSM_OK;
}

// gather: round 1 (random row reads: wants two waves per SIMD to cover the latency).  The other rounds read
// coalesced, prefetched planes; a lone wave already gets ~88 % of a SIMD's issue rate, and every lane pays one field
// inversion (~19 pair additions' worth) per round, so small rounds run better on half as many lanes with twice the
// steps (2^18: 2.62 -> 2.45 ms; neutral at 2^20, 1 % at 2^22; round 1 at 2^22 would lose 60 %).
// lone: the launch has the GPU to itself (a window group of one window with no second group beside it -- the
// 8-GPU shard).  All waves of one resident batch then move through the memory-heavy forward sweep and the ALU-heavy
// backward sweep in step; four batches of 128 steps instead of one of 512 stagger the phases (2^26, one window:
// 31.5 -> 29.2 ms).  With two groups on two streams the other stream already fills the gaps and 512 is better.
RoundGeom round_geom(const msm_ctx* ctx, uint64_t n_out, bool gather, bool lone) {
  uint64_t target = (uint64_t)ctx->n_cu * 4 * BA_WAVES * 64;  // as many lanes as the kernel's launch bounds keep resident
  // steps at full width below which a non-gather round runs on half as many lanes: every lane pays one inversion per
  // round (~13 pair additions' worth), and one wave per SIMD already gets 89 % of the multiplier's two-wave rate
  // (tools/ubench_mul2.hip).  Measured with the round-2 kernel: 2^20 4.05 -> 3.91 ms, 2^22 12.9 -> 12.4, neutral elsewhere.
  const uint64_t half_below = 128;
  if (!gather && n_out < target * half_below) target /= 2;
  uint32_t max_steps = (lone && n_out >= target * 512) ? 128 : 512;
  MSM_KNOB(max_steps, "MSM_MAX_STEPS", 1);
  uint64_t steps = (n_out + target - 1) / target;
  steps = std::max<uint64_t>(1, std::min<uint64_t>(steps, max_steps));
  uint64_t threads = (n_out + steps - 1) / steps;
  uint64_t grid = std::max<uint64_t>(1, (threads + 255) / 256);
  return RoundGeom{(uint32_t)steps, (uint32_t)grid, grid * 256};
}

// how many windows fit one group under the workspace budget
void release_workspaces(msm_ctx* ctx) {
  for (auto& w : ctx->ws) {
    (void)hipStreamSynchronize(w.stream);
    for (DevBuf* b : w.all) ctx->release(*b);
  }
}

long double window_bytes(const msm_ctx* ctx, uint64_t n, const Plan& pl) {
  // bytes per window and point (Weierstrass: 2 entries per point): digits 8, records of the sort's passes 16, slots (or the pair
  // list of a big window) ~9, tree buffers 96 + 48, prefix scratch 56; the tile-ordered round 1 of a big window (c >= 18) adds
  // the element index of every pair (4) and the element records (128 bytes per pair), and its plane buffers start one round
  // later.  Per window and bucket: counters, cursors, up to 34 offset tables, and the block histograms of the sort.
  const bool te = ctx->is_te();
  const bool big = pl.c > 16;
  long double per_point = te ? (4 + 8 + 5 + 64 + 32) : (8 + 16 + 9 + 96 + 48 + 56);
  if (big && !te) per_point += 4 + 128 - 72;
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

}  // namespace msmi
