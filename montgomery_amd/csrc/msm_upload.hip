// Host -> device transfers of big pageable buffers: staged through pinned chunks (upload_staged) and, for the scalars of a
// big MSM, pipelined behind the computation range by range of the points (PieceUpload).
// (reference: scalarsFromBytes / pointsFromBytes into shared wasm memory, src/parallel.ts:97-133)
#include "msm_internal.h"

using namespace msm;
using namespace msmi;

namespace msmi {

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

}  // namespace msmi
