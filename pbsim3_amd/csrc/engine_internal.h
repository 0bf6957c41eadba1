// engine_internal.h -- helpers of engine.cpp that its siblings (deflate_host.cpp, units.cpp, sample.cpp) use.  Internal: nothing here is part of
// include/pbsim3_amd.h.
#pragma once
#include "ctx.h"

namespace pbsim {
int upload(DevBuf &b, const void *src, size_t n, hipStream_t s);
int ensure_header_tables(pbsim_ctx *c);
int ensure_qs_tabs(pbsim_ctx *c, bool hp11);
int ensure_class_tables(pbsim_ctx *c);
int read_flags(pbsim_ctx *c, DeviceFlags *f);  // the selected slot's flags, through pinned staging
// K0 of the context's current unit on its own stream (upper-case, homopolymer lengths, census into census_out), synchronous
int prepare_reference(pbsim_ctx *c, uint8_t *d_seq, int64_t len, int keep_first_case, int64_t census_out[kHpSlots]);
int64_t batch_capacity(const pbsim_ctx *c);  // reads a batch of the current unit is sized to
// the finalized batch of the selected slot to the sink (text or members), waiting for its text emission first when deferred
// (deflate_host.cpp)
int deliver(pbsim_ctx *c, const pbsim_sink *sink);

// A driver that does not wait for a batch's text emission in finalize_text (the batch's statistics are added on the host
// meanwhile; deliver() waits before it hands text to a sink); every emission has landed when the driver returns.
struct DeferTextSync {
  pbsim_ctx *c;
  bool prev;
  explicit DeferTextSync(pbsim_ctx *ctx) : c(ctx), prev(ctx->defer_text_sync) { c->defer_text_sync = true; }
  ~DeferTextSync() {
    for (Slot &sl : c->slots)
      if (sl.stream) (void)hipStreamSynchronize(sl.stream);
    c->defer_text_sync = prev;
  }
};
}  // namespace pbsim
