// Slab halo exchange (ring over the slowest spatial axis) — RCCL send/recv, loaded lazily.
#pragma once
#include "common.hpp"

namespace xlb {
// enqueue on `stream` the refill of f's two ghost planes from the ring neighbours (or from the
// field itself when there is a single rank): populations with c_x = +1 travel "right", c_x = -1 "left".
int halo_exchange_on(xlbhip_ctx* c, int lattice, xlbhip_field* f, hipStream_t stream);
}  // namespace xlb
