// Slab halo exchange (ring over the slowest spatial axis) — RCCL send/recv (loaded lazily) or IPC-mapped peer copies.
#pragma once
#include "common.hpp"

namespace xlb {
// enqueue on `stream` the refill of f's two ghost planes from the ring neighbours (or from the
// field itself when there is a single rank): populations with c_x = +1 travel "right", c_x = -1 "left".
// depth 1: what one step pulls across the faces; depth 2: what two fused steps need (see comm.cpp)
int halo_exchange_on(xlbhip_ctx* c, int lattice, xlbhip_field* f, hipStream_t stream, int depth = 1);
// ghost planes -1 and nx of a per-cell array (nx + 2 halo planes of ny x nz elements)
int plane_exchange_on(xlbhip_ctx* c, void* base, size_t elem_bytes, int nx, int ny, int nz, int halo, hipStream_t stream);
// MIN of `value` over the ranks of the context's communicator (blocking; `*out = value` without one)
int comm_all_min(xlbhip_ctx* c, int value, int* out);
int comm_ranks(const xlbhip_ctx* c);
// ipc transport: non-zero (error set) when a device-side wait of this rank timed out since the last check
int comm_check(xlbhip_ctx* c);
// an allocation that may have been exported is about to be freed
void comm_forget_buffer(xlbhip_ctx* c, const void* alloc_base);
}  // namespace xlb
