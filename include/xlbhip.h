/*
 * xlbhip.h — C ABI of the MI355X-native LBM stepper (libxlbhip.so).
 *
 * This is the drop-in boundary for the per-timestep hot path of hsalehipour/XLB.
 * The reference has no native code: its "FFI" for this path is the set of Python
 * methods registered with @Operator.register_backend(ComputeBackend.X)
 * (reference xlb/operator/operator.py:74-133).  Every entry point below names the
 * reference method(s) whose body a `hip_implementation` replaces; the ctypes stub a
 * maintainer would add is shown in INTEGRATION.md and shipped in xlb_amd/_lib.py.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no C++/torch types.
 *  - every call returns 0 on success, non-zero on failure; the message is available
 *    from xlbhip_last_error() (thread-local).  The Python shim turns it into an
 *    exception, mirroring the reference's exception-only error model
 *    (operator.py:128-133).
 *  - calls only ENQUEUE work on the context's compute stream (asynchronous, like
 *    wp.launch in the reference) except *_download, *_sync and *_timed.
 *  - host arrays are C-order (cardinality, nx, ny, nz), population index slowest,
 *    last spatial axis fastest — the reference layout (xlb/grid/warp_grid.py:26,
 *    xlb/grid/jax_grid.py:44-46).  2-D grids (nx, ny) are passed as (1, nx, ny):
 *    the lattice's two components then act on the last two axes.
 */
#ifndef XLBHIP_H
#define XLBHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct xlbhip_ctx xlbhip_ctx;
typedef struct xlbhip_field xlbhip_field;
typedef struct xlbhip_stepper xlbhip_stepper;

/* element types of a field; mirrors Precision in xlb/precision_policy.py:13-54 */
enum {
  XLBHIP_F64 = 0,
  XLBHIP_F32 = 1,
  XLBHIP_F16 = 2,
  XLBHIP_U8 = 3,
  XLBHIP_BOOL = 4,      /* one byte per element, 0/1 */
  XLBHIP_MISSING = 5    /* missing_mask: host view (q,nx,ny,nz) u8 0/1; device: one u32 bit-set per cell */
};

/* velocity sets; xlb/velocity_set/{d2q9,d3q19,d3q27}.py */
enum { XLBHIP_D2Q9 = 0, XLBHIP_D3Q19 = 1, XLBHIP_D3Q27 = 2 };

/* collision models; xlb/operator/stepper/nse_stepper.py:78-84 */
enum {
  XLBHIP_BGK = 0,
  XLBHIP_KBC = 1,
  XLBHIP_SMAGORINSKY_LES_BGK = 2 /* xlb/operator/collision/smagorinsky_les_bgk.py:44-60 (section 8f rank 3) */
};

/* boundary-condition kinds (in scope: SURVEY.md section 8 rows a7-a9, + DoNothing) */
enum {
  XLBHIP_BC_EQUILIBRIUM = 1,  /* bc_equilibrium.py:72-80   (post-streaming) */
  XLBHIP_BC_HALFWAY_BB = 2,   /* bc_halfway_bounce_back.py:116-134 (post-streaming) */
  XLBHIP_BC_FULLWAY_BB = 3,   /* bc_fullway_bounce_back.py:50-56   (post-collision) */
  XLBHIP_BC_DO_NOTHING = 4,   /* bc_do_nothing.py:50-54    (post-streaming) */
  /* SURVEY.md section 8f rank 1 — inlet/outlet family with CONSTANT prescribed values (JAX semantics:
   * the value lives in the BC object, bc_zouhe.py:120-121, not in the f_1 aux encoding) */
  XLBHIP_BC_ZOUHE_VELOCITY = 5,        /* bc_zouhe.py:218-304, values[0..2] = velocity (3-component internal form) */
  XLBHIP_BC_ZOUHE_PRESSURE = 6,        /* values[0] = density */
  XLBHIP_BC_REGULARIZED_VELOCITY = 7,  /* bc_regularized.py:78-137 */
  XLBHIP_BC_REGULARIZED_PRESSURE = 8,
  /* ExtrapolationOutflowBC, JAX branch (bc_extrapolation_outflow.py:98-145): streaming step = the missing populations
   * take the cell's own opposite PRE-stream population; after the collision the outgoing populations are overwritten
   * with cs * f_post_stream(neighbour behind the face) + (1 - cs) * f_post_stream(cell) ("auxiliary data", read back by
   * the next step).  values[0..2] = outward face normal (internal 3-component form), values[3] = cs = 1/sqrt(3) and
   * values[4] = 1 - cs, both rounded in the compute dtype by the host side. */
  XLBHIP_BC_EXTRAPOLATION_OUTFLOW = 9,
  /* HybridBC (bc_hybrid.py:254-358; kernel backends only in the reference), 3-D lattices.  values[0..2] = wall velocity
   * (3-component internal form), values[3] != 0: moving-wall treatment, values[4] != 0: use the wall-distance weights the
   * mesh masker produced (xlbhip_stepper_set_bc_distances) */
  XLBHIP_BC_HYBRID_BB_REGULARIZED = 10,  /* interpolated bounce-back + regularisation (bc_hybrid.py:256-290) */
  XLBHIP_BC_HYBRID_BB_GRADS = 11,        /* interpolated bounce-back + Grad's approximation (:292-326) */
  XLBHIP_BC_HYBRID_NEQ_REGULARIZED = 12, /* non-equilibrium bounce-back + regularisation (:328-358) */
  /* halfway bounce-back whose wall velocity is a per-cell table (xlbhip_stepper_set_bc_profile): the reference's
     HalfwayBounceBackBC(profile=...) of the kernel backends, bc_halfway_bounce_back.py:144-169 + helper_functions_bc.py:230-250 */
  XLBHIP_BC_HALFWAY_BB_PROFILE = 13
};

/* One boundary condition as the stepper sees it.  `values` holds, in COMPUTE
 * precision widened to double (exactly representable): for EQUILIBRIUM the q
 * populations feq(rho0,u0); for HALFWAY_BB the q moving-wall terms
 * 6 w_l (c_l . u_wall) (all zero for no-slip).  Computed on the host by the
 * Python operator exactly as the reference's JAX branch does. */
typedef struct {
  int32_t id;        /* value in bc_mask, 1..253; boundary_condition_registry.py:16-27 */
  int32_t kind;      /* XLBHIP_BC_* */
  double values[27];
} xlbhip_bc_desc;

/* ---- context ------------------------------------------------------------ */
/* replaces: xlb.init backend branch, xlb/default_config.py:78-100 */
int xlbhip_create(int device, xlbhip_ctx** out);
int xlbhip_destroy(xlbhip_ctx* ctx);
int xlbhip_sync(xlbhip_ctx* ctx);                 /* wp.synchronize(), mlups_3d.py:230 */
const char* xlbhip_last_error(void);
int xlbhip_device_info(xlbhip_ctx* ctx, char* name, int name_len, int* compute_units, uint64_t* hbm_bytes);
/* tuning knobs (kernel variant selection etc.); unknown keys are an error.  The ones with semantics:
 *   exact_math      0 (default): fp64-compute D3Q27 KBC runs the tolerance-graded fast collision (rounding-level
 *                   differences from the bit-exact build, <= 4e-16 measured; north-star tolerance 1e-6); 1: bit-exact builds only
 *   fast_bgk        1 (default 0): the two-step kernel uses the tolerance-graded fast BGK body (rounding-level differences,
 *                   <= 3e-7 measured over 10 steps in fp32; +2-4 %); ignored with exact_math=1
 *   fuse2           xlbhip_run* pair their steps through the two-step kernel: 0 never, 1 where eligible and the work
 *                   items fill the chip, 2 wherever eligible
 *   external_halo   1: the caller refills the ghost planes before every xlbhip_step / xlbhip_step2 (host-staged transports)
 *   overlap         0: no overlap of the halo exchange with the interior launch (measurement)
 * Pure tuning (results identical): vec, nt_store, nt_load, plane_pad_bytes (at field creation), block_threads, block_tz,
 * xcd_swizzle, fuse2_xseg / _lpt / _xcd / _clean / _xcap / _shift / _tile / _cus (csrc/api.hip: xlbhip_create). */
int xlbhip_set_option(xlbhip_ctx* ctx, const char* key, int64_t value);
int xlbhip_get_option(xlbhip_ctx* ctx, const char* key, int64_t* value);

/* lattice tables as compiled into the kernels, for cross-checking against the
 * Python VelocitySet (velocity_set.py:63-83): c is (3,q) row-major (2-D sets have a
 * leading zero row), w (q), opp (q), cc (q,6) */
int xlbhip_lattice_info(int lattice, int* d, int* q, int32_t* c, double* w, int32_t* opp, int32_t* cc);

/* ---- fields ------------------------------------------------------------- */
/* replaces: WarpGrid.create_field, xlb/grid/warp_grid.py:17-35.
 * halo = number of ghost x-planes on each side: 0, or 1 / 2 for slab-decomposed runs (2 lets xlbhip_run fuse
 * two steps per pass across rank boundaries). */
int xlbhip_field_create(xlbhip_ctx* ctx, int cardinality, int nx, int ny, int nz, int dtype, int halo,
                        double fill_value, xlbhip_field** out);
int xlbhip_field_destroy(xlbhip_field* f);
int xlbhip_field_fill(xlbhip_field* f, double value);
int xlbhip_field_copy(xlbhip_field* dst, const xlbhip_field* src);            /* wp.copy, nse_stepper.py:124 */
/* streaming copy by a plain kernel (4 or 16 bytes per lane): bandwidth yardstick and the
 * known-byte-count calibration of the rocprofv3 FETCH_SIZE / WRITE_SIZE counters (tools/) */
int xlbhip_field_copy_kernel(xlbhip_field* dst, const xlbhip_field* src, int bytes_per_lane);
/* the same copy with the launch shape of the two-step kernel: one 704-thread block per CU, (8 x 64) tiles marching along x, a plane's
 * stores and pulls separated by a barrier — the second yardstick of bench.py (roofline.pattern_copy_*); 4-byte elements, ny % 8 == 0,
 * nz % 64 == 0.  Measurement only: nothing in the reference corresponds (its harness times the step alone, examples/performance/mlups_3d.py:225-242) */
int xlbhip_field_copy_tiles(xlbhip_field* dst, const xlbhip_field* src);
int xlbhip_field_upload(xlbhip_field* f, const void* host, size_t host_bytes);   /* interior only */
int xlbhip_field_download(const xlbhip_field* f, void* host, size_t host_bytes); /* interior only; synchronous */
/* one x-plane of one population, addressed by STORAGE plane (0 .. nx + 2 halo - 1, ghosts included):
 * host-staged halo transports and tests */
int xlbhip_field_plane_download(const xlbhip_field* f, int population, int storage_plane, void* host, size_t bytes);
int xlbhip_field_plane_upload(xlbhip_field* f, int population, int storage_plane, const void* host, size_t bytes);
/* Tell the library that somebody wrote the field behind its back — through a zero-copy alias (Field.__dlpack__ /
 * __cuda_array_interface__, the replacement of utils.py:340-447's ToJAX): bumps the contents version that the stepper's
 * per-mask caches (meta words, clean-item flags, end-plane scan) are keyed on */
int xlbhip_field_touch(xlbhip_field* f);
/* (backend-internal, no reference counterpart) free / total device memory (hipMemGetInfo): the Python stepper refuses to pair reference-style calls when the
 * temporary third field a read of the virtual f(t+1) needs would not fit */
int xlbhip_mem_info(xlbhip_ctx* ctx, uint64_t* free_bytes, uint64_t* total_bytes);
int xlbhip_field_info(const xlbhip_field* f, int* cardinality, int* nx, int* ny, int* nz, int* dtype, int* halo,
                      uint64_t* plane_stride_elems, void** device_ptr);

/* ---- whole-field operators (kernel-backend out-of-place style) ---------- */
/* Stream()(f_0, f_1): stream.py:96-125 */
int xlbhip_stream(xlbhip_ctx* ctx, int lattice, const xlbhip_field* f_src, xlbhip_field* f_dst);
/* QuadraticEquilibrium()(rho, u, f): quadratic_equilibrium.py:23-30, :91-103 */
int xlbhip_equilibrium(xlbhip_ctx* ctx, int lattice, int compute_dtype, const xlbhip_field* rho, const xlbhip_field* u,
                       xlbhip_field* f);
/* Macroscopic()(f, rho, u): macroscopic.py:21-26, :57-64 */
int xlbhip_macroscopic(xlbhip_ctx* ctx, int lattice, int compute_dtype, const xlbhip_field* f, xlbhip_field* rho,
                       xlbhip_field* u);
/* SecondMoment()(f, pi): second_moment.py:35-55 */
int xlbhip_second_moment(xlbhip_ctx* ctx, int lattice, int compute_dtype, const xlbhip_field* f, xlbhip_field* pi);
/* as xlbhip_apply_bc for a Zou-He / Regularized BC with per-cell prescribed values (see xlbhip_stepper_set_bc_profile) */
int xlbhip_apply_bc_profile(xlbhip_ctx* ctx, int lattice, int compute_dtype, const xlbhip_bc_desc* bc, const xlbhip_field* f_pre,
                            xlbhip_field* f_post, const xlbhip_field* bc_mask, const xlbhip_field* missing_mask, int64_t n,
                            const uint32_t* storage_cells, const double* values);
/* MomentumTransfer(no_slip_bc)(f_0, f_1, bc_mask, missing_mask) -> net force on the solid behind the BC
 * (force/momentum_transfer.py:167-205, JAX; f_0 = post-collision populations, i.e. the field a step returned).
 * The BC must be a halfway or fullway bounce-back; force_out = (F_x, F_y, F_z) in the internal 3-component form,
 * the sum over THIS rank's cells (slab runs add the ranks' vectors).  Synchronous. */
int xlbhip_momentum_transfer(xlbhip_ctx* ctx, int lattice, int compute_dtype, const xlbhip_bc_desc* no_slip_bc, const xlbhip_field* f_0,
                             const xlbhip_field* bc_mask, const xlbhip_field* missing_mask, double force_out[3]);
/* Vorticity()(u, bc_mask, vorticity, vorticity_magnitude): postprocess/vorticity.py:30-93 (3-D; cells one layer inside the
 * box whose six face neighbours are all fluid get the curl of u by central differences and its magnitude; every other cell of
 * the outputs is left untouched).  Arithmetic in u's dtype (fp32 / fp64). */
int xlbhip_vorticity(xlbhip_ctx* ctx, const xlbhip_field* u, const xlbhip_field* bc_mask, xlbhip_field* vorticity,
                     xlbhip_field* vorticity_magnitude);
/* GridToPoint()(grid, points, point_values): postprocess/grid_to_point.py:28-104 — trilinear interpolation of component 0 of a
 * 3-D field at n points (host float[n][3] in, host values[n] in the field's dtype out; synchronous).  Unlike the reference,
 * points whose surrounding cube leaves the field are an error instead of an out-of-bounds read. */
int xlbhip_grid_to_point(xlbhip_ctx* ctx, const xlbhip_field* grid, int64_t n, const float* points, void* point_values);
/* QCriterion()(u, bc_mask, norm_mu, q): postprocess/q_criterion.py:36-139; Q = (|Omega|^2 - |S|^2) / 2, same cells */
int xlbhip_q_criterion(xlbhip_ctx* ctx, const xlbhip_field* u, const xlbhip_field* bc_mask, xlbhip_field* norm_mu, xlbhip_field* q);
/* BGK()/KBC()(f, feq, fout, omega): bgk.py:27-32,:78-91; kbc.py:40-79 */
int xlbhip_collide(xlbhip_ctx* ctx, int lattice, int collision, int compute_dtype, const xlbhip_field* f,
                   const xlbhip_field* feq, xlbhip_field* fout, double omega);
/* bc(f_pre, f_post, bc_mask, missing_mask) -> f_post (in place): boundary_condition.py:146-180 */
int xlbhip_apply_bc(xlbhip_ctx* ctx, int lattice, int compute_dtype, const xlbhip_bc_desc* bc, const xlbhip_field* f_pre,
                    xlbhip_field* f_post, const xlbhip_field* bc_mask, const xlbhip_field* missing_mask);

/* ---- boundary masker ----------------------------------------------------- */
/* replaces: IndicesBoundaryMasker (JAX semantics), indices_boundary_masker.py:73-143.
 * For BC i: tag_idx[i] is an int32 (3, tag_count[i]) row-major array of GLOBAL cell
 * indices that receive bc_ids[i] in bc_mask (later BCs overwrite earlier ones);
 * solid_idx[i] (may be NULL / count 0) are GLOBAL indices whose every population is
 * marked missing before the mask is streamed (the "interior geometry" branch,
 * :107-121).  global_shape is the whole domain; x_offset is the global x index of this
 * rank's first interior plane (the start_index of :76,:106).  Existing contents of
 * missing_mask are streamed too, as in the reference. */
int xlbhip_build_masks(xlbhip_ctx* ctx, int lattice, int n_bc, const int32_t* bc_ids, const int32_t* const* tag_idx,
                       const int64_t* tag_count, const int32_t* const* solid_idx, const int64_t* solid_count,
                       const int32_t global_shape[3], int x_offset, xlbhip_field* bc_mask, xlbhip_field* missing_mask);

/* MeshMaskerAABB()(bc, f_1, bc_mask, missing_mask): boundary_masker/aabb.py:38-100 + mesh_boundary_masker.py:62-245.
 * vertices = n_triangles x 3 corners x (x, y, z) in lattice units (voxel i spans [i, i+1]), fully inside the domain.
 * Voxels the surface passes through become BC_SOLID (255); fluid voxels next to one get bc_id and the missing bits of the
 * directions pulled out of it; voxels of bc_id also miss the directions pulled from outside the box.  Existing mask
 * contents are kept (other BCs' ids, earlier solid voxels).  Single-rank fields.  Synchronous. */
int xlbhip_mesh_mask_aabb(xlbhip_ctx* ctx, int lattice, int bc_id, int64_t n_triangles, const float* vertices, xlbhip_field* bc_mask,
                          xlbhip_field* missing_mask);

/* MeshMaskerRay()(bc, f_1, bc_mask, missing_mask): boundary_masker/ray.py:38-76 — a voxel whose lattice link along c_l
 * (from the voxel centre, length |c_l|) crosses the surface gets bc_id and missing[opp l]; no solid voxels are marked.
 * Same conventions as xlbhip_mesh_mask_aabb. */
int xlbhip_mesh_mask_ray(xlbhip_ctx* ctx, int lattice, int bc_id, int64_t n_triangles, const float* vertices, xlbhip_field* bc_mask,
                         xlbhip_field* missing_mask);

/* Mesh voxelisation methods (xlb/operator/boundary_masker/mesh_voxelization_method.py:13-55) */
enum { XLBHIP_MESH_AABB = 1, XLBHIP_MESH_RAY = 2, XLBHIP_MESH_AABB_CLOSE = 3, XLBHIP_MESH_WINDING = 4 };

/* All mesh maskers behind one entry, with the wall distances ("weights") curved-wall BCs interpolate with.  Replaces the
 * warp_implementation of MeshMaskerAABB (aabb.py:38-100), MeshMaskerRay (ray.py:38-101), MeshMaskerWinding
 * (winding.py:19-115) and MeshMaskerAABBClose (aabb_close.py:26-365; `close_voxels` = half width of the (2h+1)^3
 * structuring element of the morphological close) — masker(bc, distances, bc_mask, missing_mask).
 * `distances`: NULL, or a (q, nx, ny, nz) fp32 field that receives, like the reference's `distances` argument,
 *   RAY        distances[l]     = t / |c_l|            at a boundary voxel whose link along c_l crosses the surface at t
 *   WINDING    distances[opp l] = (|c_l| - t) / |c_l|  at the fluid neighbour (+c_l) of a solid voxel
 *   AABB_CLOSE distances[l]     = (t - |c_l| / 2) / |c_l| within 1.5 |c_l|, 1.0 without a hit, for links ending in a solid voxel
 * (untouched elsewhere).  The Warp BVH queries are replaced by exhaustive closest-hit / exact winding-number loops over
 * the triangles: O(voxels near the surface x triangles near the voxel) for the rays, O(bounding-box voxels x triangles)
 * for the winding number. */
int xlbhip_mesh_mask(xlbhip_ctx* ctx, int lattice, int method, int bc_id, int64_t n_triangles, const float* vertices, int close_voxels,
                     xlbhip_field* bc_mask, xlbhip_field* missing_mask, xlbhip_field* distances);

/* out[i][l] = field[l][cells[i]] for n interior linear cell indices ((x * ny + y) * nz + z), element type of the field;
 * blocking.  How boundary-cell data (e.g. the distances above) reach the host without downloading whole fields. */
int xlbhip_field_gather(const xlbhip_field* f, int64_t n, const uint32_t* cells, void* out, size_t bytes);

/* ---- the stepper (the hot path) ------------------------------------------ */
/* replaces: IncompressibleNavierStokesStepper._construct_warp + launch,
 * nse_stepper.py:335-476, with the JAX step order of :237-282. */
int xlbhip_stepper_create(xlbhip_ctx* ctx, int lattice, int collision, int compute_dtype, int store_dtype, int n_bc,
                          const xlbhip_bc_desc* bcs, xlbhip_stepper** out);
int xlbhip_stepper_destroy(xlbhip_stepper* s);
/* Per-cell prescribed values of a Zou-He / Regularized BC built with a profile (bc_zouhe.py:122-124,225-232: the
 * array the profile returns, broadcast over the grid, evaluated at the BC's cells): storage_cells[i] =
 * ((x_local + halo) * ny + y) * nz + z of a cell of bc_id on this rank, values[3 i .. 3 i + 2] = its velocity vector
 * (internal 3-component form) or its density in values[3 i].  Replaces the kernel backends' encoding of these values in
 * f_1 (helper_functions_bc.py:371-499, nse_stepper.py:398-425).  May be called once per BC; tables are merged. */
int xlbhip_stepper_set_bc_profile(xlbhip_stepper* s, int bc_id, int64_t n, const uint32_t* storage_cells, const double* values);

/* Wall-distance weights of HybridBC cells: n storage cell indices (as for set_bc_profile) and q floats each, in the
 * layout of the mesh masker's `distances` (the weight of missing direction l in slot opp l).  The reference keeps them
 * in f_1 and recovers them every step (bc_hybrid.py:207-214, nse_stepper.py:398-425); here they live in a sorted table
 * that the boundary lanes of the step kernel search.  Calls accumulate (one per mesh BC). */
int xlbhip_stepper_set_bc_distances(xlbhip_stepper* s, int64_t n, const uint32_t* storage_cells, const float* weights);
/* MomentumTransfer for a HybridBC or a profile wall of this stepper (the kernel backends' path, force/momentum_transfer.py:225-262 with
 * FetchPopulations :75-92): f_post_stream = that BC applied to (own populations of f_0, populations pulled from f_0) with the stepper's
 * wall-distance and wall-velocity tables; force[3] in the internal 3-component form.  Fields without ghost planes. */
int xlbhip_stepper_momentum_transfer(xlbhip_stepper* s, int bc_id, const xlbhip_field* f_0, const xlbhip_field* bc_mask, const xlbhip_field* missing_mask,
                                     double force[3]);
/* ForcedCollision with the exact-difference scheme (forced_collision.py:44-50, exact_difference_force.py:61-83):
 * force[3] in the internal 3-component form; NULL switches forcing off */
int xlbhip_stepper_set_force(xlbhip_stepper* s, const double* force);
/* Smagorinsky constant of XLBHIP_SMAGORINSKY_LES_BGK (default 0.17, smagorinsky_les_bgk.py:36) */
int xlbhip_stepper_set_smagorinsky(xlbhip_stepper* s, double coef);
/* one step: reads f_src, writes f_dst (caller swaps); omega is cast to compute dtype (bgk.py:31) */
int xlbhip_step(xlbhip_stepper* s, const xlbhip_field* f_src, xlbhip_field* f_dst, const xlbhip_field* bc_mask,
                const xlbhip_field* missing_mask, double omega, int64_t timestep);
/* n_steps steps with the A/B swap done natively; the result is in f_a if n_steps is even, else f_b */
int xlbhip_run(xlbhip_stepper* s, xlbhip_field* f_a, xlbhip_field* f_b, const xlbhip_field* bc_mask,
               const xlbhip_field* missing_mask, double omega, int64_t first_timestep, int64_t n_steps);
/* TWO steps in one pass (f(t) in f_src -> f(t+2) in f_dst, f(t+1) never reaches HBM): what xlbhip_run uses for
 * pairs of steps where xlbhip_step2_eligible() says 1 (D3Q19 BGK fp32, basic BCs, ny % 8 == nz % 64 == 0, fields
 * with 0 or 2 ghost planes, enough tiles to fill the chip).  Same arithmetic in the same order as two xlbhip_step
 * calls (nse_stepper.py:237-282 twice): bit-identical results.  With the "external_halo" option the caller has
 * filled f_src's ghost planes to depth 2 and the ghost planes of the masks to depth 1. */
int xlbhip_step2_eligible(xlbhip_stepper* s, const xlbhip_field* f_src, xlbhip_field* f_dst, const xlbhip_field* bc_mask,
                          const xlbhip_field* missing_mask);
int xlbhip_step2(xlbhip_stepper* s, const xlbhip_field* f_src, xlbhip_field* f_dst, const xlbhip_field* bc_mask,
                 const xlbhip_field* missing_mask, double omega, int64_t timestep);
/* as xlbhip_run without the placement contract: every pair of steps may be fused (xlbhip_run keeps an even number of fused
 * passes and finishes with single steps to honour its contract); *result_in_b = 1 when the result is in f_b */
int xlbhip_run_any(xlbhip_stepper* s, xlbhip_field* f_a, xlbhip_field* f_b, const xlbhip_field* bc_mask,
                   const xlbhip_field* missing_mask, double omega, int64_t first_timestep, int64_t n_steps, int* result_in_b);
/* the same loop bracketed by HIP events on the compute stream; returns device ms for the whole loop.
 * result_in_b == NULL: xlbhip_run's placement; otherwise as xlbhip_run_any */
int xlbhip_run_timed(xlbhip_stepper* s, xlbhip_field* f_a, xlbhip_field* f_b, const xlbhip_field* bc_mask,
                     const xlbhip_field* missing_mask, double omega, int64_t first_timestep, int64_t n_steps,
                     float* device_ms, int* result_in_b);

/* ---- slab decomposition over ranks (one process per GPU) ------------------ */
/* semantics reference: xlb/distribute/distribute.py:18-48 (ring exchange of the
 * face-crossing populations along the slowest spatial axis). */
#define XLBHIP_UNIQUE_ID_BYTES 128
int xlbhip_comm_unique_id(void* out_id_bytes);     /* rank 0; broadcast by the host side */
/* n_ranks == 1 with a non-NULL id creates a real one-rank RCCL communicator (self send/recv): used to
 * exercise the RCCL code path on a single GPU; with a NULL id the ghosts are refilled by device copies */
int xlbhip_comm_init(xlbhip_ctx* ctx, int rank, int n_ranks, const void* id_bytes, int periodic_x);
/* replaces: the same two lax.ppermute calls (xlb/distribute/distribute.py:31-41) as xlbhip_comm_init, by another transport.
 * The same exchange without RCCL: every rank exports the buffers it exchanges (hipIpcGetMemHandle) and PULLS the
 * neighbours' planes on the communication stream — by one small copy kernel per exchange (option "ipc_copy" = 1, the
 * default: 8 blocks per plane, no LDS) or by plane-sized hipMemcpyAsync calls (= 0: copy engines, no compute unit;
 * "or hipMemcpyPeerAsync ... over xGMI").  Processes are ordered by sequence counters in a host
 * shared-memory control block /dev/shm/xlbhip-ipc-<token> (created by rank 0, unlinked as soon as every rank mapped
 * it) that one-lane kernels post and poll; every wait is bounded by the option "ipc_timeout_ms" (default 180 s; a timed-out wait
 * makes the next xlbhip_sync fail).  One node; the ranks may share a device.  `token`: letters / digits / '-' / '_',
 * fresh per job (the host side broadcasts a random one).  Ranks must exchange the same buffers in the same order. */
int xlbhip_comm_init_ipc(xlbhip_ctx* ctx, int rank, int n_ranks, const char* token, int periodic_x);
int xlbhip_comm_destroy(xlbhip_ctx* ctx);
/* (no counterpart in the reference's JAX path; Neon's skeleton reports its overlap mode in the benchmark's JSON, mlups_3d.py:434-488)
 * Telemetry of the slab protocol: the time the compute stream spent waiting for the halo event after the interior
 * launch, summed over `halo_waits` exchanges since the last reset (option "halo_telemetry", default on).  With a
 * working overlap it is the event overhead only (a few microseconds per exchange).  Blocking. */
int xlbhip_comm_stats(xlbhip_ctx* ctx, double* halo_wait_ms, int64_t* halo_waits, int reset);
/* fill the ghost planes of f from the neighbours (blocking w.r.t. the compute stream order) */
int xlbhip_halo_exchange(xlbhip_ctx* ctx, int lattice, xlbhip_field* f);
/* the exchange a fused pair of steps needs (fields with 2 ghost planes): every population of the neighbours' edge
 * planes and the face-crossing populations of the planes behind them */
int xlbhip_halo_exchange_wide(xlbhip_ctx* ctx, int lattice, xlbhip_field* f);

#ifdef __cplusplus
}
#endif
#endif /* XLBHIP_H */
