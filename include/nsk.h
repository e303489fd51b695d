/*
 * libnsk -- C ABI of the MI355X (gfx950) kernels for the NavierStokes.jl
 * projection-method time step.
 *
 * The reference has no FFI: its boundary is the ParallelStencil call syntax
 * `@parallel f!(args...)` (scripts/NavierStokes.jl:79-100) on device-resident,
 * column-major Float64 arrays.  Every entry point below replaces exactly one of
 * those call sites (cited per function) and keeps the Julia argument order; the
 * array extents, which Julia carries inside the array objects, follow as
 * trailing integers.
 *
 * Conventions
 *   - All array arguments are DEVICE pointers to dense column-major Float64
 *     data, A[ix,iy] at a[(iy-1)*size(A,1) + (ix-1)]  (ix contiguous).
 *   - nx, ny are the cell counts of the pressure grid of the array window the
 *     call works on:  Pr, C, divV, txx, tyy are (nx,ny); Vx (nx+1,ny);
 *     Vy (nx,ny+1); txy (nx-1,ny-1); dPrdtau, Rp (nx-2,ny-2); Vprof (nx).
 *   - Slab windows: a process that holds rows iy0+1 .. iy0+ny of a global grid
 *     of nyg rows passes (iy0, nyg); a whole domain passes iy0=0, nyg=ny.
 *     Interior/boundary guards are evaluated on the global row index, so the
 *     same kernels serve the single-GPU and the slab-decomposed run.
 *   - Every function returns 0 on success, nonzero on error; the message is
 *     available from nsk_last_error() on the calling thread.  No exceptions
 *     cross the boundary.
 *   - Calls are asynchronous on the context's HIP stream unless stated
 *     otherwise.  A context belongs to one host thread.
 */
#ifndef NSK_H
#define NSK_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nsk_ctx nsk_ctx;

/* ---- lifetime / error / stream ------------------------------------------ */

/* Replaces `@init_parallel_stencil(CUDA, Float64, 2)` (NavierStokes.jl:4-8):
 * binds a context to HIP device `device` and creates its stream and scratch. */
int nsk_create(int device, nsk_ctx **out);
int nsk_destroy(nsk_ctx *ctx);
const char *nsk_last_error(void);
const char *nsk_version(void);
/* Return values other than 0.  The first two are what every entry point may return; the last two
 * belong to the code3 time steps (nsk_bub_code3_step / nsk_bub_code3_run). */
#define NSK_ERR_HIP 1            /* a HIP runtime call failed */
#define NSK_ERR_ARGUMENT 2       /* an argument or a size is not usable */
#define NSK_ERR_FRONT_CAPACITY 3 /* the regridded front needs more than `cap` points: the place where
                                  * scripts/code3_fun.jl:462-465 resizes its vectors; see
                                  * nsk_bub_code3_run_status for how a caller continues */
#define NSK_ERR_TIMEOUT 5        /* nsk_slabs_*: a host-side wait behind an exchange or an all-reduce ran into
                                  * its deadline (nsk_slabs_set_option "deadline_ms", "first_contact_ms");
                                  * nsk_sync / nsk_download / residuals: a kernel whose workgroups wait for each
                                  * other gave up (nsk_set_option "poisson_small") */
#define NSK_ERR_FRONT_OUTSIDE 4  /* a front point left the box [0,Lx] x [0,Ly]: the reference prints a
                                  * message (scripts/code3_fun.jl:186-188) and its next array access
                                  * by that point throws; the state is not to be used */
/* Use an existing hipStream_t (e.g. the host framework's current stream)
 * instead of the context's own; NULL is HIP's default stream. */
int nsk_set_stream(nsk_ctx *ctx, void *hip_stream);
/* Return to the stream the context created for itself. */
int nsk_use_own_stream(nsk_ctx *ctx);
/* Tuning and test switches with no counterpart in the reference; none of them changes a
 * result bit.
 *   "poisson_k"            passes of the Poisson iteration per kernel launch (1..8, 0 = chosen per
 *                          grid); reading it back gives the value last used
 *   "poisson_rows"         rows per wavefront task of that kernel (0 = chosen per grid)
 *   "poisson_rows_used"    (read only) rows per task of the last launch
 *   "poisson_edge_percent" task height of the first/last column strip in % of the others (0 = 75)
 *   "poisson_xcd"          1 = task numbering that keeps neighbouring tasks on one XCD (default)
 *   "poisson_stream"       rows of Pr and dPrdtau requested with the non-temporal cache hint: 1 / 0, -1 = where
 *                          it pays by the size of the fields against the last-level cache (default)
 *   "poisson_stage"        1 = long runs of passes on dense arrays go through pitched copies (default)
 *   "poisson_div"          -1 = automatic, 0 / 2 / 3 / 4 = ask for IEEE / 2- / 3- / 4-operation
 *                          division; a form the two divisors dx, dy do not qualify for is replaced
 *                          by the next exact one (2 -> 3 -> 4 -> IEEE)
 *   "poisson_div_used"     (read only) the form the last fused Poisson call ran
 *   "poisson_small"        nsk_ns_poisson_iters on whole domains whose tiles fit on chip (up to 256 tiles of up
 *                          to 6144 cells with their frames; 153 x 255 uses 209), runs of 12 passes or
 *                          more: 1 = the whole run in one launch, the fields in LDS and registers, tiles
 *                          handing their outer cells to their neighbours every "poisson_small_halo"
 *                          passes (default), 0 = launches of "poisson_k" passes (as does a "poisson_k" or
 *                          "poisson_rows" other than 0: the caller has pinned that plan).  "poisson_k" and
 *                          "poisson_rows_used" read back the run's length and the tile height then.  A
 *                          wait between workgroups that runs out of polls ends the launch; the next
 *                          nsk_sync / nsk_download / residual of the context returns NSK_ERR_TIMEOUT.
 *                          The workgroups of such a launch must be resident together: launches of
 *                          different contexts of one process are chained on the device (an event wait,
 *                          no host wait); another PROCESS that runs the same kind of launch on the same
 *                          device at the same time can starve both until the polls run out -- switch
 *                          the option off where several processes share a device.
 *   "poisson_small_halo"   passes between two hand-overs = width of a tile's frame (0 = 8)
 *   "poisson_small_cols", "poisson_small_rows"  tile of a workgroup (0 = the smallest square tiles that
 *                          cover the grid with no more workgroups than compute units)
 *   "poisson_small_spin_limit"  polls before a workgroup gives up on a neighbour (0 = 2^22)
 *   "poisson_small_test_absent" testing: 1 + the workgroup that leaves at once (0 = none)
 *   "poisson_pairs"        launches of "poisson_k" passes on pitched storage (two-operation or IEEE division; not
 *                          nsk_ns_poisson_iters_static): 1 = the tasks of a column strip in
 *                          pairs that start at a common row and walk away from it, one up and one down the
 *                          array, handing each other that row's values of every pass through LDS, so that
 *                          neither recomputes halo rows at that end -- where the launch planner estimates
 *                          that form to be the faster one (default; 5-7 % faster at 2559 x 4095),
 *                          3 = in pairs wherever the launch can have them (for tests),
 *                          0 = every task with halo rows at both ends, 2 = the launch shape of 3 without
 *                          pairs (for timing).  NSK_POISSON_PAIRS in the environment sets the default.
 *                          Strips with boundary-column work pair chunks of "poisson_edge_percent" (0 = 62 beside pairs).
 *   "poisson_pairs_used"   (read only) 1 = the last launch of "poisson_k" passes ran its tasks in pairs
 *   "poisson_tall_percent" launches of "poisson_k" passes: interior tasks in pairs of
 *                          poisson_rows * (100 +- this) / 100 rows that start together, so that the last
 *                          round of tasks ends evenly (0 = all alike, the default: at most 2 % gained)
 *   "poisson_edge_task_rows" nsk_ns_poisson_iters_edges_first: rows per task of the edge rows' launch
 *                          (0 = 6)
 *   "sor_pipeline"         nsk_bub_solve_pressure / nsk_bub_construct_den: 1 = several lexicographic sweeps
 *                          in flight on the anti-diagonals when the grid fits LDS (default), 0 = one
 *                          sweep at a time; same results
 *   "sor_grid"             the same two solvers on grids beyond one workgroup's LDS: 1 = tiles of 64 x
 *                          "sor_grid_rows" cells on the whole chip, several sweeps in flight (default),
 *                          0 = one workgroup, one sweep at a time; same results
 *   "sor_grid_rows"        tile height of that solver (4..64, 0 = 32)
 *   "sor_grid_copies"      copies of the unknown that solver keeps (one per sweep in flight; 0 = by the grid,
 *                          else >= 3: fewer copies than sweeps fit in flight, for tests)
 *   "sor_fast_division"    the sweeps-in-flight pressure solve: 1 = quotients by the face coefficients through
 *                          their reciprocals and two fused corrections, which gives the bits of the IEEE
 *                          division while operands stay in [2^-900, 2^900] (else the solve is repeated
 *                          with IEEE divisions) (default), 0 = IEEE divisions; same results
 *   "bub_merge_launches"   nsk_bub_dns_step / nsk_bub_code3_step on grids the sweeps-in-flight solver takes:
 *                          1 = the element-wise kernels around a solve are phases of the solver's launch
 *                          (default), 0 = one launch per kernel of the reference; same results
 *   "step_fused"           nsk_ns_step: 1 = three fused kernels around the Poisson loop (default; the
 *                          stress arrays txx, tyy, txy are then not written), 0 = one launch per
 *                          call site of NavierStokes.jl:79-100 */
int nsk_set_option(nsk_ctx *ctx, const char *name, int value);
int nsk_get_option(nsk_ctx *ctx, const char *name, int *value);
/* Blocks until all work queued on the context's stream is done. */
int nsk_sync(nsk_ctx *ctx);

/* ---- memory (mirrors @zeros / Array(A) / A .= B) ------------------------- */

/* `@zeros(n1,n2)` (NavierStokes.jl:58-70). */
int nsk_zeros(nsk_ctx *ctx, int n1, int n2, double **out);
int nsk_free(nsk_ctx *ctx, double *dev);
/* `Data.Array(host)` (NavierStokes.jl:72); synchronous. */
int nsk_upload(nsk_ctx *ctx, double *dev, const double *host, size_t count);
/* `Array(A)` (NavierStokes.jl:75,109); synchronous. */
int nsk_download(nsk_ctx *ctx, double *host, const double *dev, size_t count);
/* `A_o .= A` (NavierStokes.jl:99). */
int nsk_copy(nsk_ctx *ctx, double *dst, const double *src, size_t count);

/* ---- kernel-granular entry points, 1:1 with the @parallel call sites ----- */

/* update_τ!   NavierStokes.jl:116-121, call site :79 */
int nsk_update_tau(nsk_ctx *ctx, double *txx, double *tyy, double *txy,
                   const double *Vx, const double *Vy, double mu, double dx,
                   double dy, int nx, int ny);
/* predict_V!  NavierStokes.jl:123-127, call site :80 */
int nsk_predict_V(nsk_ctx *ctx, double *Vx, double *Vy, const double *txx,
                  const double *tyy, const double *txy, double rho, double g,
                  double dt, double dx, double dy, int nx, int ny);
/* set_sphere! NavierStokes.jl:225-250, call sites :81,:97 */
int nsk_set_sphere(nsk_ctx *ctx, double *C, double *Vx, double *Vy, double a2,
                   double b2, double ox, double oy, double sinb, double cosb,
                   double lx, double ly, double dx, double dy, int nx, int ny,
                   int iy0);
/* update_∇V!  NavierStokes.jl:129-132, call site :82 */
int nsk_update_divV(nsk_ctx *ctx, double *divV, const double *Vx,
                    const double *Vy, double dx, double dy, int nx, int ny);
/* update_∇V! writing a pitched divV (leading dimension ld >= nx), as the fused
 * Poisson entry points accept it */
int nsk_update_divV_ld(nsk_ctx *ctx, double *divV, int ld, const double *Vx,
                       const double *Vy, double dx, double dy, int nx, int ny);
/* update_dPrdτ! NavierStokes.jl:134-137, call site :85 */
int nsk_update_dPrdtau(nsk_ctx *ctx, const double *Pr, double *dPrdtau,
                       const double *divV, double rho, double dt, double dtau,
                       double damp, double dx, double dy, int nx, int ny);
/* update_Pr!  NavierStokes.jl:139-142, call site :86 */
int nsk_update_Pr(nsk_ctx *ctx, double *Pr, const double *dPrdtau, double dtau,
                  int nx, int ny);
/* set_bc_Pr!  NavierStokes.jl:186-190 (bc_x! :155-159 then bc_yval! :173-177),
 * call site :87.  full_rows=0 reproduces the reference's launch range of
 * bc_x!, which covers only rows iy <= size(A,1); full_rows=1 covers all rows. */
int nsk_set_bc_Pr(nsk_ctx *ctx, double *Pr, double val, int nx, int ny,
                  int full_rows, int iy0, int nyg);
/* compute_res! NavierStokes.jl:144-147, call site :89 */
int nsk_compute_res(nsk_ctx *ctx, double *Rp, const double *Pr,
                    const double *divV, double rho, double dt, double dx,
                    double dy, int nx, int ny);
/* maximum(abs.(A)) NavierStokes.jl:90; NaN-propagating; synchronous. */
int nsk_maxabs(nsk_ctx *ctx, const double *A, size_t count, double *out);
/* correct_V!  NavierStokes.jl:149-153, call site :96 */
int nsk_correct_V(nsk_ctx *ctx, double *Vx, double *Vy, const double *Pr,
                  double dt, double rho, double dx, double dy, int nx, int ny);
/* correct_V! reading a pitched Pr (leading dimension ld >= nx) */
int nsk_correct_V_ld(nsk_ctx *ctx, double *Vx, double *Vy, const double *Pr, int ld,
                     double dt, double rho, double dx, double dy, int nx, int ny);
/* set_bc_Vel! NavierStokes.jl:179-184 (bc_y!(Vx), bc_x!(Vy), bc_yV!(Vy,Vprof)),
 * call site :98 */
int nsk_set_bc_Vel(nsk_ctx *ctx, double *Vx, double *Vy, const double *Vprof,
                   int nx, int ny, int full_rows, int iy0, int nyg);
/* advect! (+ backtrack!, lerp) NavierStokes.jl:192-223, call site :100 */
int nsk_advect(nsk_ctx *ctx, double *Vx, const double *Vx_o, double *Vy,
               const double *Vy_o, double *C, const double *C_o, double dt,
               double dx, double dy, int nx, int ny, int iy0, int nyg);

/* ---- fused entry points --------------------------------------------------- */

/* Scalars of runme(), NavierStokes.jl:14-56, plus the slab window and the
 * storage of the three Poisson fields.
 *
 * Storage: the fused entry points accept pitched fields.  Pr and divV share
 * the leading dimension ld_pr (0 = dense, nx).  dPrdtau either is the
 * reference's (nx-2, ny-2) array with leading dimension ld_dp (0 = dense,
 * nx-2; pointer at its element [1,1]), or -- dp_embedded = 1 -- an array laid
 * out exactly like Pr (same extents and ld_pr; pointer at its first element)
 * whose element for pressure cell (ix,iy) sits at the same offset as
 * Pr[ix,iy]; its rows 1, ny and columns 1, nx are unused.  The vectorised
 * kernels run when dp_embedded = 1, ld_pr is even and the three pointers are
 * 16-byte aligned; any other storage runs the same arithmetic with 8-byte
 * accesses or, for 32 passes and more, on pitched copies kept by the context. */
typedef struct nsk_ns_params {
  int nx, ny;        /* window extents (ny = rows held by this process) */
  int iy0, nyg;      /* global row offset and global row count          */
  int niter, nchk;   /* NavierStokes.jl:42-43                            */
  int full_rows;     /* 0 = reference launch range of bc_x!              */
  int ld_pr;         /* leading dimension of Pr and divV, 0 = nx         */
  int ld_dp;         /* leading dimension of dPrdtau, 0 = nx-2           */
  int dp_embedded;   /* 1 = dPrdtau stored on the Pr layout (see above)  */
  double lx, ly, rho, vin, psc, mu, g, a2, b2, ox, oy, sinb, cosb;
  double eps_it, dx, dy, dt, damp, dtau;
} nsk_ns_params;

/* `count` consecutive passes of NavierStokes.jl:85-87
 * (update_dPrdτ!, update_Pr!, set_bc_Pr!(Pr,0.0)) with no residual check.
 * Asynchronous.  Rows 1 and ny of the window are physical boundaries only
 * when they are rows 1 / nyg of the global grid; otherwise they are halo rows
 * and are left untouched (after `count` passes the outermost `count` halo
 * rows of Pr and dPrdtau are stale and must be exchanged by the caller). */
int nsk_ns_poisson_iters(nsk_ctx *ctx, double *Pr, double *dPrdtau,
                         const double *divV, const nsk_ns_params *p, int count);

/* nsk_ns_poisson_iters for callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the
 * reference.  Same arguments, same results bit for bit; the ping-pong partners of Pr and dPrdtau
 * live in a static device array of 2 x 4 Mi doubles (64 MiB per GPU, held until the process
 * ends), so the call never calls hipMalloc / hipFree.  A window whose leading dimension x rows
 * exceeds 4 Mi (4 194 304) values is rejected with an error.  Storage the vectorised kernels
 * cannot take is iterated in place by the 8-byte kernels, never on staging copies.  All calls on
 * one GPU share the array: calls that overlap in time on one GPU (two streams, two contexts, two
 * threads) race.  Asynchronous. */
int nsk_ns_poisson_iters_static(nsk_ctx *ctx, double *Pr, double *dPrdtau,
                                const double *divV, const nsk_ns_params *p, int count);

/* nsk_ns_poisson_iters for a window whose edge rows somebody is waiting for -- the halo exchange
 * of a slab decomposition (no counterpart in the reference, which is single-device).  Same values
 * bit for bit.  The last launch of the call is cut in three: the rows
 * [skip_lo, skip_lo + edge_lo) and [ny - skip_hi - edge_hi, ny - skip_hi) (0-based window rows) are
 * produced first, by short tasks on edge_stream (a hipStream_t of the context's device); the rows
 * between them on the context's stream; the first skip_lo and the last skip_hi rows -- halo rows the
 * caller is about to overwrite with received ones -- are not produced and hold undefined values
 * afterwards.  Work queued on edge_stream after the call (the transfer: keep it on this stream, a
 * third stream may share a hardware queue with the interior's) starts as soon as the edge rows are final,
 * while the interior is still being computed; work queued on the context's stream after the call
 * sees every row.  When no cut is possible (no rows between the edges, an odd number of launches,
 * dense arrays iterated on staging copies) the plain sequence runs, all rows are produced, and
 * edge_stream is ordered behind it.  Asynchronous. */
int nsk_ns_poisson_iters_edges_first(nsk_ctx *ctx, double *Pr, double *dPrdtau,
                                     const double *divV, const nsk_ns_params *p, int count,
                                     int skip_lo, int edge_lo, int edge_hi, int skip_hi,
                                     void *edge_stream);

/* How nsk_ns_poisson_iters would cut a launch on an nx x ny window: host-only, needs no GPU (no
 * counterpart in the reference; for tuning and tests).  num_simds = 4 x compute units;
 * staged = 1 for pitched storage; k_forced / rows_forced = 0 leave the choice to the planner;
 * k_limit = passes asked for.  out[6] = passes per launch, rows per task, rows per task of the
 * two edge strips, column strips, tasks, estimated ns per pass. */
int nsk_poisson_plan(int nx, int ny, int num_simds, int staged, int k_forced, int k_limit,
                     int rows_forced, int edge_percent, int *out);
/* Which evaluation of x/d the Poisson kernel uses for the divisor d (host-only, needs no GPU; no
 * counterpart in the reference, for tests): 0 = IEEE division, 2 / 3 / 4 = the 2-, 3- or
 * 4-operation FMA form, each chosen only when it yields the correctly rounded quotient for every
 * dividend (the 2-operation form after an exhaustive check of this divisor's near-midpoint
 * quotients).  rh_out / rl_out (may be NULL) receive 1/d as an unevaluated sum hi + lo. */
int nsk_division_form(double d, int *form_out, double *rh_out, double *rl_out);
/* Residual of NavierStokes.jl:89-90 on the window: writes Rp (may be NULL)
 * and returns max|Rp| over the rows this window owns (rows own_lo..own_hi,
 * 1-based window rows of Pr; pass 2 and ny-1 for a whole domain).
 * Synchronous. */
int nsk_ns_residual(nsk_ctx *ctx, double *Rp, const double *Pr,
                    const double *divV, const nsk_ns_params *p, int own_lo,
                    int own_hi, double *maxabs_out);

/* The whole Poisson loop NavierStokes.jl:84-95 on a single domain: iterates in
 * chunks of nchk, evaluates err = max|Rp|*ly^2/psc after each chunk and stops
 * when err < eps_it or err is not finite.  err_hist/iter_hist (host, capacity
 * hist_cap, may be NULL) receive the (iter, err) pairs the reference prints at
 * :92.  Synchronous. */
int nsk_ns_poisson(nsk_ctx *ctx, double *Pr, double *dPrdtau, double *Rp,
                   const double *divV, const nsk_ns_params *p, int *iters_out,
                   double *err_hist, int *iter_hist, int hist_cap,
                   int *nhist_out);

/* Device arrays of runme(), NavierStokes.jl:58-72.  Pr, dPrdtau and divV are
 * stored as the parameter block says; all other arrays are dense. */
typedef struct nsk_ns_state {
  double *Pr, *dPrdtau, *C, *C_o, *txx, *tyy, *txy, *Vx, *Vy, *Vx_o, *Vy_o,
      *divV, *Rp, *Vprof;
} nsk_ns_state;

/* One pass of the time-loop body NavierStokes.jl:79-100 on a single domain.
 * Synchronous (the Poisson loop reads the residual back).  On return Vx, Vy, C, Pr, dPrdtau, divV,
 * Rp and Vx_o, Vy_o, C_o hold what the reference's arrays hold after :100; txx, tyy, txy only
 * with the option "step_fused" = 0 (the fused kernels keep the stresses in LDS). */
int nsk_ns_step(nsk_ctx *ctx, const nsk_ns_state *s, const nsk_ns_params *p,
                int *iters_out, double *err_hist, int *iter_hist, int hist_cap,
                int *nhist_out);

/* ---- slab decomposition of the Poisson loop over several GPUs ----------------------------------
 * The reference is single-device; scripts/DNS.jl:65 (`@hide_communication`) is its only trace of
 * a multi-device intent.  These entry points run NavierStokes.jl:84-95 on a global nx x (world *
 * ny_loc) grid cut along iy into `world` slabs.  A group holds the slabs of one process:
 * first_rank .. first_rank + nlocal - 1, slab i on HIP device devices[i] (all slabs: one host
 * thread drives every GPU; one slab: one process per GPU).  The library owns the three pitched
 * window arrays of each slab (Pr, dPrdtau on the Pr layout, divV; nsk_slabs_fields hands out the
 * device pointers, nsk_slabs_window the geometry).  Between two exchanges a slab runs `halo` passes
 * on its window of ny_loc + 2*halo rows; results equal the single-domain loop bit for bit.
 * transport: 0 = automatic (device copies when nlocal == world, else RCCL), 1 = RCCL (ncclSend /
 * ncclRecv in one ncclGroup per exchange on a communication stream per slab, ncclAllReduce(max) of
 * the residual's bit pattern; librccl is loaded at run time), 2 = peer-to-peer device copies
 * (needs nlocal == world).  unique_id: the 128 bytes of nsk_rccl_unique_id from ONE process,
 * passed to all (NULL when nlocal == world).  No call blocks the host except upload, download,
 * residual, poisson and sync. */
typedef struct nsk_slabs nsk_slabs;
int nsk_rccl_unique_id(void *id128);
int nsk_slabs_create(int world, int first_rank, int nlocal, const int *devices,
                     const nsk_ns_params *global, int ny_loc, int halo, int transport,
                     const void *unique_id, nsk_slabs **out);
int nsk_slabs_destroy(nsk_slabs *g);
/* Geometry of local slab `local` (any output may be NULL): global rank, device, global 0-based row
 * of the window's first row, window rows, owned window rows [own_lo, own_hi) 0-based, leading
 * dimension of the three arrays. */
int nsk_slabs_window(nsk_slabs *g, int local, int *rank, int *device, int *iy0, int *ny_win,
                     int *own_lo, int *own_hi, int *ld);
int nsk_slabs_fields(nsk_slabs *g, int local, double **Pr, double **dPrdtau, double **divV);
/* Scatter / gather of GLOBAL host arrays in the reference's layout: field 0 = Pr (nx x nyg),
 * 1 = dPrdtau ((nx-2) x (nyg-2)), 2 = divV (nx x nyg) and, after nsk_slabs_init_flow, 3 = Vx
 * (nx+1 x nyg), 4 = Vy (nx x nyg+1), 5 = C (nx x nyg).  Upload fills each local window including
 * its halo rows; download writes only the rows (of Vy: faces) the local slabs own.  Synchronous. */
int nsk_slabs_upload(nsk_slabs *g, int field, const double *host_global);
int nsk_slabs_download(nsk_slabs *g, int field, double *host_global);
/* Refresh the halo rows of the fields in field_mask (bit 0 Pr, bit 1 dPrdtau, bit 2 divV). */
int nsk_slabs_exchange(nsk_slabs *g, int field_mask);
/* `count` passes of NavierStokes.jl:85-87 on every local slab with an exchange of Pr and dPrdtau
 * after every `halo` passes (and after the last). */
int nsk_slabs_poisson_iters(nsk_slabs *g, int count);
/* max|Rp| of NavierStokes.jl:89-90 over the whole grid (identical in every process). */
int nsk_slabs_residual(nsk_slabs *g, double *maxabs_out);
/* The loop NavierStokes.jl:84-95 (niter, nchk, eps_it, ly, psc of the global parameter block);
 * outputs as nsk_ns_poisson. */
int nsk_slabs_poisson(nsk_slabs *g, int *iters_out, double *err_hist, int *iter_hist,
                      int hist_cap, int *nhist_out);
/* The remaining arrays of runme() (NavierStokes.jl:58-72) as windows on every local slab, zero except
 * the inflow row Vy[:,1] = Vprof (:73); Vprof_host holds nx values.  Needs halo >= 4. */
int nsk_slabs_init_flow(nsk_slabs *g, const double *Vprof_host);
/* One pass of the time-loop body NavierStokes.jl:79-100 on the slab-decomposed grid (outputs as
 * nsk_ns_step).  The back-trace of advect! must stay inside the halo: reach = max|Vy| dt/dy + 2 >
 * halo - 1 is an error, not a wrong answer: the bound is tested on entry (velocities as they are) and
 * again after the projection; after the second kind of failure the windows hold a half-finished step and
 * the group has to be set up again with a wider halo.  Synchronous. */
int nsk_slabs_step(nsk_slabs *g, int *iters_out, double *err_hist, int *iter_hist, int hist_cap,
                   int *nhist_out);
int nsk_slabs_sync(nsk_slabs *g);
/* Transport check that needs no second GPU (RCCL groups only): every local slab sends the first
 * `rows` owned rows of Pr to itself and receives them into the first `rows` owned rows of divV. */
int nsk_slabs_rccl_self_exchange(nsk_slabs *g, int rows);
int nsk_slabs_info(nsk_slabs *g, int *world, int *nlocal, int *transport, int *exchanges);
/* Switches with no effect on any result: "step_fused" (default 1) -- nsk_slabs_step runs the three fused
 * kernels of nsk_ns_step around the Poisson loop, on windows (guards on window rows, cylinder mask and
 * boundary conditions on global rows), 0 = one launch per kernel of NavierStokes.jl:79-100.
 * "edges_first" (default 1) -- the last launch of every
 * block of Poisson passes produces the rows the neighbours wait for first
 * (nsk_ns_poisson_iters_edges_first), so that the halo exchange runs beside the rest of that
 * launch; 0 = whole window, then the exchange.
 * "deadline_ms" (default 60000), "first_contact_ms" (default 180000; applies until the first wait behind
 * an exchange or an all-reduce has succeeded): every host-side wait of the group (nsk_slabs_sync,
 * nsk_slabs_residual and the calls built on them) polls its stream and gives up with NSK_ERR_TIMEOUT and
 * a description (slab, device, neighbours, bytes, transport, schedule, exchanges queued, whether the last
 * one completed) after that many milliseconds instead of waiting for good; 0 = wait for good.  The group
 * is not usable after a timeout.  "test_stall_ms": the next exchange is held back on the device for that
 * long (a bounded idle kernel; for the test of the deadline). */
int nsk_slabs_set_option(nsk_slabs *g, const char *name, int value);
/* Reads a group switch back; any other name is an option of nsk_get_option, answered by the context of the
 * first local slab (e.g. "poisson_k": passes per launch its windows run with). */
int nsk_slabs_get_option(nsk_slabs *g, const char *name, int *value);

/* ---- bubble / drop solvers: scripts/DNS.jl (= scripts/code1.m) and scripts/code3_fun.jl ---
 * Grid arrays are (nx+2) x (ny+2) with one ghost layer; u is (nx+1) x (ny+2), v is
 * (nx+2) x (ny+1); fronts are vectors with points 2..Nf+1 and two periodic ghost points
 * (1-based).  DNS.jl does not run as written (undefined variables, statements cut by line
 * breaks); these entry points implement the algorithm it transcribes, scripts/code1.m:33-101. */

/* set_density!  DNS.jl:128-137 / code1.m:25-28; with m != NULL also set_density_viscosity!
 * code3_fun.jl:8-18.  x, y: cell-centre coordinates (nx+2), (ny+2). */
int nsk_bub_set_density(nsk_ctx *ctx, double *r, double *m, double xc, double yc,
                        const double *x, const double *y, double rad, double rho2,
                        double m2, int nx, int ny);
/* set_tangential!  DNS.jl:118-126 / code1.m:33-34 / code3_fun.jl:28-35 */
int nsk_bub_set_tangential(nsk_ctx *ctx, double *u, double *v, double usouth,
                           double unorth, double vwest, double veast, int nx, int ny);
/* temp_uvel! + temp_vvel!  DNS.jl:159-179 / code1.m:36-52 (ut or vt may be NULL) */
int nsk_bub_temp_vel(nsk_ctx *ctx, double *ut, double *vt, const double *u,
                     const double *v, const double *r, double dt, double dx, double dy,
                     double gx, double gy, double m0, int nx, int ny);
/* temp_uvel_adv! + temp_vvel_adv!  code3_fun.jl:37-62 */
int nsk_bub_temp_vel_adv(nsk_ctx *ctx, double *ut, double *vt, const double *u,
                         const double *v, const double *r, double dt, double dx,
                         double dy, const double *fx, const double *fy, double gx,
                         double gy, double rro, int nx, int ny);
/* temp_uvel_diff! + temp_vvel_diff!  code3_fun.jl:64-90 */
int nsk_bub_temp_vel_diff(nsk_ctx *ctx, double *ut, double *vt, const double *u,
                          const double *v, const double *r, double dt, double dx,
                          double dy, const double *m, int nx, int ny);
/* set_boundary!  DNS.jl:138-148 / code1.m:55-57 / code3_fun.jl:415-420: rt = r, lrg on borders */
int nsk_bub_set_boundary(nsk_ctx *ctx, double *rt, const double *r, double lrg, int nx, int ny);
/* compute_source!  DNS.jl:149-157 / code1.m:59-65 / code3_fun.jl:92-102 */
int nsk_bub_compute_sources(nsk_ctx *ctx, double *tmp1, double *tmp2, const double *rt,
                            const double *ut, const double *vt, double dt, double dx,
                            double dy, int nx, int ny);
/* solve_pressure! with its iteration loop and stopping test  DNS.jl:83-93,181-189 /
 * code1.m:68-78 / code3_fun.jl:104-124.  Lexicographic in-place SOR; returns the sweeps
 * performed and the last max|old-new|.  Synchronous when an output pointer is given. */
int nsk_bub_solve_pressure(nsk_ctx *ctx, double *p, const double *tmp1,
                           const double *tmp2, const double *rt, double beta, double dx,
                           double dy, int nx, int ny, int maxit, double maxError,
                           int *sweeps_out, double *change_out);
/* Red/black SOR: solve_pressure_kernel_red!/black! code3_perf_gpu.jl:129-183 with the loop
 * :609-617 (same kernels in scripts/tests/test_solve_pressure_gpu.jl:6-61).  Ordinary grids,
 * scales past one workgroup; NOT the lexicographic update order of nsk_bub_solve_pressure
 * (results agree with it only to the iteration error).  Double precision.  Synchronous. */
int nsk_bub_solve_pressure_rb(nsk_ctx *ctx, double *p, const double *tmp1,
                              const double *tmp2, const double *rt, double beta, double dx,
                              double dy, int nx, int ny, int maxit, double maxError,
                              int *sweeps_out, double *change_out);
/* update_dPdτ! + update_p! code1_perf_gpu.jl:98-118 with the loop :228-242: damped
 * pseudo-transient iteration of the variable-density pressure equation; dPdtau ((nx+2)x(ny+2))
 * persists between calls.  Double precision.  Synchronous. */
int nsk_bub_pt_pressure(nsk_ctx *ctx, double *p, double *dPdtau, const double *tmp1,
                        const double *rt, double dtau, double damp, double dx, double dy,
                        int nx, int ny, int maxit, double maxError, int *iters_out,
                        double *change_out);
/* correct_uvel! + correct_vvel!  DNS.jl:190-202 / code1.m:80-86 (u or v may be NULL) */
int nsk_bub_correct_vel(nsk_ctx *ctx, double *u, double *v, const double *ut,
                        const double *vt, const double *p, const double *r, double dt,
                        double dx, double dy, int nx, int ny);
/* advect_density!  DNS.jl:204-212 as code1.m:88-96 states it */
int nsk_bub_advect_density(nsk_ctx *ctx, double *r, const double *ro, const double *u,
                           const double *v, double dt, double dx, double dy, double m0,
                           int nx, int ny);
/* velocity!  DNS.jl:214-220 / code1.m:100-101 (all (nx+1) x (ny+1) corner values) */
int nsk_bub_velocity(nsk_ctx *ctx, double *uu, double *vv, const double *u,
                     const double *v, int nx, int ny);
/* surface tension: tangents code3_fun.jl:372-378, spreading :365-366,380-401 */
int nsk_bub_surface_tension(nsk_ctx *ctx, double *fx, double *fy, double *tx, double *ty,
                            const double *xf, const double *yf, double sigma, double dx,
                            double dy, int nx, int ny, int Nf);
/* advect_front! + move_front!  code3_fun.jl:144-195; out_of_bounds (may be NULL) counts the
 * points the reference would print a warning for */
int nsk_bub_advect_move_front(nsk_ctx *ctx, double *uf, double *vf, double *xf, double *yf,
                              const double *u, const double *v, double dx, double dy,
                              double dt, double Lx, double Ly, int nx, int ny, int Nf,
                              int *out_of_bounds);
/* fx = fy = 0 and distribute!  code3_fun.jl:433-435,197-222 */
int nsk_bub_distribute(nsk_ctx *ctx, double *fx, double *fy, const double *xf,
                       const double *yf, double rho1, double rho2, double dx, double dy,
                       int nx, int ny, int Nf);
/* construct_den!  code3_fun.jl:224-240 */
int nsk_bub_construct_den(nsk_ctx *ctx, double *r, const double *fx, const double *fy,
                          double beta, int maxit, double maxError, double dx, double dy,
                          int nx, int ny, int *sweeps_out);
/* m = m1 and update_visco!  code3_fun.jl:440-441,242-249 */
int nsk_bub_update_visco(nsk_ctx *ctx, const double *r, double *m, double m1, double m2,
                         double rho1, double rho2, int nx, int ny);
/* a = 0.5*(a + b)  code3_fun.jl:445-450 */
int nsk_bub_average(nsk_ctx *ctx, double *a, const double *b, size_t count);
/* front regridding  code3_fun.jl:455-494; xf, yf and the two scratch vectors hold `cap`
 * points; NSK_ERR_FRONT_CAPACITY is returned where the reference would resize its vectors (the
 * front as it was on entry is then in the scratch vectors, Nf + 2 points).  Synchronous. */
int nsk_bub_regrid_front(nsk_ctx *ctx, double *xf, double *yf, double *xscratch,
                         double *yscratch, double dx, double dy, int Nf, int cap,
                         int *Nf_out);

/* Whole time steps of the two bubble drivers: the kernels above in the drivers' order, queued in
 * one call with a single wait at the end for the sweep counts (the reference reads every count
 * back, and scripts/DNS.jl:83-93 synchronises per sweep).  Same values as the kernel-by-kernel
 * sequence.  Scalars of scripts/DNS.jl:16-45 and scripts/code3_fun.jl:296-309,337,340; the DNS
 * driver uses m1 as its one viscosity (m0) and ignores m2, sigma, rro. */
typedef struct nsk_bub_params {
  int nx, ny, maxit;
  double Lx, Ly, gx, gy, rho1, rho2, m1, m2, sigma, rro, usouth, unorth, vwest, veast;
  double dt, dx, dy, maxError, beta;
} nsk_bub_params;
/* device arrays of scripts/DNS.jl:52-62: u (nx+1, ny+2), v (nx+2, ny+1), uu, vv (nx+1, ny+1),
 * the others (nx+2, ny+2); r = rho, ro = its copy before advect_density!, rt = rhot */
typedef struct nsk_bub_dns_state {
  double *u, *v, *p, *ut, *vt, *tmp1, *tmp2, *r, *ro, *rt, *uu, *vv;
} nsk_bub_dns_state;
/* device arrays of scripts/code3_fun.jl:311-327,345-352: grid arrays as above (un, vn, rn, mn =
 * the copies of :357-362), front vectors of `cap` points (xs, ys: scratch of the regridding),
 * Nf = points of the front (updated by the step) */
typedef struct nsk_bub_code3_state {
  double *u, *v, *p, *ut, *vt, *tmp1, *tmp2, *fx, *fy, *un, *vn, *r, *m, *rn, *mn, *rt;
  double *xf, *yf, *xfn, *yfn, *uf, *vf, *tx, *ty, *xs, *ys;
  int Nf, cap;
} nsk_bub_code3_state;
/* scripts/DNS.jl:71-101 (form of scripts/code1.m:33-96) with the lexicographic SOR;
 * *sweeps_out = sweeps of the pressure solve (may be NULL: then the call does not wait). */
int nsk_bub_dns_step(nsk_ctx *ctx, const nsk_bub_dns_state *s, const nsk_bub_params *q,
                     int *sweeps_out);
/* scripts/code3_fun.jl:356-494; sweeps_out[0..3] = pressure and density sweeps of the two substeps
 * (may be NULL); s->Nf receives the regridded front size.  Synchronous. */
int nsk_bub_code3_step(nsk_ctx *ctx, nsk_bub_code3_state *s, const nsk_bub_params *q,
                       int *sweeps_out);
/* The time loops themselves, scripts/DNS.jl:70-102 and scripts/code3_fun.jl:354-494 (without the
 * plotting): nsteps time steps queued back to back, one wait at the end instead of one per step
 * (code3's front size stays on the device in between).  sweeps_out receives the sweep counts of
 * every step (nsteps values for DNS, 4 * nsteps for code3; may be NULL -- nsk_bub_dns_run then
 * does not wait at all).  Same values as nsteps calls of the step functions. */
int nsk_bub_dns_run(nsk_ctx *ctx, const nsk_bub_dns_state *s, const nsk_bub_params *q, int nsteps,
                    int *sweeps_out);
int nsk_bub_code3_run(nsk_ctx *ctx, nsk_bub_code3_state *s, const nsk_bub_params *q, int nsteps,
                      int *sweeps_out);
/* After nsk_bub_code3_step / nsk_bub_code3_run on this context: *steps_done = time steps of that call
 * that are complete.  When the call returned NSK_ERR_FRONT_CAPACITY, step number *steps_done + 1 is
 * complete except for its regridding (scripts/code3_fun.jl:455-494): s->Nf is still the size of the
 * front before it, its averaged front of s->Nf + 2 points is in xs, ys, its four sweep counts are in
 * sweeps_out, and nothing after it has run.  A caller continues the way the reference's resize! does:
 * larger front vectors (all ten; only xs, ys carry data), the averaged front copied into the new
 * xf, yf, nsk_bub_regrid_front on them, then the remaining steps.  No counterpart in the reference
 * (its vectors grow in place). */
int nsk_bub_code3_run_status(nsk_ctx *ctx, int *steps_done);

#ifdef __cplusplus
}
#endif
#endif /* NSK_H */
