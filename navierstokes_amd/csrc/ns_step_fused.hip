// The time-step body outside the Poisson loop as three kernels instead of eleven launches
// (scripts/NavierStokes.jl:79-82 and :96-100), on a whole domain or on a window of rows of one (a slab
// with halo rows: guards on the window's rows, the cylinder mask and the boundary conditions on global
// rows, exactly as the single kernels treat windows):
//
//   k_predictor   update_τ! + predict_V! + set_sphere! + update_∇V!   (:116-132, :225-250)
//                 out of place: reads Vx, Vy, C, writes the predicted and masked fields into the
//                 Vx_o, Vy_o, C_o arrays and ∇V.  The stresses are formed in LDS on a 64 x 8 tile
//                 with its halo and never reach memory.
//   k_corrector   correct_V! + set_sphere! + set_bc_Vel!               (:149-153, :179-184)
//                 in place on the Vx_o, Vy_o, C_o arrays.
//   k_advect_out  advect!                                              (:192-223)
//                 from the Vx_o, Vy_o, C_o arrays back into Vx, Vy, C, including the edge rows and
//                 columns advect! does not write.
//
// Two out-of-place kernels put the fields back where the caller keeps them and leave in the *_o
// arrays exactly what the reference's copies `Vx_o .= Vx` (:99) leave there, so the three copies
// disappear.  Every value is formed by the reference's expression in the reference's order, so the
// results equal the kernel-by-kernel sequence bit for bit.
#include "nsk_internal.h"

namespace {

struct StepArgs {
  int nx, ny, ld;                 // ny: rows of the window; ld: leading dimension of Pr and ∇V
  int iy0, nyg;                   // the window holds global rows iy0+1 .. iy0+ny of nyg
  int bottom, top;                // the window's first / last row is the domain's
  int xrows;                      // window rows of Vy that bc_x! reaches (NavierStokes.jl:155,181)
  double mu, dtr, rg, dx, dy;     // dtr = dt/ρ, rg = ρ*g
  double a2, b2, ox, oy, sinb, cosb, lx, ly, dt;
};

// set_sphere! (NavierStokes.jl:225-250): which of the three staggered points (ix, iy) are inside.
struct Mask {
  bool c, vx, vy;
};
__device__ __forceinline__ Mask mask_at(const StepArgs &a, int ix, int iy) {
  const double xv = (double)(ix - 1) * a.dx - a.lx / 2, yv = (double)(iy + a.iy0 - 1) * a.dy - a.ly / 2;
  const double xc = xv + a.dx / 2, yc = yv + a.dx / 2;  // (dx in yc as in the reference, :227)
  Mask m;
  {
    const double xr = (xc - a.ox) * a.cosb - (yc - a.oy) * a.sinb, yr = (xc - a.ox) * a.sinb + (yc - a.oy) * a.cosb;
    m.c = xr * xr / a.a2 + yr * yr / a.b2 < 1.05;
  }
  {
    const double xr = (xv - a.ox) * a.cosb - (yc - a.oy) * a.sinb, yr = (xv - a.ox) * a.sinb + (yc - a.oy) * a.cosb;
    m.vx = xr * xr / a.a2 + yr * yr / a.b2 < 1.0;
  }
  {
    const double xr = (xc - a.ox) * a.cosb - (yv - a.oy) * a.sinb, yr = (xc - a.ox) * a.sinb + (yv - a.oy) * a.cosb;
    m.vy = xr * xr / a.a2 + yr * yr / a.b2 < 1.0;
  }
  return m;
}

// ---- predictor ----------------------------------------------------------------------------------
// Tile of TX x TY cells (origin I0, J0, 1-based).  Index ranges held in LDS, relative to the origin:
//   old velocities    i in [-1, TX+1], j in [-1, TY+1]      (VW x VH)
//   stresses          i in [-1, TX],   j in [-1, TY]        (SW x SH)
//   new velocities    i in [ 0, TX],   j in [ 0, TY]        (NW x NH)
constexpr int TX = 64, TY = 8;
constexpr int VW = TX + 3, VH = TY + 3, SW = TX + 2, SH = TY + 2, NW = TX + 1, NH = TY + 1;

__global__ void __launch_bounds__(256) k_predictor(const double *__restrict__ Vx, const double *__restrict__ Vy,
                                                   const double *__restrict__ C, double *__restrict__ Vx_n,
                                                   double *__restrict__ Vy_n, double *__restrict__ C_n,
                                                   double *__restrict__ divV, StepArgs a) {
  __shared__ double sVx[VH][VW], sVy[VH][VW];
  __shared__ double sxx[SH][SW], syy[SH][SW], sxy[SH][SW];
  __shared__ double nVx[NH][NW], nVy[NH][NW];
  const int nx = a.nx, ny = a.ny;
  const int I0 = blockIdx.x * TX + 1, J0 = blockIdx.y * TY + 1;
  const int tid = threadIdx.x;
  // old velocities (0 outside the arrays: such values are never used by a guarded expression)
  for (int e = tid; e < VW * VH; e += 256) {
    const int li = e % VW, lj = e / VW, i = I0 - 1 + li, j = J0 - 1 + lj;
    sVx[lj][li] = (i >= 1 && i <= nx + 1 && j >= 1 && j <= ny) ? A2(Vx, nx + 1, i, j) : 0.0;
    sVy[lj][li] = (i >= 1 && i <= nx && j >= 1 && j <= ny + 1) ? A2(Vy, nx, i, j) : 0.0;
  }
  __syncthreads();
  // update_τ!  NavierStokes.jl:116-121
  const double mu2 = 2 * a.mu;
  for (int e = tid; e < SW * SH; e += 256) {
    const int li = e % SW, lj = e / SW, i = I0 - 1 + li, j = J0 - 1 + lj;  // old-velocity tile index = (li, lj)
    double txx = 0.0, tyy = 0.0, txy = 0.0;
    if (i >= 1 && i <= nx && j >= 1 && j <= ny) {
      const double dVx = sVx[lj][li + 1] - sVx[lj][li];
      const double dVy = sVy[lj + 1][li] - sVy[lj][li];
      const double div = dVx / a.dx + dVy / a.dy;
      txx = mu2 * (dVx / a.dx - div / 3.0);
      tyy = mu2 * (dVy / a.dy - div / 3.0);
      if (i <= nx - 1 && j <= ny - 1) {
        const double dyVx = sVx[lj + 1][li + 1] - sVx[lj][li + 1];
        const double dxVy = sVy[lj + 1][li + 1] - sVy[lj + 1][li];
        txy = a.mu * (dyVx / a.dy + dxVy / a.dx);
      }
    }
    sxx[lj][li] = txx;
    syy[lj][li] = tyy;
    sxy[lj][li] = txy;
  }
  __syncthreads();
  // predict_V! (:123-127) then set_sphere! (:225-250); stress tile index of (i, j) is (li + 1, lj + 1)
  for (int e = tid; e < NW * NH; e += 256) {
    const int li = e % NW, lj = e / NW, i = I0 + li, j = J0 + lj;
    double vx = sVx[lj + 1][li + 1], vy = sVy[lj + 1][li + 1];
    if (i >= 2 && i <= nx && j >= 2 && j <= ny - 1) {
      const double p = sxx[lj + 1][li + 1] - sxx[lj + 1][li];
      const double q = sxy[lj + 1][li] - sxy[lj][li];
      vx = vx + a.dtr * (p / a.dx + q / a.dy);
    }
    if (i >= 2 && i <= nx - 1 && j >= 2 && j <= ny) {
      const double p = syy[lj + 1][li + 1] - syy[lj][li + 1];
      const double q = sxy[lj][li + 1] - sxy[lj][li];
      vy = vy + a.dtr * (p / a.dy + q / a.dx - a.rg);
    }
    if (i <= nx + 1 && j <= ny + 1) {
      const Mask m = mask_at(a, i, j);
      if (m.vx) vx = 0.0;
      if (m.vy) vy = 0.0;
      if (li < TX && lj < TY) {  // this tile's own points go to memory
        if (j <= ny) A2(Vx_n, nx + 1, i, j) = vx;
        if (i <= nx) A2(Vy_n, nx, i, j) = vy;
        if (i <= nx && j <= ny) A2(C_n, nx, i, j) = m.c ? 1.0 : A2(C, nx, i, j);
      }
    }
    nVx[lj][li] = vx;
    nVy[lj][li] = vy;
  }
  __syncthreads();
  // update_∇V!  :129-132
  for (int e = tid; e < TX * TY; e += 256) {
    const int li = e % TX, lj = e / TX, i = I0 + li, j = J0 + lj;
    if (i <= nx && j <= ny)
      A2(divV, a.ld, i, j) = (nVx[lj][li + 1] - nVx[lj][li]) / a.dx + (nVy[lj + 1][li] - nVy[lj][li]) / a.dy;
  }
}

// ---- corrector ----------------------------------------------------------------------------------
// Every thread forms the corrected and masked value of its own point from its own old value and Pr,
// and stores it at its point and at the boundary points that set_bc_Vel! copies it to; nobody reads
// another point's velocity, so the update is race-free in place.  Order of the reference: bc_y!(Vx);
// bc_x!(Vy) on rows <= xrows; bc_yV!(Vy, Vprof) (bottom row = Vprof, top row = row below it, taken
// after bc_x!).
__global__ void __launch_bounds__(256) k_corrector(double *__restrict__ Vx, double *__restrict__ Vy,
                                                   double *__restrict__ C, const double *__restrict__ Pr,
                                                   const double *__restrict__ Vprof, StepArgs a) {
  const int nx = a.nx, ny = a.ny;
  const int i = blockIdx.x * 64 + (threadIdx.x & 63) + 1, j = blockIdx.y * 4 + (threadIdx.x >> 6) + 1;
  if (i > nx + 1 || j > ny + 1) return;
  const Mask m = mask_at(a, i, j);
  if (i <= nx && j <= ny && m.c) A2(C, nx, i, j) = 1.0;
  if (j <= ny) {  // Vx (nx+1, ny)
    double v = A2(Vx, nx + 1, i, j);
    if (i >= 2 && i <= nx && j >= 2 && j <= ny - 1)
      v = v - a.dtr * (A2(Pr, a.ld, i, j) - A2(Pr, a.ld, i - 1, j)) / a.dx;  // correct_V! :150
    if (m.vx) v = 0.0;
    // the domain's rows 1 and ny are copies of rows 2 and ny-1 (bc_y!); a window's other end rows are its own
    if (!(j == 1 && a.bottom) && !(j == ny && a.top)) A2(Vx, nx + 1, i, j) = v;
    if (j == 2 && a.bottom) A2(Vx, nx + 1, i, 1) = v;
    if (j == ny - 1 && a.top) A2(Vx, nx + 1, i, ny) = v;
  }
  if (i <= nx) {  // Vy (nx, ny+1)
    if (j == 1 && a.bottom) {
      A2(Vy, nx, i, 1) = Vprof[i - 1];  // bc_yV!: the bottom row is the inflow profile
    } else if (!(j == ny + 1 && a.top)) {
      double v = A2(Vy, nx, i, j);
      if (i >= 2 && i <= nx - 1 && j >= 2 && j <= ny)
        v = v - a.dtr * (A2(Pr, a.ld, i, j) - A2(Pr, a.ld, i, j - 1)) / a.dy;  // correct_V! :151
      if (m.vy) v = 0.0;
      const bool xbc = j <= a.xrows;  // bc_x! reaches this row
      const bool own = !(xbc && (i == 1 || i == nx));
      if (own) A2(Vy, nx, i, j) = v;
      if (xbc && i == 2) A2(Vy, nx, 1, j) = v;
      if (xbc && i == nx - 1) A2(Vy, nx, nx, j) = v;
      if (j == ny && a.top) {  // bc_yV!: the top row takes row ny as bc_x! left it
        if (own) A2(Vy, nx, i, ny + 1) = v;
        if (xbc && i == 2) A2(Vy, nx, 1, ny + 1) = v;
        if (xbc && i == nx - 1) A2(Vy, nx, nx, ny + 1) = v;
      }
    }
    // (the domain's row ny+1 itself: written by the thread of row ny; bc_x! on row ny+1, when it reaches that
    //  far, acts before bc_yV! overwrites the whole row, so it leaves no trace)
  }
}

// ---- advect -------------------------------------------------------------------------------------
__device__ __forceinline__ double lerp_ref(double p, double q, double t) { return q * t + p * (1 - t); }  // :204
__device__ __forceinline__ int clamp_floor(double f, int lo, int hi) {
  if (!(f >= (double)lo)) return lo;
  if (f >= (double)hi) return hi;
  return (int)f;
}
// backtrack!  NavierStokes.jl:192-202.  (ix, gy) are global indices; the source window holds global rows
// iy0+1 .. iy0+n2w of an array with n2g rows (the extra clamp to the window only acts when the window's
// halo is too narrow for the back-trace, which the caller excludes).
__device__ __forceinline__ double backtrack(const double *__restrict__ A_o, int n1, int n2w, int n2g, int iy0, double vxc,
                                            double vyc, double dt, double dx, double dy, int ix, int gy) {
  double ddx = dt * vxc / dx, ddy = dt * vyc / dy;
  const int ix1 = clamp_floor(floor((double)ix - ddx), 1, n1);
  int iy1 = clamp_floor(floor((double)gy - ddy), 1, n2g);
  const int ix2 = min(ix1 + 1, n1);
  int iy2 = min(iy1 + 1, n2g);
  iy1 = min(max(iy1 - iy0, 1), n2w);
  iy2 = min(max(iy2 - iy0, 1), n2w);
  ddx = (double)(ddx > 0) - fmod(ddx, 1.0);
  ddy = (double)(ddy > 0) - fmod(ddy, 1.0);
  const double fx1 = lerp_ref(A2(A_o, n1, ix1, iy1), A2(A_o, n1, ix2, iy1), ddx);
  const double fx2 = lerp_ref(A2(A_o, n1, ix1, iy2), A2(A_o, n1, ix2, iy2), ddx);
  return lerp_ref(fx1, fx2, ddy);
}

// advect!  NavierStokes.jl:206-223; points it leaves alone (first and last column of Vx, first and
// last row of Vy -- of the window: a window's end rows are halo rows or the domain's) keep the value they
// have in the source arrays.
__global__ void __launch_bounds__(256) k_advect_out(double *__restrict__ Vx, const double *__restrict__ Vx_o,
                                                    double *__restrict__ Vy, const double *__restrict__ Vy_o,
                                                    double *__restrict__ C, const double *__restrict__ C_o,
                                                    StepArgs a) {
  const int nx = a.nx, ny = a.ny;
  const int ix = blockIdx.x * 64 + (threadIdx.x & 63) + 1, iy = blockIdx.y * 4 + (threadIdx.x >> 6) + 1;
  if (ix > nx + 1 || iy > ny + 1) return;
  const int gy = iy + a.iy0;
  if (iy <= ny) {
    double v = A2(Vx_o, nx + 1, ix, iy);
    if (ix > 1 && ix < nx + 1) {
      const double vyc = 0.25 * (A2(Vy_o, nx, ix - 1, iy) + A2(Vy_o, nx, ix - 1, iy + 1) + A2(Vy_o, nx, ix, iy) +
                                 A2(Vy_o, nx, ix, iy + 1));
      v = backtrack(Vx_o, nx + 1, ny, a.nyg, a.iy0, v, vyc, a.dt, a.dx, a.dy, ix, gy);
    }
    A2(Vx, nx + 1, ix, iy) = v;
  }
  if (ix <= nx) {
    double v = A2(Vy_o, nx, ix, iy);
    if (iy > 1 && iy < ny + 1) {
      const double vxc = 0.25 * (A2(Vx_o, nx + 1, ix, iy - 1) + A2(Vx_o, nx + 1, ix + 1, iy - 1) +
                                 A2(Vx_o, nx + 1, ix, iy) + A2(Vx_o, nx + 1, ix + 1, iy));
      v = backtrack(Vy_o, nx, ny + 1, a.nyg + 1, a.iy0, vxc, v, a.dt, a.dx, a.dy, ix, gy);
    }
    A2(Vy, nx, ix, iy) = v;
  }
  if (ix <= nx && iy <= ny) {
    const double vxc = 0.5 * (A2(Vx_o, nx + 1, ix, iy) + A2(Vx_o, nx + 1, ix + 1, iy));
    const double vyc = 0.5 * (A2(Vy_o, nx, ix, iy) + A2(Vy_o, nx, ix, iy + 1));
    A2(C, nx, ix, iy) = backtrack(C_o, nx, ny, a.nyg, a.iy0, vxc, vyc, a.dt, a.dx, a.dy, ix, gy);
  }
}

StepArgs step_args(const nsk_ns_params *p) {
  StepArgs a;
  a.nx = p->nx;
  a.ny = p->ny;
  a.ld = p->ld_pr ? p->ld_pr : p->nx;
  a.iy0 = p->iy0;
  a.nyg = p->nyg;
  a.bottom = p->iy0 == 0;
  a.top = p->iy0 + p->ny == p->nyg;
  const int rows_vy = p->nyg + 1;
  const int xrows_global = p->full_rows ? rows_vy : (p->nx < rows_vy ? p->nx : rows_vy);
  a.xrows = xrows_global - p->iy0 < 0 ? 0 : (xrows_global - p->iy0 > p->ny + 1 ? p->ny + 1 : xrows_global - p->iy0);
  a.mu = p->mu;
  a.dtr = p->dt / p->rho;
  a.rg = p->rho * p->g;
  a.dx = p->dx;
  a.dy = p->dy;
  a.a2 = p->a2;
  a.b2 = p->b2;
  a.ox = p->ox;
  a.oy = p->oy;
  a.sinb = p->sinb;
  a.cosb = p->cosb;
  a.lx = p->lx;
  a.ly = p->ly;
  a.dt = p->dt;
  return a;
}

}  // namespace

// NavierStokes.jl:79-82: (Vx, Vy, C) -> (Vx_o, Vy_o, C_o) and ∇V.
int nsk_i_step_predictor(nsk_ctx *ctx, const nsk_ns_state *s, const nsk_ns_params *p) {
  const StepArgs a = step_args(p);
  dim3 grid(nsk_cdiv(p->nx + 1, TX), nsk_cdiv(p->ny + 1, TY));
  k_predictor<<<grid, 256, 0, ctx->stream>>>(s->Vx, s->Vy, s->C, s->Vx_o, s->Vy_o, s->C_o, s->divV, a);
  NSK_LAUNCH_CHECK();
  return 0;
}

// NavierStokes.jl:96-98 in place on (Vx_o, Vy_o, C_o).
int nsk_i_step_corrector(nsk_ctx *ctx, const nsk_ns_state *s, const nsk_ns_params *p) {
  const StepArgs a = step_args(p);
  dim3 grid(nsk_cdiv(p->nx + 1, 64), nsk_cdiv(p->ny + 1, 4));
  k_corrector<<<grid, 256, 0, ctx->stream>>>(s->Vx_o, s->Vy_o, s->C_o, s->Pr, s->Vprof, a);
  NSK_LAUNCH_CHECK();
  return 0;
}

// NavierStokes.jl:100 from (Vx_o, Vy_o, C_o) back into (Vx, Vy, C); :99 has nothing left to do.
int nsk_i_step_advect_out(nsk_ctx *ctx, const nsk_ns_state *s, const nsk_ns_params *p) {
  const StepArgs a = step_args(p);
  dim3 grid(nsk_cdiv(p->nx + 1, 64), nsk_cdiv(p->ny + 1, 4));
  k_advect_out<<<grid, 256, 0, ctx->stream>>>(s->Vx, s->Vx_o, s->Vy, s->Vy_o, s->C, s->C_o, a);
  NSK_LAUNCH_CHECK();
  return 0;
}

int nsk_i_step_corrector_advect(nsk_ctx *ctx, const nsk_ns_state *s, const nsk_ns_params *p) {
  if (int rc = nsk_i_step_corrector(ctx, s, p)) return rc;
  return nsk_i_step_advect_out(ctx, s, p);
}
