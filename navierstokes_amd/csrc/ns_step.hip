// Whole time-step body of runme() (scripts/NavierStokes.jl:79-100) behind one call.
#include "nsk_internal.h"

int nsk_i_update_divV(nsk_ctx *ctx, double *divV, int ld, const double *Vx, const double *Vy,
                      double dx, double dy, int nx, int ny);
int nsk_i_correct_V(nsk_ctx *ctx, double *Vx, double *Vy, const double *Pr, int ld, double dt,
                    double rho, double dx, double dy, int nx, int ny);

int nsk_i_check_params(const nsk_ns_params *p);
int nsk_i_step_predictor(nsk_ctx *ctx, const nsk_ns_state *s, const nsk_ns_params *p);
int nsk_i_step_corrector_advect(nsk_ctx *ctx, const nsk_ns_state *s, const nsk_ns_params *p);

#define NSK_TRY(call)            \
  do {                           \
    if (int rc_ = (call)) return rc_; \
  } while (0)

extern "C" int nsk_ns_step(nsk_ctx *ctx, const nsk_ns_state *s, const nsk_ns_params *p,
                           int *iters_out, double *err_hist, int *iter_hist, int hist_cap,
                           int *nhist_out) {
  NSK_CTX(ctx);
  NSK_REQUIRE(s != nullptr && p != nullptr, "nsk_ns_step: null argument");
  NSK_NONNULL(s->Pr, s->dPrdtau, s->C, s->C_o, s->txx, s->tyy, s->txy, s->Vx, s->Vy, s->Vx_o, s->Vy_o, s->divV,
              s->Rp, s->Vprof);
  NSK_TRY(nsk_i_check_params(p));  // before anything modifies the state
  NSK_REQUIRE(p->iy0 == 0 && p->ny == p->nyg, "nsk_ns_step works on a whole domain");
  const int nx = p->nx, ny = p->ny;
  const int ld = p->ld_pr ? p->ld_pr : nx;
  if (ctx->step_fused) {
    // Three kernels around the Poisson loop (ns_step_fused.hip); the stress arrays are not
    // materialised (nothing outside update_τ! / predict_V! reads them).
    NSK_TRY(nsk_i_step_predictor(ctx, s, p));
    NSK_TRY(nsk_ns_poisson(ctx, s->Pr, s->dPrdtau, s->Rp, s->divV, p, iters_out, err_hist, iter_hist, hist_cap,
                           nhist_out));
    NSK_TRY(nsk_i_step_corrector_advect(ctx, s, p));
    NSK_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
  }
  // one launch per call site: NavierStokes.jl:79-82
  NSK_TRY(nsk_update_tau(ctx, s->txx, s->tyy, s->txy, s->Vx, s->Vy, p->mu, p->dx, p->dy, nx, ny));
  NSK_TRY(nsk_predict_V(ctx, s->Vx, s->Vy, s->txx, s->tyy, s->txy, p->rho, p->g, p->dt, p->dx, p->dy, nx, ny));
  NSK_TRY(nsk_set_sphere(ctx, s->C, s->Vx, s->Vy, p->a2, p->b2, p->ox, p->oy, p->sinb, p->cosb, p->lx,
                         p->ly, p->dx, p->dy, nx, ny, 0));
  NSK_TRY(nsk_i_update_divV(ctx, s->divV, ld, s->Vx, s->Vy, p->dx, p->dy, nx, ny));
  // :84-95
  NSK_TRY(nsk_ns_poisson(ctx, s->Pr, s->dPrdtau, s->Rp, s->divV, p, iters_out, err_hist, iter_hist,
                         hist_cap, nhist_out));
  // :96-100
  NSK_TRY(nsk_i_correct_V(ctx, s->Vx, s->Vy, s->Pr, ld, p->dt, p->rho, p->dx, p->dy, nx, ny));
  NSK_TRY(nsk_set_sphere(ctx, s->C, s->Vx, s->Vy, p->a2, p->b2, p->ox, p->oy, p->sinb, p->cosb, p->lx,
                         p->ly, p->dx, p->dy, nx, ny, 0));
  NSK_TRY(nsk_set_bc_Vel(ctx, s->Vx, s->Vy, s->Vprof, nx, ny, p->full_rows, 0, ny));
  NSK_TRY(nsk_copy(ctx, s->Vx_o, s->Vx, (size_t)(nx + 1) * ny));
  NSK_TRY(nsk_copy(ctx, s->Vy_o, s->Vy, (size_t)nx * (ny + 1)));
  NSK_TRY(nsk_copy(ctx, s->C_o, s->C, (size_t)nx * ny));
  NSK_TRY(nsk_advect(ctx, s->Vx, s->Vx_o, s->Vy, s->Vy_o, s->C, s->C_o, p->dt, p->dx, p->dy, nx, ny, 0, ny));
  NSK_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}
