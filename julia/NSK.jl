# Julia binding of libnsk (include/nsk.h).  NOT RUN in this repository's environment (no Julia
# in the build image); it shows the `ccall` stubs a maintainer of the reference would add so
# that scripts/NavierStokes*.jl call the MI355X library instead of ParallelStencil kernels.
module NSK

const lib = get(ENV, "LIBNSK", joinpath(@__DIR__, "..", "navierstokes_amd", "libnsk.so"))

struct Ctx
    h::Ptr{Cvoid}
end

"A nonzero return of the library; code is one of the NSK_ERR_* values of include/nsk.h."
struct NskError <: Exception
    code::Cint
    msg::String
end
const ERR_FRONT_CAPACITY = Cint(3)
const ERR_FRONT_OUTSIDE = Cint(4)
check(rc) = rc == 0 || throw(NskError(rc, unsafe_string(ccall((:nsk_last_error, lib), Cstring, ()))))

function Ctx(device::Integer=0)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:nsk_create, lib), Cint, (Cint, Ptr{Ptr{Cvoid}}), device, r))
    Ctx(r[])
end
destroy(c::Ctx) = ccall((:nsk_destroy, lib), Cint, (Ptr{Cvoid},), c.h)

"Device array handle: the pointer libnsk returned plus the Julia-side extents."
struct DArray
    p::Ptr{Cdouble}
    n1::Int
    n2::Int
end

function zeros(c::Ctx, n1, n2)                       # @zeros(n1,n2)
    r = Ref{Ptr{Cdouble}}(C_NULL)
    check(ccall((:nsk_zeros, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Ptr{Cdouble}}), c.h, n1, n2, r))
    DArray(r[], n1, n2)
end
upload!(c::Ctx, d::DArray, a::Array{Float64}) =      # Data.Array(a)
    check(ccall((:nsk_upload, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Csize_t), c.h, d.p, a, length(a)))
function download(c::Ctx, d::DArray)                 # Array(A)
    a = Array{Float64}(undef, d.n1, d.n2)
    check(ccall((:nsk_download, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Csize_t), c.h, a, d.p, length(a)))
    a
end

# struct nsk_ns_params / nsk_ns_state, field for field
struct Params
    nx::Cint; ny::Cint; iy0::Cint; nyg::Cint; niter::Cint; nchk::Cint; full_rows::Cint
    ld_pr::Cint; ld_dp::Cint; dp_embedded::Cint
    lx::Cdouble; ly::Cdouble; rho::Cdouble; vin::Cdouble; psc::Cdouble; mu::Cdouble; g::Cdouble
    a2::Cdouble; b2::Cdouble; ox::Cdouble; oy::Cdouble; sinb::Cdouble; cosb::Cdouble
    eps_it::Cdouble; dx::Cdouble; dy::Cdouble; dt::Cdouble; damp::Cdouble; dtau::Cdouble
end
struct State
    Pr::Ptr{Cdouble}; dPrdtau::Ptr{Cdouble}; C::Ptr{Cdouble}; C_o::Ptr{Cdouble}
    txx::Ptr{Cdouble}; tyy::Ptr{Cdouble}; txy::Ptr{Cdouble}; Vx::Ptr{Cdouble}; Vy::Ptr{Cdouble}
    Vx_o::Ptr{Cdouble}; Vy_o::Ptr{Cdouble}; divV::Ptr{Cdouble}; Rp::Ptr{Cdouble}; Vprof::Ptr{Cdouble}
end

# one stub per @parallel call site (argument order of the Julia kernels) ------------------
update_τ!(c, τxx, τyy, τxy, Vx, Vy, μ, dx, dy) =
    check(ccall((:nsk_update_tau, lib), Cint,
        (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Cdouble, Cdouble, Cint, Cint),
        c.h, τxx.p, τyy.p, τxy.p, Vx.p, Vy.p, μ, dx, dy, τxx.n1, τxx.n2))
update_dPrdτ!(c, Pr, dPrdτ, ∇V, ρ, dt, dτ, damp, dx, dy) =
    check(ccall((:nsk_update_dPrdtau, lib), Cint,
        (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cint, Cint),
        c.h, Pr.p, dPrdτ.p, ∇V.p, ρ, dt, dτ, damp, dx, dy, Pr.n1, Pr.n2))
update_Pr!(c, Pr, dPrdτ, dτ) =
    check(ccall((:nsk_update_Pr, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Cint, Cint),
        c.h, Pr.p, dPrdτ.p, dτ, Pr.n1, Pr.n2))
set_bc_Pr!(c, Pr, val) =
    check(ccall((:nsk_set_bc_Pr, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cdouble, Cint, Cint, Cint, Cint, Cint),
        c.h, Pr.p, val, Pr.n1, Pr.n2, 0, 0, Pr.n2))
function maxabs(c, A)                                 # maximum(abs.(A))
    r = Ref{Cdouble}(0.0)
    check(ccall((:nsk_maxabs, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Csize_t, Ptr{Cdouble}), c.h, A.p, A.n1 * A.n2, r))
    r[]
end
const PD = Ptr{Cdouble}
predict_V!(c, Vx, Vy, τxx, τyy, τxy, ρ, g, dt, dx, dy) =
    check(ccall((:nsk_predict_V, lib), Cint,
        (Ptr{Cvoid}, PD, PD, PD, PD, PD, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cint, Cint),
        c.h, Vx.p, Vy.p, τxx.p, τyy.p, τxy.p, ρ, g, dt, dx, dy, τxx.n1, τxx.n2))
set_sphere!(c, C, Vx, Vy, a2, b2, ox, oy, sinβ, cosβ, lx, ly, dx, dy) =
    check(ccall((:nsk_set_sphere, lib), Cint,
        (Ptr{Cvoid}, PD, PD, PD, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble,
         Cint, Cint, Cint),
        c.h, C.p, Vx.p, Vy.p, a2, b2, ox, oy, sinβ, cosβ, lx, ly, dx, dy, C.n1, C.n2, 0))
update_∇V!(c, ∇V, Vx, Vy, dx, dy) =
    check(ccall((:nsk_update_divV, lib), Cint, (Ptr{Cvoid}, PD, PD, PD, Cdouble, Cdouble, Cint, Cint),
        c.h, ∇V.p, Vx.p, Vy.p, dx, dy, ∇V.n1, ∇V.n2))
compute_res!(c, Rp, Pr, ∇V, ρ, dt, dx, dy) =
    check(ccall((:nsk_compute_res, lib), Cint, (Ptr{Cvoid}, PD, PD, PD, Cdouble, Cdouble, Cdouble, Cdouble, Cint, Cint),
        c.h, Rp.p, Pr.p, ∇V.p, ρ, dt, dx, dy, Pr.n1, Pr.n2))
correct_V!(c, Vx, Vy, Pr, dt, ρ, dx, dy) =
    check(ccall((:nsk_correct_V, lib), Cint, (Ptr{Cvoid}, PD, PD, PD, Cdouble, Cdouble, Cdouble, Cdouble, Cint, Cint),
        c.h, Vx.p, Vy.p, Pr.p, dt, ρ, dx, dy, Pr.n1, Pr.n2))
set_bc_Vel!(c, Vx, Vy, Vprof) =                       # full_rows = 0: the reference's bc_x! row range
    check(ccall((:nsk_set_bc_Vel, lib), Cint, (Ptr{Cvoid}, PD, PD, PD, Cint, Cint, Cint, Cint, Cint),
        c.h, Vx.p, Vy.p, Vprof.p, Vy.n1, Vx.n2, 0, 0, Vx.n2))
advect!(c, Vx, Vx_o, Vy, Vy_o, C, C_o, dt, dx, dy) =
    check(ccall((:nsk_advect, lib), Cint,
        (Ptr{Cvoid}, PD, PD, PD, PD, PD, PD, Cdouble, Cdouble, Cdouble, Cint, Cint, Cint, Cint),
        c.h, Vx.p, Vx_o.p, Vy.p, Vy_o.p, C.p, C_o.p, dt, dx, dy, C.n1, C.n2, 0, C.n2))
copy!(c, dst::DArray, src::DArray) =                  # A_o .= A
    check(ccall((:nsk_copy, lib), Cint, (Ptr{Cvoid}, PD, PD, Csize_t), c.h, dst.p, src.p, src.n1 * src.n2))
free!(c, d::DArray) = check(ccall((:nsk_free, lib), Cint, (Ptr{Cvoid}, PD), c.h, d.p))
sync(c) = check(ccall((:nsk_sync, lib), Cint, (Ptr{Cvoid},), c.h))

"Whole body of the time loop (the fused entry point); returns (iterations, [(iter, err)...])."
function step!(c::Ctx, s::State, p::Params; cap=1024)
    iters = Ref{Cint}(0); nh = Ref{Cint}(0)
    errs = Vector{Cdouble}(undef, cap); its = Vector{Cint}(undef, cap)
    check(ccall((:nsk_ns_step, lib), Cint,
        (Ptr{Cvoid}, Ref{State}, Ref{Params}, Ptr{Cint}, Ptr{Cdouble}, Ptr{Cint}, Cint, Ptr{Cint}),
        c.h, s, p, iters, errs, its, cap, nh))
    n = min(nh[], cap)
    iters[], collect(zip(its[1:n], errs[1:n]))
end

# ---- fused Poisson entry points, options, streams ------------------------------------------
"`count` passes of NavierStokes.jl:85-87 without a residual check."
poisson_iters!(c, Pr, dPrdτ, ∇V, p::Params, count) =
    check(ccall((:nsk_ns_poisson_iters, lib), Cint, (Ptr{Cvoid}, PD, PD, PD, Ref{Params}, Cint),
        c.h, Pr.p, dPrdτ.p, ∇V.p, p, count))
"The same for callers that must not allocate (windows of at most 4 Mi values per field)."
poisson_iters_static!(c, Pr, dPrdτ, ∇V, p::Params, count) =
    check(ccall((:nsk_ns_poisson_iters_static, lib), Cint, (Ptr{Cvoid}, PD, PD, PD, Ref{Params}, Cint),
        c.h, Pr.p, dPrdτ.p, ∇V.p, p, count))
"""The same for a slab window whose edge rows a neighbour waits for: in the last launch the rows
`skip_lo+1:skip_lo+edge_lo` and their mirror at the other end are produced first, on `edge_stream`
(a hipStream_t), the first `skip_lo` / last `skip_hi` rows not at all."""
poisson_iters_edges_first!(c, Pr, dPrdτ, ∇V, p::Params, count, skip_lo, edge_lo, edge_hi, skip_hi, edge_stream::Ptr{Cvoid}) =
    check(ccall((:nsk_ns_poisson_iters_edges_first, lib), Cint,
        (Ptr{Cvoid}, PD, PD, PD, Ref{Params}, Cint, Cint, Cint, Cint, Cint, Ptr{Cvoid}),
        c.h, Pr.p, dPrdτ.p, ∇V.p, p, count, skip_lo, edge_lo, edge_hi, skip_hi, edge_stream))
function residual(c, Rp, Pr, ∇V, p::Params, own_lo, own_hi)     # compute_res! + maximum(abs.(Rp))
    r = Ref{Cdouble}(0.0)
    check(ccall((:nsk_ns_residual, lib), Cint, (Ptr{Cvoid}, PD, PD, PD, Ref{Params}, Cint, Cint, Ptr{Cdouble}),
        c.h, Rp.p, Pr.p, ∇V.p, p, own_lo, own_hi, r))
    r[]
end
"The Poisson loop NavierStokes.jl:84-95; returns (iterations, [(iter, err)...])."
function poisson!(c, Pr, dPrdτ, Rp, ∇V, p::Params; cap=1024)
    iters = Ref{Cint}(0); nh = Ref{Cint}(0)
    errs = Vector{Cdouble}(undef, cap); its = Vector{Cint}(undef, cap)
    check(ccall((:nsk_ns_poisson, lib), Cint,
        (Ptr{Cvoid}, PD, PD, PD, PD, Ref{Params}, Ptr{Cint}, Ptr{Cdouble}, Ptr{Cint}, Cint, Ptr{Cint}),
        c.h, Pr.p, dPrdτ.p, Rp.p, ∇V.p, p, iters, errs, its, cap, nh))
    n = min(nh[], cap)
    iters[], collect(zip(its[1:n], errs[1:n]))
end
update_∇V_ld!(c, ∇V, ld, Vx, Vy, dx, dy, nx, ny) =
    check(ccall((:nsk_update_divV_ld, lib), Cint, (Ptr{Cvoid}, PD, Cint, PD, PD, Cdouble, Cdouble, Cint, Cint),
        c.h, ∇V.p, ld, Vx.p, Vy.p, dx, dy, nx, ny))
correct_V_ld!(c, Vx, Vy, Pr, ld, dt, ρ, dx, dy, nx, ny) =
    check(ccall((:nsk_correct_V_ld, lib), Cint, (Ptr{Cvoid}, PD, PD, PD, Cint, Cdouble, Cdouble, Cdouble, Cdouble, Cint, Cint),
        c.h, Vx.p, Vy.p, Pr.p, ld, dt, ρ, dx, dy, nx, ny))
set_option!(c, name::String, v) =
    check(ccall((:nsk_set_option, lib), Cint, (Ptr{Cvoid}, Cstring, Cint), c.h, name, v))
function get_option(c, name::String)
    r = Ref{Cint}(0)
    check(ccall((:nsk_get_option, lib), Cint, (Ptr{Cvoid}, Cstring, Ptr{Cint}), c.h, name, r))
    r[]
end
set_stream!(c, hip_stream::Ptr{Cvoid}) = check(ccall((:nsk_set_stream, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), c.h, hip_stream))
use_own_stream!(c) = check(ccall((:nsk_use_own_stream, lib), Cint, (Ptr{Cvoid},), c.h))
version() = unsafe_string(ccall((:nsk_version, lib), Cstring, ()))
function poisson_plan(nx, ny; num_simds=1024, staged=1, k=0, k_limit=8, rows=0, edge_percent=0)
    out = Vector{Cint}(undef, 6)
    check(ccall((:nsk_poisson_plan, lib), Cint, (Cint, Cint, Cint, Cint, Cint, Cint, Cint, Cint, Ptr{Cint}),
        nx, ny, num_simds, staged, k, k_limit, rows, edge_percent, out))
    out
end
function division_form(d)
    f = Ref{Cint}(0); rh = Ref{Cdouble}(0.0); rl = Ref{Cdouble}(0.0)
    check(ccall((:nsk_division_form, lib), Cint, (Cdouble, Ptr{Cint}, Ptr{Cdouble}, Ptr{Cdouble}), d, f, rh, rl))
    f[], rh[], rl[]
end

# ---- slab decomposition over several GPUs (one Julia process drives all of them) -------------
struct Slabs
    h::Ptr{Cvoid}
    nlocal::Int
end
function rccl_unique_id()
    id = Vector{UInt8}(undef, 128)
    check(ccall((:nsk_rccl_unique_id, lib), Cint, (Ptr{Cvoid},), id))
    id
end
"All `world` slabs in this process on `devices` (transport 0: device copies; 1: RCCL)."
function Slabs(devices::Vector{<:Integer}, p::Params, ny_loc, halo; world=length(devices), first_rank=0, transport=0,
               unique_id=C_NULL)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    dev = Vector{Cint}(devices)
    check(ccall((:nsk_slabs_create, lib), Cint,
        (Cint, Cint, Cint, Ptr{Cint}, Ref{Params}, Cint, Cint, Cint, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}),
        world, first_rank, length(dev), dev, p, ny_loc, halo, transport, unique_id, r))
    Slabs(r[], length(dev))
end
destroy(s::Slabs) = ccall((:nsk_slabs_destroy, lib), Cint, (Ptr{Cvoid},), s.h)
function window(s::Slabs, local_)
    v = [Ref{Cint}(0) for _ = 1:7]
    check(ccall((:nsk_slabs_window, lib), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Cint}, Ptr{Cint}, Ptr{Cint}, Ptr{Cint}, Ptr{Cint}, Ptr{Cint}, Ptr{Cint}),
        s.h, local_, v[1], v[2], v[3], v[4], v[5], v[6], v[7]))
    (; rank=v[1][], device=v[2][], iy0=v[3][], ny_win=v[4][], own_lo=v[5][], own_hi=v[6][], ld=v[7][])
end
function fields(s::Slabs, local_)
    a = Ref{Ptr{Cdouble}}(C_NULL); b = Ref{Ptr{Cdouble}}(C_NULL); d = Ref{Ptr{Cdouble}}(C_NULL)
    check(ccall((:nsk_slabs_fields, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Ptr{Cdouble}}, Ptr{Ptr{Cdouble}}, Ptr{Ptr{Cdouble}}),
        s.h, local_, a, b, d))
    a[], b[], d[]
end
"field: 0 Pr (nx x nyg), 1 dPrdτ ((nx-2) x (nyg-2)), 2 ∇V -- global host arrays"
upload!(s::Slabs, field, a::Array{Float64}) =
    check(ccall((:nsk_slabs_upload, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}), s.h, field, a))
download!(s::Slabs, field, a::Array{Float64}) =
    check(ccall((:nsk_slabs_download, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}), s.h, field, a))
exchange!(s::Slabs, mask=3) = check(ccall((:nsk_slabs_exchange, lib), Cint, (Ptr{Cvoid}, Cint), s.h, mask))
poisson_iters!(s::Slabs, count) = check(ccall((:nsk_slabs_poisson_iters, lib), Cint, (Ptr{Cvoid}, Cint), s.h, count))
function residual(s::Slabs)
    r = Ref{Cdouble}(0.0)
    check(ccall((:nsk_slabs_residual, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), s.h, r))
    r[]
end
function poisson!(s::Slabs; cap=1024)
    iters = Ref{Cint}(0); nh = Ref{Cint}(0)
    errs = Vector{Cdouble}(undef, cap); its = Vector{Cint}(undef, cap)
    check(ccall((:nsk_slabs_poisson, lib), Cint, (Ptr{Cvoid}, Ptr{Cint}, Ptr{Cdouble}, Ptr{Cint}, Cint, Ptr{Cint}),
        s.h, iters, errs, its, cap, nh))
    n = min(nh[], cap)
    iters[], collect(zip(its[1:n], errs[1:n]))
end
"Velocity / tracer windows of runme() on every slab (Vy[:,1] = Vprof on the bottom one)."
init_flow!(s::Slabs, Vprof::Vector{Float64}) =
    check(ccall((:nsk_slabs_init_flow, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), s.h, Vprof))
"One time step (NavierStokes.jl:79-100) on the slabs; returns (iterations, [(iter, err)...])."
function step!(s::Slabs; cap=1024)
    iters = Ref{Cint}(0); nh = Ref{Cint}(0)
    errs = Vector{Cdouble}(undef, cap); its = Vector{Cint}(undef, cap)
    check(ccall((:nsk_slabs_step, lib), Cint, (Ptr{Cvoid}, Ptr{Cint}, Ptr{Cdouble}, Ptr{Cint}, Cint, Ptr{Cint}),
        s.h, iters, errs, its, cap, nh))
    n = min(nh[], cap)
    iters[], collect(zip(its[1:n], errs[1:n]))
end
sync(s::Slabs) = check(ccall((:nsk_slabs_sync, lib), Cint, (Ptr{Cvoid},), s.h))
rccl_self_exchange!(s::Slabs, rows) =
    check(ccall((:nsk_slabs_rccl_self_exchange, lib), Cint, (Ptr{Cvoid}, Cint), s.h, rows))
"Switches without effect on results, e.g. \"edges_first\" (default 1)."
set_option!(s::Slabs, name::String, v::Integer) =
    check(ccall((:nsk_slabs_set_option, lib), Cint, (Ptr{Cvoid}, Cstring, Cint), s.h, name, v))
function get_option(s::Slabs, name::String)
    v = Ref{Cint}(0)
    check(ccall((:nsk_slabs_get_option, lib), Cint, (Ptr{Cvoid}, Cstring, Ptr{Cint}), s.h, name, v))
    Int(v[])
end
function info(s::Slabs)
    v = [Ref{Cint}(0) for _ = 1:4]
    check(ccall((:nsk_slabs_info, lib), Cint, (Ptr{Cvoid}, Ptr{Cint}, Ptr{Cint}, Ptr{Cint}, Ptr{Cint}), s.h, v[1], v[2], v[3], v[4]))
    (; world=v[1][], nlocal=v[2][], transport=v[3][], exchanges=v[4][])
end

# struct nsk_bub_params / nsk_bub_dns_state / nsk_bub_code3_state, field for field
struct BubParams
    nx::Cint; ny::Cint; maxit::Cint
    Lx::Cdouble; Ly::Cdouble; gx::Cdouble; gy::Cdouble; rho1::Cdouble; rho2::Cdouble; m1::Cdouble; m2::Cdouble
    sigma::Cdouble; rro::Cdouble; usouth::Cdouble; unorth::Cdouble; vwest::Cdouble; veast::Cdouble
    dt::Cdouble; dx::Cdouble; dy::Cdouble; maxError::Cdouble; beta::Cdouble
end
struct DnsState
    u::Ptr{Cdouble}; v::Ptr{Cdouble}; p::Ptr{Cdouble}; ut::Ptr{Cdouble}; vt::Ptr{Cdouble}; tmp1::Ptr{Cdouble}
    tmp2::Ptr{Cdouble}; r::Ptr{Cdouble}; ro::Ptr{Cdouble}; rt::Ptr{Cdouble}; uu::Ptr{Cdouble}; vv::Ptr{Cdouble}
end
struct Code3State
    u::Ptr{Cdouble}; v::Ptr{Cdouble}; p::Ptr{Cdouble}; ut::Ptr{Cdouble}; vt::Ptr{Cdouble}; tmp1::Ptr{Cdouble}
    tmp2::Ptr{Cdouble}; fx::Ptr{Cdouble}; fy::Ptr{Cdouble}; un::Ptr{Cdouble}; vn::Ptr{Cdouble}; r::Ptr{Cdouble}
    m::Ptr{Cdouble}; rn::Ptr{Cdouble}; mn::Ptr{Cdouble}; rt::Ptr{Cdouble}
    xf::Ptr{Cdouble}; yf::Ptr{Cdouble}; xfn::Ptr{Cdouble}; yfn::Ptr{Cdouble}; uf::Ptr{Cdouble}; vf::Ptr{Cdouble}
    tx::Ptr{Cdouble}; ty::Ptr{Cdouble}; xs::Ptr{Cdouble}; ys::Ptr{Cdouble}
    Nf::Cint; cap::Cint
end
"One time step of DNS.jl:71-101 in one call; returns the sweeps of the pressure solve."
function dns_step!(c, s::DnsState, q::BubParams)
    sw = Ref{Cint}(0)
    check(ccall((:nsk_bub_dns_step, lib), Cint, (Ptr{Cvoid}, Ref{DnsState}, Ref{BubParams}, Ptr{Cint}), c.h, s, q, sw))
    sw[]
end
"One time step of code3_fun.jl:356-494 in one call; returns the state with the new front size and the four sweep counts."
function code3_step!(c, s::Code3State, q::BubParams)
    r = Ref(s)
    sw = zeros(Cint, 4)
    check(ccall((:nsk_bub_code3_step, lib), Cint, (Ptr{Cvoid}, Ref{Code3State}, Ref{BubParams}, Ptr{Cint}), c.h, r, q, sw))
    r[], sw
end
"nsteps time steps of DNS.jl:70-102 queued in one call, one wait at its end; returns the sweeps of every step."
function dns_run!(c, s::DnsState, q::BubParams, nsteps::Integer)
    sw = zeros(Cint, nsteps)
    check(ccall((:nsk_bub_dns_run, lib), Cint, (Ptr{Cvoid}, Ref{DnsState}, Ref{BubParams}, Cint, Ptr{Cint}), c.h, s, q, nsteps, sw))
    sw
end
"nsteps time steps of code3_fun.jl:354-494 in one call; returns the state with the new front size and a 4 x nsteps matrix of sweep counts."
function code3_run!(c, s::Code3State, q::BubParams, nsteps::Integer)
    r = Ref(s)
    sw = zeros(Cint, 4, nsteps)
    check(ccall((:nsk_bub_code3_run, lib), Cint, (Ptr{Cvoid}, Ref{Code3State}, Ref{BubParams}, Cint, Ptr{Cint}), c.h, r, q, nsteps, sw))
    r[], sw
end
"Complete time steps of the last code3_step! / code3_run! call on this context (after ERR_FRONT_CAPACITY: the next one lacks only its regridding)."
function code3_run_status(c)
    n = Ref{Cint}(0)
    check(ccall((:nsk_bub_code3_run_status, lib), Cint, (Ptr{Cvoid}, Ptr{Cint}), c.h, n))
    Int(n[])
end

# ---- bubble / drop solvers: scripts/DNS.jl (in the form scripts/code1.m states) and ----------
# ---- scripts/code3_fun.jl; one stub per kernel, argument order of the Julia kernels ----------
vec(c::Ctx, n) = zeros(c, n, 1)
"set_density! DNS.jl:128-137; with m also set_density_viscosity! code3_fun.jl:8-18 (m = nothing: density only)"
set_density!(c, r, m, xc, yc, x, y, rad, ρ₂, m₂, nx, ny) =
    check(ccall((:nsk_bub_set_density, lib), Cint,
        (Ptr{Cvoid}, PD, PD, Cdouble, Cdouble, PD, PD, Cdouble, Cdouble, Cdouble, Cint, Cint),
        c.h, r.p, m === nothing ? Ptr{Cdouble}(C_NULL) : m.p, xc, yc, x.p, y.p, rad, ρ₂, m₂, nx, ny))
set_tangential!(c, u, v, usouth, unorth, vwest, veast, nx, ny) =
    check(ccall((:nsk_bub_set_tangential, lib), Cint, (Ptr{Cvoid}, PD, PD, Cdouble, Cdouble, Cdouble, Cdouble, Cint, Cint),
        c.h, u.p, v.p, usouth, unorth, vwest, veast, nx, ny))
"temp_uvel! + temp_vvel! DNS.jl:159-179"
temp_vel!(c, ut, vt, u, v, r, dt, dx, dy, gx, gy, m0, nx, ny) =
    check(ccall((:nsk_bub_temp_vel, lib), Cint,
        (Ptr{Cvoid}, PD, PD, PD, PD, PD, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cint, Cint),
        c.h, ut.p, vt.p, u.p, v.p, r.p, dt, dx, dy, gx, gy, m0, nx, ny))
"temp_uvel_adv! + temp_vvel_adv! code3_fun.jl:37-62"
temp_vel_adv!(c, ut, vt, u, v, r, dt, dx, dy, fx, fy, gx, gy, rro, nx, ny) =
    check(ccall((:nsk_bub_temp_vel_adv, lib), Cint,
        (Ptr{Cvoid}, PD, PD, PD, PD, PD, Cdouble, Cdouble, Cdouble, PD, PD, Cdouble, Cdouble, Cdouble, Cint, Cint),
        c.h, ut.p, vt.p, u.p, v.p, r.p, dt, dx, dy, fx.p, fy.p, gx, gy, rro, nx, ny))
"temp_uvel_diff! + temp_vvel_diff! code3_fun.jl:64-90"
temp_vel_diff!(c, ut, vt, u, v, r, dt, dx, dy, m, nx, ny) =
    check(ccall((:nsk_bub_temp_vel_diff, lib), Cint,
        (Ptr{Cvoid}, PD, PD, PD, PD, PD, Cdouble, Cdouble, Cdouble, PD, Cint, Cint),
        c.h, ut.p, vt.p, u.p, v.p, r.p, dt, dx, dy, m.p, nx, ny))
set_boundary!(c, rt, r, lrg, nx, ny) =
    check(ccall((:nsk_bub_set_boundary, lib), Cint, (Ptr{Cvoid}, PD, PD, Cdouble, Cint, Cint), c.h, rt.p, r.p, lrg, nx, ny))
compute_sources!(c, tmp1, tmp2, rt, ut, vt, dt, dx, dy, nx, ny) =
    check(ccall((:nsk_bub_compute_sources, lib), Cint,
        (Ptr{Cvoid}, PD, PD, PD, PD, PD, Cdouble, Cdouble, Cdouble, Cint, Cint),
        c.h, tmp1.p, tmp2.p, rt.p, ut.p, vt.p, dt, dx, dy, nx, ny))
"The whole pressure loop (DNS.jl:83-93 / code3_fun.jl:104-124); returns (sweeps, last max|old-new|)."
function solve_pressure!(c, p, tmp1, tmp2, rt, β, dx, dy, nx, ny, maxit, maxError; solver=:sor)
    sw = Ref{Cint}(0); ch = Ref{Cdouble}(0.0)
    if solver == :sor
        check(ccall((:nsk_bub_solve_pressure, lib), Cint,
            (Ptr{Cvoid}, PD, PD, PD, PD, Cdouble, Cdouble, Cdouble, Cint, Cint, Cint, Cdouble, Ptr{Cint}, Ptr{Cdouble}),
            c.h, p.p, tmp1.p, tmp2.p, rt.p, β, dx, dy, nx, ny, maxit, maxError, sw, ch))
    else    # red/black order, code3_perf_gpu.jl:129-183
        check(ccall((:nsk_bub_solve_pressure_rb, lib), Cint,
            (Ptr{Cvoid}, PD, PD, PD, PD, Cdouble, Cdouble, Cdouble, Cint, Cint, Cint, Cdouble, Ptr{Cint}, Ptr{Cdouble}),
            c.h, p.p, tmp1.p, tmp2.p, rt.p, β, dx, dy, nx, ny, maxit, maxError, sw, ch))
    end
    sw[], ch[]
end
"update_dPdτ! + update_p! with their loop, code1_perf_gpu.jl:98-118,228-242"
function pt_pressure!(c, p, dPdτ, tmp1, rt, dτ, damp, dx, dy, nx, ny, maxit, maxError)
    it = Ref{Cint}(0); ch = Ref{Cdouble}(0.0)
    check(ccall((:nsk_bub_pt_pressure, lib), Cint,
        (Ptr{Cvoid}, PD, PD, PD, PD, Cdouble, Cdouble, Cdouble, Cdouble, Cint, Cint, Cint, Cdouble, Ptr{Cint}, Ptr{Cdouble}),
        c.h, p.p, dPdτ.p, tmp1.p, rt.p, dτ, damp, dx, dy, nx, ny, maxit, maxError, it, ch))
    it[], ch[]
end
"correct_uvel! + correct_vvel! DNS.jl:190-202"
correct_vel!(c, u, v, ut, vt, p, r, dt, dx, dy, nx, ny) =
    check(ccall((:nsk_bub_correct_vel, lib), Cint,
        (Ptr{Cvoid}, PD, PD, PD, PD, PD, PD, Cdouble, Cdouble, Cdouble, Cint, Cint),
        c.h, u.p, v.p, ut.p, vt.p, p.p, r.p, dt, dx, dy, nx, ny))
advect_density!(c, r, ro, u, v, dt, dx, dy, m0, nx, ny) =
    check(ccall((:nsk_bub_advect_density, lib), Cint,
        (Ptr{Cvoid}, PD, PD, PD, PD, Cdouble, Cdouble, Cdouble, Cdouble, Cint, Cint),
        c.h, r.p, ro.p, u.p, v.p, dt, dx, dy, m0, nx, ny))
velocity!(c, uu, vv, u, v, nx, ny) =
    check(ccall((:nsk_bub_velocity, lib), Cint, (Ptr{Cvoid}, PD, PD, PD, PD, Cint, Cint), c.h, uu.p, vv.p, u.p, v.p, nx, ny))
"tangents + spreading of the surface tension, code3_fun.jl:365-401"
surface_tension!(c, fx, fy, tx, ty, xf, yf, σ, dx, dy, nx, ny, Nf) =
    check(ccall((:nsk_bub_surface_tension, lib), Cint,
        (Ptr{Cvoid}, PD, PD, PD, PD, PD, PD, Cdouble, Cdouble, Cdouble, Cint, Cint, Cint),
        c.h, fx.p, fy.p, tx.p, ty.p, xf.p, yf.p, σ, dx, dy, nx, ny, Nf))
"advect_front! + move_front! code3_fun.jl:144-195"
advect_move_front!(c, uf, vf, xf, yf, u, v, dx, dy, dt, Lx, Ly, nx, ny, Nf) =
    check(ccall((:nsk_bub_advect_move_front, lib), Cint,
        (Ptr{Cvoid}, PD, PD, PD, PD, PD, PD, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cint, Cint, Cint, Ptr{Cint}),
        c.h, uf.p, vf.p, xf.p, yf.p, u.p, v.p, dx, dy, dt, Lx, Ly, nx, ny, Nf, Ptr{Cint}(C_NULL)))
"fx = fy = 0; distribute! code3_fun.jl:433-435,197-222"
distribute!(c, fx, fy, xf, yf, ρ₁, ρ₂, dx, dy, nx, ny, Nf) =
    check(ccall((:nsk_bub_distribute, lib), Cint,
        (Ptr{Cvoid}, PD, PD, PD, PD, Cdouble, Cdouble, Cdouble, Cdouble, Cint, Cint, Cint),
        c.h, fx.p, fy.p, xf.p, yf.p, ρ₁, ρ₂, dx, dy, nx, ny, Nf))
function construct_den!(c, r, fx, fy, β, maxit, maxError, dx, dy, nx, ny)
    sw = Ref{Cint}(0)
    check(ccall((:nsk_bub_construct_den, lib), Cint,
        (Ptr{Cvoid}, PD, PD, PD, Cdouble, Cint, Cdouble, Cdouble, Cdouble, Cint, Cint, Ptr{Cint}),
        c.h, r.p, fx.p, fy.p, β, maxit, maxError, dx, dy, nx, ny, sw))
    sw[]
end
update_visco!(c, r, m, m₁, m₂, ρ₁, ρ₂, nx, ny) =
    check(ccall((:nsk_bub_update_visco, lib), Cint, (Ptr{Cvoid}, PD, PD, Cdouble, Cdouble, Cdouble, Cdouble, Cint, Cint),
        c.h, r.p, m.p, m₁, m₂, ρ₁, ρ₂, nx, ny))
"a = 0.5*(a + b) on the first n values, code3_fun.jl:445-450"
average!(c, a, b, n) =
    check(ccall((:nsk_bub_average, lib), Cint, (Ptr{Cvoid}, PD, PD, Csize_t), c.h, a.p, b.p, n))
"Front regridding code3_fun.jl:455-494; returns the new Nf."
function regrid_front!(c, xf, yf, xs, ys, dx, dy, Nf, cap)
    r = Ref{Cint}(0)
    check(ccall((:nsk_bub_regrid_front, lib), Cint, (Ptr{Cvoid}, PD, PD, PD, PD, Cdouble, Cdouble, Cint, Cint, Ptr{Cint}),
        c.h, xf.p, yf.p, xs.p, ys.p, dx, dy, Nf, cap, r))
    r[]
end

end # module
