# Julia binding of libnsk (include/nsk.h).  NOT RUN in this repository's environment (no Julia
# in the build image); it shows the `ccall` stubs a maintainer of the reference would add so
# that scripts/NavierStokes*.jl call the MI355X library instead of ParallelStencil kernels.
module NSK

const lib = get(ENV, "LIBNSK", joinpath(@__DIR__, "..", "navierstokes_amd", "libnsk.so"))

struct Ctx
    h::Ptr{Cvoid}
end

check(rc) = rc == 0 || error(unsafe_string(ccall((:nsk_last_error, lib), Cstring, ())))

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

end # module
