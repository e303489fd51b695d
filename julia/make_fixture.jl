# Reference fixtures for the NavierStokes.jl path, produced by the REFERENCE's own kernels.
#
#   julia -t 4 julia/make_fixture.jl /path/to/Teslos-NavierStokes [outdir]
#
# Needs Julia with ParallelStencil.jl (the reference's dependency); nothing else.  The kernel
# definitions are not restated here: the script reads scripts/NavierStokes.jl of the checkout it is
# pointed at and evaluates that file's own text from `macro ∇V` up to the line before its trailing
# `runme(...)` call (scripts/NavierStokes.jl:115-250), on ParallelStencil's Threads backend in
# Float64.  It then
#   1. answers which rows `@parallel bc_x!(A)` touches (the kernel has ONE parallel index and is
#      launched without ranges, scripts/NavierStokes.jl:155-159,181,187),
#   2. runs every kernel once on exactly representable pseudo-data (kernel-level dumps), and
#   3. runs whole time steps of runme() (scripts/NavierStokes.jl:77-100) from its initial condition
#      on 21 x 37 cells (3 steps) and 153 x 255 cells (2 steps) -- the recipes of
#      tests/golden/make_ns_golden.py -- and dumps Pr, Vx, Vy, C, dPrdτ and the (iter, err) history.
# Output: one directory per recipe with `manifest.txt` (name, element type, extents) and one raw
# little-endian column-major file per array; tools/check_reference_fixture.py compares such a
# directory with the HIP library (and with the CPU oracle).
using ParallelStencil
using ParallelStencil.FiniteDifferences2D
@init_parallel_stencil(Threads, Float64, 2)
using Printf

const REF = length(ARGS) >= 1 ? ARGS[1] : error("usage: julia make_fixture.jl REFERENCE_CHECKOUT [outdir]")
const OUT = length(ARGS) >= 2 ? ARGS[2] : "reference_fixture"

let lines = readlines(joinpath(REF, "scripts", "NavierStokes.jl"))
    i0 = findfirst(l -> startswith(l, "macro ∇V"), lines)
    i1 = findlast(l -> startswith(strip(l), "runme("), lines)
    (i0 === nothing || i1 === nothing) && error("kernel section of scripts/NavierStokes.jl not found")
    include_string(Main, join(lines[i0:i1-1], "\n"), "scripts/NavierStokes.jl[$i0:$(i1-1)]")
end

# ---- output ------------------------------------------------------------------------------------
struct Dump
    dir::String
    manifest::Vector{String}
end
function Dump(dir)
    mkpath(dir)
    Dump(dir, String[])
end
function dump!(d::Dump, name, A::AbstractArray{T}) where {T<:Union{Float64,Int64}}
    a = Array(A)
    n1, n2 = size(a, 1), ndims(a) > 1 ? size(a, 2) : 1
    open(joinpath(d.dir, name * ".bin"), "w") do io
        write(io, htol.(vec(a)))
    end
    push!(d.manifest, @sprintf("%s %s %d %d", name, T == Float64 ? "f64" : "i64", n1, n2))
end
dump!(d::Dump, name, x::Float64) = dump!(d, name, [x])
dump!(d::Dump, name, x::Integer) = dump!(d, name, Int64[x])
finish(d::Dump) = write(joinpath(d.dir, "manifest.txt"), join(d.manifest, "\n") * "\n")

# Values every IEEE-754 implementation forms alike: a small integer hash scaled by a power of two.
pseudo(ix, iy, k) = ((ix * 7919 + iy * 104729 + k * 1299709) % 2048) / 1024.0 - 1.0
fill_pseudo(n1, n2, k, s=1.0) = Data.Array([s * pseudo(ix, iy, k) for ix = 1:n1, iy = 1:n2])

# Scalars of runme(), scripts/NavierStokes.jl:14-56, for a given grid.
function scalars(nx, ny)
    ly = 1.0; ρ = 1.0; vin = 1.0; psc = ρ * vin^2
    Re = 1e4; Fr = Inf; lx_ly = 0.6; a_ly = 0.05; b_ly = 0.05; ox_ly = 0.05; oy_ly = -0.4; β = 0 * π / 6
    lx = lx_ly * ly; ox = ox_ly * ly; oy = oy_ly * ly
    μ = 1 / Re * ρ * vin * ly; g = 1 / Fr^2 * vin^2 / ly
    a2 = (a_ly * ly)^2; b2 = (b_ly * ly)^2
    sinβ, cosβ = sincos(β)
    εit = 1e-3; niter = 50 * nx; nchk = 1 * (nx - 1)
    CFLτ = 0.9 / sqrt(2); CFL_visc = 1 / 4.1; CFL_adv = 1.0
    dx, dy = lx / nx, ly / ny
    dt = min(CFL_visc * dy^2 * ρ / μ, CFL_adv * dy / vin)
    damp = 2 / ny; dτ = CFLτ * dy
    (; ly, ρ, vin, psc, lx, ox, oy, μ, g, a2, b2, sinβ, cosβ, εit, niter, nchk, dx, dy, dt, damp, dτ)
end

# ---- 1. launch range of bc_x! ------------------------------------------------------------------
function probe_bc_x(d::Dump, nx, ny, name)
    A = Data.Array([Float64(ix + 1000 * iy) for ix = 1:nx, iy = 1:ny])
    B = copy(A)
    @parallel bc_x!(A)
    rows = Int64[iy for iy = 1:ny if A[1, iy] != B[1, iy] || A[end, iy] != B[end, iy]]
    println("bc_x! on a $nx x $ny array ($name) changed rows ",
            isempty(rows) ? "none" : "$(first(rows))..$(last(rows)) ($(length(rows)) of $ny)",
            rows == collect(1:min(nx, ny)) ? "  => rows 1..size(A,1): the single index runs over the first launch range" :
            rows == collect(1:ny) ? "  => every row" : "  => NEITHER of the two expected patterns")
    dump!(d, "bc_x_rows_" * name, rows)
end

# ---- 2. every kernel once ----------------------------------------------------------------------
function kernel_dumps(nx, ny)
    d = Dump(joinpath(OUT, "kernels_$(nx)x$(ny)"))
    s = scalars(nx, ny)
    dump!(d, "nx", nx); dump!(d, "ny", ny)
    probe_bc_x(d, nx, ny, "Pr"); probe_bc_x(d, nx, ny + 1, "Vy")
    Vx = fill_pseudo(nx + 1, ny, 1); Vy = fill_pseudo(nx, ny + 1, 2)
    τxx = @zeros(nx, ny); τyy = @zeros(nx, ny); τxy = @zeros(nx - 1, ny - 1)
    @parallel update_τ!(τxx, τyy, τxy, Vx, Vy, s.μ, s.dx, s.dy)
    dump!(d, "tau_txx", τxx); dump!(d, "tau_tyy", τyy); dump!(d, "tau_txy", τxy)
    τxx = fill_pseudo(nx, ny, 3); τyy = fill_pseudo(nx, ny, 4); τxy = fill_pseudo(nx - 1, ny - 1, 5)
    @parallel predict_V!(Vx, Vy, τxx, τyy, τxy, s.ρ, s.g, s.dt, s.dx, s.dy)
    dump!(d, "predict_Vx", Vx); dump!(d, "predict_Vy", Vy)
    C = fill_pseudo(nx, ny, 6); Vx = fill_pseudo(nx + 1, ny, 1); Vy = fill_pseudo(nx, ny + 1, 2)
    @parallel set_sphere!(C, Vx, Vy, s.a2, s.b2, s.ox, s.oy, s.sinβ, s.cosβ, s.lx, s.ly, s.dx, s.dy)
    dump!(d, "sphere_C", C); dump!(d, "sphere_Vx", Vx); dump!(d, "sphere_Vy", Vy)
    Vx = fill_pseudo(nx + 1, ny, 1); Vy = fill_pseudo(nx, ny + 1, 2); ∇V = @zeros(nx, ny)
    @parallel update_∇V!(∇V, Vx, Vy, s.dx, s.dy)
    dump!(d, "div_divV", ∇V)
    Pr = fill_pseudo(nx, ny, 7); dPrdτ = fill_pseudo(nx - 2, ny - 2, 8); ∇V = fill_pseudo(nx, ny, 9)
    @parallel update_dPrdτ!(Pr, dPrdτ, ∇V, s.ρ, s.dt, s.dτ, s.damp, s.dx, s.dy)
    dump!(d, "dPrdtau_dPrdtau", dPrdτ)
    @parallel update_Pr!(Pr, dPrdτ, s.dτ)
    dump!(d, "updatePr_Pr", Pr)
    set_bc_Pr!(Pr, 0.0)
    dump!(d, "bcPr_Pr", Pr)
    Rp = @zeros(nx - 2, ny - 2)
    @parallel compute_res!(Rp, Pr, ∇V, s.ρ, s.dt, s.dx, s.dy)
    dump!(d, "res_Rp", Rp); dump!(d, "res_maxabs", maximum(abs.(Rp)))
    Vx = fill_pseudo(nx + 1, ny, 1); Vy = fill_pseudo(nx, ny + 1, 2); Pr = fill_pseudo(nx, ny, 7)
    @parallel correct_V!(Vx, Vy, Pr, s.dt, s.ρ, s.dx, s.dy)
    dump!(d, "correct_Vx", Vx); dump!(d, "correct_Vy", Vy)
    Vprof = Data.Array([pseudo(ix, 0, 10) for ix = 1:nx])
    set_bc_Vel!(Vx, Vy, Vprof)
    dump!(d, "bcVel_Vx", Vx); dump!(d, "bcVel_Vy", Vy)
    # velocities of a few cells per time step, both signs, so that the back-trace crosses cells and clamps
    Vx_o = fill_pseudo(nx + 1, ny, 1, 2.5); Vy_o = fill_pseudo(nx, ny + 1, 2, 2.5); C_o = fill_pseudo(nx, ny, 6)
    Vx = copy(Vx_o); Vy = copy(Vy_o); C = copy(C_o)
    @parallel advect!(Vx, Vx_o, Vy, Vy_o, C, C_o, s.dt, s.dx, s.dy)
    dump!(d, "advect_Vx", Vx); dump!(d, "advect_Vy", Vy); dump!(d, "advect_C", C)
    finish(d)
end

# ---- 3. whole time steps -----------------------------------------------------------------------
function time_steps(nx, ny, nsteps)
    d = Dump(joinpath(OUT, "steps_$(nx)x$(ny)_$(nsteps)"))
    s = scalars(nx, ny)
    Pr = @zeros(nx, ny); dPrdτ = @zeros(nx - 2, ny - 2); C = @zeros(nx, ny); C_o = @zeros(nx, ny)
    τxx = @zeros(nx, ny); τyy = @zeros(nx, ny); τxy = @zeros(nx - 1, ny - 1)
    Vx = @zeros(nx + 1, ny); Vy = @zeros(nx, ny + 1); Vx_o = @zeros(nx + 1, ny); Vy_o = @zeros(nx, ny + 1)
    ∇V = @zeros(nx, ny); Rp = @zeros(nx - 2, ny - 2)
    Vprof = Data.Array([4 * s.vin * x / s.lx * (1.0 - x / s.lx) for x = LinRange(0.5 * s.dx, s.lx - 0.5 * s.dx, nx)])
    Vy[:, 1] .= Vprof
    yc = LinRange(-(s.ly - s.dy) / 2, (s.ly - s.dy) / 2, ny)
    Pr .= .-(yc' .- s.ly / 2) .* s.ρ .* s.g
    dump!(d, "Vprof", Vprof)
    iters = Int64[]; hist_iter = Int64[]; hist_err = Float64[]; hist_step = Int64[]
    for it = 1:nsteps
        @parallel update_τ!(τxx, τyy, τxy, Vx, Vy, s.μ, s.dx, s.dy)
        @parallel predict_V!(Vx, Vy, τxx, τyy, τxy, s.ρ, s.g, s.dt, s.dx, s.dy)
        @parallel set_sphere!(C, Vx, Vy, s.a2, s.b2, s.ox, s.oy, s.sinβ, s.cosβ, s.lx, s.ly, s.dx, s.dy)
        @parallel update_∇V!(∇V, Vx, Vy, s.dx, s.dy)
        done = 0
        for iter = 1:s.niter
            @parallel update_dPrdτ!(Pr, dPrdτ, ∇V, s.ρ, s.dt, s.dτ, s.damp, s.dx, s.dy)
            @parallel update_Pr!(Pr, dPrdτ, s.dτ)
            set_bc_Pr!(Pr, 0.0)
            done = iter
            if iter % s.nchk == 0
                @parallel compute_res!(Rp, Pr, ∇V, s.ρ, s.dt, s.dx, s.dy)
                err = maximum(abs.(Rp)) * s.ly^2 / s.psc
                push!(hist_step, it); push!(hist_iter, iter); push!(hist_err, err)
                if err < s.εit || !isfinite(err) break end
            end
        end
        push!(iters, done)
        @parallel correct_V!(Vx, Vy, Pr, s.dt, s.ρ, s.dx, s.dy)
        @parallel set_sphere!(C, Vx, Vy, s.a2, s.b2, s.ox, s.oy, s.sinβ, s.cosβ, s.lx, s.ly, s.dx, s.dy)
        set_bc_Vel!(Vx, Vy, Vprof)
        Vx_o .= Vx; Vy_o .= Vy; C_o .= C
        @parallel advect!(Vx, Vx_o, Vy, Vy_o, C, C_o, s.dt, s.dx, s.dy)
        @printf("%d x %d: step %d, %d Poisson passes\n", nx, ny, it, done)
    end
    dump!(d, "nx", nx); dump!(d, "ny", ny); dump!(d, "steps", nsteps)
    dump!(d, "iters", iters); dump!(d, "hist_step", hist_step); dump!(d, "hist_iter", hist_iter); dump!(d, "hist_err", hist_err)
    dump!(d, "Pr", Pr); dump!(d, "Vx", Vx); dump!(d, "Vy", Vy); dump!(d, "C", C); dump!(d, "dPrdtau", dPrdτ)
    for (k, v) in pairs(s)
        v isa Float64 && dump!(d, "scalar_" * replace(String(k), "ρ" => "rho", "μ" => "mu", "β" => "b", "ε" => "eps", "τ" => "tau"), v)
    end
    finish(d)
end

kernel_dumps(21, 37)
time_steps(21, 37, 3)
time_steps(153, 255, 2)
println("fixtures written to ", OUT)
# Last, because it may throw: with more columns than rows a bc_x! that runs over 1:size(A,1) leaves its array.
try
    probe_bc_x(Dump(joinpath(OUT, "probe_wide")), 40, 23, "wide")
catch e
    println("bc_x! on a 40 x 23 array threw ", typeof(e), "  => its index runs past the last row, i.e. over 1:size(A,1)")
end
