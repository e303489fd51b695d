# Drivers with the reference's surface, `runme(; n, do_vis, do_save)`, on top of libnsk.
# NOT RUN here (no Julia in the build image).  `script = :NavierStokes` takes the scalars of
# scripts/NavierStokes.jl:14-56 (153 x 255, Re = 1e4, no `n`), `script = :NavierStokes_Re6` those
# of scripts/NavierStokes_Re6.jl:14-56 (nx = 160n-1, ny = 256n-1, Re = 1e6).
include("NSK.jl")
using .NSK, MAT, Printf

function scalars(script::Symbol, n)
    re6 = script == :NavierStokes_Re6
    ly, ρ, vin = 1.0, 1.0, 1.0
    Re = re6 ? 1e6 : 1e4
    lx = 0.6ly; μ = 1/Re*ρ*vin*ly
    ny = re6 ? n*256-1 : 255
    nx = re6 ? n*160-1 : ceil(Int, ny*0.6)
    dx, dy = lx/nx, ly/ny
    dt = min(1/4.1*dy^2*ρ/μ, 1.0*dy/vin)
    (; nx, ny, lx, ly, ρ, vin, μ, dx, dy, dt, psc=ρ*vin^2, g=0.0, a2=(0.05ly)^2, b2=(0.05ly)^2,
       ox=0.05ly, oy=-0.4ly, sinβ=0.0, cosβ=1.0, εit=1e-3, niter=50nx, nchk=(re6 ? 4 : 1)*(nx-1),
       damp=2/ny, dτ=0.9/sqrt(2)*dy, nt=(re6 ? 10000 : 2000), nsave=(re6 ? 50 : 10))
end

function runme(; n=1, do_vis=true, do_save=false, script::Symbol=:NavierStokes_Re6)
    q = scalars(script, n); nx, ny = q.nx, q.ny
    c = NSK.Ctx(0)
    z(a, b) = NSK.zeros(c, a, b)
    Pr, dPrdτ, C, C_o = z(nx,ny), z(nx-2,ny-2), z(nx,ny), z(nx,ny)
    τxx, τyy, τxy = z(nx,ny), z(nx,ny), z(nx-1,ny-1)
    Vx, Vy, Vx_o, Vy_o = z(nx+1,ny), z(nx,ny+1), z(nx+1,ny), z(nx,ny+1)
    ∇V, Rp, Vprof = z(nx,ny), z(nx-2,ny-2), z(nx,1)
    vp = [4*q.vin*x/q.lx*(1.0-x/q.lx) for x = LinRange(0.5q.dx, q.lx-0.5q.dx, nx)]
    NSK.upload!(c, Vprof, reshape(vp, nx, 1))
    vy0 = zeros(nx, ny+1); vy0[:,1] .= vp; NSK.upload!(c, Vy, vy0)
    p = NSK.Params(nx, ny, 0, ny, q.niter, q.nchk, 0, 0, 0, 0, q.lx, q.ly, q.ρ, q.vin, q.psc, q.μ, q.g,
                   q.a2, q.b2, q.ox, q.oy, q.sinβ, q.cosβ, q.εit, q.dx, q.dy, q.dt, q.damp, q.dτ)
    s = NSK.State(Pr.p, dPrdτ.p, C.p, C_o.p, τxx.p, τyy.p, τxy.p, Vx.p, Vy.p, Vx_o.p, Vy_o.p, ∇V.p, Rp.p, Vprof.p)
    save(it) = matwrite("out_vis/step_$it.mat", Dict("Pr"=>NSK.download(c,Pr), "Vx"=>NSK.download(c,Vx),
                        "Vy"=>NSK.download(c,Vy), "C"=>NSK.download(c,C), "dx"=>q.dx, "dy"=>q.dy))
    do_save && (!ispath("./out_vis") && mkdir("./out_vis"); save(0))
    for it = 1:q.nt
        println("#it = $it")
        _, hist = NSK.step!(c, s, p)
        for (iter, err) in hist; @printf("  #iter = %d, err = %1.3e\n", iter, err); end
        do_save && it % q.nsave == 0 && save(it)
    end
    NSK.destroy(c)
end

# scripts/NavierStokes_Re6.jl:252 ends with  runme(; n=16, do_vis=false, do_save=true);
# scripts/NavierStokes.jl:252 with           runme(; do_vis=true, do_save=false)  ->  script=:NavierStokes
if abspath(PROGRAM_FILE) == @__FILE__
    runme(; n=16, do_vis=false, do_save=true)
end
