# scripts/DNS.jl on libnsk: `main(; do_vis, do_save)` with the parameter block DNS.jl:16-41 and the
# loop :70-113; every `@parallel f!(...)` is one call into the library.  NOT RUN here (no Julia in
# the build image).  DNS.jl does not run as written (undefined variables, statements cut by line
# breaks); the kernels called implement the algorithm it transcribes, scripts/code1.m:33-101.
include("NSK.jl")
using .NSK, MAT, Printf

function main(; do_vis=true, do_save=false, pressure_solver=:sor)
    # physics DNS.jl:16-31
    Lx = 1.0; Ly = 1.0; gx = 0.0; gy = -100.0; ρ₁ = 1.0; ρ₂ = 2.0; m0 = 0.01
    unorth = 0.0; usouth = 0.0; veast = 0.0; vwest = 0.0
    time = 0.0; nstep = 1; rad = 0.15; xcenter = 0.5; ycenter = 0.7
    # numerics DNS.jl:34-41
    nx = 32; ny = 32; dt = 0.00125; nsteps = 100; nvis = 10; maxit = 400; maxerr = 1e-3; beta = 1.2
    dx = Lx/nx; dy = Ly/ny
    c = NSK.Ctx(0)
    z(a, b) = NSK.zeros(c, a, b)
    # allocate DNS.jl:52-62
    ρ, ρₜ, ρo = z(nx+2,ny+2), z(nx+2,ny+2), z(nx+2,ny+2)
    u, v, p = z(nx+1,ny+2), z(nx+2,ny+1), z(nx+2,ny+2)
    ut, vt = z(nx+1,ny+2), z(nx+2,ny+1)
    uu, vv = z(nx+1,ny+1), z(nx+1,ny+1)
    tmp1, tmp2, dPdτ = z(nx+2,ny+2), z(nx+2,ny+2), z(nx+2,ny+2)
    xc, yc = NSK.vec(c, nx+2), NSK.vec(c, ny+2)
    NSK.upload!(c, xc, reshape([dx*(i-1.5) for i = 1:nx+2], nx+2, 1))     # code1.m:22
    NSK.upload!(c, yc, reshape([dy*(j-1.5) for j = 1:ny+2], ny+2, 1))
    save(k) = matwrite("out_dns/step_$k.mat", Dict("Vx"=>NSK.download(c,uu), "Vy"=>NSK.download(c,vv),
                       "C"=>NSK.download(c,ρ), "dx"=>dx, "dy"=>dy))
    do_save && (!ispath("./out_dns") && mkdir("./out_dns"); save(0))
    NSK.upload!(c, ρ, fill(ρ₁, nx+2, ny+2))                                 # ρ .= ρ₁   DNS.jl:64
    NSK.set_density!(c, ρ, nothing, xcenter, ycenter, xc, yc, rad, ρ₂, 0.0, nx, ny)
    for is = 1:nsteps
        time += dt
        NSK.set_tangential!(c, u, v, usouth, unorth, vwest, veast, nx, ny)
        NSK.temp_vel!(c, ut, vt, u, v, ρ, dt, dx, dy, gx, gy, m0, nx, ny)
        NSK.set_boundary!(c, ρₜ, ρ, 1000.0, nx, ny)
        NSK.compute_sources!(c, tmp1, tmp2, ρₜ, ut, vt, dt, dx, dy, nx, ny)
        if pressure_solver == :pt
            it, _ = NSK.pt_pressure!(c, p, dPdτ, tmp1, ρₜ, 0.9/sqrt(2)*dy, 0.72, dx, dy, nx, ny, maxit, maxerr)
        else
            it, _ = NSK.solve_pressure!(c, p, tmp1, tmp2, ρₜ, beta, dx, dy, nx, ny, maxit, maxerr; solver=pressure_solver)
        end
        it == maxit && @printf("Warning: pressure solver did not converge in %d iterations\n", it)
        NSK.correct_vel!(c, u, v, ut, vt, p, ρ, dt, dx, dy, nx, ny)
        NSK.copy!(c, ρo, ρ)
        NSK.advect_density!(c, ρ, ρo, u, v, dt, dx, dy, m0, nx, ny)
        NSK.velocity!(c, uu, vv, u, v, nx, ny)
        if is % nvis == 0
            do_save && (save(nstep); nstep += 1)
            println("time = $time")
        end
    end
    NSK.destroy(c)
end

if abspath(PROGRAM_FILE) == @__FILE__
    main(; do_vis=false, do_save=true)
end
