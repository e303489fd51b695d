# scripts/code3_fun.jl on libnsk: the script body :296-516 (parameters :296-309, front :340-351,
# time loop :354-494) with every kernel call replaced by one call into the library.  NOT RUN here
# (no Julia in the build image).  The front vectors are device arrays of capacity `cap`; where the
# reference resizes its host vectors (:462-465) the library returns ERR_FRONT_CAPACITY with the front as
# it was before the regridding in xs, ys, and the driver regrids again on vectors twice as long.
include("NSK.jl")
using .NSK, Printf

function code3(; nstep=300, cap=1024)
    Lx = 1.0; Ly = 1.0; gx = 0.0; gy = -100.0; rho1 = 1.0; rho2 = 2.0; m1 = 0.01; m2 = 0.05; sigma = 10.0
    rro = rho1; unorth = 0.0; usouth = 0.0; veast = 0.0; vwest = 0.0
    rad = 0.15; xc = 0.5; yc = 0.7
    nx = 32; ny = 32; maxit = 200; maxError = 0.001; beta = 1.2; Nf = 100
    dx = Lx/nx; dy = Ly/ny
    dt = min(0.00125, 0.25*min(dx,dy)^2*min(rho1,rho2)/m1)                 # code3_fun.jl:337, m == m1 there
    c = NSK.Ctx(0)
    z(a, b) = NSK.zeros(c, a, b)
    u, v, p = z(nx+1,ny+2), z(nx+2,ny+1), z(nx+2,ny+2)
    ut, vt, un, vn = z(nx+1,ny+2), z(nx+2,ny+1), z(nx+1,ny+2), z(nx+2,ny+1)
    tmp1, tmp2, fx, fy = z(nx+2,ny+2), z(nx+2,ny+2), z(nx+2,ny+2), z(nx+2,ny+2)
    r, m, rn, mn, rt = z(nx+2,ny+2), z(nx+2,ny+2), z(nx+2,ny+2), z(nx+2,ny+2), z(nx+2,ny+2)
    NSK.upload!(c, r, fill(rho1, nx+2, ny+2)); NSK.upload!(c, m, fill(m1, nx+2, ny+2))   # :332-333
    xf, yf, xfn, yfn = NSK.vec(c, cap), NSK.vec(c, cap), NSK.vec(c, cap), NSK.vec(c, cap)
    uf, vf, tx, ty, xs, ys = (NSK.vec(c, cap) for _ = 1:6)
    x0 = zeros(cap, 1); y0 = zeros(cap, 1)                                 # set_front! :20-26
    for l = 1:Nf+2
        x0[l] = xc - rad*sin(2.0*pi*(l-1)/Nf); y0[l] = yc + rad*cos(2.0*pi*(l-1)/Nf)
    end
    NSK.upload!(c, xf, x0); NSK.upload!(c, yf, y0)
    time = 0.0
    for is = 1:nstep
        for (a, b) in ((un,u), (vn,v), (rn,r), (mn,m), (xfn,xf), (yfn,yf)); NSK.copy!(c, a, b); end   # :356-361
        for substep = 1:2
            NSK.surface_tension!(c, fx, fy, tx, ty, xf, yf, sigma, dx, dy, nx, ny, Nf)
            NSK.set_tangential!(c, u, v, usouth, unorth, vwest, veast, nx, ny)
            NSK.temp_vel_adv!(c, ut, vt, u, v, r, dt, dx, dy, fx, fy, gx, gy, rro, nx, ny)
            NSK.temp_vel_diff!(c, ut, vt, u, v, r, dt, dx, dy, m, nx, ny)
            NSK.set_boundary!(c, rt, r, 1000.0, nx, ny)
            NSK.compute_sources!(c, tmp1, tmp2, rt, ut, vt, dt, dx, dy, nx, ny)
            NSK.solve_pressure!(c, p, tmp1, tmp2, rt, beta, dx, dy, nx, ny, maxit, maxError)
            NSK.correct_vel!(c, u, v, ut, vt, p, r, dt, dx, dy, nx, ny)
            NSK.advect_move_front!(c, uf, vf, xf, yf, u, v, dx, dy, dt, Lx, Ly, nx, ny, Nf)
            NSK.distribute!(c, fx, fy, xf, yf, rho1, rho2, dx, dy, nx, ny, Nf)
            NSK.construct_den!(c, r, fx, fy, beta, maxit, maxError, dx, dy, nx, ny)
            NSK.update_visco!(c, r, m, m1, m2, rho1, rho2, nx, ny)
        end
        for (a, b) in ((u,un), (v,vn), (r,rn), (m,mn)); NSK.average!(c, a, b, a.n1*a.n2); end            # :445-450
        NSK.average!(c, xf, xfn, Nf+2); NSK.average!(c, yf, yfn, Nf+2)
        Nf = try                                                                                         # :455-494
            NSK.regrid_front!(c, xf, yf, xs, ys, dx, dy, Nf, cap)
        catch e
            (e isa NSK.NskError && e.code == NSK.ERR_FRONT_CAPACITY) || rethrow()
            xo = NSK.download(c, xs); yo = NSK.download(c, ys)              # the front before the regridding
            nn = -1
            while nn < 0                                                     # resize!, :462-465
                cap *= 2
                xf, yf, xfn, yfn, uf, vf, tx, ty, xs, ys = (NSK.vec(c, cap) for _ = 1:10)
                x0 = zeros(cap, 1); y0 = zeros(cap, 1); x0[1:Nf+2] = xo[1:Nf+2]; y0[1:Nf+2] = yo[1:Nf+2]
                NSK.upload!(c, xf, x0); NSK.upload!(c, yf, y0)
                try
                    nn = NSK.regrid_front!(c, xf, yf, xs, ys, dx, dy, Nf, cap)
                catch e2
                    (e2 isa NSK.NskError && e2.code == NSK.ERR_FRONT_CAPACITY) || rethrow()
                end
            end
            nn
        end
        time += dt
        @printf("step %d  time %.5f  front points %d\n", is, time, Nf)
    end
    NSK.destroy(c)
end

if abspath(PROGRAM_FILE) == @__FILE__
    code3()
end
