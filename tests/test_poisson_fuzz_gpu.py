"""A seeded random sweep of the multi-pass Poisson kernel against the oracle (tools/fuzz_poisson.py:
grid sizes around the strip and chunk boundaries, both storage forms, every pass count per launch,
task heights, slab windows, tiny / huge / zero-laden data), compared as bit patterns -- signed zeros
included.  The tool itself runs tens of thousands of cases; the suite keeps a few hundred."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("seed,max_n", [(11, 700), (12, 2000)])
def test_random_sweep(oracle, seed, max_n):
    import torch
    from navierstokes_amd import _lib
    import fuzz_poisson
    ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
    try:
        failures = fuzz_poisson.sweep(ctx, oracle, 400, seed, max_n, verbose=False)
    finally:
        ctx.close()
    assert not failures, failures[:5]


def test_random_sweep_of_whole_time_steps():
    """tools/fuzz_step.py: random grids, physical scalars, ellipse geometry, initial fields and loop
    lengths through nsk_ns_step (fused kernels or one launch per call site), two steps each, every
    array of the reference compared as bit patterns."""
    import fuzz_step
    failures = fuzz_step.sweep(150, 21, 260, verbose=False)
    assert not failures, failures[:5]


def test_random_sweep_of_slab_groups(gpu, oracle):
    """tools/fuzz_slabs.py: 2-6 slabs on cuda:0 through nsk_slabs_* with random heights, halo widths
    and pass counts (Poisson passes with residual; whole time steps) against the single domain."""
    import fuzz_slabs
    failures = fuzz_slabs.sweep(oracle, 250, 31, verbose=False)
    assert not failures, failures[:5]


def test_random_sweep_of_the_bubble_drivers():
    """tools/fuzz_bubble.py: DNS.jl / code3_fun.jl steps with random grids (on both sides of what the
    several-sweeps-in-flight SOR takes), densities, wall velocities, SOR settings, solvers and fronts."""
    import fuzz_bubble
    failures = fuzz_bubble.sweep(150, 41, verbose=False)
    assert not failures, failures[:5]
