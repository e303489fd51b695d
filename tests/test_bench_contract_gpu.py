"""bench.py's one-line JSON contract: the keys and types a harness reads, on a short run of the real
command (a few hundred passes instead of a whole check interval, a one-second CPU sample)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line():
    j = run_bench("--steps", "2", "--warmup", "1", "--passes", "240", "--time-steps", "1", "--cpu-seconds", "1")
    assert j["metric"] == "T_eff" and j["unit"] == "GB/s" and j["higher_is_better"] is True
    assert (j["n_gpus"], j["steps"], j["warmup"]) == (1, 2, 1)
    assert j["dtype"] == "f64" and j["data"] == "synthetic" and j["scaling"] == "weak" and j["vs_baseline"] is None
    assert "2559x4095" in j["config"]["workload"] and "model" not in j["config"]
    assert j["value"] > 5000 and 5 < j["t_iter_us"] < 80          # the 70 % target is 74.9 us per pass
    assert abs(j["ms_per_step"] - j["t_iter_us"] * 240 / 1e3) < 1e-6 * j["ms_per_step"] + 1e-9
    r = j["roofline"]
    assert r["bound"] in ("hbm", "valu") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["launch_us"] * 1e-6) / 1e9) < 1e-6 * r["achieved"]
    assert r["algorithmic_bytes_per_launch"] == 40.0 * 2559 * 4095 * r["passes_per_launch"]
    # the kernel's time comes from the timed loop itself: it cannot exceed the loop's wall time
    assert r["launches_per_step"] == 240 / r["passes_per_launch"]
    assert r["launch_us"] * r["launches_per_step"] <= j["ms_per_step"] * 1e3 * (1 + 1e-9)
    assert r["launch_us"] * r["launches_per_step"] > 0.8 * j["ms_per_step"] * 1e3      # and is most of it
    assert "fabric" in r["traffic_kind"] and "hbm_frac" not in r
    if r["traffic"] is not None:        # counters taken from this very kernel build
        assert 0.2 < r["fabric_frac"] < 1.0 and 0.2 < r["valu_frac"] < 1.0
        assert r["valu_frac_sustained"] is None or r["valu_frac"] * 0.9 < r["valu_frac_sustained"] < 1.0
        assert abs(r["frac_of_bound"] - max(r["fabric_frac"], r["valu_frac"], r["valu_frac_sustained"] or 0.0)) < 1e-12
        assert r["traffic"] >= 0.15 * r["algorithmic_bytes_per_launch"]
    else:
        assert r["fabric_frac"] is None and r["valu_frac"] is None and r["frac_of_bound"] is None
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "GB/s" and 1 <= c["cores"] <= c["cores_present"] and c["value"] > 0
    assert "passes" in c["sample"]
    t = j["time_steps"]
    assert t["steps"] == 1 and t["poisson_iters"][0] % 10232 == 0 and t["steps_per_s"] > 0
    # the reference's own default grid, both forms of the Poisson passes: same pass counts, the one-launch form faster
    d = j["time_steps_default_grid"]
    a, b = d["one_launch_per_run_of_passes"], d["launch_per_k_passes"]
    assert d["grid"] == [153, 255] and a["poisson_iters"] == b["poisson_iters"] and all(i % 152 == 0 for i in a["poisson_iters"])
    assert 0 < a["us_per_pass_whole_step"] < b["us_per_pass_whole_step"]


def test_two_rank_line_as_the_driver_launches_it():
    """N > 1 the way a harness starts it (torch.distributed.run, one rank per process), rehearsed on the
    one GPU: both ranks share cuda:0 and the halo rows travel through gloo -- the decomposition, the
    schedule, the max over ranks and the extra keys of the line are the ones of a real node, the rate is
    not a measurement."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--passes", "150", "--backend", "gloo", "--one-gpu"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]               # rank 0 only
    j = json.loads(lines[0])
    assert (j["n_gpus"], j["steps"], j["warmup"], j["scaling"]) == (2, 2, 1, "weak")
    assert j["config"]["parallelism"] == "slab2" and j["config"]["halo_rows"] == 60
    assert j["config"]["grid_per_gpu"] == [2559, 4095] and j["schedule"] == "edges"
    assert j["exchanges_timed"] == 2 * 3 and j["comm_exposed_us_per_exchange"] > 0      # ceil(150 / 60) per step
    assert abs(j["value"] - 40.0 * 2559 * 4095 * 2 / (j["t_iter_us"] * 1e-6) / 1e9) < 1e-6 * j["value"]
    assert "cpu_baseline" not in j and "time_steps" not in j
    # who took part, gathered from every rank
    assert j["world_size"] == 2 and j["backend"] == "gloo" and [e["rank"] for e in j["ranks"]] == [0, 1]
    assert all(e["device"] == "cuda:0" and e["name"] for e in j["ranks"]) and j["distinct_devices"] == 1
    assert len({e["pid"] for e in j["ranks"]}) == 2
    r = j["roofline"]
    assert r["launch_us"] * r["launches_per_step"] <= j["ms_per_step"] * 1e3 * (1 + 1e-9)


def test_first_contact_has_a_deadline():
    """A rank that never arrives (here: rank 1 sleeps instead of joining the process group) must not hang
    the run: the other rank's watchdog says where it was and the command ends with a non-zero status
    within the deadline."""
    import socket
    import time
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "1", "--warmup", "0", "--passes", "60", "--backend", "gloo", "--one-gpu",
           "--first-contact-s", "6"]
    t0 = time.time()
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT,
                         env=dict(os.environ, NSK_BENCH_TEST_STALL="1"))
    # (two interpreters start and import torch before the 6 s run; on a busy box that alone has taken most of a minute)
    assert out.returncode != 0 and time.time() - t0 < 180, (out.returncode, time.time() - t0, out.stderr[-2000:])
    assert "no completion within 6 s -- rank 0 of 2 on cuda:0" in out.stderr, out.stderr[-3000:]
    assert "phase: process group set-up" in out.stderr and "peers [1]" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]        # no result line


def test_traffic_record_matches_the_build():
    """profiles/poisson_traffic.json is only used when it was measured on the kernel build in the tree."""
    from navierstokes_amd._lib import poisson_kernel_hash
    with open(os.path.join(ROOT, "profiles", "poisson_traffic.json")) as f:
        rec = json.load(f)
    entries = {k: v for k, v in rec.items() if k.startswith("k_poisson")}
    assert entries
    assert any(v.get("kernel_hash") == poisson_kernel_hash() for v in entries.values()), \
        "re-run tools/profile_round.sh + tools/make_traffic.py: the kernel source changed after the counters were taken"
