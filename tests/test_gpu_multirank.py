"""N > 1 on the GPU: two rank processes share the box's one MI355X, each calls rt_trace / rt_trace_strips on its part,
rank 0 gathers (sharding.py, gloo with host staging) and the frame must equal the full-frame GPU trace and the oracle.
Also runs `python bench.py --gpus 2` directly (no launcher): bench.py must start its own ranks and print one line."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(world, argv, timeout=300):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="4")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multirank_worker.py")] + argv, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{outs[r][-3000:]}"


# 1080-style ragged split (53 rows / 2 ranks: bands of 26 and 27 rows; 7 strips -> 4 + 3) and an even one
@pytest.mark.parametrize("partition", ["bands", "strips"])
@pytest.mark.parametrize("w,h", [(256, 192), (97, 53)])
def test_two_ranks_one_gpu(tmp_path, partition, w, h):
    out = str(tmp_path / "res.json")
    _run_ranks(2, [partition, str(w), str(h), out])
    res = json.load(open(out))
    assert res["gathered_equals_full_gpu"], res
    assert res["gathered_equals_oracle"], res
    assert res["counters_sum"] == res["counters_full_gpu"] == res["counters_oracle"], res
    assert res["nonblack"] > 0


@pytest.mark.parametrize("partition", ["bands", "strips"])
def test_bench_self_launch_two_ranks(partition):
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: one JSON line with n_gpus = 2 and 2 ranks seen."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--grid", "100", "--width", "640", "--height", "360", "--no-extras", "--no-cpu-baseline",
                        "--inflight", "2", "--partition", partition],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks"]["world_size"] == 2 and d["ranks"]["self_launched"]
    assert d["ranks"]["partition"] == partition and d["value"] > 0


def test_bench_one_rank_rccl_api():
    """One rank, RCCL initialised and the per-frame gather issued through it (RT_BENCH_FORCE_DIST=1): the collective
    calls bench.py makes at N > 1 are accepted by the nccl backend (both partitions)."""
    for partition in ("bands", "strips"):
        env = dict(os.environ, RT_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
            env.pop(k, None)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--grid", "100",
                            "--width", "640", "--height", "360", "--no-extras", "--partition", partition],
                           env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
        assert d["ranks"]["backend"] == "nccl" and d["ranks"]["world_size"] == 1
        assert d["cpu_baseline"]["parity"]["gpu_frame_rows_equal_oracle"] is True
        assert d["cpu_baseline"]["parity"]["sum_box_tri_tests_equal_oracle"] is True


def test_bench_line_schema():
    """The one JSON line of `python bench.py` (N = 1, a small workload): every field of the driver's contract plus the
    roofline and cpu_baseline objects, and the parity verdict the CPU-baseline leg attaches."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--grid", "120",
                        "--width", "640", "--height", "360"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "serial_mrays", "build_ms"):
        assert k in d, k
    assert d["unit"] == "Mrays/s" and d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "l1" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "parity"):
        assert k in c, k
    assert c["kind"] == "port" and c["parity"]["gpu_frame_rows_equal_oracle"] is True
    assert c["parity"]["sum_box_tri_tests_equal_oracle"] is True
    assert abs(d["value"] - 640 * 360 * 6 / (d["ms_per_step"] * 6 * 1e-3) / 1e6) / d["value"] < 0.01
