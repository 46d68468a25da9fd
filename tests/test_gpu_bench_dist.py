"""bench.py's N > 1 leg end to end on ONE GPU: two ranks launched by torch.distributed.run exactly as the bench contract says
(the launcher is all that is used of torch), the shared-memory transport instead of RCCL (CTK_DIST_BACKEND=shm: RCCL refuses
two ranks on one device), weak scaling = two members concatenated on the time axis.  The tracked count must equal the
one-call result on the concatenated slab."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_two_ranks_weak_scaling():
    env = dict(os.environ, CTK_DIST_BACKEND="shm", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "era5_1deg_90",
           "--strong-steps", "24"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]                       # ONE JSON line on stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["unit"] == "timesteps/s"
    assert out["config"]["total_timesteps"] == 180 and out["value"] > 0
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(out["roofline"])
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["cores"] == 1
    assert out["config"]["distinct_devices"] == 1
    assert out["config"]["transport"] == "shm" and out["config"]["collectives_per_step"]["allgathers"] >= 4
    # the run proves itself: every rank's flag shard checksummed against the one-call result on the concatenated slab
    assert out["config"]["parity_checked"] is True, out["config"]["parity"]
    assert out["config"]["parity"]["shards_equal"] == [True, True]
    assert out["config"]["parity"]["n_tracked_one_call"] == out["config"]["n_tracked"]
    # strong scaling of one device-generated 0.25 degree slab, measured in the same launch
    sb = out["strong_025deg"]
    assert sb["n_gpus"] == 2 and sb["n_tracked_equal"] is True and sb["ms_per_step"] > 0 and sb["ms_per_step_1gpu"] > 0
    assert sb["speedup_vs_1gpu"] == pytest.approx(sb["ms_per_step_1gpu"] / sb["ms_per_step"])
    assert sb["collectives_per_step"]["allgathers"] >= 4

    from contrack_amd import _native, synth
    from contrack_amd.contrack import row_weights
    a = np.concatenate([synth.smooth_field(90, 181, 360, seed=r) for r in range(2)], axis=0)
    lat, _ = synth.grid(181, 360)
    w = row_weights(lat, np.float32(1.0), np.float32(1.0))
    with _native.Tracker(0) as t:
        _, n = t.track(a, np.full(180, 160.0), 0, w, 0.5, 5, True)
    assert out["config"]["n_tracked"] == n


@pytest.mark.gpu
def test_bench_gpus_2_without_a_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment: the ranks are started by bench.py itself -- the line says
    n_gpus 2 (never a silent 1-GPU line), proves itself (parity_checked) and carries the CPU leg"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(CTK_DIST_BACKEND="shm", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "era5_1deg_90", "--strong-steps", "0"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["total_timesteps"] == 180
    assert out["config"]["parity_checked"] is True and out["config"]["parity"]["shards_equal"] == [True, True]
    assert out["cpu_baseline"]["value"] > 0
    assert "strong_025deg" not in out


@pytest.mark.gpu
def test_bench_refuses_a_launcher_of_another_size():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "era5_1deg_90"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and p.stdout.strip() == "" and "refusing" in p.stderr


def test_bench_gpus_n_without_devices_exits_nonzero():
    """(runs on CPU too) RCCL wants one device per rank: `--gpus 8` on a box with fewer devices must fail, not print a line"""
    from contrack_amd import _native
    if _native.device_count() >= 8:
        pytest.skip("eight devices visible: the real launch would start")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "CTK_DIST_BACKEND")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and p.stdout.strip() == "" and "RCCL wants one device per rank" in p.stderr
