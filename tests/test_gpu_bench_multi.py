"""bench.py's N > 1 step (horizonator_amd/sharding.py: azimuth sectors, sparse strips, StripExchange
with preallocated bins, device-side ordering against the library's streams) on the GPU that is
there: two and three gloo ranks sharing it, and the RCCL-shaped path (device tensors, side stream,
self-exchange) with the one rank a single GPU allows.  bench.py itself compares the gathered
panorama with the single-GPU render of the same view, byte for byte, and says so in its line."""
import json
import os
import subprocess
import sys

import pytest

import hzutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(*args, timeout=600):
    if not hzutil.hip_available():
        pytest.skip("no HIP device")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    # a child process of its own (never an exec from a process that has touched the GPU)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "cfg2", "--steps", "4", "--warmup", "2",
                        "--no-cpu-baseline", "--no-host", "--no-extra", *args],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("ranks", [2, 3])
@pytest.mark.parametrize("wire", ["sparse", "packed"])
def test_sector_ranks_sharing_one_gpu_assemble_the_single_gpu_panorama(ranks, wire):
    line = _bench("--gpus", str(ranks), "--backend", "gloo", "--same-gpu", "--wire", wire)
    assert line["n_gpus"] == ranks and line["steps"] == 4
    assert line["gathered_panorama_equals_single_gpu_render"] is True
    assert len(line["config"]["sector_widths"]) == ranks and sum(line["config"]["sector_widths"]) == 8000
    assert line["value"] > 0 and line["scaling"] == "strong"
    # the line carries both gather modes (value: --gather's default, rotate) and the latency of one panorama at this N
    assert line["config"]["gather"] == "rotate"
    assert line["gather_root0"]["gathered_panorama_equals_single_gpu_render"] is True and line["gather_root0"]["value"] > 0
    assert line["single_panorama_latency_ms"]["value"] > 0


def test_the_rccl_shaped_exchange_with_one_rank():
    line = _bench("--gpus", "1", "--exchange-anyway")
    assert line["gathered_panorama_equals_single_gpu_render"] is True
    assert isinstance(line["config"]["strip_resends"], int)


def test_the_series_driven_from_c_with_one_rank():
    """--loop c: horizonator_rccl_render_series over an RCCL communicator of the library's own instead of the Python loop"""
    line = _bench("--gpus", "1", "--exchange-anyway", "--loop", "c")
    assert line["gathered_panorama_equals_single_gpu_render"] is True
    assert line["loop"]["host_us_per_panorama"] > 0


@pytest.mark.parametrize("ranks", [2, 3])
def test_the_series_driven_from_c_with_ranks_sharing_one_gpu(ranks):
    """horizonator_rccl_render_series with world > 1: RCCL refuses ranks that share a GPU, so the strips travel over gloo
    (horizonator_rccl_series_t::exchange) - the C loop's slots, its rotation of the gathering rank and its ordering against
    the contexts' streams meet a second and a third rank all the same; both gather modes against the single-GPU render"""
    line = _bench("--gpus", str(ranks), "--backend", "gloo", "--same-gpu", "--loop", "c")
    assert line["n_gpus"] == ranks
    assert line["gathered_panorama_equals_single_gpu_render"] is True
    assert line["gather_root0"]["gathered_panorama_equals_single_gpu_render"] is True
    assert line["loop"]["driver"].startswith("horizonator_rccl_render_series")
