"""bench.py end to end on the GPU box: the one-line JSON contract, and the N > 1 control flow (barriers, max-over-ranks reduction,
one frame per rank) with two ranks sharing the single GPU over gloo (IPK_BENCH_SHARE_GPU=1, a development hook; the driver's
multi-GPU runs use one GPU per rank over RCCL)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def test_bench_contract_single_gpu():
    r = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--width", "6000", "--height", "4000", "--cpu-seconds", "1"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["unit"] == "MP/s" and d["higher_is_better"] is True
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(d["cpu_baseline"])
    assert "bit-identical" in d["parity_check"] and d["value"] > 1000 and 0 < d["roofline"]["frac"] < 1


def test_bench_two_ranks_control_flow():
    env = dict(os.environ, IPK_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29533",
                        "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--width", "2048", "--height", "1024", "--no-cpu-baseline"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["frames_per_step"] == 2 and d["scaling"] == "weak" and d["value"] > 0


@pytest.mark.parametrize("cfa,H,W,nproc", [("RGGB", 150, 600, 2), ("GBRG", 301, 258, 3), ("GGRGGBGGBGGRBRGRBGGGBGGRGGRGGBRBGBRG", 180, 300, 4)])
def test_one_frame_banded_over_ranks_matches_oracle(cfa, H, W, nproc):
    """row-band sharding of ONE frame with the real kernel: band plan aligned to the CFA period, 1-row halo exchange between
    neighbours (dist.batch_isend_irecv), band form of the fused kernel, all-gather of the output -- ranks share the GPU over gloo"""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
                        "--master-port", str(29540 + nproc), os.path.join("tests", "helpers", "band_worker.py"), cfa, str(H), str(W)],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "BANDED_OK" in r.stdout, (r.stdout + r.stderr)[-3000:]


@pytest.mark.parametrize("n_frames,nproc", [(7, 2), (8, 4)])
def test_frame_batch_sharded_over_ranks(n_frames, nproc):
    """BASELINE.json configs[3] in small: a batch of frames, frame i -> rank i % N, no data-path collective; every frame equals the oracle"""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
                        "--master-port", str(29550 + nproc), os.path.join("tests", "helpers", "batch_worker.py"), str(n_frames), "200", "600"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "BATCH_OK" in r.stdout, (r.stdout + r.stderr)[-3000:]
