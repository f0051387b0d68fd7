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


def test_bench_gpus_n_invoked_plainly_starts_n_ranks_or_refuses():
    """`python bench.py --gpus 2` WITHOUT a launcher: on this one-GPU box it must refuse (non-zero exit, no JSON line) rather than print a line for
    one rank; with the development hook that lets ranks share the GPU it starts the two ranks itself (torch.distributed.run) and the line says so --
    n_gpus 2, and the rank count the library's own communicator saw (ipk_comm_info)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "IPK_BENCH_SHARE_GPU")}
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--width", "2048", "--height", "1024", "--no-cpu-baseline", "--prewarm-ms", "0"]
    import torch
    if torch.cuda.device_count() < 2:
        r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")], (r.returncode, r.stdout[-500:])
        assert "refusing" in r.stderr
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=dict(env, IPK_BENCH_SHARE_GPU="1"))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["frames_per_step"] == 2 and d["config"]["ipk_comm"]["ranks"] == 2
    # a WORLD_SIZE that is not --gpus is refused as well
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=600,
                       env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.parametrize("nproc", [1, 2])
def test_bench_default_control_flow_with_extras(nproc):
    """the DEFAULT command's whole control flow -- parity check, cold-clock leg, copy ceiling, the other data kinds, the 64-frame batch
    leg with its gather -- on small frames (IPK_BENCH_DEV_SMALL), alone and with two ranks sharing the GPU: every extra object is present
    and the ranks stay in step (a rank that skipped a barrier would hang this test)"""
    env = dict(os.environ, IPK_BENCH_DEV_SMALL="1", IPK_BENCH_SHARE_GPU="1")
    cmd = [sys.executable, "bench.py", "--gpus", str(nproc), "--steps", "3", "--warmup", "1", "--cpu-seconds", "0.5", "--prewarm-ms", "10"]
    if nproc > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1", "--master-port", "29535"] + cmd[1:]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == nproc and d["scaling"] == "weak" and d["config"]["frames_per_step"] == nproc
    assert "cold_ms" in d["config"] and "copy_ceiling_GBps" in d["roofline"] and set(d["other_data"]) == {"smooth", "photo"}
    b = d["batch_64x24MP"]
    assert b["value"] > 0 and b["scaling"] == "strong" and "bit-identical" in b["parity_check"]       # the batch launch itself against the oracle
    assert "ipk_copy_probe_GBps" in d["roofline"]["copy_ceiling_detail"] and d["value_from_median_step"] > 0
    if nproc > 1:
        g = b["with_gather"]
        # the library's own gather (host transport here: the ranks share the GPU) must have moved the frames, f32 and 8-bit
        assert g["ipk_f32"]["ok"] and g["ipk_u8"]["ok"] and g["ipk_f32"]["every_rank_holds_the_same_frames"], g
        assert g["ipk_u8"]["gathered_bytes_per_rank_per_step"] * 4 == g["ipk_f32"]["gathered_bytes_per_rank_per_step"]
        # the multi-GPU claim, first-class in the line: the batch on N GPUs against the same batch on one, per delivery of the results, each with
        # the xGMI arithmetic beside the measurement
        sc = d["scale"]
        assert sc["n_gpus"] == 2 and sc["n1_batch_ms"] > 0 and "compute_only" in sc["claim"]
        assert sc["compute_only"]["ms"] > 0 and sc["compute_only"]["expected_speedup"] == 2.0 and sc["compute_only"]["speedup"] > 0
        for k in ("gather_to_root_u8", "gather_to_root_u8_overlapped", "all_gather_f32", "all_gather_f32_overlapped"):
            assert sc[k]["ms"] > 0 and sc[k]["expected_ms"] > 0 and sc[k]["link_ms"] > 0 and sc[k]["speedup"] > 0, (k, sc[k])
        assert sc["all_gather_f32"]["link_ms"] == pytest.approx(4 * sc["gather_to_root_u8"]["link_ms"], rel=0.02)
        assert g["ipk_u8_to_root"]["root"] == 0 and g["ipk_f32_overlapped"]["overlapped"] is True
        # the banded single-frame leg runs in child processes on the RCCL transport: with two ranks on ONE GPU RCCL refuses the communicator,
        # and the leg must say so loudly (ok false, named in failed_legs) instead of costing the line or passing silently
        assert d["band_mode"]["ok"] is False and "band_mode" in d["failed_legs"], d.get("band_mode")
        assert d["config"]["ipk_comm"] == {"ranks": 2, "transport": "host"}
        # ... and the same batch over the same devices from ONE process (rank 0's child, two contexts on the shared GPU here)
        sp = sc["single_process"]
        assert sp["n_gpus"] == 2 and sp["devices"] == [0, 0] and sp["compute_only"]["speedup"] > 0 and "bit-identical" in sp["parity_check"], sp
        assert sp["host_in_host_out_u8"]["same_bytes_as_one_context"] is True
        assert d["scale_is_the_claim"] is True
    else:
        assert d["failed_legs"] == []
    assert "bit-identical" in d["parity_check"] and "cpu_baseline" in d


def test_bench_single_process_over_two_contexts():
    """python bench.py --gpus 2 --single-process: ONE process, one ipk_ctx per device (two on this box's one GPU), frame i -> member i mod 2 through
    ipk_pipeline_run_batch_multi; the line keeps the contract and carries the `scale` object of that mode, device-resident and host-to-host"""
    env = dict(os.environ, IPK_BENCH_DEV_SMALL="1")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--single-process", "--devices", "0,0", "--steps", "3", "--warmup", "1", "--cpu-seconds", "0.5",
                        "--prewarm-ms", "10"], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _last_json(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 2 and d["config"]["single_process"] is True and d["config"]["devices"] == [0, 0] and d["value"] > 0
    sc = d["scale"]
    assert sc["n_gpus"] == 2 and sc["distinct_devices"] == 1 and sc["n1_batch_ms"] > 0 and sc["compute_only"]["ms"] > 0 and sc["fused_batch_launches"] is True
    assert "bit-identical" in sc["parity_check"] and sc["host_in_host_out_u8"]["same_bytes_as_one_context"] is True and d["scale_is_the_claim"] is True
    # a device list that does not name N devices, or names one the box does not have, is refused without a line
    for extra in (["--devices", "0"], ["--devices", "0,7"]):
        r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--single-process", "--steps", "1"] + extra, cwd=ROOT, capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_reports_the_box_state_and_the_host_boundary():
    """config.shader_clock_GHz / socket_power_W (the box's state under the timed workload: a slow box and a slow kernel can be told apart) and the
    host_boundary object (host Vec in, host Vec out: what a drop-in behind Pipeline::run crosses) in the line"""
    r = subprocess.run([sys.executable, "bench.py", "--config", "c2", "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--host-boundary"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _last_json(r.stdout)
    c = d["config"]
    assert 0.3 < c["shader_clock_GHz"] < 3.0 and "ipk_clock_probe" in c["clock_power_method"], c
    if "socket_power_W" in c:                                   # hwmon is there on the MI355X boxes; a container without /sys access reports the clock alone
        assert 50 < c["socket_power_W"] < 2000
    hb = d["host_boundary"]
    for k in ("to_u8", "to_f32"):
        e = hb[k]
        assert e["ms_per_frame_batched"] > 0 and e["ms_per_frame_single"] > 0 and 0 < e["frac_of_pcie_floor"]["batched"] <= 1.2 and e["used_fused"] is True, e
    assert hb["to_u8"]["link_measured"]["up_GBps"] > 5 and hb["to_u8"]["link_measured"]["both_at_once_ms"] > 0 and hb["to_f32"]["frac_of_measured_link"] > 0.3
    assert "output_8bit" in hb["to_u8"]["parity_check"] and hb["to_f32"]["pcie_bytes_per_frame"]["down"] == 4 * hb["to_u8"]["pcie_bytes_per_frame"]["down"]
    assert d["failed_legs"] == []


def test_bench_batch_mode_and_configs():
    """bench.py --batch (BASELINE.json configs[3] shape, small here; the full 64 x 24 MP batch is tests/test_gpu_fused.py's) and the
    per-config modes: one JSON line each, strong scaling for a fixed batch, a roofline object per config"""
    r = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--batch", "6", "--width", "1200", "--height", "800", "--no-cpu-baseline",
                        "--prewarm-ms", "0"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d["scaling"] == "strong" and d["config"]["frames_per_step"] == 6 and d["value"] > 0 and "with_gather" in d and "bit-identical" in d["parity_check"]
    for cfg, kern in (("c5", "k_raw_scaled_demosaic"), ("c5b", "k_fused_bayer"), ("c2", "k_fused_bayer")):
        r = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--config", cfg, "--no-cpu-baseline", "--no-check", "--prewarm-ms", "0"],
                           cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        d = _last_json(r.stdout)
        assert kern in d["roofline"]["kernel"] and 0 < d["roofline"]["frac"] < 1 and d["config"]["baseline_config"].startswith("BASELINE.json configs[")


def test_bench_batch_two_ranks():
    env = dict(os.environ, IPK_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29534",
                        "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "6", "--width", "1024", "--height", "512", "--no-cpu-baseline", "--prewarm-ms", "0"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["frames_per_step"] == 6 and d["scaling"] == "strong" and d["value"] > 0
    assert d["with_gather"]["ipk_f32"]["ok"] and d["with_gather"]["ipk_u8"]["ok"]        # the library's gather; torch's over gloo may refuse device tensors
    assert d["scale"]["compute_only"]["speedup"] > 0 and d["scale"]["all_gather_f32_overlapped"]["ms"] > 0
    assert d["scale_is_the_claim"] is True and "`scale` object" in d["value_basis"]      # an N > 1 line says which of its numbers carries the multi-GPU claim


@pytest.mark.parametrize("cfa,H,W,nproc", [("RGGB", 150, 600, 2), ("GBRG", 301, 258, 3), ("GGRGGBGGBGGRBRGRBGGGBGGRGGRGGBRBGBRG", 180, 300, 4)])
def test_one_frame_banded_over_ranks_matches_oracle(cfa, H, W, nproc):
    """row-band sharding of ONE frame through the C entry points with the real kernel: ipk_band_plan aligned to the CFA period, the 1-row
    halo exchange in place on the slab (ipk_band_exchange_halo), the band form of the fused kernel writing into its rows of the frame,
    ipk_band_gather / ipk_band_gather_begin in place (f32 and 8-bit) -- ranks share the GPU, so the communicator runs its host transport"""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
                        "--master-port", str(29540 + nproc), os.path.join("tests", "helpers", "band_worker.py"), cfa, str(H), str(W)],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "BANDED_OK" in r.stdout, (r.stdout + r.stderr)[-3000:]


@pytest.mark.parametrize("cfa,H,W,nW,nH,nproc", [("GGRGGBGGBGGRBRGRBGGGBGGRGGRGGBRBGBRG", 576, 864, 216, 144, 2), ("GGRGGBGGBGGRBRGRBGGGBGGRGGRGGBRBGBRG", 580, 870, 217, 145, 3),
                                                 ("RGGB", 400, 600, 150, 100, 4)])
def test_scaled_path_banded_over_ranks_matches_oracle(cfa, H, W, nW, nH, nproc):
    """config 5's shape (X-Trans -> 4x smaller) sharded by OUTPUT-row bands: ipk_band_plan_scaled gives each rank the source rows its
    windows read (scaling.rs:84-94), ipk_raw_scaled_demosaic_band computes the band, ipk_band_gather assembles the frame on rank 0"""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
                        "--master-port", str(29560 + nproc), os.path.join("tests", "helpers", "band_worker.py"), cfa, str(H), str(W), "scaled", str(nW), str(nH)],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "BANDED_OK" in r.stdout, (r.stdout + r.stderr)[-3000:]


@pytest.mark.parametrize("nranks,w,h", [(2, 600, 150), (3, 1024, 258), (4, 300, 26)])
def test_banded_frame_from_a_plain_cpp_host(nranks, w, h):
    """tests/cpp/comm_test.cpp in gpu mode: ranks are threads of ONE C++ process sharing device 0 (no Python, no torch): device slabs,
    ipk_band_exchange_halo, the band kernel into its rows of the frame, ipk_band_gather_begin / ipk_comm_wait -- every rank's gathered
    frame equals the whole-frame launch bit for bit"""
    exe = os.path.join(ROOT, "tests", "cpp", "build", "comm_test")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")])
    r = subprocess.run([exe, str(nranks), str(w), str(h), "gpu"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "COMM_OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("nranks,w,h", [(2, 600, 150), (3, 1024, 258), (4, 300, 26), (4, 512, 64), (8, 256, 96)])
def test_rccl_transport_protocol_with_several_ranks(nranks, w, h):
    """The library's RCCL transport with 2..8 ranks on this single-GPU box: comm_test in rccl mode (ranks = threads sharing device 0) with
    the TEST DOUBLE tests/cpp/mock_rccl.cpp first on LD_LIBRARY_PATH -- the real RCCL refuses two ranks on one device.  The double keeps
    NCCL's group / FIFO-matching / in-place all-gather semantics and fails on a count mismatch, so what is verified here is ipk_comm.cpp's
    side: peers, offsets, counts and stream order of the grouped ncclSend/ncclRecv halo exchange, the in-place ncclAllGather (equal bands:
    4 x 512 x 64) and the ragged and rooted gathers, ipk_comm_selftest included.  The real RCCL is driven by the single-rank test below."""
    exe = os.path.join(ROOT, "tests", "cpp", "build", "comm_test")
    mock = os.path.join(ROOT, "tests", "cpp", "build", "mockrccl")
    if not os.path.exists(exe) or not os.path.exists(os.path.join(mock, "librccl.so.1")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")])
    env = dict(os.environ, LD_LIBRARY_PATH=mock + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe, str(nranks), str(w), str(h), "rccl"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "COMM_OK" in r.stdout and "rccl" in r.stdout, r.stdout + r.stderr


def test_bench_band_child_on_the_real_rccl_with_one_rank():
    """bench.py's banded single-frame leg (the child process `bench.py --gpus N` starts per rank on the library's RCCL transport), here with the
    one rank this box allows: real ncclCommInitRank with an id whose root lives in THIS process, ipk_comm_selftest, halo exchange, band kernel,
    gather -- and the frame compared with one whole-frame launch"""
    import ctypes as C
    from imagepipe_amd import _lib
    idb = C.create_string_buffer(_lib.COMM_ID_BYTES)
    _lib.check(_lib.load().ipk_comm_unique_id(idb), "ipk_comm_unique_id")
    r = subprocess.run([sys.executable, "bench.py", "--band-child", "0", "1", "0", idb.raw.hex(), "--width", "2048", "--height", "1024", "--steps", "4"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("BAND_CHILD ")]
    assert r.returncode == 0 and line, (r.stdout + r.stderr)[-3000:]
    d = json.loads(line[-1][len("BAND_CHILD "):])
    assert d["gathered_frame_bit_identical_to_one_launch"] is True and d["ranks"] == 1 and d["value"] > 0


def test_rccl_transport_single_rank():
    """the RCCL transport end to end on the one GPU of this box: ncclGetUniqueId, ncclCommInitRank (one rank), a self ncclSend/ncclRecv
    group and an ncclAllGather on device buffers (ipk_comm_selftest), the band entry points as no-ops"""
    code = ("import torch, ctypes as C, imagepipe_amd as ipa; from imagepipe_amd import _lib; ipa.init(0); L=_lib.load();"
            "idb=C.create_string_buffer(128); _lib.check(L.ipk_comm_unique_id(idb),'id'); h=C.c_void_p();"
            "_lib.check(L.ipk_comm_init_rccl(idb.raw,0,1,C.byref(h)),'init'); _lib.check(L.ipk_comm_selftest(h),'selftest');"
            "r=C.c_int(); n=C.c_int(); t=C.c_int(); L.ipk_comm_info(h,C.byref(r),C.byref(n),C.byref(t)); assert (r.value,n.value,t.value)==(0,1,0);"
            "L.ipk_comm_free(h); print('RCCL_OK')")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout + r.stderr)[-3000:]


@pytest.mark.parametrize("n_frames,nproc", [(7, 2), (8, 4)])
def test_frame_batch_sharded_over_ranks(n_frames, nproc):
    """BASELINE.json configs[3] in small: a batch of frames, frame i -> rank i % N, no data-path collective; every frame equals the oracle"""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
                        "--master-port", str(29550 + nproc), os.path.join("tests", "helpers", "batch_worker.py"), str(n_frames), "200", "600"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "BATCH_OK" in r.stdout, (r.stdout + r.stderr)[-3000:]
