"""World-size-2 gloo test (CPU) of the multi-GPU path: contiguous batch sharding, independent per-rank sampling, gather.
The sample function here is a deterministic per-row stand-in (the HIP sampler needs a GPU); what is under test is that the
N > 1 path partitions rows exactly once, needs no data-path collective, and reassembles results in batch order."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from drmnet_amd.dist import gather_results, sample_sharded, shard_indices, shard_rows


def test_shard_rows_partition():
    for n in (0, 1, 5, 32, 2048, 2049):
        for w in (1, 2, 3, 8):
            spans = [shard_rows(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_rows(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_sampler(x):
    # per-row "chain": depends only on the row itself (like GroupNorm/attention/convergence in the real path)
    Lr0 = x * 2 + x.flatten(1).sum(1)[:, None, None, None]
    zK = x.flatten(1)[:, :6].clone()
    K = (x.flatten(1).abs().sum(1) * 10).to(torch.int32)
    return Lr0, zK, K


def test_strided_sharding_deals_rows_round_robin():
    for n in (0, 1, 7, 32):
        for w in (1, 2, 3, 8):
            owned = [shard_indices(n, w, r, strided=True) for r in range(w)]
            assert sorted(torch.cat(owned).tolist()) == list(range(n))
            assert all(ix.tolist() == list(range(r, n, w)) for r, ix in enumerate(owned))


def _worker(rank, world, port, n):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        x = torch.randn((n, 3, 4, 8), generator=g)  # same full batch on every rank
        calls = []

        def fn(xs):
            calls.append(xs.shape[0])
            return _fake_sampler(xs)

        Lr0, zK, K = sample_sharded(fn, [x])
        ref = _fake_sampler(x)
        a, b = shard_rows(n, world, rank)
        assert calls == [b - a]  # each rank sampled only its own rows, once
        for got, want in zip((Lr0, zK, K), ref):
            assert got.shape == want.shape and torch.equal(got, want)
        # ragged gather in isolation
        mine = torch.full((b - a, 2), float(rank))
        (full,) = gather_results([mine], n)
        assert full.shape[0] == n and torch.equal(full[a:b], mine)
        # strided dealing (the early-exit layout): same results, rows rank, rank + world, ...
        calls.clear()
        got = sample_sharded(fn, [x], strided=True)
        assert calls == [len(range(rank, n, world))]
        for g_, want in zip(got, ref):
            assert torch.equal(g_, want)
        # bench.py's own aggregation: the job's time is the SLOWEST rank's, its work the SUM over ranks
        import bench

        t, units = bench.rank_aggregate(0.5 + 0.25 * rank, 32.0 * 10 * (rank + 1), dist, None)
        assert t == 0.5 + 0.25 * (world - 1) and units == 32.0 * 10 * sum(range(1, world + 1))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [5, 8])
def test_two_rank_gloo_sharded_sampling(n):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n), nprocs=2, join=True)


def test_bench_aggregate_single_process_and_self_launch_refusal(capsys):
    """Single process: aggregation is the identity.  `--gpus 2` with no launcher on a box without two GPUs must refuse loudly
    (exit code 2) before spawning anything -- the check runs on torch.cuda.device_count(), which does not initialise the GPU."""
    import argparse

    import bench

    assert bench.rank_aggregate(0.25, 320.0) == (0.25, 320.0)
    if torch.cuda.device_count() < 2:
        assert bench.self_launch(argparse.Namespace(gpus=2)) == 2
        assert "GPU(s) are visible" in capsys.readouterr().err


def test_a_failing_rank_ends_the_job_with_rc_and_no_result_line(tmp_path, capfd):
    """under bench.py's own launcher a rank that dies must give rc != 0 and NO JSON line -- rank 0 prints the line only
    after the last barrier, and the launcher terminates the ranks left parked there instead of waiting out the collective's timeout.  The
    children here are stand-ins with the launcher's environment contract (RANK / WORLD_SIZE / MASTER_*): rank 1 fails, rank 0 would print a
    result line after its 'barrier'."""
    import argparse
    import sys
    import time

    import bench

    child = tmp_path / "rank.py"
    child.write_text(
        "import os, sys, time\n"
        "rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
        "assert world == 2 and os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0\n"
        "mode = sys.argv[1]\n"
        "if mode == 'fail' and rank == 1:\n"
        "    sys.exit(3)\n"
        "time.sleep(30 if mode == 'fail' else 0.2)  # rank 0 parked at the barrier the dead rank never reaches\n"
        "if rank == 0:\n"
        "    print('{\"metric\": \"x\"}', flush=True)\n")
    t0 = time.time()
    rc = bench.self_launch(argparse.Namespace(gpus=2), command=[sys.executable, str(child), "fail"], have=2, poll_s=0.05)
    out = capfd.readouterr()
    assert rc == 3 and time.time() - t0 < 15
    assert '"metric"' not in out.out and "no result line" in out.err
    rc = bench.self_launch(argparse.Namespace(gpus=2), command=[sys.executable, str(child), "ok"], have=2, poll_s=0.05)
    out = capfd.readouterr()
    assert rc == 0 and out.out.count('"metric"') == 1


def test_live_traffic_falls_back_quietly(monkeypatch):
    """bench.py measures roofline.traffic with two rocprofv3 --pmc child runs of its own command; where that cannot be done -- no rocprofv3, the
    run is itself under the profiler, a child fails -- it returns {} and the committed (hash-gated) profile is imported instead."""
    import argparse
    import shutil
    import subprocess

    import bench

    args = argparse.Namespace(precision="auto", batch=32, height=128, width=256)
    dom = "void drm::conv_split2_kernel<9, 16, 16, 4, 2, 2, 2, 2, 3, 2, false, false, false>"
    # under a profiler: no nested rocprofv3
    monkeypatch.setenv("ROCPROFILER_REGISTER_FORCE_LOAD", "1")
    assert bench.under_profiler() and bench.live_traffic(args, dom) == {}
    monkeypatch.delenv("ROCPROFILER_REGISTER_FORCE_LOAD")
    for k in list(os.environ):
        if k.startswith(("ROCPROF", "ROCP_")):
            monkeypatch.delenv(k)
    monkeypatch.setenv("LD_PRELOAD", "")
    assert not bench.under_profiler()
    # no rocprofv3 on the box
    monkeypatch.setattr(shutil, "which", lambda name: None)
    real_exists = os.path.exists
    monkeypatch.setattr(os.path, "exists", lambda p: False if p.endswith("rocprofv3") else real_exists(p))
    assert bench.live_traffic(args, dom) == {}
    # a child run that fails (no GPU here): rc != 0 -> {}
    monkeypatch.setattr(os.path, "exists", real_exists)
    monkeypatch.setattr(shutil, "which", lambda name: "/bin/false" if name == "rocprofv3" else None)
    calls = []
    real_popen = subprocess.Popen

    def fake_popen(cmd, **kw):
        calls.append((cmd, kw))
        return real_popen(["/bin/false"], **{k: v for k, v in kw.items() if k in ("stdout", "stderr", "start_new_session")})

    monkeypatch.setattr(subprocess, "Popen", fake_popen)
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("MASTER_PORT", "12345")
    monkeypatch.setenv("DRM_BENCH_DIST", "1")
    assert bench.live_traffic(args, dom) == {}
    cmd, kw = calls[0]
    assert "--pmc" in cmd and "FETCH_SIZE" in cmd and "--no-live-traffic" in cmd and "--kernel-trace" in cmd
    assert not any(f in cmd for f in ("--sys-trace", "-s", "--runtime-trace", "-r", "--hip-trace"))  # (counters in their own pass: the pool's rule)
    # [r5, ADVICE r4] the child is a session of its own (a timeout kills rocprofv3 AND the python it started) and does not inherit the rank /
    # rendezvous variables of a distributed parent
    assert kw.get("start_new_session") is True
    assert not any(k in kw["env"] for k in ("RANK", "WORLD_SIZE", "MASTER_PORT", "DRM_BENCH_DIST"))


def test_pmc_child_kills_the_whole_process_group_on_timeout(monkeypatch, tmp_path):
    """A child that outlives its budget: rocprofv3 would be killed alone by subprocess.run(timeout=...) and leave the profiled python holding the GPU."""
    import argparse
    import shutil
    import subprocess
    import time

    import bench

    for k in list(os.environ):
        if k.startswith(("ROCPROF", "ROCP_")):
            monkeypatch.delenv(k)
    monkeypatch.setenv("LD_PRELOAD", "")
    marker = tmp_path / "grandchild.pid"
    script = tmp_path / "fake_rocprofv3"
    script.write_text(f"#!/bin/bash\n(sleep 30 & echo $! > {marker}; wait) \n")
    script.chmod(0o755)
    monkeypatch.setattr(shutil, "which", lambda name: str(script) if name == "rocprofv3" else None)
    args = argparse.Namespace(precision="f16mx", batch=32, height=128, width=256)
    t0 = time.time()
    assert bench.pmc_child(args, ["FETCH_SIZE"], timeout_s=1) is None
    assert time.time() - t0 < 10
    pid = int(marker.read_text())

    def running(p):  # (a killed process may linger as a zombie until its new parent reaps it)
        try:
            with open(f"/proc/{p}/stat") as f:
                return f.read().rsplit(")", 1)[1].split()[0] != "Z"
        except OSError:
            return False

    for _ in range(40):
        if not running(pid):
            break
        time.sleep(0.05)
    assert not running(pid)  # the grandchild went with the group
