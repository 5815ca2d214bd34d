"""N>1 path on CPU: world_size-2 gloo run of the batch-sharding / timing-aggregation logic
bench.py uses (no GPU work here — the per-image compute is covered by the -m gpu tests)."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

from chan_vese_amd import batch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_images_partitions_exactly():
    for B, R in [(64, 8), (64, 1), (7, 2), (5, 8), (0, 3)]:
        parts = [batch.shard_images(B, r, R) for r in range(R)]
        flat = [b for p in parts for b in p]
        assert flat == list(range(B))
    assert batch.shard_images(64, 3, 8) == list(range(24, 32))      # C5: images 8r .. 8r+7 on GPU r
    with pytest.raises(ValueError):
        batch.shard_images(4, 2, 2)


def test_aggregate_throughput():
    recs = [[8, 8 * 500 * 4096.0 * 4096, 1.0], [8, 8 * 500 * 4096.0 * 4096, 1.25]]
    assert batch.aggregate_throughput(recs, 1.25) == pytest.approx(16 * 500 * 4096 * 4096 / 1.25 / 1e6)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_gloo_gather_and_max(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent(f"""
        import sys, json
        sys.path.insert(0, {ROOT!r})
        from chan_vese_amd import batch
        dist, rank, world, local = batch.init_distributed(backend="gloo")
        mine = batch.shard_images(64, rank, world)
        batch.barrier(dist)
        elapsed = 1.0 + 0.5 * rank
        recs = batch.gather_records(dist, [len(mine), len(mine) * 1000.0, elapsed])
        emax = batch.max_over_ranks(dist, elapsed)
        batch.barrier(dist)
        if rank == 0:
            print(json.dumps(dict(world=world, mine=mine[:2], recs=recs, emax=emax,
                                  value=batch.aggregate_throughput(recs, emax))))
        dist.destroy_process_group()
    """))
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["world"] == 2 and d["mine"] == [0, 1]
    assert d["recs"] == [[32.0, 32000.0, 1.0], [32.0, 32000.0, 1.5]]
    assert d["emax"] == 1.5
    assert d["value"] == pytest.approx(64000.0 / 1.5 / 1e6)


def test_bench_gpus_flag_spawns_ranks_dry_run():
    """`python bench.py --gpus 2` started directly (no launcher) must start 2 ranks itself and report
    n_gpus == 2.  CHANVESE_BENCH_DRYRUN=1 skips the GPU work, so this covers the launch / barrier / gather /
    report path on CPU with gloo; the line is marked "dry-run" and carries no measurement."""
    import json
    env = dict(os.environ, CHANVESE_DIST_BACKEND="gloo", CHANVESE_BENCH_DRYRUN="1", OMP_NUM_THREADS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                       # ONE line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["ranks_in_group"] == 2 and d["data"] == "dry-run"
    assert d["config"]["images_per_gpu"] == 8 and d["config"]["images_total"] == 16   # C5 share per GPU
    assert len(d["config"]["per_rank_mpx_it_s"]) == 2
    # whole-job value = all pixel-iterations / slowest rank's time (rank 1 reports 1.25 s in the dry run)
    assert d["value"] == pytest.approx(2 * 8 * 3 * 4096.0 * 4096.0 / 1.25 / 1e6)
