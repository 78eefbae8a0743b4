"""world_size-2 gloo tests (CPU) of the N > 1 path of bench.py: replicas only, one stream per rank, whole-job
rate = units of all ranks / max-over-ranks time (chronoclust_amd/multi.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from chronoclust_amd import multi


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        r, w, lr = multi.rank_info()
        assert (r, w, lr) == (rank, world, rank)
        import bench
        # each rank has its own stream: different seeds give different data of the same shape
        X = bench.make_blobs(multi.stream_seed(42, rank), 2000, 5, 10)
        t = torch.tensor([float(X.sum())], dtype=torch.float64)
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        assert len({float(g.item()) for g in gathered}) == world
        dist.barrier()
        elapsed = 1.0 + rank  # rank 1 is the slow one
        worst = multi.max_over_ranks(elapsed, dist)
        assert worst == float(world)
        out[rank] = multi.whole_job_rate(1000, 3, world, worst)
    finally:
        dist.destroy_process_group()


def test_two_rank_replicas_gloo():
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        assert out[0] == out[1] == pytest.approx(2 * 1000 * 3 / 2.0)


def test_single_process_defaults():
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    assert multi.rank_info() == (0, 1, 0)
    assert multi.max_over_ranks(1.5, None) == 1.5


def _partition_worker(rank, world, port, out):
    """Each rank asks the library for ITS block of the three partitions of the exact multi-GPU path (cc_shard_rows:
    the arithmetic k_scan, cc_offline and cc_assoc_argmin use) and the ranks exchange them over gloo."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        from chronoclust_amd import _lib
        cases = [(n, unit) for unit in (1, 64) for n in (0, 1, 2, 63, 64, 65, 1000, 5000, 50_000, 123_457)]
        mine = torch.tensor([_lib.shard_rows(n, world, rank, unit) for n, unit in cases], dtype=torch.int64)
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        for i, (n, unit) in enumerate(cases):
            blocks = [tuple(int(v) for v in g[i]) for g in gathered]
            # the blocks tile [0, n) in rank order, every block but the last non-empty one is a whole share
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[r][1] == blocks[r + 1][0] for r in range(world - 1))
            share = max(hi - lo for lo, hi in blocks)
            assert share % unit == 0 or share == n
            assert all(hi - lo == share for lo, hi in blocks if hi < n)
        out[rank] = [tuple(int(v) for v in mine[6])]  # n = 1000, unit 1
    finally:
        dist.destroy_process_group()


def test_two_rank_partition_arithmetic_gloo():
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_partition_worker, args=(world, port, out), nprocs=world, join=True)
        assert out[0] == [(0, 500)] and out[1] == [(500, 1000)]


def _id_worker(rank, world, port, out):
    """The channel that carries the 128-byte RCCL id from rank 0 to the other ranks (multi.broadcast_bytes), and the
    all-ranks-equal check bench.py applies to the digests of the ranks' final states, over gloo."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        payload = bytes(range(128)) if rank == 0 else None
        got = multi.broadcast_bytes(payload, dist)
        assert got == bytes(range(128)) and len(got) == 128
        assert multi.all_ranks_equal("same-digest", dist)
        assert not multi.all_ranks_equal("digest-of-rank-%d" % rank, dist)
        out[rank] = multi.one_stream_rate(1000, 2, 0.5)
    finally:
        dist.destroy_process_group()


def test_two_rank_id_broadcast_and_digest_check_gloo():
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_id_worker, args=(world, port, out), nprocs=world, join=True)
        assert out[0] == out[1] == 4000.0


def test_bench_command_line_parses_without_a_gpu():
    """`python bench.py --help` (argument surface of the driver contract: --gpus / --steps / --warmup) needs neither a
    GPU nor torch."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--help"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--no-one-stream", "--no-relaxed", "--relaxed-minibatch"):
        assert flag in out.stdout


_GUARD_SCRIPT = """
import os, signal, sys
sys.path.insert(0, %r)
import bench
fd = os.dup(1)
os.dup2(2, 1)
g = bench.LineGuard(fd)
print("noise on fd 1 from a native library")
g.provisional({"metric": "m", "value": 1.0})
mode = sys.argv[1]
if mode == "final":
    g.final({"metric": "m", "value": 1.0, "one_stream_exact": {"value": 2.0}})
elif mode == "killed":
    os.kill(os.getpid(), signal.SIGKILL)
elif mode == "terminated":
    os.killpg(os.getpgid(0), signal.SIGTERM)  # the launcher ends the whole group: the keeper still prints
"""


def test_bench_line_survives_the_death_of_the_process():
    """bench.py's stdout line is held by a keeper process: one line in every case - the complete one after a normal
    end, the headline marked incomplete when the process is killed after the headline was measured."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mode in ("final", "killed", "terminated"):
        p = subprocess.Popen([sys.executable, "-c", _GUARD_SCRIPT % root, mode], stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, text=True, start_new_session=True)
        out, _ = p.communicate(timeout=60)
        lines = [x for x in out.splitlines() if x.strip()]
        assert len(lines) == 1, (mode, out)
        obj = json.loads(lines[0])
        assert obj["value"] == 1.0
        assert ("incomplete" in obj) == (mode != "final")
        assert ("one_stream_exact" in obj) == (mode == "final")
