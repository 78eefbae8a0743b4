"""Multi-process CPU tests of the N > 1 path of bench.py: the helpers of chronoclust_amd/multi.py over world_size-2 gloo
(torch.distributed) and over the torch-free host group bench.py itself uses (chronoclust_amd/rendezvous.py, worlds of
2 and 4): rank info, barrier, max-over-ranks time, the channel of the 128-byte RCCL id, the all-ranks-equal check of
the state digests, the partition arithmetic of the exact multi-GPU path."""
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


def _host_group_worker(rank, world, rdzv_file, out):
    """What bench.py does between its legs, on the group it uses: no torch in this process."""
    import sys
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), CHRONOCLUST_RDZV_FILE=rdzv_file)
    from chronoclust_amd import _lib, rendezvous
    g = rendezvous.from_env(timeout=60.0)
    try:
        assert (g.get_rank(), g.get_world_size()) == (rank, world) and multi.rank_info() == (rank, world, rank)
        g.barrier()
        assert multi.max_over_ranks(1.0 + rank, g) == float(world)
        payload = bytes(range(128)) if rank == 0 else None
        assert multi.broadcast_bytes(payload, g) == bytes(range(128))
        assert multi.all_ranks_equal("same-digest", g) and not multi.all_ranks_equal("digest-of-rank-%d" % rank, g)
        assert g.all_equal(b"ok") and (world == 1 or not g.all_equal(b"failed:%d" % rank))
        parts = g.all_gather_bytes(b"rank%d" % rank * (rank + 1))  # ragged payloads
        assert parts == [b"rank%d" % r * (r + 1) for r in range(world)]
        # the partition of the exact multi-GPU path, exchanged through the group
        mine = [_lib.shard_rows(n, world, rank, unit) for unit in (1, 64) for n in (0, 1, 63, 65, 5000, 123_457)]
        box = [None] * world
        g.all_gather_object(box, mine)
        for i in range(len(mine)):
            blocks = [b[i] for b in box]
            assert blocks[0][0] == 0 and all(blocks[r][1] == blocks[r + 1][0] for r in range(world - 1))
        out[rank] = ("torch" in sys.modules, multi.whole_job_rate(1000, 3, world, multi.max_over_ranks(1.0 + rank, g)))
    finally:
        g.close()


@pytest.mark.parametrize("world", [2, 4])
def test_host_group_without_torch(world, tmp_path):
    import multiprocessing
    ctx = multiprocessing.get_context("spawn")
    rdzv = str(tmp_path / "rdzv")
    with open(rdzv, "w") as f:
        f.write("1 stale-token-of-an-earlier-job\n")  # a stale file must not confuse the joiners
    with ctx.Manager() as mgr:
        out = mgr.dict()
        procs = [ctx.Process(target=_host_group_worker, args=(r, world, rdzv, out)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
        assert all(p.exitcode == 0 for p in procs)
        assert all(out[r][1] == pytest.approx(world * 1000 * 3 / float(world)) for r in range(world))
    assert not os.path.exists(rdzv)  # rank 0 removes the rendezvous file


def test_host_group_peer_gone_raises_instead_of_hanging(tmp_path):
    """A rank that never shows up: the others get a TimeoutError after the deadline."""
    from chronoclust_amd import rendezvous
    with pytest.raises(TimeoutError):
        rendezvous.HostGroup(0, 2, rdzv_file=str(tmp_path / "r"), timeout=0.5)
    with pytest.raises(TimeoutError):
        rendezvous.HostGroup(1, 2, rdzv_file=str(tmp_path / "nobody"), timeout=0.5)


def test_default_rendezvous_file_is_the_same_for_every_rank_of_a_job(monkeypatch):
    """The file rank 0 publishes its port in must have the same name in every rank, whatever started them: with a
    MASTER_PORT (every launcher of the contract sets one) the name does not involve the parent's pid - ranks behind a
    wrapper script have different parents -, two jobs of one user differ in the port, CHRONOCLUST_RDZV_FILE overrides."""
    from chronoclust_amd import rendezvous
    monkeypatch.delenv("CHRONOCLUST_RDZV_FILE", raising=False)
    monkeypatch.setenv("MASTER_PORT", "29517")
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "job7")
    a = rendezvous.default_rdzv_file()
    assert str(os.getppid()) not in os.path.basename(a).split("_") and "29517" in a and "job7" in a
    monkeypatch.setattr(os, "getppid", lambda: 424242)
    assert rendezvous.default_rdzv_file() == a
    monkeypatch.setenv("MASTER_PORT", "29518")
    assert rendezvous.default_rdzv_file() != a
    monkeypatch.delenv("MASTER_PORT")
    assert "424242" in rendezvous.default_rdzv_file()  # no port: the launcher's pid tells jobs apart
    monkeypatch.setenv("CHRONOCLUST_RDZV_FILE", "/tmp/explicit_rdzv")
    assert rendezvous.default_rdzv_file() == "/tmp/explicit_rdzv"


def test_bench_imports_no_torch():
    """north_star: no PyTorch on the path - bench.py and the package import neither torch nor the oracle at import time."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import sys; sys.path.insert(0, %r); import bench, chronoclust_amd, chronoclust_amd.multi, " \
           "chronoclust_amd.rendezvous, chronoclust_amd.app; " \
           "assert 'torch' not in sys.modules and 'oracle' not in sys.modules and 'oracle.oracle' not in sys.modules" % root
    subprocess.run([sys.executable, "-c", code], check=True, timeout=120)


def test_bench_command_line_parses_without_a_gpu():
    """`python bench.py --help` (argument surface of the driver contract: --gpus / --steps / --warmup) needs neither a
    GPU nor torch."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--help"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--no-one-stream", "--no-relaxed", "--relaxed-minibatch",
                 "--no-c2-legs", "--only-leg"):
        assert flag in out.stdout


_GUARD_SCRIPT = """
import os, signal, sys
sys.path.insert(0, %r)
import bench
bench.DETAIL_FILE = sys.argv[2]
fd = os.dup(1)
os.dup2(2, 1)
g = bench.LineGuard(fd)
print("noise on fd 1 from a native library")
g.provisional({"metric": "m", "value": 1.0})
mode = sys.argv[1]
if mode == "final":
    g.final({"metric": "m", "value": 1.0, "strong_scaling": {"one_stream_exact": {"value": 2.0, "n_gpus": 1}}})
elif mode == "killed":
    os.kill(os.getpid(), signal.SIGKILL)
elif mode == "terminated":
    os.killpg(os.getpgid(0), signal.SIGTERM)  # the launcher ends the whole group: the keeper still prints
"""


def test_bench_line_survives_the_death_of_the_process(tmp_path):
    """bench.py's stdout line is held by a keeper process: one line in every case - the complete one after a normal
    end, the headline marked incomplete when the process is killed after the headline was measured."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mode in ("final", "killed", "terminated"):
        p = subprocess.Popen([sys.executable, "-c", _GUARD_SCRIPT % root, mode, str(tmp_path / "detail.json")], stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, text=True, start_new_session=True)
        out, _ = p.communicate(timeout=60)
        lines = [x for x in out.splitlines() if x.strip()]
        assert len(lines) == 1, (mode, out)
        obj = json.loads(lines[0])
        assert obj["value"] == 1.0
        assert ("incomplete" in obj) == (mode != "final")
        assert ("one_stream_exact" in obj.get("strong_scaling", {})) == (mode == "final")


def _bench(*argv, env=None, timeout=120):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "CHRONOCLUST_RDZV_FILE")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + list(argv), capture_output=True, text=True,
                          timeout=timeout, env=e)


@pytest.mark.parametrize("n", [2, 4])
def test_bench_starts_its_own_ranks(n):
    """`python bench.py --gpus N` without a launcher (the form the driver uses at N = 1): the process itself starts N fresh
    ranks - before anything touched the GPU - and relays rank 0's ONE line.  --dry-launch: the ranks only meet."""
    import json
    out = _bench("--gpus", str(n), "--dry-launch")
    assert out.returncode == 0, out.stderr
    lines = [x for x in out.stdout.splitlines() if x.strip()]
    assert len(lines) == 1, out.stdout
    obj = json.loads(lines[0])
    assert obj["dry_launch"] is True and obj["n_gpus"] == n and obj["launcher"] == "bench.py"
    assert [r["rank"] for r in obj["ranks"]] == list(range(n)) == [r["local_rank"] for r in obj["ranks"]]
    assert len({r["pid"] for r in obj["ranks"]}) == n  # n processes, none of them this one's child re-executed in place


def test_bench_under_an_external_launcher_does_not_spawn():
    """With RANK / WORLD_SIZE in the environment (torch.distributed.run, srun, ...) bench.py is one rank of that job."""
    import json
    out = _bench("--gpus", "1", "--dry-launch", env=dict(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"))
    assert out.returncode == 0, out.stderr
    obj = json.loads(out.stdout.strip())
    assert obj["n_gpus"] == 1 and obj["launcher"] == "external" and len(obj["ranks"]) == 1


def test_bench_launcher_reports_a_failed_rank():
    """A rank that dies takes the job's exit status with it (here: every rank fails at once on an unknown flag)."""
    out = _bench("--gpus", "2", "--dry-launch", "--no-such-flag")
    assert out.returncode != 0
