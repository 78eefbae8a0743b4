#!/usr/bin/env python3
"""bench.py — ChronoClust hot path on MI355X: points clustered per second (20-dim), config C2 of BASELINE.json.

A step = one timepoint of the hot path over one batch of synthetic input that is already resident in HBM:
reset to an empty HDDStream, the exact per-point online phase over N points (cc_online_run) and the offline
PreDeCon phase (cc_offline) with the export of all clusters.

At N GPUs the headline `value` is N independent event streams of the C2 shape, one per rank (weak scaling, no
data-path collective: the online phase of one stream is a sequential chain over its points).  The same JSON line
also carries, as side legs (strong scaling: ONE stream, all N GPUs):
  one_stream_exact            the stress config's shape (d = 40, 50 000 microclusters), exact: snapshot scans split by
                              table rows, one RCCL all-gather of 64 B per window point, offline pair matrices split by
                              rows, with a check that every rank ended with the same bytes
  events_sharded_relaxed      the WNV shape (d = 14, 2 000 microclusters), EVENTS sharded over the GPUs with an RCCL
                              all-reduce of the CF deltas per super-step - relaxed semantics, reported with its agreement
                              with the exact path
  one_stream_exact_c2         the headline's own 20-dim shape (C2) as ONE stream on all GPUs, exact: at 5 000 x 20 the
                              library does not split the scan (a window's scan is shorter than the all-gather that would
                              follow it), so every rank repeats the work; the same leg with the split forced is beside it
  events_sharded_relaxed_c2   the C2 shape, relaxed event-sharded, with its agreement
Every leg carries `n_gpus`, `collective`, `rccl_ranks_seen` and a `roofline` object of its own scan kernel.

No torch anywhere: the timing bracket is cc_sync (HIP stream synchronise through the C-ABI), barrier / max over
ranks / the 128-byte RCCL id travel over chronoclust_amd.rendezvous (TCP on 127.0.0.1).

    python bench.py [--gpus N] [--steps K] [--warmup W]
        N > 1 and no launcher's RANK / WORLD_SIZE in the environment: THIS process starts the N ranks itself - fresh
        child processes, one per GPU (LOCAL_RANK = device), before anything here has touched HIP or RCCL -, relays
        rank 0's line and exits with the worst child's status (launch_ranks below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
        any launcher that sets RANK / WORLD_SIZE / LOCAL_RANK will do as well (only the launcher of torch is used)
    python bench.py --gpus N --dry-launch
        the ranks only meet (rendezvous, barrier) and rank 0 prints who came: the launch path without a GPU

Prints ONE JSON line (< 4 KB: compact_line) on rank 0 and writes the full record - per-kernel counters, notes, the legs'
own objects - to bench_detail.json beside this file (CHRONOCLUST_BENCH_DETAIL overrides the path; the line names it).
Exit status: 0 when the headline was measured and every leg returned (a leg that FAILED is named in `legs_failed`, its
error in `leg_errors`; --strict-legs makes that status 3); 3 when a leg was abandoned (a collective that never
completed) - the line is still printed.
"""
import argparse
import hashlib
import json
import os
import signal
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VALU_PEAK_TOPS = 39.3     # 78.6 TFLOP/s FP64 vector counts an FMA as 2: 39.3 T instruction-lanes/s
PMC_TRAFFIC_GLOB = os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")
PMC_BENCH_STEP_GLOB = os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_bench_step.json")  # tools/pmc_bench_step.py: counted on bench.py itself
PMC_VALU_GLOB = os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_valu_d%d%s.json")  # (% (d, "" | "_plain")), tools/pmc_valu_summary.py
PRUNED_CHAIN = ("k_seed", "k_seed_merge", "k_scan_a", "k_scan_p", "k_scan_p2", "k_prefix16", "k_scan_p3")  # kernels of a pruned snapshot scan as the PMC summaries name them
SCAN_SOURCES = ("cc_scan.h", "cc_common.h", "cc_div.h")  # what defines the snapshot-scan kernels the PMC files describe
REFERENCE_RATE_FILE = os.path.join(ROOT, "profiles", "reference_py_rate.json")
EXIT_LEG_FAILED = 3


def kernel_short(name):
    """rocprofv3 leaves some template instances mangled (_Z9k_scan_p3ILi20ELi4ELb0ELb0EEv...): the function's own name, marked as
    a template instance, so that prefix matches written for demangled names ("k_scan_p3<") keep working."""
    import re
    name = re.sub(r"^void ", "", str(name))
    m = re.match(r"_Z(\d+)", name)
    if m:
        n = int(m.group(1))
        return name[m.end():m.end() + n] + "<mangled>"
    return re.sub(r"\(.*", "", name)


def csrc_digest(names=None):
    """SHA-256 over the kernel sources (all of csrc/, or the named files): a PMC figure is only quoted for the kernels it
    was measured on."""
    m = hashlib.sha256()
    d = os.path.join(ROOT, "chronoclust_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip")) and (names is None or name in names):
            with open(os.path.join(d, name), "rb") as f:
                m.update(name.encode() + b"\0" + f.read())
    return m.hexdigest()


def scan_digest():
    """The sources of the snapshot-scan kernels alone - and the host function that sets their launch geometry (sub-range
    split, phase-B grouping, the row count from which phase A is a kernel of its own: launch_scan_dp in cc_api.hip; the
    instructions per (wave, row) depend on it) -: a change to the validation or offline kernels leaves the scan's
    counters valid."""
    m = hashlib.sha256(csrc_digest(SCAN_SOURCES).encode())
    try:
        with open(os.path.join(ROOT, "chronoclust_amd", "csrc", "cc_api.hip")) as f:
            text = f.read()
        a = text.index("void launch_scan_dp(")
        b = text.index("void launch_scan(", a)
        m.update(text[a:b].encode())
    except (OSError, ValueError):
        m.update(b"launch_scan_dp not found")
    return m.hexdigest()


def pmc_matches(pm):
    """A PMC summary counts for this build when it was measured on these scan sources (files of round 5 on carry
    `scan_sha256`; earlier ones only the digest of all of csrc/)."""
    if "scan_sha256" in pm:
        return pm["scan_sha256"] == scan_digest()
    return pm.get("csrc_sha256") == csrc_digest()


def newest_pmc(pattern):
    """(summary, file name) of the newest round's PMC file of this pattern that matches the build, else (None, why)."""
    import glob
    why = "no PMC file %s" % os.path.basename(pattern)
    for path in sorted(glob.glob(pattern), reverse=True):
        try:
            with open(path) as f:
                pm = json.load(f)
        except (OSError, ValueError):
            continue
        if pmc_matches(pm):
            return pm, os.path.basename(path)
        why = "profiles/%s was measured on other scan-kernel sources (digest differs)" % os.path.basename(path)
    return None, why


DETAIL_FILE = os.environ.get("CHRONOCLUST_BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json"))
LINE_LIMIT = 4096   # bytes of the ONE stdout line (tests/test_host_logic.py::test_bench_line_is_compact holds it)
LEG_NAMES = ("one_stream_exact", "events_sharded_relaxed", "one_stream_exact_c2", "events_sharded_relaxed_c2")


def _r(x, digits=6):
    """Floats of the line with 6 significant digits (the detail file keeps them all)."""
    if isinstance(x, float):
        return float("%.*g" % (digits, x))
    if isinstance(x, dict):
        return {k: _r(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, digits) for v in x]
    return x


def compact_line(full, detail_file=None):
    """The ONE stdout line: the contract's keys, a flat `roofline` and `cpu_baseline`, one small object per strong-scaling
    leg - below LINE_LIMIT bytes whatever the run did.  Everything else (per-kernel counters, notes, parts, the legs' own
    objects and rooflines) is in the detail file this line names."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data")
    line = {k: full.get(k) for k in keep}
    cfg = full.get("config") or {}
    line["config"] = {k: cfg.get(k) for k in ("workload", "points", "dim", "microclusters", "clusters", "streams") if k in cfg}
    line["config"]["workload"] = str(cfg.get("workload", ""))[:160]
    for k in ("online_only_points_per_s", "value_with_transfers", "filter_on_points_per_s"):
        if full.get(k) is not None:
            line[k] = full[k]
    rf = full.get("roofline")
    if rf:
        ex = (rf.get("executed") or {}).get("kernels") or {}
        dom = ex.get("k_scan_u") or ex.get("k_scan_p3") or ex.get("k_scan_p") or {}  # (by time the plain scan of the start-up windows dominates since round 6)
        hbm = rf.get("hbm") or {}
        line["roofline"] = {
            "bound": rf.get("bound"), "kernel": str(rf.get("kernel", ""))[:120], "achieved": rf.get("achieved"),
            "peak": rf.get("peak"), "unit": str(rf.get("unit", ""))[:60], "frac": rf.get("frac"),
            "step_frac": rf.get("step_frac"), "model_frac": (rf.get("model") or {}).get("frac"),
            "valu_busy": dom.get("valu_busy_fraction"), "effective_frac": (rf.get("effective") or {}).get("frac"),
            "hbm_frac": hbm.get("frac"), "hbm_achieved_gbs": hbm.get("achieved"), "hbm_peak_gbs": hbm.get("peak"),
            "traffic": rf.get("traffic"), "algorithmic_bytes": hbm.get("algorithmic_bytes_per_launch"),
            "launches": rf.get("launches"), "avg_launch_us": rf.get("avg_launch_us"),
            "pmc_source": rf.get("pmc_source"), "derived": rf.get("derived")}
    cb = full.get("cpu_baseline")
    if cb:
        ref = cb.get("reference_py") or {}
        allc = cb.get("all_cores") or {}
        line["cpu_baseline"] = {"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"),
                                "kind": cb.get("kind"), "sample": str(cb.get("sample", ""))[:140],
                                "labels_match_gpu_prefix": cb.get("labels_match_gpu_prefix"),
                                "all_cores_value": allc.get("value"), "all_cores": allc.get("cores"),
                                "steady_value": (cb.get("steady") or {}).get("value"), "steady_rows": (cb.get("steady") or {}).get("rows"),
                                "reference_py": ref.get("value")}
    ss = full.get("strong_scaling")
    if ss:
        legs = {}
        for name in LEG_NAMES:
            leg = ss.get(name)
            if leg:
                legs[name] = {k: leg.get(k) for k in ("value", "ms_per_step", "n_gpus", "rccl_ranks_seen",
                                                     "all_ranks_bit_identical", "semantics", "sharded_windows_per_step",
                                                     "agreement_with_exact_by_cluster", "predicted_value") if leg.get(k) is not None}
                st = leg.get("split_threshold")
                if st:
                    legs[name]["split_threshold"] = st.get("row_dims_plain_scan")
        line["strong_scaling"] = dict(legs, unit="points/s of ONE stream on all n_gpus ranks")
    for name in LEG_NAMES:
        if isinstance(full.get(name), dict) and "error" in full[name]:
            line.setdefault("leg_errors", {})[name] = str(full[name]["error"])[:120]
    line["legs_failed"] = list(full.get("legs_failed", []))
    if full.get("incomplete"):
        line["incomplete"] = str(full["incomplete"])[:100]
    line["detail_file"] = detail_file
    line = _r(line)
    text = json.dumps(line, separators=(",", ":"))
    if len(text) >= LINE_LIMIT:  # (cannot happen with the bounded fields above; the contract's keys survive whatever does)
        for k in ("strong_scaling", "leg_errors", "cpu_baseline"):
            if len(text) >= LINE_LIMIT and k in line:
                line[k] = "see detail_file"
                text = json.dumps(line, separators=(",", ":"))
    return text


def write_detail(full, path=None):
    """The full record beside the line (atomically: a reader never sees half a file).  Returns the path, or None when the
    directory cannot be written (the line then says so by carrying no detail_file)."""
    path = path or DETAIL_FILE
    try:
        tmp = "%s.%d.tmp" % (path, os.getpid())
        with open(tmp, "w") as f:
            json.dump(full, f, indent=1)
            f.write("\n")
        os.replace(tmp, path)
        return os.path.relpath(path, ROOT) if path.startswith(ROOT + os.sep) else path
    except OSError as e:
        sys.stderr.write("[bench] detail file %s not written: %s\n" % (path, e))
        return None


class LineGuard:
    """Keeper of the ONE line on stdout.  A small child process, forked before anything touches the GPU, holds the
    real stdout and reads what this process sends it: the headline as soon as it is measured (provisional), the
    complete line at the end (final).  When the pipe closes it prints the last thing it was given - so the headline
    also reaches stdout when a later leg takes the process down (a fault inside a collective, the launcher ending
    the rank after a peer died), marked as incomplete.  The child never uses the GPU, the oracle or torch."""

    def __init__(self, out_fd):
        r, w = os.pipe()
        # a group-wide signal must not end the keeper before the line is out: blocked across the fork, ignored in the child
        sigs = (signal.SIGTERM, signal.SIGINT, signal.SIGHUP)
        before = signal.pthread_sigmask(signal.SIG_BLOCK, sigs)
        pid = os.fork()
        if pid == 0:
            status = 0
            try:
                os.close(w)
                for sig in sigs:
                    signal.signal(sig, signal.SIG_IGN)
                signal.pthread_sigmask(signal.SIG_SETMASK, before)
                buf = b""
                while True:
                    chunk = os.read(r, 1 << 16)
                    if not chunk:
                        break
                    buf += chunk
                lines = [x for x in buf.split(b"\n") if x]
                if lines:
                    kind, payload = lines[-1][:1], lines[-1][1:]
                    if kind != b"F":
                        obj = json.loads(payload)
                        obj["incomplete"] = "the process ended before the remaining legs of the run had finished"
                        payload = json.dumps(obj, separators=(",", ":")).encode()
                    data = payload + b"\n"
                    while data:  # (a short write - a non-blocking or full pipe - must not cut the line)
                        try:
                            data = data[os.write(out_fd, data):]
                        except BlockingIOError:
                            time.sleep(0.01)
            except BaseException:  # noqa: BLE001 - nothing to report to
                status = 1
            os._exit(status)
        signal.pthread_sigmask(signal.SIG_SETMASK, before)
        os.close(r)
        self._w, self._pid = w, pid

    def _send(self, kind, obj):
        """`obj`: the full record; the keeper gets the compact line, the detail file gets the rest."""
        data = kind + compact_line(obj, write_detail(obj)).encode() + b"\n"
        while data:
            data = data[os.write(self._w, data):]

    def provisional(self, obj):
        self._send(b"P", obj)

    def final(self, obj):
        self._send(b"F", obj)
        self.close()

    def close(self):
        """Ends the keeper (it prints what it holds) and waits until the line is out."""
        if self._w is not None:
            os.close(self._w)
            self._w = None
            os.waitpid(self._pid, 0)


def launch_ranks(n_ranks, argv):
    """`bench.py --gpus N` started without a launcher: N ranks as fresh child processes of this one, which has not
    loaded the HIP library and never will (a process that has touched the GPU must not be re-executed, and does not
    have to be).  Every child gets RANK / LOCAL_RANK / WORLD_SIZE, a rendezvous file in a private directory and the
    same command line; all devices stay visible to every rank (RCCL wants to see its peers) and a rank picks device
    LOCAL_RANK.  Rank 0 inherits stdout - its ONE line is this job's line -, the other ranks' stdout goes to stderr.
    Returns the exit status: 0 when every rank left with 0, else the worst one (a signal counts as 128 + number).
    A rank that fails takes the job down: the others get 60 s to notice through their own bounded waits, then SIGTERM."""
    import shutil
    import subprocess
    import tempfile
    tmp = tempfile.mkdtemp(prefix="chronoclust_bench_")
    env = dict(os.environ)
    env["WORLD_SIZE"] = str(n_ranks)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env["CHRONOCLUST_RDZV_FILE"] = os.path.join(tmp, "rendezvous")
    env["CHRONOCLUST_BENCH_LAUNCHER"] = "bench.py"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (dmabuf IPC: what RCCL needs between processes on this host driver)
    cmd = [sys.executable, os.path.abspath(__file__)] + list(argv)
    procs = []
    try:
        for r in range(n_ranks):
            procs.append(subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                                          stdout=None if r == 0 else sys.stderr))

        def forward(sig, _frame):
            for p in procs:
                if p.poll() is None:
                    p.send_signal(sig)

        for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
            signal.signal(sig, forward)
        first_failure = None
        while any(p.poll() is None for p in procs):
            time.sleep(0.05)
            if first_failure is None and any(p.poll() not in (None, 0) for p in procs):
                first_failure = time.monotonic()
            if first_failure is not None and time.monotonic() - first_failure > 60.0:
                for p in procs:
                    if p.poll() is None:
                        p.terminate()
                grace = time.monotonic() + 10.0
                while any(p.poll() is None for p in procs) and time.monotonic() < grace:
                    time.sleep(0.05)
                for p in procs:
                    if p.poll() is None:
                        p.kill()
        codes = [p.wait() for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        shutil.rmtree(tmp, ignore_errors=True)
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        sys.stderr.write("[bench] ranks that did not exit cleanly: %s\n" % ", ".join("rank %d -> %d" % rc for rc in bad))
    return max((c if c >= 0 else 128 - c) for c in codes)


def make_blobs(seed, n, d, g, sigma=0.01):
    """BASELINE.md section 4 generator (same recipe as tests/scenarios.py)."""
    rng = np.random.default_rng(seed)
    centres = rng.uniform(0.1, 0.9, (g, d))
    lab = rng.integers(0, g, n)
    return np.ascontiguousarray(np.clip(centres[lab] + rng.normal(0.0, sigma, (n, d)), 0.0, 1.0))


def blob_config(n, beta=0.5, promote_after=10):
    return dict(beta=beta, delta=0.05, epsilon=0.05, k=4, mu=promote_after / (beta * n), pi=0, omicron=0.0,
                upsilon=6.5, **{"lambda": 0.5})


def set_params(h, cfg, n, d):
    """Derived parameters with the reference's expressions (hddstream.py:45-52, 107-126)."""
    eps = float(cfg["epsilon"])
    ups = float(cfg["upsilon"]) * eps
    delta = float(cfg["delta"])
    pi = d if float(cfg["pi"]) <= 0 else round(float(cfg["pi"]))
    h.set_params(eps ** 2, delta ** 2, float(cfg["k"]), float(cfg["beta"]), float(cfg["mu"]) * n,
                 cfg["omicron"] * 0, ups, ups ** 2, delta, pi)


def cpu_baseline(cfg, X, sample, cores, gpu_uid, gpu_tables=None, steady_sample=30000):
    """The oracle (oracle/chrono_oracle.c, a scalar C port of the reference's loop) on a prefix of the same stream:
    one thread, then one independent copy of the same work on every host core (the algorithm is a sequential chain,
    so cores can only be used by independent streams - as the GPU replicas do)."""
    from oracle import oracle as O
    m = min(sample, X.shape[0])
    o = O.OracleHDDStream(cfg)
    o.set_dataset_dependent_parameters(X)  # thresholds of the full timepoint (mu = mu_cfg * N), like the GPU run
    t1 = time.perf_counter()
    o.online_microcluster_maintenance(X[:m], 0, reset_param=False, offline=False)
    one = time.perf_counter() - t1
    rows = o.table(O.PCORE)["id"].shape[0] + o.table(O.OUTLIER)["id"].shape[0]
    out = {"value": m / one, "unit": "points/s", "cores": 1, "kind": "port",
           "sample": "first %d points of the same stream (microclusters grow 0 -> %d), online phase only, "
                     "oracle/chrono_oracle.c single thread" % (m, rows),
           "labels_match_gpu_prefix": None if gpu_uid is None else bool(np.array_equal(o.labels_uid, gpu_uid[:m]))}
    if cores > 1:
        workers = [O.OracleHDDStream(cfg) for _ in range(cores)]
        for w in workers:
            w.set_dataset_dependent_parameters(X)
        threads = [threading.Thread(target=w.online_microcluster_maintenance, args=(X[:m], 0),
                                    kwargs=dict(reset_param=False, offline=False)) for w in workers]
        t2 = time.perf_counter()
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        allc = time.perf_counter() - t2
        out["all_cores"] = {"value": cores * m / allc, "unit": "points/s", "cores": cores,
                            "note": "%d independent copies of the same sample, one thread per usable host core, "
                                    "ctypes threads; the sequential chain of one stream cannot use more than one "
                                    "core" % cores}
    if gpu_tables is not None and steady_sample > 0:
        # The C port in the STEADY state - the table the stream ends on (5 000 pcore microclusters at C2), every point joins
        # an existing microcluster: what the GPU path's steady state (tools/steady.py) stands beside.  The final tables of
        # the GPU run are injected (co_inject_mc, list order), the same stream's first points run again on them (same
        # daystamp: no decay), one thread.
        s_o = O.OracleHDDStream(cfg)
        s_o.set_dataset_dependent_parameters(X)
        s_o._push_params()
        n_rows = 0
        for kind, t in ((O.PCORE, gpu_tables[0]), (O.OUTLIER, gpu_tables[1])):
            for i in range(len(t["id"])):
                s_o.inject(kind, t["cf1"][i], t["cf2"][i], t["cen"][i], t["pref"][i], t["w"][i], t["id"][i], t["uid"][i])
            n_rows += len(t["id"])
        ms = min(steady_sample, X.shape[0])
        t3 = time.perf_counter()
        s_o.online_microcluster_maintenance(X[:ms], 0, reset_param=False, offline=False)
        st = time.perf_counter() - t3
        out["steady"] = {"value": ms / st, "unit": "points/s", "cores": 1, "rows": n_rows,
                         "sample": "%d points on the stream's final table (%d microclusters, injected from the GPU run), one thread" % (ms, n_rows)}
    # the Python reference itself, measured in the build container (1 core, d = 20, 100 - 400 microclusters;
    # BASELINE.md section 2): it cannot be imported on the GPU box
    try:
        with open(REFERENCE_RATE_FILE) as f:
            out["reference_py"] = json.load(f)  # written by tools/time_reference.py in the build container
    except (OSError, ValueError):
        out["reference_py"] = None
    return out


def with_transfers(h, X, steps, step):
    """The same step with its transfers inside the timed region (the seam of app.py:170-178: a timepoint arrives as a host
    array, its per-point labels go back to the host): per step the upload of the step's points - started one step earlier
    through cc_points_prefetch (page-locked staging buffers, a stream of its own), so that it runs beside the previous
    step's clustering -, reset + online + offline phases, and the download of the labels.  Two host arrays take turns (the
    library recognises a prefetched array by its address)."""
    Xs = [X, X.copy()]
    h.points_upload(Xs[0])
    h.points_prefetch(Xs[1])
    step()  # (fills the pipeline: the first timed step's points are on their way while this one runs)
    h.labels_download()
    h.sync()
    wait_s = lab_s = 0.0
    t0 = time.perf_counter()
    for i in range(steps):
        t1 = time.perf_counter()
        h.points_upload(Xs[(i + 1) % 2])   # adopts the prefetched copy: waits for what is left of it
        h.points_prefetch(Xs[i % 2])       # the next step's points, beside this step's kernels
        t2 = time.perf_counter()
        step()
        t3 = time.perf_counter()
        h.labels_download()
        lab_s += time.perf_counter() - t3
        wait_s += t2 - t1
    h.sync()
    elapsed = time.perf_counter() - t0
    h.points_upload(Xs[steps % 2])  # (takes over the last prefetch: nothing left in flight)
    n, d = X.shape
    return {"value": n * steps / elapsed, "unit": "points/s", "ms_per_step": 1e3 * elapsed / steps, "steps": steps,
            "upload_mb_per_step": n * d * 8 / 1e6, "labels_mb_per_step": n * 9 / 1e6,
            "waiting_for_upload_ms_per_step": 1e3 * wait_s / steps, "labels_download_ms_per_step": 1e3 * lab_s / steps,
            "note": "host array in (pageable, 8 d bytes per point) through cc_points_prefetch one step ahead, labels out "
                    "(uid int64 + path int8 per point); everything else as the headline step"}


def general_regimes(h, cfg, n, d, step, steps):
    """The headline step (reset + online + offline phases, inputs resident) with the parameters moved out of the common case:
    pi = d - 2 (the tentative-add pdim filter decides which pcore microclusters a point may join) and k = 3 (every distance
    term an IEEE division).  The handle's parameters are put back afterwards."""
    out = {}
    for name, over in (("filter_pi_d_minus_2", {"pi": d - 2}), ("k_3", {"k": 3.0}), ("filter_and_k_3", {"pi": d - 2, "k": 3.0})):
        c = dict(cfg, **over)
        set_params(h, c, n, d)
        step()
        h.sync()
        t0 = time.perf_counter()
        acc = {"scan_p_launches": 0, "scan_launches": 0, "truncated": 0, "windows": 0}
        for _ in range(steps):
            s, n_clusters = step()
            for k in acc:
                acc[k] += s[k]
        h.sync()
        el = time.perf_counter() - t0
        out[name] = {"value": n * steps / el, "unit": "points/s", "ms_per_step": 1e3 * el / steps, "steps": steps, "parameters": over,
                     "clusters": n_clusters, "windows_per_step": acc["windows"] / steps, "truncated_windows_per_step": acc["truncated"] / steps,
                     "snapshot_scans_per_step": acc["scan_launches"] / steps, "pruned_scans_per_step": acc["scan_p_launches"] / steps}
    set_params(h, cfg, n, d)
    return out


def digest_of(h):
    """Bytes that pin the state a run ended in: labels, both tables, id counters."""
    from chronoclust_amd import _lib
    m = hashlib.sha256()
    uid, path = h.labels_download()
    m.update(uid.tobytes())
    m.update(path.tobytes())
    for kind in (_lib.PCORE, _lib.OUTLIER):
        t = h.export(kind)
        for key in ("id", "uid", "w", "cf1", "cf2", "cen", "pref"):
            m.update(np.ascontiguousarray(t[key]).tobytes())
    m.update(repr(h.counters()).encode())
    return m.hexdigest()


def load_pmc_valu(d):
    """The per-kernel PMC figures of the scan kernels at dimensionality d (pruned chain, plain scan) - only when they were
    measured on these kernel sources.  Returns (kernels, files, full_rows_frac_under_pmc) or (None, why, None)."""
    out, why, files, full = {}, None, [], None
    for suffix in ("", "_plain"):
        pm, name = newest_pmc(PMC_VALU_GLOB % (d, suffix))
        if pm is None:
            why = name
            continue
        files.append(name)
        if suffix == "" and pm.get("rows_evaluated_in_full_frac") is not None:
            full = float(pm["rows_evaluated_in_full_frac"])
        for kname, k in pm["kernels"].items():
            out[kname] = dict(k, source=name)
    # (the window's pruned scan is k_scan_p2 since round 6; k_scan_p where that is switched off or does not apply)
    if "k_scan_u" in out and ("k_scan_p" in out or "k_scan_p2" in out or "k_scan_p3" in out):
        return out, files, full
    return None, why or "incomplete PMC files for d = %d" % d, None


def scan_roofline(acc, d, kernel):
    """The `roofline` object of a snapshot scan from the library's own HIP-event timing of every launch (cc_stats,
    time_kernels = 1).  The binding roof is fp64 / VALU issue, not HBM (intensity ~1.5 M flop/B).
      achieved / frac   EXECUTED VALU instruction-lanes per second over all timed launches: per kernel of the chain the
                        instructions per (wave of 64 points, table row) that rocprofv3's SQ_INSTS_VALU pass counted
                        (profiles/r04_pmc_valu_*.json, quoted only for the kernel sources they were measured on) x the
                        (wave, row) pairs this run's launches covered x 64 lanes.  What the hardware did.
      effective         the ALGORITHMIC rate: 3 fp64 instructions per (point, microcluster, dim) - sub, mul, fma; k = 4 is
                        a power of two, so x2 * 2^e + acc in one rounding is the reference's double - whether or not they
                        were issued.  A pruned chain abandons a row as soon as its partial sum provably exceeds the
                        point's threshold, so this figure can exceed what an every-pair scan could reach (or 1).
      parts             the plain scans (start-up: the table is filling, most rows would be completed) and the pruned
                        chains (steady state) apart, with their launch counts
      executed          the per-kernel counters behind `achieved`: instructions per (wave, row), VALU- and LDS-busy
                        fractions of full windows running alone, the named co-limiter"""
    secs = acc["scan_ms"] * 1e-3
    if not secs or not acc["scan_launches"]:
        return None
    pd_all, pd_p = acc["scan_pair_dims"], acc.get("scan_pair_dims_pruned", 0.0)
    ms_p, n_p = acc.get("scan_ms_pruned", 0.0), int(acc.get("scan_launches_pruned", 0))
    ms_u, n_u = acc["scan_ms"] - ms_p, int(acc["scan_launches"]) - n_p
    issued = 3.0 * pd_all / secs / 1e12
    out = {"bound": "fp64_valu", "kernel": kernel, "achieved": None, "peak": FP64_VALU_PEAK_TOPS,
           "unit": "T VALU instruction-lanes/s (executed)", "frac": None, "traffic": None,
           "launches": int(acc["scan_launches"]), "avg_launch_us": 1e3 * acc["scan_ms"] / acc["scan_launches"],
           "pair_dims_per_launch": pd_all / acc["scan_launches"],
           "effective": {"achieved": issued, "frac": issued / FP64_VALU_PEAK_TOPS, "instr_per_pair_dim": 3.0,
                         "unit": "T fp64 VALU instruction-lanes/s (algorithmic: 3 per (point, microcluster, dim), issued or not)"},
           "pruning": prune_note(acc)}
    parts = {}
    if n_u:
        parts["plain"] = {"kernel": "k_scan_u (start-up: every row evaluated in full)", "launches": n_u,
                          "avg_launch_us": 1e3 * ms_u / n_u, "effective_frac": 3.0 * (pd_all - pd_p) / (ms_u * 1e-3) / 1e12 / FP64_VALU_PEAK_TOPS}
    if n_p:
        parts["pruned"] = {"kernel": "k_scan_p2 (k_seed + k_seed_merge before it while thresholds are seeded; steady state: rows abandoned on a prefix)", "launches": n_p,
                           "avg_launch_us": 1e3 * ms_p / n_p, "effective_frac": 3.0 * pd_p / (ms_p * 1e-3) / 1e12 / FP64_VALU_PEAK_TOPS}
    out["parts"] = parts
    pm, why, full_pmc = load_pmc_valu(d)
    if pm is None:
        out["executed"] = None
        out["derived"] = "achieved / frac not quoted: %s; re-run tools/profile_round.sh (SECTIONS=valu)" % why
        return out
    out["pmc_source"] = "profiles/" + " + ".join(why)
    out["derived"] = "achieved = VALU instructions per (wave,row) of pmc_source x this run's (wave,row) pairs x 64 lanes / " \
                     "this run's HIP-event scan time"
    # instruction-lanes: instructions per (wave, row) x (wave, row) pairs x 64 lanes = instructions per (wave, row) x pairs
    pairs_u, pairs_p = (pd_all - pd_p) / d, pd_p / d
    full = (acc["pruned_scan_full_rows"] / acc["pruned_scan_rows"]) if acc.get("pruned_scan_rows") else 0.0
    # a pruned launch of this run may complete more rows than the steady-state launches the counters were taken on (the
    # first pruned windows of a stream): every completed row beyond that share is charged the plain scan's row
    if full_pmc is None:  # (a file that does not record the share of rows its own launches completed: no extra charge)
        full_pmc = full
    extra = max(0.0, full - full_pmc) * pm["k_scan_u"]["valu_instructions_per_wave_row"]
    per_row_p = sum(pm[k]["valu_instructions_per_wave_row"] for k in PRUNED_CHAIN if k in pm) + extra
    lanes_u = pairs_u * pm["k_scan_u"]["valu_instructions_per_wave_row"]
    lanes_p = pairs_p * per_row_p
    executed = (lanes_u + lanes_p) / secs / 1e12
    out["achieved"], out["frac"] = executed, executed / FP64_VALU_PEAK_TOPS
    if n_u:
        parts["plain"]["executed_frac"] = lanes_u / (ms_u * 1e-3) / 1e12 / FP64_VALU_PEAK_TOPS
    if n_p:
        parts["pruned"]["executed_frac"] = lanes_p / (ms_p * 1e-3) / 1e12 / FP64_VALU_PEAK_TOPS
        parts["pruned"]["valu_instr_per_wave_row"] = per_row_p
    keys = ("valu_instructions_per_wave_row", "salu_instructions_per_wave_row", "lds_instructions_per_wave_row",
            "valu_busy_fraction", "lds_busy_fraction", "lds_array_busy_fraction", "wave_time_parked_on_waitcnt",
            "co_limiter", "avg_us_under_pmc", "source")
    out["executed"] = {"kernels": {n: {k: pm[n][k] for k in keys if k in pm[n]} for n in ("k_scan_u",) + PRUNED_CHAIN if n in pm},
                       "instruction_lanes_per_launch": (lanes_u + lanes_p) / acc["scan_launches"],
                       "note": "counters of full windows running alone (tools/steady.py under rocprofv3 --pmc, three passes); "
                               "busy fractions are of the kernel's own run time, `frac` above is over this run's launches "
                               "(short start-up windows and co-running validation kernels included)"}
    return out


def counted_on_bench(rf, cfg, scan_ms_per_step, ms_per_step):
    """The roofline from instructions COUNTED on the bench itself (tools/pmc_bench_step.py: rocprofv3 --pmc SQ_INSTS_VALU over
    `bench.py --steps 1 --warmup 1`, per kernel): the snapshot-scan kernels' instruction-lanes of one step / this run's
    HIP-event scan time per step -> achieved / frac; the lanes of ALL kernels of a step / this run's step time -> step_frac.
    Every step is the same deterministic work, so the counts of that pass are this run's; they are quoted only for the
    csrc/ digest and the workload they were counted on.  The model figure (instructions per (wave, row) of full windows
    running alone x this run's pairs) stays beside it as `model`."""
    import glob
    why = "no profiles/rNN_pmc_bench_step.json"
    for path in sorted(glob.glob(PMC_BENCH_STEP_GLOB), reverse=True):
        try:
            with open(path) as f:
                pm = json.load(f)
        except (OSError, ValueError):
            continue
        name = os.path.basename(path)
        if pm.get("csrc_sha256") != csrc_digest():
            why = "profiles/%s was counted on other kernel sources (csrc/ digest differs)" % name
            continue
        if (pm.get("points"), pm.get("dim"), pm.get("microclusters"), pm.get("window")) != (
                cfg["points"], cfg["dim"], cfg["microclusters"], cfg["window"]):
            why = "profiles/%s was counted on another workload" % name
            continue
        lanes_scan = pm["snapshot_scan"]["instruction_lanes_per_step"]
        lanes_all = pm["all_kernels"]["instruction_lanes_per_step"]
        achieved = lanes_scan / (scan_ms_per_step * 1e-3) / 1e12
        rf["model"] = {"achieved": rf.get("achieved"), "frac": rf.get("frac"), "pmc_source": rf.get("pmc_source"),
                       "derived": rf.get("derived")}
        rf["achieved"], rf["frac"] = achieved, achieved / FP64_VALU_PEAK_TOPS
        rf["step_frac"] = lanes_all / (ms_per_step * 1e-3) / 1e12 / FP64_VALU_PEAK_TOPS
        if rf["model"]["frac"]:
            rf["model_over_counted"] = rf["model"]["frac"] / rf["frac"]
        rf["pmc_source"] = "profiles/" + name
        rf["derived"] = "achieved = SQ_INSTS_VALU x 64 lanes of the snapshot-scan kernels of ONE bench step, counted by rocprofv3 " \
                        "on bench.py itself (pmc_source) / this run's HIP-event scan time per step; step_frac = the same count over " \
                        "ALL kernels of the step / this run's step time / peak"
        rf["counted"] = {"scan_instruction_lanes_per_step": lanes_scan, "all_instruction_lanes_per_step": lanes_all,
                         "scan_launches_per_step_counted": pm["snapshot_scan"]["launches_per_step"],
                         "kernels_by_lanes": {k: v["instruction_lanes_per_step"] for k, v in list(pm["kernels"].items())[:10]}}
        return
    rf["step_frac"] = None
    rf["counted_note"] = "not quoted: %s; run tools/r6_pmc_bench_step.sh on the GPU box" % why


def scan_kernel_name(s, d):
    if s.get("scan_p_launches", 0) > 0:
        if s.get("scan_p2_launches", 0) > 0:
            return "k_scan_u<%d,4> (plain: start-up) / k_prefix16 + k_scan_p3<%d,4> (pruned, MFMA prefix test: steady state)" % (d, d)
        return "k_seed<%d, 4> + k_seed_merge + k_scan_p<%d, 4> (pruned) / k_scan_u<%d, 4>" % (d, d, d)
    return ("k_scan_u<%d, 4>" % d) if s.get("scan_u_launches", 0) > 0 else "k_scan<%d, DIRTY=false>" % d


def prune_note(acc):
    """How much of the scans' algorithmic work the pruned launches really executed (sampled by the library)."""
    if not acc.get("scan_p_launches"):
        return None
    return {"pruned_launches": int(acc["scan_p_launches"]), "of_scan_launches": int(acc["scan_u_launches"]),
            "rows_evaluated_in_full_frac": (acc["pruned_scan_full_rows"] / acc["pruned_scan_rows"]) if acc["pruned_scan_rows"] else None,
            "note": "a pruned launch (k_seed -> k_seed_merge -> k_scan_p, timed as one) abandons a row as soon as its partial "
                    "sum provably exceeds the point's threshold and evaluates only this fraction of the (wave, row) pairs "
                    "over all dimensions - exactly (same labels, same tables).  `effective` counts the algorithmic 3 fp64 "
                    "instructions per (point, microcluster, dim) for those rows as well, `achieved` does not"}


def join_group(h, rank, world, group):
    """Makes `h` a member of the RCCL group of all ranks (one rank alone: a communicator of one, the same code path)."""
    from chronoclust_amd import _lib, multi
    if world > 1:
        multi.join_stream_group(h, group)
    else:
        h.comm_init_rccl(_lib.comm_unique_id(), 0, 1)
    seen = group.all_gather_bytes(b"%d" % h.comm_info()["world"])
    return [int(x) for x in seen]


def split_note(s, info, world, d, min_row_dims):
    """What the leg's line says about the row split - from what the library did (cc_stats, cc_comm_info), not assumed."""
    rows = int(s["rows"])
    if s["sharded_windows"]:
        return ", each of %d rank(s) scans 1/%d of the table rows per window (%d windows split)" % (
            info["world"], info["world"], int(s["sharded_windows"]))
    if info["transport"] == "none":
        return "; one rank without a group: nothing to split"
    if rows * d < min_row_dims:
        return "; the final table (%d rows x %d dims = %d) is below this leg's split threshold of %d (row, dim) entries: " \
               "every rank scans all rows" % (rows, d, rows * d, min_row_dims)
    return "; no window was split although the final table (%d x %d) is above the threshold of %d" % (rows, d, min_row_dims)


def exact_leg(args, rank, world, local_rank, group, sync, n, d, g, seed, label, force_split=False):
    """ONE stream on all ranks (exact multi-GPU path).  Every rank generates the same input."""
    from chronoclust_amd import multi
    from chronoclust_amd import _lib
    X = make_blobs(seed, n, d, g)
    cfg = blob_config(n)
    h = _lib.Handle(local_rank)
    h.set_tuning(time_kernels=1)
    if world > 1 or force_split:
        seen = join_group(h, rank, world, group)
    else:  # one rank, no group: what every rank's library says about itself (cc_comm_info), gathered like the others
        seen = [int(x) for x in group.all_gather_bytes(b"%d" % h.comm_info()["world"])]
    # the split threshold of this leg (the library's own default is the same figure): stated in the line below
    # (default: the thresholds cc_comm_init_rccl derived from its own measurement of this group's all-gather and scan -
    # cc_comm_calibrate; a group of one keeps the library's constant)
    h.set_shard_thresholds(0 if force_split else (args.shard_min_row_dims if args.shard_min_row_dims is not None else -1),
                           0 if force_split else -1)
    thresholds = h.stats()
    set_params(h, cfg, n, d)
    h.points_upload(X)
    del X

    def step():
        h.reset()
        h.online_run()
        s = h.stats()
        arrays, _ = h.offline_arrays()
        return s, len(arrays[2])

    for _ in range(args.stream_warmup):
        step()
    sync(h)
    t0 = time.perf_counter()
    acc = dict(scan_ms=0.0, scan_launches=0, comm_ms=0.0, comm_launches=0, run_ms=0.0, scan_pair_dims=0.0,
               scan_p_launches=0, scan_u_launches=0, pruned_scan_rows=0, pruned_scan_full_rows=0, sharded_windows=0,
               scan_ms_pruned=0.0, scan_launches_pruned=0, scan_pair_dims_pruned=0.0)
    for _ in range(args.stream_steps):
        s, n_clusters = step()
        for k in acc:
            acc[k] += s[k]
    s = dict(s, sharded_windows=acc["sharded_windows"])
    sync(h)
    elapsed = time.perf_counter() - t0
    elapsed = multi.max_over_ranks(elapsed, group)
    dg = digest_of(h)
    agree = group.all_equal(dg.encode())
    info = h.comm_info()
    if info["transport"] != "none":
        h.comm_destroy()
    h.close()
    return {
        "workload": "%s: ONE stream, 1 timepoint, %dx%d synthetic blobs, %d microclusters, exact sequential "
                    "semantics; online + offline phases per step; every rank holds the table and the points%s" % (
                        label, n, d, g, split_note(s, info, world, d, 0 if force_split else int(thresholds["split_threshold_row_dims"]))),
        "split_threshold": {"row_dims_plain_scan": int(thresholds["split_threshold_row_dims"]),
                            "row_dims_pruned_chain": int(thresholds["split_threshold_row_dims_pruned"]),
                            "measured_allgather_us": thresholds["calib_allgather_us"],
                            "measured_scan_ns_per_row_dim": thresholds["calib_scan_ns_per_row_dim"],
                            "source": "forced (0)" if force_split else ("--shard-min-row-dims" if args.shard_min_row_dims is not None else
                                      "cc_comm_calibrate: all-gather of a window's records x world / (world - 1) / plain scan per (row, dim), group maxima; "
                                      "a group of one keeps the library's constant")},
        "value": multi.one_stream_rate(n, args.stream_steps, elapsed), "unit": "points/s", "scaling": "strong",
        "n_gpus": world, "rccl_ranks_seen": seen, "steps": args.stream_steps, "warmup": args.stream_warmup,
        "ms_per_step": 1e3 * elapsed / args.stream_steps,
        "collective": "none (one rank, no group)" if info["transport"] == "none" else (
            "%s all-gather of 64 B per window point on the scan's stream, %d per step; offline: all-gather of preference "
            "vectors + reachability bitmask" % (info["transport"], int(acc["comm_launches"] / args.stream_steps))),
        "transport": info["transport"],
        "microclusters": int(s["rows"]), "clusters": n_clusters, "windows_per_step": int(s["windows"]),
        "sharded_windows_per_step": int(s["sharded_windows"] / args.stream_steps),
        "pruned_scan_launches_per_step": int(acc["scan_p_launches"] / args.stream_steps),
        "rank0_online_ms_per_step": acc["run_ms"] / args.stream_steps,
        "rank0_scan_ms_per_step": acc["scan_ms"] / args.stream_steps,
        "rank0_exchange_ms_per_step": acc["comm_ms"] / args.stream_steps,
        "roofline": scan_roofline(acc, d, scan_kernel_name(s, d)),
        "all_ranks_bit_identical": bool(agree), "state_sha256": dg[:16],
    }


def one_stream_exact(args, rank, world, local_rank, group, sync):
    return exact_leg(args, rank, world, local_rank, group, sync, args.stream_points, args.stream_dim, args.stream_blobs,
                     4242, "C5-shaped")


def one_stream_exact_c2(args, rank, world, local_rank, group, sync):
    """The headline's shape as ONE stream on all GPUs: default thresholds (5 000 x 20 is not split), then the split
    forced - the number that shows why the library does not split a table this small."""
    n, d, g = args.points, args.dim, args.blobs
    out = exact_leg(args, rank, world, local_rank, group, sync, n, d, g, 42, "C2-shaped")
    forced = exact_leg(args, rank, world, local_rank, group, sync, n, d, g, 42, "C2-shaped, split forced", force_split=True)
    out["split_forced"] = {k: forced[k] for k in ("value", "ms_per_step", "collective", "sharded_windows_per_step",
                                                   "rank0_scan_ms_per_step", "rank0_exchange_ms_per_step",
                                                   "all_ranks_bit_identical", "state_sha256", "rccl_ranks_seen")}
    out["split_note"] = "state_sha256 of the two runs must agree: the split changes where rows are scanned, not a result"
    return out


def relaxed_leg(args, rank, world, local_rank, group, sync, n, d, g, seed, label):
    """ONE stream, its EVENTS sharded over the ranks: the relaxed mode (cc_comm_set_relaxed).  Not the reference's
    semantics - the agreement with the exact path is reported beside the rate (rank 0 runs the exact path on the
    same input after the timed region)."""
    from chronoclust_amd import _lib, multi
    X = make_blobs(seed, n, d, g)
    cfg = blob_config(n)
    h = _lib.Handle(local_rank)
    h.set_tuning(time_kernels=1)
    seen = join_group(h, rank, world, group)
    h.comm_set_relaxed(args.relaxed_minibatch)
    set_params(h, cfg, n, d)
    h.points_upload(X)

    def step():
        h.reset()
        h.online_run()
        s = h.stats()
        arrays, _ = h.offline_arrays()
        return s, arrays

    for _ in range(args.stream_warmup):
        step()
    sync(h)
    t0 = time.perf_counter()
    acc = dict(scan_ms=0.0, scan_launches=0, scan_pair_dims=0.0, scan_p_launches=0, scan_u_launches=0,
               pruned_scan_rows=0, pruned_scan_full_rows=0, scan_ms_pruned=0.0, scan_launches_pruned=0,
               scan_pair_dims_pruned=0.0)
    for _ in range(args.stream_steps):
        s, arrays = step()
        for k in acc:
            acc[k] += s[k]
    sync(h)
    elapsed = time.perf_counter() - t0
    elapsed = multi.max_over_ranks(elapsed, group)
    rs = h.relaxed_stats()
    dg = digest_of(h)
    agree_ranks = group.all_equal(dg.encode())
    uid, _ = h.labels_download()
    pc = h.export(_lib.PCORE)
    pci = multi.point_cluster_index(uid, pc["id"], pc["uid"], arrays[0], arrays[1])
    n_clusters = len(arrays[2])
    transport = h.comm_info()["transport"]
    h.comm_destroy()
    h.close()
    out = {
        "workload": "%s: ONE stream, 1 timepoint, %dx%d synthetic blobs, %d microclusters; its events sharded "
                    "over %d rank(s) in contiguous blocks; online + offline phases per step" % (label, n, d, g, world),
        "semantics": "RELAXED (not the reference's): per super-step of %d points per rank the ranks cluster against "
                     "the shared table without seeing each other's adds, CF deltas are all-reduced, points that no "
                     "MC absorbs are clustered on every rank redundantly" % args.relaxed_minibatch,
        "value": multi.one_stream_rate(n, args.stream_steps, elapsed), "unit": "points/s", "scaling": "strong",
        "n_gpus": world, "rccl_ranks_seen": seen, "steps": args.stream_steps, "warmup": args.stream_warmup,
        "ms_per_step": 1e3 * elapsed / args.stream_steps,
        "collective": "RCCL all-reduce (sum, f64) of [%d, 2 x %d + 1] CF deltas + all-gather of the set-aside point "
                      "indices per super-step, %d super-steps per step" % (int(s["rows"]), d, rs["super_steps"]),
        "microclusters": int(s["rows"]), "clusters": n_clusters,
        "set_aside_points_per_step": rs["deferred_points"], "all_ranks_bit_identical": bool(agree_ranks),
        "transport": transport,
        "roofline": scan_roofline(acc, d, scan_kernel_name(s, d)),
    }
    if rank == 0:
        # the exact path on the same input, one GPU: what the relaxed result is measured against
        e = _lib.Handle(local_rank)
        set_params(e, cfg, n, d)
        e.points_upload(X)
        t1 = time.perf_counter()
        e.reset()
        e.online_run()
        e_arrays, _ = e.offline_arrays()
        exact_s = time.perf_counter() - t1
        e_uid, _ = e.labels_download()
        e_pc = e.export(_lib.PCORE)
        e_pci = multi.point_cluster_index(e_uid, e_pc["id"], e_pc["uid"], e_arrays[0], e_arrays[1])
        out["exact_path_one_gpu_points_per_s"] = n / exact_s
        out["agreement_with_exact_by_cluster"] = multi.label_agreement(pci, e_pci)
        out["agreement_with_exact_by_microcluster"] = multi.label_agreement(uid, e_uid)
        out["exact_clusters"] = len(e_arrays[2])
        e.close()
    return out


def events_sharded_relaxed(args, rank, world, local_rank, group, sync):
    return relaxed_leg(args, rank, world, local_rank, group, sync, args.relaxed_points, 14, 2000, 777, "C4-shaped")


def events_sharded_relaxed_c2(args, rank, world, local_rank, group, sync):
    return relaxed_leg(args, rank, world, local_rank, group, sync, args.points, args.dim, args.blobs, 42, "C2-shaped")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=20)
    ap.add_argument("--blobs", type=int, default=5000)
    ap.add_argument("--window", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=0)
    ap.add_argument("--segments", type=int, default=0)
    ap.add_argument("--lookahead", type=int, default=0, help="0/1 on (default), 2 off, 3 forced")
    ap.add_argument("--early-window", type=int, default=0, help="window while the table grows / is being promoted (0: 4096)")
    ap.add_argument("--windows-per-sync", type=int, default=0, help="windows enqueued between host read-backs (0: 16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-general-regimes", action="store_true", help="skip the step with the pdim filter on / k not a power of two")
    ap.add_argument("--no-transfers", action="store_true", help="skip the transfer-inclusive measurement (upload + labels inside the timed region)")
    ap.add_argument("--cpu-sample", type=int, default=100_000)
    ap.add_argument("--cpu-cores", type=int, default=16, help="threads of the all-cores CPU column (at most the usable cores)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--share-gpu", action="store_true", help="testing: every rank uses GPU 0")
    ap.add_argument("--no-one-stream", action="store_true", help="skip the one-stream-on-all-GPUs leg (C5-shaped)")
    ap.add_argument("--stream-points", type=int, default=2_000_000)
    ap.add_argument("--stream-dim", type=int, default=40)
    ap.add_argument("--stream-blobs", type=int, default=50_000)
    ap.add_argument("--stream-steps", type=int, default=2)
    ap.add_argument("--stream-warmup", type=int, default=1)
    ap.add_argument("--stream-timeout", type=float, default=120.0,
                    help="seconds after which a one-stream leg that has not finished is abandoned")
    ap.add_argument("--strict-legs", action="store_true",
                    help="exit with status 3 when a side leg fails or is abandoned (default: the headline line is printed, "
                         "the leg's object carries the error, `legs_failed` names it, exit status 0)")
    ap.add_argument("--no-relaxed", action="store_true", help="skip the event-sharded relaxed leg (C4-shaped)")
    ap.add_argument("--relaxed-points", type=int, default=5_000_000)
    ap.add_argument("--relaxed-minibatch", type=int, default=262144,
                    help="points per rank and super-step (eight windows: lookahead scans inside a super-step; 65 536 - two windows per call - measured half the rate on one rank)")
    ap.add_argument("--no-c2-legs", action="store_true", help="skip the two C2-shaped strong-scaling legs")
    ap.add_argument("--only-leg", default=None, help="profiling: only this leg, after a token headline (20 000 points, 100 microclusters, no CPU baseline)")
    ap.add_argument("--dry-launch", action="store_true", help="the ranks only rendezvous and rank 0 prints who came (no GPU is touched)")
    ap.add_argument("--shard-min-row-dims", type=int, default=None,
                    help="one-stream legs: a snapshot scan is split over the ranks from this many (row, dim) entries on "
                         "(default: what the library measured when the group was formed, cc_comm_calibrate)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # no launcher: start the N ranks from here (nothing in this process has loaded the HIP library)
        sys.stdout.flush()
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    # stdout carries exactly ONE line, the JSON: native libraries (RCCL prints its path when a communicator is
    # created) and anything else that writes to file descriptor 1 during the run go to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    from chronoclust_amd import multi, rendezvous
    rank, world, local_rank = multi.rank_info()
    if world != args.gpus:
        sys.stderr.write("[bench rank %d] --gpus %d, but the launcher started %d rank(s): the launcher's count is what "
                         "runs and what the line reports\n" % (rank, args.gpus, world))
    if args.dry_launch:
        group = rendezvous.from_env(timeout=60.0)
        came = group.all_gather_bytes(json.dumps({"rank": rank, "local_rank": local_rank, "pid": os.getpid()}).encode())
        group.barrier()
        if rank == 0:
            os.write(json_fd, (json.dumps({"dry_launch": True, "n_gpus": world, "gpus_requested": args.gpus,
                                           "launcher": os.environ.get("CHRONOCLUST_BENCH_LAUNCHER", "external"),
                                           "ranks": [json.loads(x) for x in came]}) + "\n").encode())
        group.close()
        sys.exit(0)
    guard = LineGuard(json_fd) if rank == 0 else None  # (before HIP or RCCL exist in this process)

    def emit(obj):
        guard.final(obj)

    group = rendezvous.from_env()  # barrier, max over ranks, the RCCL id: plain TCP, no torch
    if args.share_gpu:
        local_rank = 0
    os.environ.setdefault("CHRONOCLUST_HIP_COMM_TIMEOUT_S", str(int(args.stream_timeout)))

    from chronoclust_amd import _lib
    n, d, g = args.points, args.dim, args.blobs
    if args.only_leg:
        n, g = min(n, 20_000), min(g, 100)  # (a token headline: the profile should hold the leg's launches, not the headline's)
    X = make_blobs(multi.stream_seed(42, rank), n, d, g)
    cfg = blob_config(n)
    h = _lib.Handle(local_rank)
    h.set_tuning(window=args.window, rounds=args.rounds, segments=args.segments, lookahead=args.lookahead,
                 early_window=args.early_window, windows_per_sync=args.windows_per_sync,
                 time_kernels=0 if args.no_kernel_timing else 1)
    set_params(h, cfg, n, d)
    h.points_upload(X)  # inputs are resident in HBM before the timed region

    def step():
        h.reset()
        h.online_run()
        s = h.stats()
        arrays, _ = h.offline_arrays()  # cc_offline + cc_clusters_export: every cluster's members and CF vectors on the host
        return s, len(arrays[2])

    def sync(handle):
        """The timing bracket: everything this rank enqueued is done (cc_sync: the handle's HIP streams), every
        rank has got here (barrier), and once more in case the barrier let a straggler's work overlap."""
        handle.sync()
        group.barrier()
        handle.sync()

    for _ in range(args.warmup):
        step()
    sync(h)
    t0 = time.perf_counter()
    table_rows = 0.0
    online_ms = 0.0
    pacc = dict(scan_ms=0.0, scan_launches=0, scan_pair_dims=0.0, scan_p_launches=0, scan_u_launches=0, pruned_scan_rows=0,
                pruned_scan_full_rows=0, scan_ms_pruned=0.0, scan_launches_pruned=0, scan_pair_dims_pruned=0.0)
    for _ in range(args.steps):
        s, n_clusters = step()
        for k in pacc:
            pacc[k] += s[k]
        table_rows += s["table_rows_scanned"]
        online_ms += s["run_ms"]
    sync(h)
    elapsed = time.perf_counter() - t0
    elapsed = multi.max_over_ranks(elapsed, group)
    uid, _ = h.labels_download()

    out = None
    if rank == 0:
        out = {
            "metric": "points clustered/sec (20-dim)" if d == 20 else "points clustered/sec (%d-dim)" % d,
            "value": multi.whole_job_rate(n, args.steps, world, elapsed),
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "C2: 1 timepoint, %dx%d synthetic blobs (seed 42+rank), %d microclusters; "
                                   "exact sequential semantics; online + offline phases per step; one independent "
                                   "stream per GPU" % (n, d, g),
                       "points": n, "dim": d, "microclusters": int(s["rows"]), "clusters": n_clusters,
                       "streams": world, "window": int(s["window"]), "windows_per_step": int(s["windows"]),
                       "lookahead_windows_per_step": int(s["lookahead_windows"]),
                       "validation_rounds_per_step": int(s["rounds"]), "truncated_windows_per_step": int(s["truncated"])},
            "online_only_points_per_s": n * args.steps / (online_ms * 1e-3) if online_ms else None,
            "harness": "no torch: cc_sync for the timing bracket, chronoclust_amd.rendezvous (TCP) for barrier / max "
                       "over ranks / RCCL id",
        }
        if pacc["scan_launches"]:
            # the dominant kernel: the snapshot scan (see scan_roofline for what achieved / effective / parts / executed are)
            scan_ms, scan_launches = pacc["scan_ms"], pacc["scan_launches"]
            n_pts = n * args.steps
            alg_bytes = n_pts * (8 * d + 4) + table_rows * (16 * d + 8)
            hbm = alg_bytes / (scan_ms * 1e-3) / 1e9
            out["roofline"] = scan_roofline(pacc, d, scan_kernel_name(s, d))
            out["roofline"].update({
                "hbm": {"bound": "hbm", "achieved": hbm, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_launch": alg_bytes / scan_launches,
                        "note": "algorithmic bytes: the window's points (8d + 4 label bytes each) + the table columns "
                                "the distance reads (16d + 8 bytes per row); this roof does not bind: at ~1.5 M flop/B "
                                "the path is FP64-VALU-bound for M >> 10, so north_star's '>= 60 % of the HBM roofline' "
                                "cannot be met by an exact implementation at these table sizes (README.md)"},
                "launch_note": "every launch of the kernel is timed with HIP events on its stream: lookahead scans "
                               "(second stream, beside the validation kernels of the previous window, including the "
                               "few that go unused) and in-place scans (short windows of the start-up phase included)"})
            counted_on_bench(out["roofline"], out["config"], scan_ms / args.steps, out["ms_per_step"])
            # HBM traffic of the same kernel from the rocprofv3 PMC passes of this round (FETCH_SIZE / WRITE_SIZE in
            # separate runs); only quoted when it was measured on this workload shape AND on these kernel sources
            pmc, name = newest_pmc(PMC_TRAFFIC_GLOB)
            if pmc is None:
                out["roofline"]["traffic_note"] = "not quoted: %s" % name
            elif (pmc.get("points"), pmc.get("dim"), pmc.get("window")) == (n, d, out["config"]["window"]):
                out["roofline"]["traffic"] = pmc["k_scan_clean_bytes_per_launch"]
                out["roofline"]["traffic_note"] = "profiles/%s: %s" % (name, pmc["note"])
        if world == 1 and not args.no_transfers and not args.only_leg:
            wt = with_transfers(h, X, max(2, min(args.steps, 10)), step)
            out["with_transfers"] = wt
            out["value_with_transfers"] = wt["value"]
        if world == 1 and not args.no_cpu_baseline and not args.only_leg:
            # (the GPU box gives a one-GPU job 16 of the host's cores; os.cpu_count() reports the whole machine)
            usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            from chronoclust_amd import _lib as _L
            out["cpu_baseline"] = cpu_baseline(cfg, X, args.cpu_sample, max(1, min(usable, args.cpu_cores)), uid,
                                               gpu_tables=(h.export(_L.PCORE), h.export(_L.OUTLIER)))
            out["cpu_baseline"]["host_cpu_count"] = os.cpu_count()
        if world == 1 and not args.no_transfers and not args.no_general_regimes and not args.only_leg:
            # the same step outside the friendliest regime (pi = d, k = 4): the pdim filter of hddstream.py:317-321 on, k not a
            # power of two - pruned scans there since round 6 (k_scan_p3<GENERAL> behind seeded thresholds)
            out["general_regimes"] = general_regimes(h, cfg, n, d, step, max(2, min(args.steps, 6)))
            out["filter_on_points_per_s"] = out["general_regimes"]["filter_pi_d_minus_2"]["value"]
    h.close()
    del X
    if rank == 0:
        out["strong_scaling"] = {
            "note": "`value` above is N independent streams (one per GPU, no data-path collective); the entries here are ONE "
                    "stream clustered by all %d rank(s): points of that one stream per second.  The 20-dim metric's own "
                    "shape: one_stream_exact_c2 (exact) and events_sharded_relaxed_c2 (relaxed, not the reference's "
                    "semantics).  Full objects: the keys of the same names" % world,
            "n_gpus": world, "rccl_ranks_seen": None}
        guard.provisional(out)  # from here on the headline is safe whatever happens to the legs below

    legs = []
    if not args.no_one_stream:
        legs.append(("one_stream_exact", one_stream_exact))
    if not args.no_relaxed:
        legs.append(("events_sharded_relaxed", events_sharded_relaxed))
    if not args.no_c2_legs:
        legs.append(("one_stream_exact_c2", one_stream_exact_c2))
        legs.append(("events_sharded_relaxed_c2", events_sharded_relaxed_c2))
    if args.only_leg:
        legs = [(nm, fn) for nm, fn in legs if nm == args.only_leg]
    status = 0
    for name, fn in legs:
        # A collective that never completes must not take the headline measurement with it: every rank arms a
        # timer; if the leg is still running when it fires, rank 0 prints the line without it and all ranks leave -
        # with a non-zero status: a hung collective is not a clean exit.  (Inside the library every wait for a
        # collective is bounded as well and returns CC_ERR_COMM; the timer is the backstop for everything else.)
        done = threading.Event()

        def abandon(name=name, done=done):
            if done.is_set():
                return
            if rank == 0:
                out[name] = {"error": "not finished after %.0f s, abandoned" % (1.5 * args.stream_timeout)}
                out.setdefault("legs_failed", []).append(name)
                emit(out)
            else:
                sys.stderr.write("[bench rank %d] leg %s not finished after %.0f s, abandoned\n" % (
                    rank, name, 1.5 * args.stream_timeout))
            os._exit(EXIT_LEG_FAILED)  # (a hung collective is never a clean exit; the line is out)

        timer = threading.Timer(1.5 * args.stream_timeout, abandon)
        timer.daemon = True
        timer.start()
        try:
            leg = fn(args, rank, world, local_rank, group, sync)
        except Exception as e:  # noqa: BLE001 - reported in the line
            leg = {"error": "%s: %s" % (type(e).__name__, e)}
            sys.stderr.write("[bench rank %d] leg %s failed: %s\n" % (rank, name, leg["error"]))
        done.set()
        timer.cancel()
        if rank == 0:
            out[name] = leg
            if "error" not in leg:
                # the strong-scaling figures - ONE stream on all ranks - at the top level, beside the replicas headline
                out["strong_scaling"]["rccl_ranks_seen"] = leg.get("rccl_ranks_seen")
                out["strong_scaling"][name] = {
                    k: leg.get(k) for k in ("value", "unit", "ms_per_step", "n_gpus", "rccl_ranks_seen", "transport",
                                            "sharded_windows_per_step", "pruned_scan_launches_per_step",
                                            "all_ranks_bit_identical", "agreement_with_exact_by_cluster", "split_threshold") if k in leg}
                out["strong_scaling"][name]["semantics"] = "relaxed" if "relaxed" in name else "exact"
        # the ranks must agree on whether to go on: one that failed may have left the others' group
        failed = not group.all_equal(b"ok" if "error" not in leg else b"failed:" + str(rank).encode()) or "error" in leg
        if failed:
            # The headline has been measured and is printed whatever happens to a side leg (its object then carries
            # "error", and the line lists it under "legs_failed"): the exit status stays 0 so that a launcher does not
            # discard the line - unless --strict-legs asks for EXIT_LEG_FAILED (tools/n2_harness.sh does).
            if args.strict_legs:
                status = EXIT_LEG_FAILED
            if rank == 0:
                out.setdefault("legs_failed", []).append(name)
            if world > 1:
                break  # the ranks may no longer be in step: no further collective legs
    if rank == 0:
        emit(out)
    group.close()
    sys.exit(status)


if __name__ == "__main__":
    main()
