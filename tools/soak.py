"""Soak run of the fuzz parity cases beyond the seeds the test suite holds (tests/test_fuzz_parity.py: seeds 0-191;
tests/test_pruned_scan.py::test_forced_pruning_fuzz: 0-95; tests/test_sequential.py::test_register_resident_sequential_kernel_fuzz:
0-95; tests/test_hip_parity.py::test_skewed_streams_fuzz: 0-11, here on every eighth seed; ::test_long_chains_fuzz: 0-15, here on
every second seed): the same case generators and
checks, other seeds.
Usage: python tools/soak.py <first seed> <last seed> [log file]   (on the GPU box from the repo root; failures and a
progress line every 25 seeds are printed and appended to the log file - default gpurun_out/soak.log -, exit status 1 if
any case failed)"""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    a, b = int(sys.argv[1]), int(sys.argv[2])
    log_path = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", "soak.log")
    os.makedirs(os.path.dirname(log_path), exist_ok=True)
    log = open(log_path, "a")

    def say(text):
        print(text, flush=True)
        log.write(text + "\n")
        log.flush()
    import pytest  # noqa: F401  (the test modules import it)
    import test_fuzz_parity as F
    import test_pruned_scan as P
    import test_sequential as S
    import test_hip_parity as H
    import test_wide_dims as W
    bad, n, t0 = [], 0, time.time()
    for seed in range(a, b):
        cases = [("fuzz la=3", lambda s: F.test_fuzz_case(s, 3)),
                 ("fuzz la=2", lambda s: F.test_fuzz_case(s, 2)),
                 ("forced pruning", lambda s: P.test_forced_pruning_fuzz(s)),
                 # (round 6: the pruned chain outside the common case - pdim filter on, k not a power of two; the suite holds seeds 0-71)
                 ("forced pruning, filter / any k", lambda s: P.test_forced_pruning_fuzz_with_the_pdim_filter_and_any_k(s)),
                 ("register sequential kernel", lambda s: S.test_register_resident_sequential_kernel_fuzz(s)),
                 # (the general fuzz cases with the sequential kernels forced: k_seq, and k_seq_g beyond its image)
                 ("fuzz sequential", lambda s: S.test_fuzz_case_sequential(s))]
        cases.append(("wide streams", lambda s: W.test_wide_streams_fuzz(s)))  # (d = 65 .. 128 on k_seq_g; the suite holds seeds 0-9)
        if seed % 2 == 0:  # (few microclusters, long chains at every compiled width: tests/test_hip_parity.py holds seeds 0-15)
            cases.append(("long chains", lambda s: H.test_long_chains_fuzz(s)))
        if seed % 8 == 0:  # (two timepoints of 40-60 k points against 1 100-2 600 microclusters: seconds per case)
            cases.append(("skewed streams", lambda s: H.test_skewed_streams_fuzz(s)))
        for name, fn in cases:
            try:
                fn(seed)
                n += 1
            except BaseException as e:  # noqa: BLE001 - pytest.skip raises too
                if type(e).__name__ == "Skipped":
                    continue
                bad.append((seed, name))
                say("FAILED seed %d %s: %s" % (seed, name, "".join(traceback.format_exception_only(type(e), e)).strip()[:400]))
        if seed % 25 == 0:
            say("seed %d: %d cases, %d failures, %.0f s" % (seed, n, len(bad), time.time() - t0))
    say("seeds %d-%d: %d cases, %d failures: %s" % (a, b - 1, n, len(bad), bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
