"""Runs named fuzz cases one after the other with a line per case (time, verdict) - for finding the one that is slow or stuck.
Usage: python tools/one_case.py <module> <function> <first seed> <last seed>"""
import importlib
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

if __name__ == "__main__":
    import pytest  # noqa: F401
    mod = importlib.import_module(sys.argv[1])
    fn = getattr(mod, sys.argv[2])
    for seed in range(int(sys.argv[3]), int(sys.argv[4])):
        print("seed %d ..." % seed, flush=True)
        t0 = time.time()
        try:
            fn(seed)
            print("seed %d ok, %.1f s" % (seed, time.time() - t0), flush=True)
        except BaseException as e:  # noqa: BLE001
            print("seed %d FAILED after %.1f s: %s" % (seed, time.time() - t0, "".join(traceback.format_exception_only(type(e), e)).strip()[:600]), flush=True)
