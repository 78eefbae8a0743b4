"""Helpers to read the golden state dumps written by tests/golden/make_golden.py."""
import hashlib
import os

import numpy as np

import scenarios

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class StateDump(object):
    def __init__(self, path):
        self.z = np.load(path, allow_pickle=False)
        self.n_timepoints = int(self.z["n_timepoints"][0])

    def get(self, t, key):
        return self.z["t%d_%s" % (t, key)]

    def has(self, t, key):
        return ("t%d_%s" % (t, key)) in self.z.files

    def clusters(self, t):
        off = self.get(t, "cl_offsets")
        mem = self.get(t, "cl_members")
        return [dict(members=mem[off[i]:off[i + 1]], w=self.get(t, "cl_w")[i], cf1=self.get(t, "cl_cf1")[i],
                     cf2=self.get(t, "cl_cf2")[i], cen=self.get(t, "cl_cen")[i], pref=self.get(t, "cl_pref")[i])
                for i in range(len(off) - 1)]


def blob_inputs(name, dump):
    """Regenerates a blob scenario's inputs from its seed and checks them against the recorded hash."""
    sc = scenarios.BLOB_SCENARIOS[name]
    Xs = scenarios.make_blob_timepoints(sc)
    if sc.get("normalise"):
        return [dump.get(t, "X") for t in range(dump.n_timepoints)]  # scaled inputs were stored
    for t, X in enumerate(Xs):
        sha = np.frombuffer(hashlib.sha256(np.ascontiguousarray(X).tobytes()).digest(), dtype=np.uint8)
        assert (sha == dump.get(t, "xsha")).all(), "numpy Generator stream changed: regenerate goldens"
    return Xs


def assert_tables_equal(got, dump, t, kind, exact=True):
    """got: dict(id, uid, w, cf1, cf2, cen, pref) in list order; dump keys '<kind>_<name>'."""
    for key in ("id", "uid"):
        np.testing.assert_array_equal(got[key], dump.get(t, "%s_%s" % (kind, key)), err_msg="%s %s t=%d" % (kind, key, t))
    for key in ("w", "cf1", "cf2", "cen", "pref"):
        exp = dump.get(t, "%s_%s" % (kind, key))
        if exact:
            assert np.array_equal(got[key], exp), "%s %s t=%d differs bitwise (max abs %g)" % (
                kind, key, t, np.max(np.abs(got[key] - exp)) if exp.size else 0.0)
        else:
            np.testing.assert_allclose(got[key], exp, rtol=0, atol=1e-6)
