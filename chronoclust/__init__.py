"""`chronoclust` — the reference's import path, served by the MI355X build.

`from chronoclust import app; app.run(...)` (sample_run_script/sample_run.py:1, chronoclust/setup.py: package name
`chronoclust`) resolves to `chronoclust_amd.app`; the sub-modules a user of the reference imports
(`chronoclust.clustering.hddstream`, `chronoclust.tracking.cluster_tracker`, `chronoclust.scaling.scaler`,
`chronoclust.objects.cluster`, `chronoclust.objects.microcluster`) are the very module objects of `chronoclust_amd`
(registered in sys.modules below), so state and classes are shared whichever name was used for the import.

The per-vector numba helpers of the reference (`chronoclust.utilities.*`, `chronoclust.clustering.predecon`,
`chronoclust.objects.predecon_mc`) have no counterpart: that arithmetic lives in the HIP kernels behind
include/chronoclust_hip.h (seam B4)."""
import importlib
import sys

_ALIASES = ("app", "clustering", "clustering.hddstream", "tracking", "tracking.cluster_tracker", "scaling",
            "scaling.scaler", "objects", "objects.cluster", "objects.microcluster")

for _name in _ALIASES:
    _mod = importlib.import_module("chronoclust_amd." + _name)
    sys.modules[__name__ + "." + _name] = _mod
    if "." not in _name:
        globals()[_name] = _mod
del _name, _mod


class _NoCounterpart(object):
    """Import hook: the reference modules that have no counterpart here fail with a message that says why (instead of a
    bare ModuleNotFoundError).  SURVEY.md section 8b, seam B4: not a viable GPU boundary - one microcluster x one point
    per call."""
    _NAMES = ("clustering.predecon", "objects.predecon_mc", "utilities", "utilities.mc_functions",
              "utilities.predeconmc_functions")

    def find_spec(self, fullname, path=None, target=None):
        for pkg in (__name__, "chronoclust_amd"):
            if fullname.startswith(pkg + ".") and fullname[len(pkg) + 1:] in self._NAMES:
                raise ImportError(
                    "%s has no counterpart in the MI355X build of chronoclust: the per-vector numba helpers and the "
                    "PreDeCon classes of the reference (chronoclust/utilities/*.py, clustering/predecon.py, "
                    "objects/predecon_mc.py) are HIP kernels behind the C-ABI of include/chronoclust_hip.h here.  Use "
                    "chronoclust.app.run, chronoclust.clustering.hddstream.HDDStream (offline_clustering / final_clusters) "
                    "or the chronoclust.tracking.cluster_tracker classes; see INTEGRATION.md." % fullname, name=fullname)
        return None


sys.meta_path.insert(0, _NoCounterpart())
