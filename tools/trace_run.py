import os, sys
sys.path.insert(0, os.getcwd())
import bench
from chronoclust_amd import _lib
n, d, g = 1_000_000, 20, 5000
X = bench.make_blobs(42, n, d, g)
cfg = bench.blob_config(n)
h = _lib.Handle(0)
bench.set_params(h, cfg, n, d)
h.points_upload(X)
h.online_run()
print(h.stats())
