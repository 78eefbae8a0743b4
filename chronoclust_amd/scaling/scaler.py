"""MinMax [0, 1] scaling fitted on all timepoints: the interface of chronoclust/scaling/scaler.py:11-54.

The reference delegates to scikit-learn's MinMaxScaler; the arithmetic feeds the hot path, so it is restated
here operation by operation (sklearn/preprocessing/_data.py, MinMaxScaler.partial_fit / transform /
inverse_transform): scale_ = 1 / range (range < 10 eps -> 1), min_ = 0 - data_min * scale_,
transform = X * scale_ + min_ (two roundings), inverse = (X - min_) / scale_.
tests/test_host_logic.py checks bit equality against scikit-learn."""
import numpy as np
import pandas as pd


def read_timepoint(filename):
    """One timepoint as float64 [N, d].  CSV with a header row as in the reference (app.py:170, scaler.py:31);
    `.npy` files (binary side input: no text parse) are taken as they are."""
    if str(filename).endswith(".npy"):
        return np.ascontiguousarray(np.load(filename), dtype=np.float64)
    return pd.read_csv(filename, header=0, sep=',').to_numpy()


class Scaler(object):
    def __init__(self, data_files=None, handle=None):
        """handle: a chronoclust_amd._lib.Handle.  With it the fit reduces every file's columns on the device
        (cc_col_minmax) and the parsed files are kept for the run, so that no file is parsed twice."""
        self.scale_ = None
        self.min_ = None
        self.input_data = []
        self._handle = handle
        self.parsed = {}
        if data_files is not None and handle is not None:
            lo = hi = None
            for filename in data_files:
                X = np.ascontiguousarray(read_timepoint(filename), dtype=np.float64)
                self.parsed[filename] = X
                if X.shape[0] == 0:
                    continue
                mn, mx = handle.col_minmax(X)
                lo = mn if lo is None else np.fmin(lo, mn)
                hi = mx if hi is None else np.fmax(hi, mx)
            self._finish_fit(lo, hi)
        elif data_files is not None:
            rows = []
            for filename in data_files:
                rows.append(read_timepoint(filename))
            self.fit_scaler(np.concatenate(rows, axis=0) if rows else np.empty((0, 0)))

    def _finish_fit(self, data_min, data_max):
        data_range = data_max - data_min
        safe = data_range.copy()
        safe[safe < 10 * np.finfo(np.float64).eps] = 1.0
        self.scale_ = (1.0 - 0.0) / safe
        self.min_ = 0.0 - data_min * self.scale_
        self.data_min_, self.data_max_, self.data_range_ = data_min, data_max, data_range

    def fit_scaler(self, data):
        X = np.asarray(data, dtype=np.float64)
        self._finish_fit(np.nanmin(X, axis=0), np.nanmax(X, axis=0))
        self.set_input_data(data)

    def scale_data(self, data):
        X = np.array(data, dtype=np.float64)  # copy
        X *= self.scale_
        X += self.min_
        return X

    def reverse_scaling(self, data):
        X = np.array(data, dtype=np.float64)
        X -= self.min_
        X /= self.scale_
        return X

    def set_input_data(self, data):
        self.input_data = data

    def get_input_data(self):
        return self.input_data
