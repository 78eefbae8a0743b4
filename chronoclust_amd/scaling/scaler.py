"""MinMax [0, 1] scaling fitted on all timepoints: the interface of chronoclust/scaling/scaler.py:11-54.

The reference delegates to scikit-learn's MinMaxScaler; the arithmetic feeds the hot path, so it is restated
here operation by operation (sklearn/preprocessing/_data.py, MinMaxScaler.partial_fit / transform /
inverse_transform): scale_ = 1 / range (range < 10 eps -> 1), min_ = 0 - data_min * scale_,
transform = X * scale_ + min_ (two roundings), inverse = (X - min_) / scale_.
tests/test_host_logic.py checks bit equality against scikit-learn."""
import numpy as np
import pandas as pd


class Scaler(object):
    def __init__(self, data_files=None):
        self.scale_ = None
        self.min_ = None
        self.input_data = []
        if data_files is not None:
            rows = []
            for filename in data_files:
                rows.append(pd.read_csv(filename, header=0, sep=',').to_numpy())
            self.fit_scaler(np.concatenate(rows, axis=0) if rows else np.empty((0, 0)))

    def fit_scaler(self, data):
        X = np.asarray(data, dtype=np.float64)
        data_min = np.nanmin(X, axis=0)
        data_max = np.nanmax(X, axis=0)
        data_range = data_max - data_min
        safe = data_range.copy()
        safe[safe < 10 * np.finfo(np.float64).eps] = 1.0
        self.scale_ = (1.0 - 0.0) / safe
        self.min_ = 0.0 - data_min * self.scale_
        self.data_min_, self.data_max_, self.data_range_ = data_min, data_max, data_range
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
