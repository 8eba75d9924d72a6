"""norm.dmp handling (utils/features_utils.py:23-26): ``joblib.dump([means(F,), stds(F,)])`` written by
preprocess_all.py:250-251 and applied as (x - means) / stds (utils/dataset_utils.py:218)."""
import numpy as np

__all__ = ['load_normalization', 'save_normalization']


def load_normalization(norm_path):
    import joblib
    with open(norm_path, 'rb') as f:
        means, stds = joblib.load(f)
    return np.asarray(means, dtype=np.float32), np.asarray(stds, dtype=np.float32)


def save_normalization(norm_path, means, stds):
    import joblib
    joblib.dump([np.asarray(means), np.asarray(stds)], norm_path)
