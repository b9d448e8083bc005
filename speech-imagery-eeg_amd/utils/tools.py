"""Host-side helpers of the experiment harness (checkpoint contract of IGN/utils/tools.py:9-77)."""
import os

import numpy as np
import torch


class EarlyStopping:
    """Keeps the best ``state_dict`` at ``<path>/checkpoint.pth`` (weights only -- IGN/utils/tools.py:19-38).

    Call with the quantity to MINIMISE (the harness passes ``-val_accuracy``).  ``np.inf`` instead of the
    reference's ``np.Inf`` (removed in NumPy 2).
    """

    def __init__(self, patience=7, verbose=False, delta=0):
        self.patience, self.verbose, self.delta = patience, verbose, delta
        self.counter = 0
        self.best_score = None
        self.early_stop = False
        self.val_loss_min = np.inf

    def __call__(self, val_loss, model, path):
        score = -val_loss
        if self.best_score is None or score >= self.best_score + self.delta:
            self.best_score = score
            self.save_checkpoint(val_loss, model, path)
            self.counter = 0
        else:
            self.counter += 1
            print(f'EarlyStopping counter: {self.counter} out of {self.patience}')
            if self.counter >= self.patience:
                self.early_stop = True

    def save_checkpoint(self, val_loss, model, path):
        if self.verbose:
            print(f'Validation loss decreased ({self.val_loss_min:.6f} --> {val_loss:.6f}).  Saving model ...')
        torch.save(model.state_dict(), os.path.join(path, 'checkpoint.pth'))
        self.val_loss_min = val_loss


def convert_to_hms(seconds):
    s = int(seconds)
    return f"{s // 3600:02d}:{(s % 3600) // 60:02d}:{s % 60:02d}"


def gini_coefficient(w):
    """Mean Gini coefficient over the rows of a non-negative matrix (IGN/utils/tools.py:54-77)."""
    w = np.asarray(w)
    if w.shape[1] == 0:
        return 0.0
    vals = []
    for row in w:
        x = np.sort(np.asarray(row, dtype=np.float64))
        n = len(x)
        vals.append((2 * np.sum(np.arange(1, n + 1) * x)) / (n * x.sum()) - (n + 1) / n)
    return float(np.mean(vals))
