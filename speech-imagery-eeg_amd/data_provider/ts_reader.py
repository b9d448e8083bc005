"""sktime-free reader for the UEA/UCR ``.ts`` format (the reference reads it through
``sktime.datasets.load_from_tsfile_to_dataframe``, IGN/data_factory/data_loader.py:668-670; sktime is not
installed here -- SURVEY D13).

Format: ``#`` comments; ``@key value`` header lines (``@problemName``, ``@timeStamps``, ``@missing``,
``@univariate``, ``@dimensions``, ``@equalLength``, ``@seriesLength``, ``@classLabel true a b c``); ``@data``;
then one case per line: ``dim_1:dim_2:...:dim_D:label`` where each ``dim_i`` is a comma-separated series and
``?`` marks a missing value.  Only the un-timestamped form used by the UEA multivariate archive is supported.
"""
import numpy as np


class TsFormatError(ValueError):
    pass


def load_ts(path, missing="NaN"):
    """-> (cases, labels, meta): cases = list over samples of list over dimensions of float64 arrays,
    labels = list of str (or None when the file has no class label)."""
    meta = {}
    cases, labels = [], []
    in_data = False
    has_label = False
    with open(path, "r", encoding="utf-8") as f:
        for lineno, raw in enumerate(f, 1):
            line = raw.strip()
            if not line or line.startswith("#"):
                continue
            if not in_data:
                if not line.startswith("@"):
                    raise TsFormatError(f"{path}:{lineno}: expected a header line, got {line[:40]!r}")
                parts = line.split()
                key = parts[0][1:].lower()
                if key == "data":
                    in_data = True
                    has_label = bool(meta.get("classlabel", [False])[0])
                    continue
                if key == "classlabel":
                    flag = len(parts) > 1 and parts[1].lower() == "true"
                    meta[key] = [flag] + parts[2:]
                elif key == "timestamps":
                    if len(parts) > 1 and parts[1].lower() == "true":
                        raise TsFormatError(f"{path}: timestamped .ts files are not supported")
                    meta[key] = False
                else:
                    meta[key] = parts[1] if len(parts) == 2 else parts[1:]
                continue
            fields = line.split(":")
            if has_label:
                labels.append(fields[-1].strip())
                fields = fields[:-1]
            dims = []
            for fld in fields:
                vals = [v.strip() for v in fld.split(",")]
                arr = np.array([np.nan if v == "?" else float(v) for v in vals if v != ""], dtype=np.float64)
                dims.append(arr)
            if not dims:
                raise TsFormatError(f"{path}:{lineno}: case without data")
            cases.append(dims)
    if not in_data:
        raise TsFormatError(f"{path}: no @data section")
    ndim = {len(c) for c in cases}
    if len(ndim) > 1:
        raise TsFormatError(f"{path}: inconsistent number of dimensions per case {sorted(ndim)}")
    return cases, (labels if has_label else None), meta


def write_ts(path, X, y, problem="synthetic", class_labels=None):
    """Write (n, C, T) data (or a list of (C, T_i) arrays for ragged series) in UEA ``.ts`` format -- used to
    build BasicMotions-shaped fixtures for tests (the real archive cannot be fetched: no network)."""
    labels = [str(v) for v in y]
    classes = class_labels or sorted(set(labels))
    lens = {np.asarray(x).shape[-1] for x in X}
    with open(path, "w", encoding="utf-8") as f:
        f.write(f"@problemName {problem}\n@timeStamps false\n@missing false\n")
        f.write(f"@univariate {'true' if np.asarray(X[0]).shape[0] == 1 else 'false'}\n")
        f.write(f"@dimensions {np.asarray(X[0]).shape[0]}\n@equalLength {'true' if len(lens) == 1 else 'false'}\n")
        if len(lens) == 1:
            f.write(f"@seriesLength {lens.pop()}\n")
        f.write("@classLabel true " + " ".join(classes) + "\n@data\n")
        for x, lab in zip(X, labels):
            x = np.asarray(x)
            f.write(":".join(",".join(repr(float(v)) for v in row) for row in x) + ":" + lab + "\n")
