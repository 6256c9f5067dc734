"""Parity bookkeeping: every comparison of a HIP result with the oracle / a reference-generated golden is recorded with
its error DISTRIBUTION, and the table is printed at the end of the pytest run (conftest.pytest_terminal_summary) and
written to gpurun_out/parity_report.json, so the margin under each tolerance is visible in the GPU test log.

Per comparison:  d = |got - ref| element-wise;  rel = d / max(|ref|, floor)  with floor = `floor_frac` * max|ref|
(elements near zero -- ReLU zeros, silent bins -- have no meaningful relative error; the floor is stated per row).
Recorded: n, max d / max|ref| (norm-wise), max rel, 99.9th percentile of rel, number of elements with rel > 1e-4."""
import json
import os

import numpy as np

ROWS = []


def record(name, got, ref, floor_frac=1e-2, floor_abs=None):
    a = np.asarray(got.detach().cpu().numpy() if hasattr(got, "detach") else got, dtype=np.float64).ravel()
    r = np.asarray(ref.detach().cpu().numpy() if hasattr(ref, "detach") else ref, dtype=np.float64).ravel()
    assert a.shape == r.shape, (name, a.shape, r.shape)
    fin = np.isfinite(r)
    a, r = a[fin], r[fin]
    if a.size == 0:
        return None
    d = np.abs(a - r)
    scale = float(np.abs(r).max())
    floor = floor_abs if floor_abs is not None else max(floor_frac * scale, 1e-30)
    rel = d / np.maximum(np.abs(r), floor)
    row = {"name": name, "n": int(a.size), "normwise": float(d.max() / max(scale, 1e-30)), "max_rel": float(rel.max()),
           "p999_rel": float(np.percentile(rel, 99.9)), "beyond_1e-4": int((rel > 1e-4).sum()), "floor": float(floor)}
    ROWS.append(row)
    return row


def note(name, **values):
    ROWS.append(dict(name=name, **values))


def summary_lines():
    out = ["parity report (rel = |d| / max(|ref|, floor)):",
           f"  {'comparison':<58} {'n':>9} {'normwise':>9} {'max rel':>9} {'p99.9':>9} {'>1e-4':>7}  floor"]
    for r in ROWS:
        if "n" in r:
            out.append(f"  {r['name'][:58]:<58} {r['n']:>9d} {r['normwise']:>9.2e} {r['max_rel']:>9.2e} {r['p999_rel']:>9.2e} "
                       f"{r['beyond_1e-4']:>7d}  {r['floor']:.1e}")
        else:
            out.append("  " + r["name"] + ": " + ", ".join(f"{k}={v:.4g}" if isinstance(v, float) else f"{k}={v}"
                                                           for k, v in r.items() if k != "name"))
    return out


def dump():
    if not ROWS:
        return
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "parity_report.json"), "w") as f:
            json.dump(ROWS, f, indent=1)
    except OSError:
        pass
