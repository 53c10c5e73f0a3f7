"""On-disk evaluation format of the reference: one `.mat` per image with keys
`x0, y0, ATy0` (float [..,H,W,2], real/imag last), `mask` [H,W], `gt` [1,H,W]
(/root/reference/dataset/datasets.py:135-207).  Filenames carry the task, e.g. `..._4_10...mat` -> "4x_10"
(acceleration x noise level; `extract_task` datasets.py:13-16).  scipy.io reads/writes it; no sample file ships with the
reference, so the writer here is what the tests round-trip."""
from __future__ import annotations

import os
import re
from typing import Dict, List, Sequence, Tuple

import numpy as np

OPTIMAL_TASKS = ["2x_5", "2x_10", "2x_15", "4x_5", "4x_10", "4x_15", "8x_5", "8x_10", "8x_15"]   # datasets.py:175
FLEX_TASKS = ["rtg_1.5", "rtg_3", "rtg_3.5", "rtg_4", "rtg_4.5", "rtg_5"]                          # datasets.py:136
OPTIMAL_RTG_RANGE = (-1.08, 16.6)       # datasets.py:178-179
FLEX_RTG_RANGE = (-1.8, 5.0)            # datasets.py:139-140
_MAT_KEYS = ("x0", "y0", "mask", "ATy0", "gt")


def task_from_filename(fn: str) -> str:
    """`\\d+_\\d+` in the file name -> '<accel>x_<sigma>' (datasets.py:13-16, :188-189)."""
    m = re.search(r"\d+_\d+", os.path.basename(fn))
    if m is None:
        raise ValueError(f"no '<accel>_<sigma>' task tag in file name {fn!r}")
    t = m.group()
    return t[0] + "x" + t[1:]           # exactly the reference's string surgery (single-digit acceleration)


def normalised_rtg(target: float, flex: bool = False) -> float:
    lo, hi = FLEX_RTG_RANGE if flex else OPTIMAL_RTG_RANGE
    return (float(target) - lo) / (hi - lo)


def save_mat(path: str, problem: Dict[str, np.ndarray], index: int = 0) -> None:
    """Write slice `index` of a `synthetic.make_problem` dict in the reference's `.mat` layout.  `x0` is stored as the
    raw zero-filled reconstruction (= ATy0, negative pixels included), as a TFPnP-style file holds it; the loader clips."""
    from scipy.io import savemat
    savemat(path, {"x0": problem["ATy0"][index], "y0": problem["y0"][index], "ATy0": problem["ATy0"][index],
                   "mask": problem["mask"].astype(np.uint8), "gt": problem["gt"][index]})


def load_mat(path: str) -> Dict[str, np.ndarray]:
    """One image as the reference's datasets return it (datasets.py:153-160,191-199): `x0` clipped at 0 is what the
    environment gets (`action_dict['x0']`, :160,199), while the policy's first state token is the UNclipped real part
    of the stored x0 (:162,201 read `mat['x0'][..., 0]` after the clip only rebound the dict entry) - kept here as
    `x0_raw` [..,H,W] so a reference checkpoint sees the same first observation."""
    from scipy.io import loadmat
    mat = loadmat(path)
    missing = [k for k in _MAT_KEYS if k not in mat]
    if missing:
        raise KeyError(f"{path}: missing keys {missing}")
    out = {k: np.asarray(mat[k]) for k in _MAT_KEYS}
    out["x0_raw"] = np.array(out["x0"][..., 0], dtype=np.float32)
    out["x0"] = np.clip(out["x0"], a_min=0, a_max=None)
    return out


def list_dir(data_dir: str, limit: int = 0) -> List[str]:
    """The `.mat` files of a directory in the reference's order (sorted, datasets.py:146-147), first `limit` if given."""
    fns = sorted(f for f in os.listdir(data_dir) if f.endswith(".mat"))
    if limit:
        fns = fns[:limit]
    if not fns:
        raise FileNotFoundError(f"no .mat files in {data_dir}")
    return fns


def load_dir(data_dir: str, limit: int = 0, start: int = 0, stop: int = None) -> Tuple[Dict[str, np.ndarray], List[str]]:
    """The `.mat` files [start, stop) of a directory (of its first `limit` files, sorted) stacked into one batch dict
    {x0,y0,ATy0: [N,1,H,W,2], mask [H,W], gt [N,1,H,W], x0_raw [N,1,H,W]} + their task names.  A rank of a sharded run
    reads only its own files."""
    fns = list_dir(data_dir, limit)[start:stop]
    if not fns:
        raise FileNotFoundError(f"no .mat files in {data_dir} [{start}:{stop}]")
    items = [load_mat(os.path.join(data_dir, f)) for f in fns]
    h, w = items[0]["mask"].shape[-2:]
    for f, it in zip(fns, items):
        if it["mask"].shape[-2:] != (h, w) or not np.array_equal(it["mask"] != 0, items[0]["mask"] != 0):
            raise ValueError(f"{f}: all files of one batch must share size and sampling mask")
    batch = {k: np.stack([it[k].reshape(1, h, w, 2) for it in items]).astype(np.float32) for k in ("x0", "y0", "ATy0")}
    batch["gt"] = np.stack([it["gt"].reshape(1, h, w) for it in items]).astype(np.float32)
    batch["x0_raw"] = np.stack([it["x0_raw"].reshape(1, h, w) for it in items]).astype(np.float32)
    batch["mask"] = items[0]["mask"].reshape(h, w) != 0
    return batch, [task_from_filename(f) for f in fns]


def task_tokens(tasks: Sequence[str], flex_target: float = None) -> np.ndarray:
    if flex_target is not None:
        key = "rtg_" + (str(int(flex_target)) if float(flex_target).is_integer() else str(flex_target))
        return np.full(len(tasks), FLEX_TASKS.index(key), dtype=np.int64)
    return np.array([OPTIMAL_TASKS.index(t) for t in tasks], dtype=np.int64)
