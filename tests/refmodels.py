"""The reference's OWN modules and decoder functions for the parity tests (test infrastructure): clair3/model.py Clair3_P /
Clair3_F and clair3/CallVariants.py possible_outcome_probabilites_from, imported from the staged copy oracle/_ref (GPU box) or
the checkout (build container).  Nothing here is on the product path."""
import sys

import numpy as np


import contextlib


@contextlib.contextmanager
def _reference_on_path(root):
    """the reference's packages importable inside the block, and gone from sys.path / sys.modules after it (the suite's own
    modules of the same top-level names -- `shared`, `clair3` -- must not be shadowed for later tests).  The reference imports
    lazily (clair3/model.py:93 imports shared.param_p inside __init__), so construction belongs inside the block too."""
    sys.path.insert(0, root)
    try:
        yield
    finally:
        sys.path.remove(root)
        for k in [k for k in sys.modules if k == "clair3" or k.startswith("clair3.") or k == "shared" or k.startswith("shared.")]:
            del sys.modules[k]


def reference_root_or_skip():
    import pytest
    from oracle.stage_reference import reference_root
    root = reference_root()
    if root is None:
        pytest.skip("no reference modules (oracle/_ref is staged by __graft_entry__.build() in the build container)")
    return root


def reference_model(root, kind, sd, indel, channels):
    """the reference's module in eval() with the state dict loaded strictly (clair3/CallVariantsFromCffi.py:19-28)"""
    import torch
    import importlib
    with _reference_on_path(root):
        model_py = importlib.import_module("clair3.model")
        cls = model_py.Clair3_P if kind == "pileup" else model_py.Clair3_F
        m = cls(add_indel_length=indel, predict=True, input_channels=channels)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    m.eval()
    return m


def reference_rows(model, x):
    """clair3/CallVariantsFromCffi.py:48-52 _torch_predict on the CPU: fp32 rows of the reference's own arithmetic"""
    import torch
    with torch.inference_mode():
        return model(torch.from_numpy(np.ascontiguousarray(x))).detach().cpu().numpy()


def outcome_probabilities(root, row, reference_base, indel):
    """Every joint outcome probability the reference's decoder ranks for one row (clair3/CallVariants.py:510-660
    possible_outcome_probabilites_from), flattened in the function's own order -> 1-D float64 array.  The call the decoder makes
    is a function of the ORDER of these numbers (output_from walks them from the largest down, :722-751), so two rows give the
    same call whenever they order them the same way."""
    import importlib
    row = np.asarray(row, dtype=np.float32)
    gt21, zyg = row[0:21], row[21:24]
    l1, l2 = (row[24:57], row[57:90]) if indel else (None, None)
    with _reference_on_path(root):
        cv = importlib.import_module("clair3.CallVariants")
        out = cv.possible_outcome_probabilites_from(gt21, zyg, l1, l2, reference_base=reference_base, alt_info_dict={},
                                                    add_indel_length=indel)
    flat = []
    for item in out:
        if isinstance(item, (list, tuple)):
            if len(item) and isinstance(item[0], (float, np.floating)):
                flat.extend(float(v) for v in item)
        elif isinstance(item, (float, np.floating)):
            flat.append(float(item))
    return np.asarray(flat, dtype=np.float64)


def order_inversions(p_ref, p_got):
    """pairs (i, j) the two rows rank differently, with the reference's gap |p_ref[i] - p_ref[j]| of each: a differing call is a
    near-tie only if every inverted pair is closer than the parity tolerance of the rows allows"""
    n = len(p_ref)
    inv = []
    order = np.argsort(-p_ref, kind="stable")[: min(n, 64)]  # calls are made among the largest outcomes
    for a in range(len(order)):
        for b in range(a + 1, len(order)):
            i, j = order[a], order[b]
            if p_ref[i] == p_ref[j]:
                continue
            if (p_ref[i] - p_ref[j]) * (p_got[i] - p_got[j]) <= 0:
                inv.append((int(i), int(j), abs(float(p_ref[i] - p_ref[j]))))
    return inv
