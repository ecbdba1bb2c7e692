#!/usr/bin/env python3
"""Golden fingerprints of the reference's stage transitions (prog/helpers.py load_slice_clone_ema, load_super), run HERE
on /root/reference with deterministic weights (tests/_detfill.py), for tests/test_growth.py.
  python tools/gen_golden_growth.py   ->  tests/golden/growth.npz"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tools.ref_import import load_reference            # noqa: E402
from tools.gen_golden import tiny_volo                 # noqa: E402
from _detfill import fill_state_dict, fingerprint      # noqa: E402


class _Wrap:                                            # ModelEma-like: unwrap_model() looks for .module
    def __init__(self, m):
        self.module = m


def main():
    ns = load_reference()
    out = {}
    cases = [("deeper", "volo_h2_l4", "volo_h2_l6"), ("wider_deeper", "volo_h2_l4", "volo_h4_l7"), ("wider", "volo_h2_l6", "volo_h4_l6")]
    for name, src_v, dst_v in cases:
        src = tiny_volo(ns, src_v, 64, 16)
        emas = [tiny_volo(ns, src_v, 64, 16) for _ in range(4)]
        dst = tiny_volo(ns, dst_v, 64, 16)
        src.load_state_dict(fill_state_dict(src.state_dict(), 0))
        for i, e in enumerate(emas):
            e.load_state_dict(fill_state_dict(e.state_dict(), i + 1))
        dst.load_state_dict(fill_state_dict(dst.state_dict(), 9))
        # reference call shape: load_slice_clone_ema(model, prev_ema_list[3], prev_ema_list)  (main_prog.py:1382)
        ns.helpers.load_slice_clone_ema(dst, emas[3], [_Wrap(e) for e in emas])
        for k, v in dst.state_dict().items():
            out["%s/%s" % (name, k)] = np.asarray(fingerprint(v), dtype=np.float64)
    # supernet -> sub-network extraction (equal widths)
    for name, sup_v, sub_v, base in [("sub_l4_of_l7", "volo_h2_l7", "volo_h2_l4", 4), ("sub_l5_of_l7", "volo_h2_l7", "volo_h2_l5", 4)]:
        sup = tiny_volo(ns, sup_v, 64, 16)
        sub = tiny_volo(ns, sub_v, 64, 16)
        sup.load_state_dict(fill_state_dict(sup.state_dict(), 3))
        sub.load_state_dict(fill_state_dict(sub.state_dict(), 8))
        ns.helpers.load_super(sub, sup, base_layer=base, model_name="volo")
        for k, v in sub.state_dict().items():
            out["%s/%s" % (name, k)] = np.asarray(fingerprint(v), dtype=np.float64)
    # search helpers of main_prog.py: the file cannot be imported (timm / tlt / apex), so the three self-contained functions are
    # cut out with ast and executed as they are
    import ast
    import json
    import random
    src = open(os.path.join(os.environ.get("AUTOPROG_REFERENCE", "/root/reference"), "main_prog.py")).read()
    ns_fn = {"random": random}
    for node in ast.parse(src).body:
        if isinstance(node, ast.FunctionDef) and node.name in ("get_divisor", "no_repeats", "sample_configs"):
            exec(compile(ast.Module(body=[node], type_ignores=[]), "main_prog.py", "exec"), ns_fn)
    rec = {"get_divisor": [[n, f, ns_fn["get_divisor"](n, f)] for n in (1, 2, 4, 6, 8, 12, 16) for f in (0.0, 0.1, 0.26, 0.5, 0.51, 0.77, 1.0)],
           "no_repeats": [[v, ns_fn["no_repeats"](list(v))] for v in ([128, 128, 160, 192, 192, 224], [12, 12, 12], [], [3, 1, 3, 2, 1])],
           "sample_configs": []}
    for seed in range(6):
        random.seed(seed)
        l_list, r_list = [12, 15, 18], [160, 192, 224]
        cfgs = [ns_fn["sample_configs"](l_list, r_list, mode="random") for _ in range(5)]
        rec["sample_configs"].append([seed, [[c[0], c[1], c[2]] for c in cfgs]])
    rec["sample_smallest"] = list(ns_fn["sample_configs"]([9, 12], [128, 160], mode="smallest"))
    with open(os.path.join(ROOT, "tests", "golden", "search.json"), "w") as fh:
        json.dump(rec, fh)
    path = os.path.join(ROOT, "tests", "golden", "growth.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "fingerprints,", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
