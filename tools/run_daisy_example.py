#!/usr/bin/env python
"""Drive the reference's own `run_examples/test.py` with the HIP path behind its model classes
(INTEGRATION.md §1): nothing in daisyRec is edited; `daisyrec_amd.dropin.install()` rebinds
`MF` / `FM` / `NeuMF` / `LightGCN` / `Item2Vec` (and, with --native-sampler, `BasicNegtiveSampler`).

    python tools/run_daisy_example.py --daisy /path/to/daisyRec -- --algo_name mf --epochs 5

Everything after `--` is the reference's command line.  The three compatibility shims the current
numpy / pandas / scipy releases make necessary for the UNMODIFIED reference (not for this package) are
applied first: `np.asfarray` (metrics.py:206), `pd.Series.iteritems` (sampler.py:136),
`scipy.sparse.dok_matrix._update` (LightGCNRecommender.py:89); `colorlog` / `colorama` fall back to the
stubs under tests/golden/_shims when they are not installed.
"""
import argparse
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--daisy", required=True, help="checkout of AmazingDD/daisyRec (contains daisy/, run_examples/, data/)")
    ap.add_argument("--script", default="run_examples/test.py")
    ap.add_argument("--native-sampler", action="store_true", help="also rebind daisy.utils.sampler.BasicNegtiveSampler")
    ap.add_argument("--native-front-end", action="store_true",
                    help="also rebind daisy.utils.utils.get_ur / get_ir / build_candidates_set (no Python row loops)")
    ap.add_argument("--extra-path", action="append", default=[],
                    help="directories put in front of sys.path (e.g. a stand-in for a package the image lacks)")
    ap.add_argument("rest", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    rest = a.rest[1:] if a.rest[:1] == ["--"] else a.rest

    import numpy as np
    import pandas as pd
    if not hasattr(np, "asfarray"):
        np.asfarray = lambda x, dtype=np.float64: np.asarray(x, dtype=dtype)
    if not hasattr(pd.Series, "iteritems"):
        pd.Series.iteritems = pd.Series.items
    try:
        import scipy.sparse as sp
        if not hasattr(sp.dok_matrix, "_update"):
            sp.dok_matrix._update = lambda self, data: self._dict.update(data)
    except ImportError:
        pass
    for mod in ("colorlog", "colorama"):
        try:
            __import__(mod)
        except ImportError:
            sys.path.insert(0, os.path.join(ROOT, "tests", "golden", "_shims"))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.abspath(a.daisy))
    for extra in a.extra_path:
        sys.path.insert(0, os.path.abspath(extra))

    import daisyrec_amd.dropin as dropin
    dropin.install(sampler=a.native_sampler, front_end=a.native_front_end)

    os.chdir(os.path.abspath(a.daisy))                      # data_path etc. are relative (basic.yaml)
    sys.argv = [a.script] + rest
    runpy.run_path(os.path.join(os.path.abspath(a.daisy), a.script), run_name="__main__")


if __name__ == "__main__":
    main()
