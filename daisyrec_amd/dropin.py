"""Make an unmodified daisyRec driver (run_examples/test.py, tune.py) train MF through
the HIP path: rebinds the names those scripts import (test.py:4,21-22; tune.py:7).

    import daisyrec_amd.dropin as d; d.install()          # before `import run_examples.test`

or, where the reference checkout exists:
    python tools/run_daisy_example.py --daisy /path/to/daisyRec -- --algo_name mf ...
"""
from __future__ import annotations

import importlib


def install(sampler=False, front_end=False):
    """Patch `daisy.model.MFRecommender.MF` (and the other mirrored recommenders) in place.
    sampler: also the negative sampler (device generator: same distribution, different stream than MT19937).
    front_end: also `daisy.utils.utils.get_ur / get_ir` (same dicts, built from one sort instead of a Python row
    loop) and `build_candidates_set` (device generator) - what makes the drivers usable beyond ml-100k sizes
    (SURVEY.md section 8f rank 1).  Call before the driver module is imported (it binds these names at import)."""
    from .model.MFRecommender import MF
    from .model.FMRecommender import FM

    ref_mf = importlib.import_module("daisy.model.MFRecommender")
    ref_mf.MF = MF
    ref_fm = importlib.import_module("daisy.model.FMRecommender")
    ref_fm.FM = FM
    from .model.NeuMFRecommender import NeuMF
    ref_nm = importlib.import_module("daisy.model.NeuMFRecommender")
    ref_nm.NeuMF = NeuMF
    from .model.Item2VecRecommender import Item2Vec
    ref_iv = importlib.import_module("daisy.model.Item2VecRecommender")
    ref_iv.Item2Vec = Item2Vec
    from .model.LightGCNRecommender import LightGCN
    try:                                   # imports scipy; the reference module itself needs it too
        ref_lg = importlib.import_module("daisy.model.LightGCNRecommender")
        ref_lg.LightGCN = LightGCN
    except ImportError:
        pass
    if sampler:
        from .utils.sampler import BasicNegtiveSampler

        from .utils.sampler import SkipGramNegativeSampler

        ref_s = importlib.import_module("daisy.utils.sampler")
        ref_s.BasicNegtiveSampler = BasicNegtiveSampler
        ref_s.SkipGramNegativeSampler = SkipGramNegativeSampler
    if front_end:
        from .utils import utils as U

        ref_u = importlib.import_module("daisy.utils.utils")
        ref_u.get_ur, ref_u.get_ir = U.get_ur, U.get_ir
        ref_u.build_candidates_set = U.build_candidates_set
    return MF
