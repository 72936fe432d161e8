import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


# A hung kernel must not hang the GPU box until the driver's own limit.  pytest-timeout's 'thread' method is the only one
# that works against a host thread blocked inside hipStreamSynchronize (the 'signal' method needs the interpreter to
# return to bytecode): it dumps every thread's stack and os._exit()s the WHOLE pytest process - the remaining tests of
# the session do not run and no summary is printed; the dump names the hung test.  Without the plugin (it is listed in
# tests/requirements.txt) no limit is applied.
GPU_TEST_TIMEOUT_S = 900


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        if config.pluginmanager.hasplugin("timeout"):          # pytest-timeout: dump the stacks and leave the process
            for item in items:
                if "gpu" in item.keywords and item.get_closest_marker("timeout") is None:
                    item.add_marker(pytest.mark.timeout(GPU_TEST_TIMEOUT_S, method="thread"))
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def kat_steps():
    return np.load(os.path.join(GOLDEN, "kat_steps.npz"))


@pytest.fixture(scope="session")
def rank_kat():
    return np.load(os.path.join(GOLDEN, "rank_kat.npz"))


@pytest.fixture(scope="session")
def ml100k():
    return np.load(os.path.join(GOLDEN, "ml100k_c1.npz"))


def mf_config(**over):
    """The flat config dict test.py builds (basic.yaml <- mf.yaml <- args), minus file paths."""
    import logging
    cfg = dict(gpu="0", seed=2022, reproducibility=True, algo_name="mf", topk=50, cand_num=1000,
               sample_method="uniform", sample_ratio=0, num_ng=1, batch_size=256, loss_type="BPR",
               init_method="default", optimizer="default", early_stop=False, UID_NAME="user",
               IID_NAME="item", INTER_NAME="rating", TID_NAME="timestamp", factors=32, epochs=3,
               lr=0.01, reg_1=0.001, reg_2=0.001, logger=logging.getLogger("test"), progress=False)
    cfg.update(over)
    return cfg
