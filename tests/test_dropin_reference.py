"""The drop-in route of INTEGRATION.md §1 end to end on the host: the reference's UNMODIFIED
run_examples/test.py, driven by tools/run_daisy_example.py, builds its config / data / sampler and
reaches `fit` of the HIP-backed class - which refuses to run without a device (no CPU fallback).
Needs the reference checkout (build container only); skipped elsewhere."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("DAISY_REFERENCE", "/root/reference")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "daisy")), reason="reference checkout not present")
@pytest.mark.skipif(torch.cuda.is_available(), reason="host-only check (with a device the run would train)")
@pytest.mark.parametrize("algo", ["mf", "neumf"])
def test_reference_driver_reaches_the_hip_classes(tmp_path, algo):
    d = tmp_path / "daisy_checkout"                        # writable cwd: test.py writes ./log ./res
    d.mkdir()
    for name in ("daisy", "run_examples", "data"):
        os.symlink(os.path.join(REF, name), d / name)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_daisy_example.py"), "--daisy", str(d), "--",
                        "--algo_name", algo, "--epochs", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "daisyrec_amd" in r.stderr and "no HIP device visible" in r.stderr, r.stderr[-2000:]
    assert "model.fit(train_loader)" in r.stderr            # failed inside the reference driver's fit call


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "daisy")), reason="reference checkout not present")
@pytest.mark.skipif(torch.cuda.is_available(), reason="host-only check (with a device the run would train)")
def test_reference_driver_with_the_front_end_rebound(tmp_path):
    """--native-front-end: the driver's own `get_ur(train_set)` / `get_ur(test_set)` calls (test.py:68-71) go through
    the mirrors (one sort instead of a row loop) and the run still reaches the HIP `fit` with a well-formed config"""
    d = tmp_path / "daisy_checkout"
    d.mkdir()
    for name in ("daisy", "run_examples", "data"):
        os.symlink(os.path.join(REF, name), d / name)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_daisy_example.py"), "--daisy", str(d),
                        "--native-front-end", "--", "--algo_name", "mf", "--epochs", "1"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "no HIP device visible" in r.stderr and "model.fit(train_loader)" in r.stderr, r.stderr[-2000:]



@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "daisy")), reason="reference checkout not present")
@pytest.mark.skipif(torch.cuda.is_available(), reason="host-only check (with a device the run would train)")
@pytest.mark.parametrize("algo", ["mf", "neumf"])
def test_reference_tune_driver_reaches_the_hip_classes(tmp_path, algo):
    """run_examples/tune.py:136-221 UNMODIFIED (optuna is absent here, so the tune path was unpinned): with
    a stand-in `optuna` that runs ONE trial, `objective(trial)` executes its own call order - suggest the parameters
    named in --tune_pack, ValidationSplitter fold, get_ur, `model_config[algo](config)`, BasicNegtiveSampler,
    BasicDataset, get_dataloader, `model.fit(train_loader)` - and reaches `fit` of the HIP-backed class, which refuses
    to run without a device."""
    d = tmp_path / "daisy_checkout"
    d.mkdir()
    for name in ("daisy", "run_examples", "data"):
        os.symlink(os.path.join(REF, name), d / name)
    pack = '{"factors": [16, 32], "lr": {"min": 0.01, "max": 0.05, "step": null}, "num_ng": {"min": 1, "max": 2, "step": 1}}'
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_daisy_example.py"), "--daisy", str(d),
                        "--script", "run_examples/tune.py",
                        "--extra-path", os.path.join(ROOT, "tests", "golden", "_shims_optuna"), "--",
                        "--algo_name", algo, "--epochs", "1", "--hyperopt_trail", "1", "--tune_pack", pack],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "daisyrec_amd" in r.stderr and "no HIP device visible" in r.stderr, r.stderr[-3000:]
    assert "model.fit(train_loader)" in r.stderr and "in objective" in r.stderr, r.stderr[-3000:]
