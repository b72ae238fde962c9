"""Trainer.fit on the GPU through the Lightning boundary (see tests/lightning_fit_script.py): with the strict
pytorch_lightning stand-in and with tacorl_amd.lightning.MiniTrainer."""
import os
import sys

import pytest

from tests.proc_util import run_group

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("flavour", ["standin", "mini"])
def test_trainer_fit_and_resume(flavour):
    env = dict(os.environ)
    pp = [ROOT] + ([os.path.join(ROOT, "tests", "fake_pl")] if flavour == "standin" else [])
    env["PYTHONPATH"] = os.pathsep.join(pp + [env.get("PYTHONPATH", "")])
    out = run_group([sys.executable, os.path.join(ROOT, "tests", "lightning_fit_script.py"), flavour], env, ROOT, timeout=420)
    assert out.returncode == 0 and "ALL OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
