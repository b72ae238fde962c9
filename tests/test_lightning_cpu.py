"""The Lightning side of the drop-in boundary without a GPU (reference scripts/train.py:28-66,
cql_offline_lightning.py:24,115-116,553-574): the module classes are LightningModules (of the real package when it is
importable - here of a strict stand-in on PYTHONPATH - else of tacorl_amd.lightning's own), are built from the
reference's composed dict configs by `instantiate`, hand torch.optim.Optimizers to the trainer, survive Trainer.fit's
checks and round-trip through a PL-layout checkpoint.  Modules are built on the CPU: parameters, optimiser state and
configuration are host-testable; stepping needs the GPU (tests/test_lightning_gpu.py)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("flavour", ["standin", "mini"])
def test_lightning_surface(flavour):
    env = dict(os.environ)
    pp = [ROOT] + ([os.path.join(ROOT, "tests", "fake_pl")] if flavour == "standin" else [])
    env["PYTHONPATH"] = os.pathsep.join(pp + [env.get("PYTHONPATH", "")])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "lightning_script.py"), "cpu", flavour], env=env,
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ALL OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
