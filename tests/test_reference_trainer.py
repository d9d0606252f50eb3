"""CPU, build container only: the reference's UNMODIFIED UNetTrainer drives the drop-in modules end to end (SURVEY.md §7
step 7, Appendix B; trainer.py:93-440) — train iterations, validation under no_grad, the reference's save_checkpoint,
strict load into a fresh drop-in model, resume through the trainer's own `resume=` path."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT
from ref_import import reference_available

pytestmark = pytest.mark.skipif(not reference_available(), reason="/root/reference only exists in the build container")


@pytest.mark.timeout(300)
def test_unmodified_reference_trainer_runs_the_drop_in(tmp_path):
    env = dict(os.environ, OMP_NUM_THREADS="4")
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "drive_reference_trainer.py"), str(tmp_path), "cpu"],
                          capture_output=True, text=True, timeout=280, env=env)
    assert proc.returncode == 0, proc.stderr[-3000:]
    line = [ln for ln in proc.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    r = json.loads(line[len("RESULT "):])
    assert r["iterations"] == 5                       # max_num_iterations=4 -> stops after the 5th (trainer.py:296-299)
    assert r["params_moved"] == r["n_params"] == r["checkpoint_keys"]  # every parameter trained and checkpointed
    assert r["eval_calls"] >= 2 * 2 + 4               # 2 validation passes x 2 batches + the per-iteration train metric
    assert r["resumed_from_iteration"] == 4 and r["resumed_to_iteration"] > 4
    assert r["wrapped_dataparallel"] is False
