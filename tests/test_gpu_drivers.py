"""drivers/train.py end to end on the GPU (the trainer had no GPU test and ignored --validiter):
a few steps of the reference architecture on small clips, the validation pass of train.py:243-264 (CC / SIM / AUC_Judd
of the last frames), a TF-format checkpoint with its `checkpoint` state file, and --pretrain resuming from it."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _train(cwd, *args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "drivers", "train.py"), "--batch", "2", "--imagesize", "32", "32",
                        "--steps", "4", "--plotiter", "2", "--validclips", "2"] + list(args),
                       cwd=cwd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    return r.stdout


def test_train_driver_validates_saves_and_resumes(tmp_path):
    out = _train(str(tmp_path), "--info", "a", "--validiter", "2", "--saveiter", "4")
    assert out.count("Doing validation...") == 2
    m = re.findall(r"Step: (\d+)\s+Metrics: (\S+) (\S+) (\S+)", out)
    assert [int(s) for s, *_ in m] == [2, 4]
    for _, cc, sim, auc in m:
        cc, sim, auc = float(cc), float(sim), float(auc)
        assert -1.0 <= cc <= 1.0 and 0.0 <= sim <= 1.0 and 0.0 <= auc <= 1.0
    d = tmp_path / "model" / "a"
    assert (d / "checkpoint").read_text().startswith('model_checkpoint_path: "p3d_4.ckpt"')
    assert (d / "p3d_4.ckpt.index").exists() and (d / "p3d_4.ckpt.data-00000-of-00001").exists()
    from sap3d_tensorflow_amd import tf_checkpoint as tfc
    names = [n for n, _, _ in tfc.list_variables(str(d / "p3d_4.ckpt"))]
    assert "firstconv1" in names and "batch_normalization/moving_mean" in names and len(names) > 900
    losses_a = [float(v) for v in re.findall(r"Training Loss (\S+)", out)]
    # resume from the directory, as `--pretrain <run>` does in the reference (train.py:204-210)
    out_b = _train(str(tmp_path), "--info", "b", "--pretrain", str(d), "--validiter", "100", "--saveiter", "100")
    assert "Using this model to retrain" in out_b
    losses_b = [float(v) for v in re.findall(r"Training Loss (\S+)", out_b)]
    assert np.all(np.isfinite(losses_a)) and np.all(np.isfinite(losses_b))
    assert losses_b[0] < losses_a[0]            # the restored model starts from where the first run got to
