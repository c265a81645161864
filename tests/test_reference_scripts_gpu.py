"""The reference's own Model_pipeline scripts on the GPU, when they have been staged into the git-ignored .ref_stage/ (reference
sources are never committed; /root/reference does not exist on the GPU box -- see tools/reference_scripts_gpu.py, whose full log
is profiles/r02_reference_scripts.txt).  Without the staging directory the test is skipped."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGE = os.path.join(ROOT, ".ref_stage")


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(os.path.join(STAGE, "DeepFM.py")), reason="reference scripts not staged into .ref_stage/")
def test_staged_reference_scripts_train_to_the_oracles_variables(dev):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import reference_scripts_gpu as R
    os.makedirs(R.DATA, exist_ok=True)
    with open(os.path.join(R.DATA, "va.libsvm"), "w") as f:
        f.write(R.synth_lines(1024, 2))
    assert R.part2() <= 2e-5
