"""One reference script's train run on the staged copy (see tools/reference_scripts_gpu.py): examples/sec lines only.
usage (GPU box): python tools/ref_one.py DeepFM.py [batch_size]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.reference_scripts_gpu as R

script = sys.argv[1] if len(sys.argv) > 1 else "DeepFM.py"
bs = sys.argv[2] if len(sys.argv) > 2 else "256"
R.make_data()
extra = dict(R.RUNS)[script] + ["--clear_existing_model=True", "--batch_size=" + bs, "--num_epochs=2"]
R.run(script, extra, "train", "/tmp/ref_model_one/")
