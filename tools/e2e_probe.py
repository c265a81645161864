"""bench.py's end_to_end block alone (tf.estimator.Estimator.train over a libsvm text file; argv: config [c2], epochs [40]): whole-call and steady-state rates.
A/B knobs: DCTR_EST_STEP_THREAD=0 (steps enqueued by the Python loop itself), DCTR_EST_MAIN_STREAM=0 (legacy default stream)."""
import json
import sys

sys.path.insert(0, '.')
import bench

cfg = sys.argv[1] if len(sys.argv) > 1 else 'c2'
C1 = dict(model="deepfm", field_size=39, feature_size=117_581, embedding_size=8, batch=256, deep_layers=(400, 400, 400), dropout=(0.5, 0.5, 0.5),
          l2_reg=1e-4, learning_rate=5e-4, optimizer="Adam")           # BASELINE configs[0]: the reference's own run (run.sh / DeepFM.py defaults)
r = bench.end_to_end(dict(C1 if cfg == 'c1' else bench.CONFIGS[cfg]), epochs=int(sys.argv[2]) if len(sys.argv) > 2 else 40)
print(json.dumps({k: v for k, v in r.items() if k != 'what'}))
