#!/usr/bin/env python
"""TF-1.x-style canned-estimator script (LinearClassifier / DNNClassifier / DNNLinearCombinedClassifier over Criteo CSV),
written against the `tensorflow` surface that tf_repos_amd.tf_shim provides -- the same surface the reference's
deep_ctr/Model_pipeline/wide_n_deep.py uses (tf.decode_csv, tf.feature_column.*, the canned estimators).

    python examples/wide_deep_estimator.py --model_type=wide_n_deep --task_type=train --data_dir=/data/criteo_csv/ \
        --embedding_size=32 --deep_layers=256,128,64 --batch_size=128

CSV rows: label, 13 numeric values, 26 integer ids (each in [0, 10000); others fall back to id 0)."""
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tf_repos_amd.tf_shim as _shim  # noqa: E402

_shim.install()
import tensorflow as tf  # noqa: E402

flags = tf.app.flags
FLAGS = flags.FLAGS
flags.DEFINE_string("model_type", "wide_n_deep", "wide | deep | wide_n_deep")
flags.DEFINE_integer("embedding_size", 32, "embedding dimension of the 26 id columns")
flags.DEFINE_string("deep_layers", "256,128,64", "hidden widths of the DNN side")
flags.DEFINE_integer("num_epochs", 1, "epochs")
flags.DEFINE_integer("batch_size", 128, "batch size")
flags.DEFINE_integer("log_steps", 100, "log every n steps")
flags.DEFINE_string("data_dir", "", "directory with tr*csv / va*csv / te*csv")
flags.DEFINE_string("model_dir", "/tmp/wide_deep_model", "checkpoint dir")
flags.DEFINE_string("servable_model_dir", "/tmp/wide_deep_servable", "export dir")
flags.DEFINE_string("task_type", "train", "train | eval | predict | export")

NUMERIC = ["I%d" % i for i in range(1, 14)]
IDS = ["C%d" % i for i in range(14, 40)]
LABEL = "clicked"
COLUMNS = [LABEL] + NUMERIC + IDS
DEFAULTS = [[0.0]] + [[0.0]] * len(NUMERIC) + [[0]] * len(IDS)
BUCKETS = 10000


def input_fn(filenames, num_epochs=1, batch_size=128):
    def decode(line):
        values = tf.decode_csv(line, record_defaults=DEFAULTS)
        features = dict(zip(COLUMNS, values))
        return features, features.pop(LABEL)

    ds = tf.data.TextLineDataset(filenames).map(decode, num_parallel_calls=10).prefetch(100000)
    ds = ds.repeat(num_epochs).batch(batch_size)
    return ds.make_one_shot_iterator().get_next()


def feature_columns():
    numeric = [tf.feature_column.numeric_column(n) for n in NUMERIC]
    ids = [tf.feature_column.categorical_column_with_identity(key=n, num_buckets=BUCKETS, default_value=0) for n in IDS]
    embedded = [tf.feature_column.embedding_column(c, dimension=FLAGS.embedding_size) for c in ids]
    return numeric + ids, numeric + embedded          # (wide columns, deep columns)


def build_estimator(model_dir, model_type):
    wide, deep = feature_columns()
    hidden = [int(h) for h in FLAGS.deep_layers.split(",")]
    config = tf.estimator.RunConfig().replace(log_step_count_steps=FLAGS.log_steps, save_summary_steps=FLAGS.log_steps)
    if model_type == "wide":
        return tf.estimator.LinearClassifier(feature_columns=wide, model_dir=model_dir, config=config)
    if model_type == "deep":
        return tf.estimator.DNNClassifier(hidden_units=hidden, feature_columns=deep, model_dir=model_dir, config=config)
    return tf.estimator.DNNLinearCombinedClassifier(model_dir=model_dir, linear_feature_columns=wide, dnn_feature_columns=deep,
                                                    dnn_hidden_units=hidden, config=config)


def main(_):
    files = lambda prefix: sorted(glob.glob(os.path.join(FLAGS.data_dir, prefix + "*csv")))
    est = build_estimator(FLAGS.model_dir, FLAGS.model_type)
    if FLAGS.task_type == "train":
        train = tf.estimator.TrainSpec(input_fn=lambda: input_fn(files("tr"), FLAGS.num_epochs, FLAGS.batch_size))
        evals = tf.estimator.EvalSpec(input_fn=lambda: input_fn(files("va"), 1, FLAGS.batch_size), steps=None)
        tf.estimator.train_and_evaluate(est, train, evals)
    elif FLAGS.task_type == "eval":
        print(est.evaluate(input_fn=lambda: input_fn(files("va"), 1, FLAGS.batch_size)))
    elif FLAGS.task_type == "predict":
        with open(os.path.join(FLAGS.data_dir, "pred.txt"), "w") as out:
            for p in est.predict(input_fn=lambda: input_fn(files("te"), 1, FLAGS.batch_size), predict_keys="probabilities"):
                out.write("%f\n" % p["probabilities"][1])
    elif FLAGS.task_type == "export":
        wide, deep = feature_columns()
        spec = tf.feature_column.make_parse_example_spec(wide + deep)
        est.export_savedmodel(FLAGS.servable_model_dir, tf.estimator.export.build_parsing_serving_input_receiver_fn(spec))


if __name__ == "__main__":
    tf.logging.set_verbosity(tf.logging.INFO)
    tf.app.run()
