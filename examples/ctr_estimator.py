#!/usr/bin/env python
"""TF-1.x-style tf.estimator script for the deep_ctr model family, written against the `tensorflow` surface that
tf_repos_amd.tf_shim provides (the same surface the reference's deep_ctr/Model_pipeline scripts use).

    python examples/ctr_estimator.py --model=deepfm --task_type=train --data_dir=/data/criteo/ --field_size=39 \
        --feature_size=117581 --embedding_size=32 --deep_layers=400,400,400 --dropout=0.5,0.5,0.5 --batch_size=256

`import tensorflow` resolves to the MI355X engine front end; every tf.* call below only builds a graph."""
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tf_repos_amd.tf_shim as _shim  # noqa: E402

_shim.install()
import tensorflow as tf  # noqa: E402

flags = tf.app.flags
FLAGS = flags.FLAGS
flags.DEFINE_string("model", "deepfm", "deepfm | fnn | ipnn | nfm | dcn")
flags.DEFINE_integer("feature_size", 0, "rows of the embedding tables")
flags.DEFINE_integer("field_size", 0, "id:val tokens per line")
flags.DEFINE_integer("embedding_size", 32, "K")
flags.DEFINE_integer("num_epochs", 1, "epochs")
flags.DEFINE_integer("batch_size", 64, "batch size")
flags.DEFINE_integer("log_steps", 100, "log every n steps")
flags.DEFINE_float("learning_rate", 0.0005, "learning rate")
flags.DEFINE_float("l2_reg", 0.0001, "l2 on the tables")
flags.DEFINE_string("optimizer", "Adam", "Adam | Adagrad | Momentum | ftrl")
flags.DEFINE_string("deep_layers", "256,128,64", "hidden widths")
flags.DEFINE_integer("cross_layers", 3, "DCN cross layers")
flags.DEFINE_string("dropout", "0.5,0.5,0.5", "keep_prob per hidden layer")
flags.DEFINE_string("data_dir", "", "directory with tr*/va*/te*libsvm")
flags.DEFINE_string("model_dir", "/tmp/ctr_model", "checkpoint dir")
flags.DEFINE_string("servable_model_dir", "/tmp/ctr_servable", "export dir")
flags.DEFINE_string("task_type", "train", "train | eval | infer | export")


def input_fn(filenames, batch_size=32, num_epochs=1, perform_shuffle=False):
    def decode(line):
        cols = tf.string_split([line], " ")
        label = tf.string_to_number(cols.values[0], out_type=tf.float32)
        kv = tf.string_split(cols.values[1:], ":")
        pairs = tf.reshape(kv.values, kv.dense_shape)
        ids, vals = tf.split(pairs, num_or_size_splits=2, axis=1)
        return {"feat_ids": tf.string_to_number(ids, out_type=tf.int32), "feat_vals": tf.string_to_number(vals, out_type=tf.float32)}, label

    ds = tf.data.TextLineDataset(filenames).map(decode, num_parallel_calls=10).prefetch(500000)
    if perform_shuffle:
        ds = ds.shuffle(buffer_size=256)
    ds = ds.repeat(num_epochs).batch(batch_size)
    return ds.make_one_shot_iterator().get_next()


def _mlp(x, widths, keep, training):
    for i, h in enumerate(widths):
        x = tf.contrib.layers.fully_connected(inputs=x, num_outputs=h, scope="mlp%d" % i)
        if training:
            x = tf.nn.dropout(x, keep_prob=keep[i])
    return x


def model_fn(features, labels, mode, params):
    F, V, K = params["field_size"], params["feature_size"], params["embedding_size"]
    widths = [int(h) for h in params["deep_layers"].split(",")]
    keep = [float(k) for k in params["dropout"].split(",")]
    kind = params["model"]
    training = mode == tf.estimator.ModeKeys.TRAIN
    ids = tf.reshape(features["feat_ids"], shape=[-1, F])
    vals = tf.reshape(features["feat_vals"], shape=[-1, F])
    table = tf.get_variable("emb", shape=[V, K], initializer=tf.glorot_normal_initializer())
    e = tf.multiply(tf.nn.embedding_lookup(table, ids), tf.reshape(vals, shape=[-1, F, 1]))     # [B,F,K]
    regularized = [table]
    if kind != "dcn":
        bias = tf.get_variable("bias", shape=[1], initializer=tf.constant_initializer(0.0))
        linear = tf.get_variable("linear", shape=[V], initializer=tf.glorot_normal_initializer())
        y_linear = tf.reduce_sum(tf.multiply(tf.nn.embedding_lookup(linear, ids), vals), 1)
        regularized.append(linear)
    flat = tf.reshape(e, shape=[-1, F * K])
    with tf.variable_scope("deep"):
        if kind == "deepfm":
            y_fm = 0.5 * tf.reduce_sum(tf.subtract(tf.square(tf.reduce_sum(e, 1)), tf.reduce_sum(tf.square(e), 1)), 1)
            h = _mlp(flat, widths, keep, training)
        elif kind == "fnn":
            h = _mlp(flat, widths, keep, training)
        elif kind == "ipnn":
            rows = [i for i in range(F - 1) for _ in range(i + 1, F)]
            cols = [j for i in range(F - 1) for j in range(i + 1, F)]
            inner = tf.reshape(tf.reduce_sum(tf.gather(e, rows, axis=1) * tf.gather(e, cols, axis=1), [-1]), [-1, len(rows)])
            h = _mlp(tf.concat([flat, inner], 1), widths, keep, training)
        elif kind == "nfm":
            bi = 0.5 * tf.subtract(tf.square(tf.reduce_sum(e, 1)), tf.reduce_sum(tf.square(e), 1))
            if training:
                bi = tf.nn.dropout(bi, keep_prob=keep[0])
            h = _mlp(bi, widths, keep, training)
        elif kind == "dcn":
            cross_b = tf.get_variable("cross_b", shape=[params["cross_layers"], F * K], initializer=tf.glorot_normal_initializer())
            cross_w = tf.get_variable("cross_w", shape=[params["cross_layers"], F * K], initializer=tf.glorot_normal_initializer())
            regularized += [cross_b, cross_w]
            xl = flat
            for l in range(params["cross_layers"]):
                xl = flat * tf.matmul(xl, tf.reshape(cross_w[l], shape=[-1, 1])) + xl + cross_b[l]
            h = tf.concat([xl, _mlp(flat, widths, keep, training)], 1)
        else:
            raise ValueError(kind)
        y_deep = tf.reshape(tf.contrib.layers.fully_connected(inputs=h, num_outputs=1, activation_fn=tf.identity, scope="out"), shape=[-1])
    if kind == "dcn":
        y = y_deep
    else:
        y = bias * tf.ones_like(y_deep, dtype=tf.float32) + y_linear + y_deep
        if kind == "deepfm":
            y = y + y_fm
    predictions = {"prob": tf.sigmoid(y)}
    if mode == tf.estimator.ModeKeys.PREDICT:
        return tf.estimator.EstimatorSpec(mode=mode, predictions=predictions,
                                          export_outputs={"serving_default": tf.estimator.export.PredictOutput(predictions)})
    loss = tf.reduce_mean(tf.nn.sigmoid_cross_entropy_with_logits(logits=y, labels=labels))
    for var in regularized:
        loss = loss + params["l2_reg"] * tf.nn.l2_loss(var)
    if mode == tf.estimator.ModeKeys.EVAL:
        return tf.estimator.EstimatorSpec(mode=mode, predictions=predictions, loss=loss,
                                          eval_metric_ops={"auc": tf.metrics.auc(labels, predictions["prob"])})
    opt = {"Adam": lambda: tf.train.AdamOptimizer(learning_rate=params["learning_rate"], beta1=0.9, beta2=0.999, epsilon=1e-8),
           "Adagrad": lambda: tf.train.AdagradOptimizer(learning_rate=params["learning_rate"], initial_accumulator_value=1e-8),
           "Momentum": lambda: tf.train.MomentumOptimizer(learning_rate=params["learning_rate"], momentum=0.95),
           "ftrl": lambda: tf.train.FtrlOptimizer(params["learning_rate"])}[params["optimizer"]]()
    return tf.estimator.EstimatorSpec(mode=mode, predictions=predictions, loss=loss,
                                      train_op=opt.minimize(loss, global_step=tf.train.get_global_step()))


def build_estimator(params, model_dir, log_steps=100):
    config = tf.estimator.RunConfig().replace(log_step_count_steps=log_steps, save_summary_steps=log_steps)
    return tf.estimator.Estimator(model_fn=model_fn, model_dir=model_dir, params=params, config=config)


def main(_):
    params = {k: getattr(FLAGS, k) for k in ("model", "field_size", "feature_size", "embedding_size", "learning_rate", "l2_reg",
                                               "deep_layers", "dropout", "cross_layers", "optimizer")}
    est = build_estimator(params, FLAGS.model_dir, FLAGS.log_steps)
    tr = sorted(glob.glob("%s/tr*libsvm" % FLAGS.data_dir))
    va = sorted(glob.glob("%s/va*libsvm" % FLAGS.data_dir))
    te = sorted(glob.glob("%s/te*libsvm" % FLAGS.data_dir))
    if FLAGS.task_type == "train":
        tf.estimator.train_and_evaluate(
            est, tf.estimator.TrainSpec(input_fn=lambda: input_fn(tr, num_epochs=FLAGS.num_epochs, batch_size=FLAGS.batch_size)),
            tf.estimator.EvalSpec(input_fn=lambda: input_fn(va, num_epochs=1, batch_size=FLAGS.batch_size), steps=None))
    elif FLAGS.task_type == "eval":
        print(est.evaluate(input_fn=lambda: input_fn(va, num_epochs=1, batch_size=FLAGS.batch_size)))
    elif FLAGS.task_type == "infer":
        with open(FLAGS.data_dir + "/pred.txt", "w") as fo:
            for p in est.predict(input_fn=lambda: input_fn(te, num_epochs=1, batch_size=FLAGS.batch_size), predict_keys="prob"):
                fo.write("%f\n" % (p["prob"]))
    elif FLAGS.task_type == "export":
        spec = {"feat_ids": tf.placeholder(dtype=tf.int64, shape=[None, FLAGS.field_size], name="feat_ids"),
                "feat_vals": tf.placeholder(dtype=tf.float32, shape=[None, FLAGS.field_size], name="feat_vals")}
        print(est.export_savedmodel(FLAGS.servable_model_dir, tf.estimator.export.build_raw_serving_input_receiver_fn(spec)))


if __name__ == "__main__":
    tf.logging.set_verbosity(tf.logging.INFO)
    tf.app.run()
