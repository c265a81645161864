"""DIN-style (field-wise sum pooling) and ESMM-style estimators over TFRecord input, written against the TF-1.x Estimator API
the way the reference's DIN.py / DeepCvrMTL.py are, and served by the MI355X engine through tf_repos_amd.tf_shim:

    import tf_repos_amd.tf_shim as shim; shim.install()
    python examples/multihot_estimator.py --task=esmm --data_dir=... --model_dir=...

Each Example holds: y, z (float), feat_ids int64[field_size], four weighted multi-hot user features (u_*ids / u_*vals),
three single ad ids (a_catids, a_shopids, a_brandids) and one unweighted multi-hot ad feature (a_intids)."""
import glob
import sys

import tensorflow as tf

USER_MULTI = ("u_cat", "u_shop", "u_brand", "u_int")
AD_SINGLE = ("a_cat", "a_shop", "a_brand")


def input_fn(filenames, batch_size=32, num_epochs=1, perform_shuffle=False, field_size=11, with_z=True):
    def parse(record):
        spec = {"y": tf.FixedLenFeature([], tf.float32), "z": tf.FixedLenFeature([], tf.float32),
                "feat_ids": tf.FixedLenFeature([field_size], tf.int64), "a_intids": tf.VarLenFeature(tf.int64)}
        for u in USER_MULTI:
            spec[u + "ids"] = tf.VarLenFeature(tf.int64)
            spec[u + "vals"] = tf.VarLenFeature(tf.float32)
        for a in AD_SINGLE:
            spec[a + "ids"] = tf.FixedLenFeature([], tf.int64)
        parsed = tf.parse_single_example(record, spec)
        y, z = parsed.pop("y"), parsed.pop("z")
        return parsed, ({"y": y, "z": z} if with_z else y)

    ds = tf.data.TFRecordDataset(filenames).map(parse, num_parallel_calls=8).prefetch(100000)
    if perform_shuffle:
        ds = ds.shuffle(buffer_size=256)
    return ds.repeat(num_epochs).batch(batch_size).make_one_shot_iterator().get_next()


def _attend(table, sp_ids, sp_vals, query, params, mode):
    """DIN-style attention pooling of one behaviour list against the candidate ad's embedding: every behaviour is scored by a
    small MLP on [behaviour, behaviour - ad, ad], squashed by a sigmoid, and the list is summed with those scores."""
    K = params["embedding_size"]
    widths = [int(v) for v in params["attention_layers"].split(",")]
    keep = [float(v) for v in params["dropout"].split(",")]
    ids = tf.sparse_tensor_to_dense(sp_ids)
    weights = tf.expand_dims(tf.sparse_tensor_to_dense(sp_vals), axis=-1)
    behaviours = tf.multiply(tf.nn.embedding_lookup(table, ids), weights)                 # [B, P, K], zero at the padding
    present = tf.expand_dims(tf.cast(ids > 0, tf.float32), axis=-1)
    longest = tf.shape(ids)[1]
    flat = tf.reshape(behaviours, [-1, K])
    ad = tf.reshape(tf.tile(query, [1, longest]), [-1, K])
    h = tf.concat([flat, flat - ad, ad], axis=1)
    for i, width in enumerate(widths):
        h = tf.contrib.layers.fully_connected(h, width, scope="score_fc%d" % i)
        if mode == tf.estimator.ModeKeys.TRAIN:
            h = tf.nn.dropout(h, keep_prob=keep[i])
    score = tf.reshape(tf.contrib.layers.fully_connected(h, 1, activation_fn=tf.sigmoid, scope="score_out"), [-1, longest, 1])
    return tf.reduce_sum(tf.multiply(tf.multiply(behaviours, score), present), 1)


def _embed(features, params, mode=None, attention=False):
    table = tf.get_variable("embeddings", [params["feature_size"], params["embedding_size"]], initializer=tf.glorot_normal_initializer())
    K = params["embedding_size"]
    ads = [tf.nn.embedding_lookup(table, features[a + "ids"]) for a in AD_SINGLE]
    ads.append(tf.nn.embedding_lookup_sparse(table, sp_ids=features["a_intids"], sp_weights=None, combiner="sum"))
    parts = [tf.reshape(tf.nn.embedding_lookup(table, features["feat_ids"]), [-1, params["field_size"] * K])]
    with tf.variable_scope("pooling", reuse=tf.AUTO_REUSE):
        for u, ad in zip(USER_MULTI, ads):
            if attention:
                parts.append(_attend(table, features[u + "ids"], features[u + "vals"], ad, params, mode))
            else:
                parts.append(tf.nn.embedding_lookup_sparse(table, sp_ids=features[u + "ids"], sp_weights=features[u + "vals"], combiner="sum"))
    return table, tf.concat(parts + ads, axis=1)


def _tower(x, params, mode, prefix):
    layers = [int(v) for v in params["deep_layers"].split(",")]
    keep = [float(v) for v in params["dropout"].split(",")]
    for i, width in enumerate(layers):
        x = tf.contrib.layers.fully_connected(x, width, scope="%smlp%d" % (prefix, i))
        if mode == tf.estimator.ModeKeys.TRAIN:
            x = tf.nn.dropout(x, keep_prob=keep[i])
    return tf.reshape(tf.contrib.layers.fully_connected(x, 1, activation_fn=tf.identity, scope=prefix + "out"), [-1])


def _optimizer(params):
    lr = params["learning_rate"]
    kind = params.get("optimizer", "Adam")
    if kind == "Adam":
        return tf.train.AdamOptimizer(lr, beta1=0.9, beta2=0.999, epsilon=1e-8)
    if kind == "Adagrad":
        return tf.train.AdagradOptimizer(lr, initial_accumulator_value=1e-8)
    if kind == "Momentum":
        return tf.train.MomentumOptimizer(lr, momentum=0.95)
    return tf.train.FtrlOptimizer(lr)


def din_model_fn(features, labels, mode, params):
    table, x = _embed(features, params, mode, attention=bool(params.get("attention_layers")))
    logit = _tower(x, params, mode, "din_")
    predictions = {"prob": tf.sigmoid(logit)}
    if mode == tf.estimator.ModeKeys.PREDICT:
        return tf.estimator.EstimatorSpec(mode=mode, predictions=predictions)
    loss = tf.reduce_mean(tf.nn.sigmoid_cross_entropy_with_logits(logits=logit, labels=labels)) + params["l2_reg"] * tf.nn.l2_loss(table)
    if mode == tf.estimator.ModeKeys.EVAL:
        return tf.estimator.EstimatorSpec(mode=mode, predictions=predictions, loss=loss,
                                          eval_metric_ops={"auc": tf.metrics.auc(labels, predictions["prob"])})
    train_op = _optimizer(params).minimize(loss, global_step=tf.train.get_global_step())
    return tf.estimator.EstimatorSpec(mode=mode, predictions=predictions, loss=loss, train_op=train_op)


def esmm_model_fn(features, labels, mode, params):
    table, x = _embed(features, params)
    y_cvr = _tower(x, params, mode, "cvr_")
    y_ctr = _tower(x, params, mode, "ctr_")
    pctr, pcvr = tf.sigmoid(y_ctr), tf.sigmoid(y_cvr)
    predictions = {"pctr": pctr, "pcvr": pcvr, "pctcvr": pctr * pcvr}
    if mode == tf.estimator.ModeKeys.PREDICT:
        return tf.estimator.EstimatorSpec(mode=mode, predictions=predictions)
    y, z = labels["y"], labels["z"]
    w = params["ctr_task_wgt"]
    ctr_loss = tf.reduce_mean(tf.nn.sigmoid_cross_entropy_with_logits(logits=y_ctr, labels=y))
    cvr_loss = tf.reduce_mean(tf.losses.log_loss(predictions=predictions["pctcvr"], labels=z))
    loss = w * ctr_loss + (1 - w) * cvr_loss + params["l2_reg"] * tf.nn.l2_loss(table)
    if mode == tf.estimator.ModeKeys.EVAL:
        metrics = {"CTR_AUC": tf.metrics.auc(y, pctr), "CVR_AUC": tf.metrics.auc(z, pcvr), "CTCVR_AUC": tf.metrics.auc(z, predictions["pctcvr"])}
        return tf.estimator.EstimatorSpec(mode=mode, predictions=predictions, loss=loss, eval_metric_ops=metrics)
    train_op = _optimizer(params).minimize(loss, global_step=tf.train.get_global_step())
    return tf.estimator.EstimatorSpec(mode=mode, predictions=predictions, loss=loss, train_op=train_op)


def build_estimator(task, params, model_dir, log_steps=100):
    config = tf.estimator.RunConfig().replace(log_step_count_steps=log_steps)
    return tf.estimator.Estimator(model_fn=esmm_model_fn if task == "esmm" else din_model_fn, model_dir=model_dir, params=params, config=config)


def main(_):
    F = tf.app.flags.FLAGS
    params = dict(field_size=F.field_size, feature_size=F.feature_size, embedding_size=F.embedding_size, deep_layers=F.deep_layers,
                  dropout=F.dropout, l2_reg=F.l2_reg, learning_rate=F.learning_rate, optimizer=F.optimizer, ctr_task_wgt=F.ctr_task_wgt,
                  attention_layers=F.attention_layers if F.task == "din" else "")
    est = build_estimator(F.task, params, F.model_dir)
    tr = glob.glob("%s/tr/*tfrecord" % F.data_dir)
    va = glob.glob("%s/te/*tfrecord" % F.data_dir)
    fn = lambda files, epochs: (lambda: input_fn(files, F.batch_size, epochs, field_size=F.field_size, with_z=F.task == "esmm"))
    est.train(fn(tr, F.num_epochs))
    print(est.evaluate(fn(va, 1)))


if __name__ == "__main__":
    fl = tf.app.flags
    fl.DEFINE_string("task", "esmm", "din | esmm")
    fl.DEFINE_string("data_dir", "", "")
    fl.DEFINE_string("model_dir", "/tmp/multihot_model", "")
    fl.DEFINE_integer("field_size", 11, "")
    fl.DEFINE_integer("feature_size", 4500000, "")
    fl.DEFINE_integer("embedding_size", 16, "")
    fl.DEFINE_integer("batch_size", 1024, "")
    fl.DEFINE_integer("num_epochs", 1, "")
    fl.DEFINE_string("deep_layers", "256,128,64", "")
    fl.DEFINE_string("dropout", "0.5,0.5,0.5", "")
    fl.DEFINE_float("l2_reg", 1e-4, "")
    fl.DEFINE_float("learning_rate", 5e-4, "")
    fl.DEFINE_float("ctr_task_wgt", 0.5, "")
    fl.DEFINE_string("optimizer", "Adam", "")
    fl.DEFINE_string("attention_layers", "64", "din: widths of the attention-pooling MLP ('' = plain sum pooling)")
    tf.app.run(main)
