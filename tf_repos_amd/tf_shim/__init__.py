"""`import tensorflow as tf` for the reference's deep_ctr scripts, served by the MI355X engine.

    import tf_repos_amd.tf_shim as shim; shim.install()      # registers sys.modules['tensorflow']
    # ... then the reference-style script runs unchanged at its tf.* call sites (SURVEY 8b symbol census).

Only the TF-1.x symbols the deep_ctr Model_pipeline touches exist; they build a symbolic graph that is lowered onto
libdeepctr_hip.so (lowering.py).  Unknown symbols raise AttributeError -- never a silent fallback.
"""
from __future__ import annotations

import sys
import types

from .. import errors as _errors
from . import canned as _canned
from . import data as _data
from . import estimator as _est
from . import flags as FLAGS_MODULE
from . import graph as _g
from . import logging


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    return m


def build_module():
    tf = _mod("tensorflow", __version__="1.4.0-tf_repos_amd")
    # dtypes / graph / variables
    for n in ("float32", "int32", "int64", "string"):
        setattr(tf, n, getattr(_g, n))
    tf.bool = _g.bool_
    for n in ("get_variable", "variable_scope", "name_scope", "glorot_normal_initializer", "glorot_uniform_initializer",
              "constant_initializer", "zeros_initializer", "ones_initializer", "placeholder", "constant", "reshape", "multiply",
              "add", "subtract", "square", "reduce_sum", "reduce_mean", "matmul", "concat", "stack", "transpose", "gather", "einsum",
              "ones_like", "identity", "sigmoid", "cast", "cond", "split", "string_split", "string_to_number", "decode_csv"):
        setattr(tf, n, getattr(_g, n))
    for n in ("FixedLenFeature", "VarLenFeature", "parse_single_example", "sparse_tensor_to_dense", "AUTO_REUSE", "expand_dims", "shape",
              "tile"):
        setattr(tf, n, getattr(_g, n))
    tf.losses = _mod("tensorflow.losses", log_loss=_g.log_loss)
    tf.summary = _mod("tensorflow.summary", scalar=_g.summary_scalar)
    tf.Graph = _g.Graph
    tf.nn = _mod("tensorflow.nn", embedding_lookup=_g.embedding_lookup, dropout=_g.dropout, softmax=_g.softmax, l2_loss=_g.l2_loss,
                 sigmoid_cross_entropy_with_logits=_g.sigmoid_cross_entropy_with_logits, relu=_g.relu, sigmoid=_g.sigmoid,
                 embedding_lookup_sparse=_g.embedding_lookup_sparse)
    layers = _mod("tensorflow.contrib.layers", fully_connected=_g.fully_connected, l2_regularizer=_g.l2_regularizer, batch_norm=_g.batch_norm)
    tf.contrib = _mod("tensorflow.contrib", layers=layers)
    from .. import tfrecord as _tfr
    tf.train = _mod("tensorflow.train", AdamOptimizer=_g.AdamOptimizer, AdagradOptimizer=_g.AdagradOptimizer,
                    MomentumOptimizer=_g.MomentumOptimizer, FtrlOptimizer=_g.FtrlOptimizer, get_global_step=_g.get_global_step,
                    get_or_create_global_step=_g.get_or_create_global_step,
                    # the message classes of Feature_pipeline/get_tfrecord.py:52-95
                    Example=_tfr.Example, Features=_tfr.Features, Feature=_tfr.Feature, Int64List=_tfr.Int64List,
                    FloatList=_tfr.FloatList, BytesList=_tfr.BytesList)
    tf.python_io = _mod("tensorflow.python_io", TFRecordWriter=_tfr.TFRecordWriter)
    tf.metrics = _mod("tensorflow.metrics", auc=_g.metrics_auc)
    tf.data = _mod("tensorflow.data", TextLineDataset=_data.TextLineDataset, TFRecordDataset=_data.TFRecordDataset)
    export = _mod("tensorflow.estimator.export", PredictOutput=_est.PredictOutput, ServingInputReceiver=_est.ServingInputReceiver,
                  build_raw_serving_input_receiver_fn=_est.build_raw_serving_input_receiver_fn,
                  build_parsing_serving_input_receiver_fn=_canned.build_parsing_serving_input_receiver_fn)
    tf.estimator = _mod("tensorflow.estimator", Estimator=_est.Estimator, EstimatorSpec=_est.EstimatorSpec, ModeKeys=_est.ModeKeys,
                        RunConfig=_est.RunConfig, TrainSpec=_est.TrainSpec, EvalSpec=_est.EvalSpec,
                        train_and_evaluate=_est.train_and_evaluate, export=export,
                        LinearClassifier=_canned.LinearClassifier, DNNClassifier=_canned.DNNClassifier,
                        DNNLinearCombinedClassifier=_canned.DNNLinearCombinedClassifier)
    tf.feature_column = _mod("tensorflow.feature_column", numeric_column=_canned.numeric_column,
                             categorical_column_with_identity=_canned.categorical_column_with_identity,
                             embedding_column=_canned.embedding_column, make_parse_example_spec=_canned.make_parse_example_spec)
    tf.ConfigProto = _est.ConfigProto
    sigc = _mod("tensorflow.saved_model.signature_constants", DEFAULT_SERVING_SIGNATURE_DEF_KEY="serving_default")
    tf.saved_model = _mod("tensorflow.saved_model", signature_constants=sigc)
    tf.app = _mod("tensorflow.app", flags=FLAGS_MODULE, run=FLAGS_MODULE.run)
    tf.flags = FLAGS_MODULE
    tf.logging = logging
    tf.errors = _errors
    return tf


def install(force: bool = False):
    """Registers the shim as `tensorflow` (refuses to shadow a real TensorFlow unless force=True)."""
    if "tensorflow" in sys.modules and not force and not getattr(sys.modules["tensorflow"], "__version__", "").endswith("tf_repos_amd"):
        raise RuntimeError("a real tensorflow is already imported")
    tf = build_module()
    sys.modules["tensorflow"] = tf
    for sub in ("nn", "contrib", "train", "metrics", "data", "estimator", "saved_model", "app", "feature_column", "losses", "summary", "python_io"):
        sys.modules["tensorflow." + sub] = getattr(tf, sub)
    sys.modules["tensorflow.contrib.layers"] = tf.contrib.layers
    sys.modules["tensorflow.estimator.export"] = tf.estimator.export
    return tf
