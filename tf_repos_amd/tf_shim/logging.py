"""tf.logging.{set_verbosity, INFO, info, ...} (DeepFM.py:369)."""
import sys
import time

DEBUG, INFO, WARN, ERROR, FATAL = 10, 20, 30, 40, 50
_level = WARN


def set_verbosity(v):
    global _level
    _level = v


def _log(tag, lvl, msg, *args):
    if lvl >= _level:
        sys.stderr.write("%s:tensorflow:%s\n" % (tag, (msg % args) if args else msg))


def info(msg, *a): _log("INFO", INFO, msg, *a)
def warn(msg, *a): _log("WARNING", WARN, msg, *a)
def warning(msg, *a): _log("WARNING", WARN, msg, *a)
def error(msg, *a): _log("ERROR", ERROR, msg, *a)
def debug(msg, *a): _log("DEBUG", DEBUG, msg, *a)
