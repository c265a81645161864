"""tf.app.flags / tf.app.run (DeepFM.py:34-60, 368-370): module-level DEFINE_* and a global FLAGS object."""
from __future__ import annotations

import sys


class _Flags:
    def __init__(self):
        object.__setattr__(self, "_defs", {})
        object.__setattr__(self, "_vals", {})
        object.__setattr__(self, "_parsed", False)

    def _define(self, name, default, help_, typ):
        self._defs[name] = (typ, default, help_)
        self._vals[name] = default

    def __getattr__(self, name):
        vals = object.__getattribute__(self, "_vals")
        if name in vals:
            return vals[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        self._vals[name] = value

    def _parse(self, argv):
        rest = [argv[0]] if argv else []
        i = 1
        while i < len(argv):
            a = argv[i]
            if a.startswith("--"):
                body = a[2:]
                if "=" in body:
                    k, v = body.split("=", 1)
                else:
                    k, v = body, None
                if k in self._defs:
                    typ = self._defs[k][0]
                    if v is None:
                        if typ is bool:
                            v = "true"
                        else:
                            i += 1
                            v = argv[i]
                    if typ is bool:
                        self._vals[k] = str(v).lower() in ("1", "true", "t", "yes")
                    else:
                        self._vals[k] = typ(v)
                    i += 1
                    continue
                if k.startswith("no") and k[2:] in self._defs and self._defs[k[2:]][0] is bool:
                    self._vals[k[2:]] = False
                    i += 1
                    continue
            rest.append(a)
            i += 1
        object.__setattr__(self, "_parsed", True)
        return rest

    def _reset(self):
        self._defs.clear()
        self._vals.clear()


FLAGS = _Flags()


def DEFINE_integer(name, default, help=""): FLAGS._define(name, default, help, int)
def DEFINE_float(name, default, help=""): FLAGS._define(name, default, help, float)
def DEFINE_string(name, default, help=""): FLAGS._define(name, default, help, str)
def DEFINE_boolean(name, default, help=""): FLAGS._define(name, default, help, bool)


DEFINE_bool = DEFINE_boolean


def run(main=None, argv=None):
    rest = FLAGS._parse(list(sys.argv if argv is None else argv))
    main = main or sys.modules["__main__"].main
    sys.exit(main(rest))
