"""``app.flags`` — the reference's flag surface (tfdist_between.py:11-13).

``DEFINE_string / DEFINE_integer / DEFINE_float / DEFINE_boolean`` register flags
on a process-wide registry; ``FLAGS.name`` parses ``sys.argv`` lazily on first
attribute access (the reference never calls ``tf.app.run()``), accepting both
``--name=value`` and ``--name value``.  Unknown arguments are kept in
``FLAGS.unparsed`` instead of aborting, as with ``parse_known_args``.
``--help`` / ``-h`` / ``--helpfull`` print the registered flags with their defaults and exit.
"""
from __future__ import annotations

import os
import sys
from typing import Any, Callable, Dict, List, Optional


def _parse_bool(v: str) -> bool:
    if v.lower() in ("1", "true", "t", "yes", "y"):
        return True
    if v.lower() in ("0", "false", "f", "no", "n"):
        return False
    raise ValueError(f"not a boolean: {v!r}")


class _FlagValues:
    def __init__(self):
        object.__setattr__(self, "_defs", {})      # name -> (parser, default, help)
        object.__setattr__(self, "_values", {})
        object.__setattr__(self, "_parsed", False)
        object.__setattr__(self, "unparsed", [])

    def _define(self, name: str, parser: Callable[[str], Any], default: Any, help_: str):
        self._defs[name] = (parser, default, help_)
        self._values[name] = default
        object.__setattr__(self, "_parsed", False)

    def _parse(self, argv: Optional[List[str]] = None):
        args = list(sys.argv[1:] if argv is None else argv)
        rest: List[str] = []
        i = 0
        while i < len(args):
            a = args[i]
            if a in ("--help", "-h", "-?", "--helpfull", "--helpshort"):
                sys.stdout.write(self._usage())
                sys.exit(0)
            if a.startswith("--"):
                body = a[2:]
                name, eq, val = body.partition("=")
                key = name.replace("-", "_")
                negated = False
                if key not in self._defs and key.startswith("no") and key[2:] in self._defs:
                    key, negated = key[2:], True
                if key in self._defs:
                    parser = self._defs[key][0]
                    if parser is _parse_bool and not eq:
                        self._values[key] = not negated
                    else:
                        if not eq:
                            i += 1
                            if i >= len(args):
                                raise ValueError(f"flag --{name} needs a value")
                            val = args[i]
                        self._values[key] = parser(val)
                    i += 1
                    continue
            rest.append(a)
            i += 1
        object.__setattr__(self, "unparsed", rest)
        object.__setattr__(self, "_parsed", True)

    def _usage(self) -> str:
        prog = sys.argv[0] if sys.argv and sys.argv[0] else "program"
        lines = [f"usage: {prog} [flags]", "", "flags:"]
        for name in sorted(self._defs):
            parser, default, help_ = self._defs[name]
            kind = {str: "string", int: "integer", float: "float", _parse_bool: "boolean"}.get(parser, "value")
            head = f"  --[no]{name}" if parser is _parse_bool else f"  --{name}=<{kind}>"
            lines.append(f"{head}: {help_}".rstrip(": ") if not help_ else f"{head}: {help_}")
            lines.append(f"    (default: {default!r})")
        return "\n".join(lines) + "\n"

    def __getattr__(self, name: str):
        if name.startswith("_"):
            raise AttributeError(name)
        if not self._parsed:
            self._parse()
        if name not in self._values:
            raise AttributeError(f"unknown flag {name!r}")
        return self._values[name]

    def __setattr__(self, name: str, value: Any):
        if name in self._defs:
            self._values[name] = value
        else:
            object.__setattr__(self, name, value)

    def flag_values_dict(self) -> Dict[str, Any]:
        if not self._parsed:
            self._parse()
        return dict(self._values)

    def _reset(self):
        self._defs.clear()
        self._values.clear()
        object.__setattr__(self, "_parsed", False)


FLAGS = _FlagValues()


def DEFINE_string(name: str, default: Optional[str], help: str = ""):  # noqa: A002,N802
    FLAGS._define(name, str, default, help)


def DEFINE_integer(name: str, default: Optional[int], help: str = ""):  # noqa: A002,N802
    FLAGS._define(name, int, default, help)


def DEFINE_float(name: str, default: Optional[float], help: str = ""):  # noqa: A002,N802
    FLAGS._define(name, float, default, help)


def DEFINE_boolean(name: str, default: Optional[bool], help: str = ""):  # noqa: A002,N802
    FLAGS._define(name, _parse_bool, default, help)


DEFINE_bool = DEFINE_boolean


def DEFINE_enum(name: str, default: Optional[str], enum_values, help: str = ""):  # noqa: A002,N802
    allowed = list(enum_values)

    def parse(v: str) -> str:
        if v not in allowed:
            raise ValueError(f"flag --{name}={v!r}: value should be one of {allowed}")
        return v

    FLAGS._define(name, parse, default, (help + " " if help else "") + "<" + "|".join(allowed) + ">")


class _FlagsModule:
    """Object exposed as ``app.flags`` (``tf.app.flags.DEFINE_string`` ... ``tf.app.flags.FLAGS``)."""

    FLAGS = FLAGS
    DEFINE_string = staticmethod(DEFINE_string)
    DEFINE_integer = staticmethod(DEFINE_integer)
    DEFINE_float = staticmethod(DEFINE_float)
    DEFINE_boolean = staticmethod(DEFINE_boolean)
    DEFINE_bool = staticmethod(DEFINE_boolean)
    DEFINE_enum = staticmethod(DEFINE_enum)


flags = _FlagsModule()


def run(main=None, argv=None):
    """``tf.app.run``: parse flags, call ``main(argv)`` and exit with its return value."""
    FLAGS._parse(argv[1:] if argv else None)
    main = main or sys.modules["__main__"].main
    prof_dir = os.environ.get("DTF_CPROFILE_DIR")
    if not prof_dir:
        sys.exit(main([sys.argv[0]] + FLAGS.unparsed))
    # DTF_CPROFILE_DIR=<dir>: the task runs under cProfile and leaves <dir>/<script>_<job>_<task>.txt (functions by
    # own time and by cumulative time) -- where the host time of a per-batch tier goes, per task of a cluster
    import cProfile
    import io
    import pstats

    pr = cProfile.Profile()
    rc = 0
    try:
        rc = pr.runcall(main, [sys.argv[0]] + FLAGS.unparsed)
    finally:
        os.makedirs(prof_dir, exist_ok=True)
        tag = "_".join(str(v) for v in (os.path.basename(sys.argv[0]).replace(".py", ""),
                                        getattr(FLAGS, "job_name", "") or "local", getattr(FLAGS, "task_index", 0)))
        buf = io.StringIO()
        st = pstats.Stats(pr, stream=buf)
        st.sort_stats("tottime").print_stats(45)
        st.sort_stats("cumtime").print_stats(r"distributed_tensorflow_amd|tfdist|tfsingle", 60)
        with open(os.path.join(prof_dir, tag + ".txt"), "w") as f:
            f.write(buf.getvalue())
    sys.exit(rc)
