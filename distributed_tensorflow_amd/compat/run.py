"""Run a TF-1.x script unmodified on this framework.

    python -m distributed_tensorflow_amd.compat.run SCRIPT.py [--dtf-set NAME=VALUE ...] [script args]

``--dtf-set epochs=1`` overrides a module-level ``NAME = <literal>`` assignment of the script (the reference
scripts keep their hyper-parameters as module constants, tfdist_between.py:19-22) without editing the file.
``settings`` (``ps_svrs`` / ``worker_svrs``) can be overridden with DTF_PS_SVRS / DTF_WORKER_SVRS.
"""
from __future__ import annotations

import os
import re
import sys
import types


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        print(__doc__)
        return 2
    script = argv.pop(0)
    overrides = {}
    rest = []
    i = 0
    while i < len(argv):
        if argv[i] == "--dtf-set" and i + 1 < len(argv):
            k, _, v = argv[i + 1].partition("=")
            overrides[k] = v
            i += 2
        else:
            rest.append(argv[i])
            i += 1
    from distributed_tensorflow_amd import compat

    compat.install(force=True)
    if os.environ.get("DTF_PS_SVRS") is not None or os.environ.get("DTF_WORKER_SVRS") is not None:
        s = types.ModuleType("settings")
        s.ps_svrs = [a for a in os.environ.get("DTF_PS_SVRS", "").split(",") if a]
        s.worker_svrs = [a for a in os.environ.get("DTF_WORKER_SVRS", "").split(",") if a]
        sys.modules["settings"] = s
    with open(script) as f:
        src = f.read()
    for k, v in overrides.items():
        src, n = re.subn(rf"(?m)^{re.escape(k)}\s*=\s*.+$", f"{k} = {v}", src, count=1)
        if n == 0:
            raise SystemExit(f"--dtf-set {k}: no module-level assignment found in {script}")
    sys.argv = [script] + rest
    sys.path.insert(0, os.path.dirname(os.path.abspath(script)))
    glb = {"__name__": "__main__", "__file__": script, "__builtins__": __builtins__}
    exec(compile(src, script, "exec"), glb)
    return 0


if __name__ == "__main__":
    sys.exit(main())
