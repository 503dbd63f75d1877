"""Drop-in aliasing: run scripts written against ``import tensorflow as tf`` (TF 1.x) on this framework.

``install()`` registers this package under the module names the reference scripts import:

* ``tensorflow``                                   -> ``distributed_tensorflow_amd``
* ``tensorflow.examples.tutorials.mnist.input_data`` -> ``distributed_tensorflow_amd.data.mnist``
  (``input_data.read_data_sets("MNIST_data", one_hot=True)``)

so that e.g. the reference's ``tfsingle.py`` / ``tfdist_between.py`` / ``tfdist_between_sync.py`` run
unmodified::

    python -m distributed_tensorflow_amd.compat.run /path/to/tfdist_between.py --job_name=worker --task_index=0

Nothing is installed if a real ``tensorflow`` is importable, unless ``force=True``.
"""
from __future__ import annotations

import importlib
import sys
import types


def install(force: bool = False) -> bool:
    if "tensorflow" in sys.modules and not force:
        return getattr(sys.modules["tensorflow"], "__dtf_compat__", False)
    if not force:
        try:
            if importlib.util.find_spec("tensorflow") is not None:
                return False
        except (ImportError, ValueError):
            pass
    import distributed_tensorflow_amd as dtf
    from distributed_tensorflow_amd.data import mnist as mnist_mod

    dtf.__dtf_compat__ = True
    sys.modules["tensorflow"] = dtf
    chain = ["tensorflow.examples", "tensorflow.examples.tutorials", "tensorflow.examples.tutorials.mnist"]
    parent = dtf
    for name in chain:
        mod = types.ModuleType(name)
        sys.modules[name] = mod
        setattr(parent, name.rsplit(".", 1)[-1], mod)
        parent = mod
    sys.modules["tensorflow.examples.tutorials.mnist.input_data"] = mnist_mod
    parent.input_data = mnist_mod
    return True
