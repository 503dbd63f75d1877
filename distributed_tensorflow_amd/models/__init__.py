"""Model families: the reference's 784-100-10 MLP (one or two launches per step) and fully connected stacks of any
depth and width on the dense-layer kernels."""
from .dense import DenseStackEngine  # noqa: F401
from .mlp import MLPEngine  # noqa: F401
