"""Model families.  The reference has exactly one: the 784-100-10 MLP."""
from .mlp import MLPEngine  # noqa: F401
