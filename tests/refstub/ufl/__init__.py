"""Recording stub of `ufl` (only what FenicsSolver imports by name); see tests/refstub/dolfin."""
from . import tensors  # noqa: F401
