"""Host-side mirror of the reference's ``las`` package (las/__init__.py): ``las.ops`` and ``las.model``."""
from . import ops  # noqa: F401
