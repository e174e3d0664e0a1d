"""Host-side mirror of the reference's ``dataset`` package for the eval path (``from dataset import reds``)."""
from . import reds  # noqa: F401
