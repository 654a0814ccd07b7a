"""Test-only stand-in for torch_geometric (see ../README.md)."""
__version__ = "2.5.0"
from . import data, nn, utils  # noqa
