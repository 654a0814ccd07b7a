"""ultra_amd -- MI355X-native engine for ULTRA's relational message passing (rspmm + NBFNet layers).

The package mirrors the reference's module layout for the hot path only:
  ultra_amd.rspmm    <- ultra/rspmm/rspmm.py        (generalized_rspmm, RSPMM*Function, `rspmm` exports)
  ultra_amd.layers   <- ultra/layers.py             (GeneralizedRelationalConv)
  ultra_amd.models   <- ultra/models.py, ultra/base_nbfnet.py (Ultra, RelNBFNet, EntityNBFNet, QueryNBFNet)
  ultra_amd.tasks    <- ultra/tasks.py              (evaluation glue around the path)
Importing it loads libultra_amd.so and fails loudly when the library is missing.
"""
from . import _lib  # noqa: F401  (loads the HIP library; raises ImportError if it was not built)
from .rspmm import generalized_rspmm  # noqa: F401

__all__ = ["generalized_rspmm"]
