"""ultra_amd -- MI355X-native engine for ULTRA's relational message passing (rspmm + NBFNet layers).

The package mirrors the reference's module layout for the hot path only:
  ultra_amd.rspmm    <- ultra/rspmm/rspmm.py        (generalized_rspmm, RSPMM*Function, `rspmm` exports)
  ultra_amd.layers   <- ultra/layers.py             (GeneralizedRelationalConv)
  ultra_amd.models   <- ultra/models.py, ultra/base_nbfnet.py (Ultra, RelNBFNet, EntityNBFNet, QueryNBFNet)
  ultra_amd.tasks    <- ultra/tasks.py              (evaluation glue around the path)

`ultra_amd.build` (hipcc driver) must be importable before the library exists, so the HIP library is
loaded by the first functional submodule (`ultra_amd._lib`, pulled in by rspmm / layers / models / dense),
which raises ImportError when libultra_amd.so is missing or stale: there is no fallback path.
"""

__all__ = ["generalized_rspmm"]


def __getattr__(name):
    if name == "generalized_rspmm":
        from .rspmm import generalized_rspmm
        return generalized_rspmm
    raise AttributeError(name)
