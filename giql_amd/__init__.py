"""giql_amd -- MI355X-native execution backend for GIQL's INTERSECTS range join.

Only what the hot path needs lives here: ``csrc/`` (HIP kernels + the C ABI of
``include/giql_hip.h``), the ctypes loader, the device engine, and the host-side
mirror of the reference's ``Table`` / ``transpile(dialect="hip")`` interface.
"""

from ._lib import GiqlHipError, GiqlHipUnavailable  # noqa: F401



def __getattr__(name):
    # the table handles of giql_amd.execute, imported on first use (they pull numpy / pyarrow in)
    if name in ("pin", "PinnedTable", "clear_caches", "cache_info"):
        from . import execute as _execute_mod

        return getattr(_execute_mod, name)
    raise AttributeError(f"module 'giql_amd' has no attribute {name!r}")


__all__ = ["GiqlHipError", "GiqlHipUnavailable", "pin", "PinnedTable", "clear_caches", "cache_info"]
__version__ = "0.1.0"
