"""giql_amd -- MI355X-native execution backend for GIQL's INTERSECTS range join.

Only what the hot path needs lives here: ``csrc/`` (HIP kernels + the C ABI of
``include/giql_hip.h``), the ctypes loader, the device engine, and the host-side
mirror of the reference's ``Table`` / ``transpile(dialect="hip")`` interface.
"""

from ._lib import GiqlHipError, GiqlHipUnavailable  # noqa: F401

__all__ = ["GiqlHipError", "GiqlHipUnavailable"]
__version__ = "0.1.0"
