"""nemoflux_amd -- MI355X-native transect-flux engine: a drop-in for nemoflux's hot path.

    from nemoflux_amd import mint            # replaces `import mint` (Grid, PolylineIntegral)
    from nemoflux_amd.field import Field     # replaces nemoflux/field.py
    from nemoflux_amd.horizgrid import HorizGrid

Importing this package needs the in-tree HIP library (nemoflux_amd/libnemoflux_amd.so); there is no CPU
fallback.  See DESIGN.md and INTEGRATION.md.
"""
from . import _lib  # noqa: F401  (fails loudly if the HIP extension is missing)
from . import mint  # noqa: F401
from ._lib import DeviceArray, DeviceBuffer, NemofluxError  # noqa: F401

__all__ = ['mint', 'DeviceArray', 'DeviceBuffer', 'NemofluxError']
