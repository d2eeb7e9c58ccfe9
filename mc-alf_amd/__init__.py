"""mc-alf_amd: MI355X (gfx950) implementation of the MC-ALF likelihood hot path.

Layout mirrors the reference package for the path it replaces:
`routines.hires_fitter.als_fitter` (reference: mcalf/routines/hires_fitter.py:30).
The directory name contains a hyphen; `import mcalf_amd` (the alias module at the
repository root) resolves to this package.
"""
from . import _lib  # noqa: F401
from . import adapters  # noqa: F401
from . import broker  # noqa: F401
from . import dist  # noqa: F401
from . import routines  # noqa: F401
from . import workloads  # noqa: F401
from .routines import hires_fitter  # noqa: F401
from .routines.hires_fitter import als_fitter  # noqa: F401

__version__ = "0.1"
