from . import hires_fitter  # noqa: F401
