"""vszip on MI355X: hand-written gfx950 HIP kernels behind the vszip filter API.

The package directory name carries a hyphen (it mirrors the upstream repo name),
so import it through the `vszip_amd` shim at the repo root:  `import vszip_amd`.
"""
from . import capi  # noqa: F401
from .capi import Device, DevPlane, VszipError  # noqa: F401
from . import cluster  # noqa: F401
