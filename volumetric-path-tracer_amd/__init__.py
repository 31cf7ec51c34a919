"""vpt-mi355x: MI355X-native replacement for the hot path of
sergeneren/Volumetric-Path-Tracer (`volume_rt_kernel`), behind the C ABI of
include/vpt_abi.h.  See DESIGN.md.

The directory name contains a hyphen, so it is imported through `__graft_entry__.load_package()`
(registered as module `vpt_amd`).
"""
from . import abi  # noqa: F401
from .host import ABI_SYMBOLS, LIB_PATH, Context, VptError, load_library  # noqa: F401
from . import scene  # noqa: F401
from . import dist  # noqa: F401
from . import atmosphere  # noqa: F401
from . import io  # noqa: F401
from . import host  # noqa: F401
