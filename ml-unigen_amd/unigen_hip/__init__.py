"""unigen_hip: host-side glue (ctypes binding, tensor wrappers, engine, nn.Module façade) over
libunigen_hip.so.  Import is cheap; the shared library is loaded on first use and its absence is a
hard error (there is no CPU implementation of this path)."""
from .lib import UniGenHipError, build, load  # noqa: F401
