from .dg import dg, dg_raw  # noqa: F401
from .hash import hash_chunks  # noqa: F401
