"""truss_mi355 -- MI355X-native batched 2-D truss FEM environment (host side).

`BatchedTruss` steps thousands of independent truss designs per HIP launch through the C ABI of
include/truss_mi355.h; `TrussTopology` mirrors the reset-time output of the reference's structure
builder.  The reference-named drop-in modules (truss2D_ENV, truss2D_GEN, FEM_2Dtruss, utils,
master_DDPG_truss2D_MO) live one directory up and are built on these two classes.
"""
from ._lib import TrussError, load, F_NO_DECODE, F_CLAMP_INPLACE, F_EMIT_OBS  # noqa: F401
from .topology import TrussTopology, SECTION_TABLE_CM, sections_si, YOUNG_MODULUS, LONG_STRESS  # noqa: F401
from .batched import BatchedTruss  # noqa: F401
from . import genes  # noqa: F401
