"""montgomery_amd: MI355X-native MSM engine behind the reference's msm() API shape.

Host-side mirror of the reference's curve modules (src/parallel.ts, src/concrete/*) over the C ABI
of libmsm_hip.so.  See DESIGN.md and INTEGRATION.md.
"""
from .api import (  # noqa: F401
    BLS12377,
    TwistedEdwards,
    Weierstrass,
    MsmContext,
    compute_msm,
    compute_msm_ed,
    create_weierstrass,
    startThreads,
    stopThreads,
)
from ._lib import MsmError  # noqa: F401
