"""Loss-side kernels.  The Loss module itself stays the reference's (models/losses/loss.py);
`compute_LNCC` is the drop-in for models/losses/ncc.py."""
from .ncc import compute_LNCC  # noqa: F401
