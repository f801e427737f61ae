"""Loss side of a training step: `Loss` is the drop-in for models/losses/loss.py (one launch forward, one
backward), `compute_LNCC` for models/losses/ncc.py."""
from .loss import Loss  # noqa: F401
from .ncc import compute_LNCC  # noqa: F401
