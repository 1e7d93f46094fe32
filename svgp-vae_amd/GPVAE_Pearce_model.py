"""`GPVAE_Pearce_model.py` of the reference (`build_pearce_elbo_graphs`, GPVAE_Pearce_model.py:89; imported by
BALL_experiment.py:15).  The exact per-video GP (`build_1d_gp`, :8-86) runs as `k_pearce_fwd / k_pearce_bwd` of the HIP
library; the host side lives with the other moving-ball engines in ball.py."""
from .ball import PearceStepEngine, build_pearce_elbo_graphs  # noqa: F401
