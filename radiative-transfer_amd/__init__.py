"""MI355X-native Monte Carlo dust photon random walk (drop-in for the reference's RunPhotons / runMC)."""
from . import _abi
from .engine import DEFAULT_SEED, H_PLANCK, Engine, RtError, SweepResult, make_params

__all__ = ["Engine", "RtError", "SweepResult", "make_params", "H_PLANCK", "DEFAULT_SEED", "_abi"]
