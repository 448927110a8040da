"""MI355X-native key-estimation hot path: CQT front end + PitchClassNet forward on HIP kernels.

Import as ``import ake_amd`` (the directory name carries a hyphen; ``ake_amd.py`` at the repo
root aliases it).  Reference-shaped modules: ``models`` (PitchClassNet), ``KeyDataset``
(KeyDataset + loaders), ``cqt`` (librosa.cqt stand-in), ``metrics`` (MIREX score).
"""
from . import _lib  # noqa: F401
from .audio import Resampler, get_resampler, prepare as prepare_audio  # noqa: F401
from .cqt import CQTPlan, cqt_logmag, get_plan, hop_for  # noqa: F401
from .KeyDataset import DatasetLoader, KeyDataset, SyntheticSineMixLoader, WaveformLoader  # noqa: F401
from .metrics import KEY_SIGNATURE_MAP, mirex_score  # noqa: F401
from .models import PitchClassNet  # noqa: F401
from .optim import FusedAdam  # noqa: F401
from .lightning_shim import Trainer  # noqa: F401
from .pipeline import KeyEstimator  # noqa: F401

__all__ = ["PitchClassNet", "KeyDataset", "DatasetLoader", "SyntheticSineMixLoader", "WaveformLoader", "CQTPlan",
           "cqt_logmag", "get_plan", "hop_for", "KEY_SIGNATURE_MAP", "mirex_score", "KeyEstimator", "Resampler", "get_resampler",
           "prepare_audio"]
