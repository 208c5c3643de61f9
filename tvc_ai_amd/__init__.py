"""tvc_ai_amd -- MI355X-native hot path of NIKHILSAI71/TVC-AI: vectorised rocket-TVC env step and
SAC learner as hand-written HIP kernels behind the reference's Python surface.

(The directory is named ``tvc_ai_amd`` because a Python package cannot carry the hyphen of
``tvc-ai_amd``.)
"""
import os as _os

# hardware queues for the two-stream train step (tvc_ai_amd/streams.py): read by the HIP runtime when it initialises, so this only
# takes effect if the process has not touched the GPU yet; an explicit setting of the user's wins
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from .env import (EnhancedRocketTVCEnv, VecRocketTVCEnv, make_debug_env, make_enhanced_tvc_env,  # noqa: F401
                  make_evaluation_env, make_training_env, CURRICULUM_STAGES, PHASE_NAMES, MissionPhase, MissionSuccess,
                  SuccessCriteria)
from ._native import TvcError  # noqa: F401

__all__ = ["EnhancedRocketTVCEnv", "VecRocketTVCEnv", "make_training_env", "make_evaluation_env",
           "make_debug_env", "make_enhanced_tvc_env", "TvcError", "CURRICULUM_STAGES", "PHASE_NAMES", "MissionPhase",
           "MissionSuccess", "SuccessCriteria"]
