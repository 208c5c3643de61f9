"""Drop-in for the reference module ``env/enhanced_rocket_tvc_env.py``: the names a caller imports from it
(scripts/train.py:44; env/__init__.py:26), re-exported from the MI355X-native implementation.  No arithmetic here."""
from tvc_ai_amd.env import (EnhancedRocketTVCEnv, MissionPhase, MissionSuccess, SuccessCriteria,  # noqa: F401
                            make_enhanced_tvc_env)

__all__ = ["EnhancedRocketTVCEnv", "MissionPhase", "MissionSuccess", "SuccessCriteria", "make_enhanced_tvc_env"]
