"""Drop-in for the reference package ``env`` (env/__init__.py:24-111): same names, backed by the HIP env of tvc_ai_amd.

Put ``<repo>/dropin`` and ``<repo>`` on sys.path (before the reference's own root) and ``scripts/train.py:44``
(``from env.enhanced_rocket_tvc_env import EnhancedRocketTVCEnv, MissionPhase``) resolves here with zero edits.
"""
from .enhanced_rocket_tvc_env import (EnhancedRocketTVCEnv, MissionPhase, MissionSuccess, SuccessCriteria,  # noqa: F401
                                      make_enhanced_tvc_env)
from tvc_ai_amd.env import make_debug_env, make_evaluation_env, make_training_env  # noqa: F401

_GYM_IDS = {  # ref env/__init__.py:28-64
    "EnhancedRocketTVC-v0": dict(enable_hierarchical=True, enable_curiosity=True, enable_physics_informed=True, debug=False),
    "EnhancedRocketTVC-Eval-v0": dict(enable_hierarchical=False, enable_curiosity=False, enable_physics_informed=False, debug=False),
    "EnhancedRocketTVC-Debug-v0": dict(enable_hierarchical=True, enable_curiosity=True, enable_physics_informed=True, debug=True),
}
try:  # gymnasium is optional here (absent from the build image); with it, the three ids of the reference are registered
    from gymnasium.envs.registration import register as _register
    for _id, _kw in _GYM_IDS.items():
        _register(id=_id, entry_point="env.enhanced_rocket_tvc_env:EnhancedRocketTVCEnv", max_episode_steps=1000, kwargs=_kw)
except ImportError:
    pass

__all__ = ["EnhancedRocketTVCEnv", "MissionPhase", "make_training_env", "make_evaluation_env", "make_debug_env"]
