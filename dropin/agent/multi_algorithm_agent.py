"""Drop-in for the reference module ``agent/multi_algorithm_agent.py``: ``scripts/train.py:45``
(``from agent.multi_algorithm_agent import MultiAlgorithmAgent``) resolves here with zero edits."""
from tvc_ai_amd.agent import MultiAlgorithmAgent  # noqa: F401
from tvc_ai_amd.curiosity import SafetyLayer  # noqa: F401
from tvc_ai_amd.hierarchical import HierarchicalPolicy as HierarchicalAgent  # noqa: F401

__all__ = ["MultiAlgorithmAgent", "HierarchicalAgent", "SafetyLayer"]
