"""Drop-in for the reference package ``agent`` (agent/__init__.py:22-34).  ``TransformerPolicyNetwork`` has no standalone
counterpart: the policy lives inside the fused SAC handle (tvc_ai_amd.agent.NativeSAC)."""
from .multi_algorithm_agent import HierarchicalAgent, MultiAlgorithmAgent, SafetyLayer  # noqa: F401

__all__ = ["MultiAlgorithmAgent", "HierarchicalAgent", "SafetyLayer"]
