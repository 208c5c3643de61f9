"""Curriculum driver for the vector env.

Restates the stage table and advancement rule of the reference's scripts/curriculum_manager.py (:60-95 stage
building from the YAML dict, :191-222 should_advance_stage, :248-290 update) as plain host logic, and -- unlike the
reference, whose manager output never reaches its env -- applies the active stage to a
VecRocketTVCEnv through tvc_env_set_dr (wind, mass variation, initial tilt; takes effect at each env's next reset).
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional


@dataclass
class Stage:
    name: str
    duration_steps: int
    conditions: Dict
    success_criteria: Dict
    completed: bool = False
    performance_history: List[float] = field(default_factory=list)


class CurriculumDriver:
    def __init__(self, curriculum_config: Optional[dict], env=None):
        self.config = curriculum_config or {}
        self.env = env
        self.stages = self._stages()
        self.current_stage_idx = 0
        self.current_step = 0
        self.stage_transition_steps: List[int] = []
        self._apply()

    def _stages(self) -> List[Stage]:
        if not self.config.get("enabled", False):
            return []
        out = []
        sc = self.config.get("stages", {})
        if isinstance(sc, dict):
            for key, sd in sc.items():
                if not isinstance(sd, dict):
                    continue
                envc = sd.get("environment", {}) or {}
                out.append(Stage(
                    name=sd.get("name", key),
                    duration_steps=sd.get("episodes", 200) * 1000,  # episodes -> steps (ref :78)
                    conditions={"max_initial_tilt": envc.get("initial_tilt_max", 0.1),
                                "wind_force": envc.get("wind_force", 0),
                                "mass_variation": envc.get("mass_variation", 0),
                                "domain_randomization": envc.get("mass_variation", 0) > 0,
                                "wind_enabled": envc.get("wind_force", 0) > 0},
                    success_criteria={"min_success_rate": envc.get("success_threshold", 0.7), "min_avg_reward": 100.0,
                                      "evaluation_episodes": 50}))
        return out

    def get_current_stage(self) -> Optional[Stage]:
        return self.stages[self.current_stage_idx] if self.current_stage_idx < len(self.stages) else None

    def should_advance_stage(self, eval_metrics: Dict) -> bool:
        st = self.get_current_stage()
        if st is None or st.completed:
            return False
        stage_steps = self.current_step - sum(s.duration_steps for s in self.stages[:self.current_stage_idx])
        if stage_steps < st.duration_steps * 0.5:  # at least half of the stage must have elapsed (ref :206)
            return False
        c = st.success_criteria
        return (eval_metrics.get("eval_success_rate", 0.0) >= c["min_success_rate"]
                and eval_metrics.get("eval_reward_mean", -float("inf")) >= c["min_avg_reward"])

    def advance_stage(self) -> bool:
        st = self.get_current_stage()
        if st:
            st.completed = True
            self.stage_transition_steps.append(self.current_step)
        self.current_stage_idx += 1
        self._apply()
        return self.get_current_stage() is not None

    def update(self, step: int, eval_metrics: Optional[Dict] = None) -> Dict:
        """update(step, eval_metrics) -> active stage conditions (+ '_curriculum_info'), ref :248-290"""
        self.current_step = step
        st = self.get_current_stage()
        if st is None:
            return {}
        if eval_metrics:
            st.performance_history.append(eval_metrics.get("eval_reward_mean", 0.0))
            if self.should_advance_stage(eval_metrics):
                self.advance_stage()
        # quirk kept from the reference (:274-288): after an advance the conditions are the NEW stage's, while the
        # progress record still names the stage that was current when update() was entered
        cur = self.get_current_stage()
        cfg = dict(cur.conditions) if cur is not None else {}
        start = sum(s.duration_steps for s in self.stages[:self.current_stage_idx])
        cfg["_curriculum_info"] = {"stage_name": st.name, "stage_index": self.current_stage_idx,
                                   "stage_progress": min((step - start) / st.duration_steps, 1.0),
                                   "total_stages": len(self.stages)}
        return cfg

    def _apply(self):
        st = self.get_current_stage()
        if self.env is None or st is None:
            return
        c = st.conditions
        self.env.set_domain_randomization(dr_enabled=1, dr_mass_var=float(c["mass_variation"]),
                                          dr_wind_std=float(c["wind_force"]), dr_init_tilt_max=float(c["max_initial_tilt"]))
