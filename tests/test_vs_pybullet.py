"""The one command that can pin SURVEY rows a4 / a5: oracle/tvc_oracle.c's restatement of `p.stepSimulation`
(env/enhanced_rocket_tvc_env.py:477) against pybullet ITSELF.

pybullet is not installed in the build image or on the GPU box and there is no package index, so this file SKIPS here; a
maintainer with `pybullet>=3.2.6,<3.3.0` (requirements.txt:11) runs

    python -m pytest tests/test_vs_pybullet.py -q

Build-authored harness (nothing of the reference is imported): the body is created with the constants the reference's own
createMultiBody / changeDynamics / setPhysicsEngineParameter calls were RECORDED to pass (tests/golden/step_ref_meta.json, written
by tests/golden/gen_step_golden.py from env/...:324-352, 409-464), the external wrench of every control step is the oracle's own
(`tvc_oracle_wrench`, pinned to the reference's applyExternalForce / applyExternalTorque arguments by tests/test_step_golden.py),
applied the way the reference applies it (env/...:524-527 gravity as an extra force, :557-559 thrust at the COM in the world frame +
torque, :571-585 drag / aerodynamic torque), and the 13-state after each `stepSimulation` is compared with the oracle's.

Bars: free flight <= 1e-4 relative on every state component over 100 steps (north_star's figure); the contact phase is
reported, and asserted only loosely (the oracle's contact model is build-defined)."""
import json
import os

import numpy as np
import pytest

pybullet = pytest.importorskip("pybullet", reason="pybullet is not installed here: SURVEY a4 / a5 stay PARITY UNPINNED")

from oracle import envoracle as eo  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
META = json.load(open(os.path.join(HERE, "golden", "step_ref_meta.json")))["constants"]


class BulletRocket:
    """the reference's world (env/...:324-352) and body (:409-464), from the recorded constants"""

    def __init__(self, plane=True):
        p = pybullet
        self.cid = p.connect(p.DIRECT)
        c = META
        p.setGravity(*c["gravity"], physicsClientId=self.cid)
        e = c["engine"]
        p.setPhysicsEngineParameter(fixedTimeStep=e["fixedTimeStep"], numSubSteps=e["numSubSteps"],
                                    enableConeFriction=e["enableConeFriction"],
                                    contactBreakingThreshold=e["contactBreakingThreshold"],
                                    enableFileCaching=e["enableFileCaching"], physicsClientId=self.cid)
        if plane:
            try:
                import pybullet_data
                p.setAdditionalSearchPath(pybullet_data.getDataPath(), physicsClientId=self.cid)
                self.ground = p.loadURDF(c["plane"], physicsClientId=self.cid)
            except Exception:  # no data package: an equivalent infinite plane
                shape = p.createCollisionShape(p.GEOM_PLANE, physicsClientId=self.cid)
                self.ground = p.createMultiBody(0, shape, physicsClientId=self.cid)
            p.changeDynamics(self.ground, -1, physicsClientId=self.cid, **c["plane_dynamics"])
        col = p.createCollisionShape(p.GEOM_CYLINDER, radius=c["collision"]["radius"], height=c["collision"]["height"],
                                     physicsClientId=self.cid)
        b = c["body"]
        self.body = p.createMultiBody(baseMass=b["mass"], baseCollisionShapeIndex=col, basePosition=b["position"],
                                      baseOrientation=b["orientation"], baseInertialFramePosition=[0, 0, 0],
                                      baseInertialFrameOrientation=[0, 0, 0, 1], physicsClientId=self.cid)
        d = dict(c["body_dynamics"])
        p.changeDynamics(self.body, -1, localInertiaDiagonal=d.pop("localInertiaDiagonal"), physicsClientId=self.cid)
        p.changeDynamics(self.body, -1, physicsClientId=self.cid, **d)

    def set_state13(self, s):
        p = pybullet
        p.resetBasePositionAndOrientation(self.body, list(s[0:3]), list(s[3:7]), physicsClientId=self.cid)
        p.resetBaseVelocity(self.body, list(s[7:10]), list(s[10:13]), physicsClientId=self.cid)

    def state13(self):
        p = pybullet
        pos, quat = p.getBasePositionAndOrientation(self.body, physicsClientId=self.cid)
        lin, ang = p.getBaseVelocity(self.body, physicsClientId=self.cid)
        return np.array(list(pos) + list(quat) + list(lin) + list(ang), dtype=np.float64)

    def step(self, F, tau):
        """one control step of the reference: wrench about the COM in the WORLD frame, then stepSimulation (4 substeps)"""
        p = pybullet
        pos, _ = p.getBasePositionAndOrientation(self.body, physicsClientId=self.cid)
        p.applyExternalForce(self.body, -1, list(F), list(pos), p.WORLD_FRAME, physicsClientId=self.cid)
        p.applyExternalTorque(self.body, -1, list(tau), p.WORLD_FRAME, physicsClientId=self.cid)
        p.stepSimulation(physicsClientId=self.cid)

    def close(self):
        pybullet.disconnect(self.cid)


def _rel(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


def _run(actions, start13, plane, contact):
    """oracle and pybullet stepped side by side, each on its OWN trajectory (free-running) -> per-step relative errors"""
    orc = eo.OracleEnv(contact=1 if contact else 0)
    orc.reset()
    orc.set_state13(start13)
    bul = BulletRocket(plane=plane)
    bul.set_state13(start13)
    errs = []
    for a in actions:
        F, tau = orc.wrench(a)          # wrench for the clipped action at the ORACLE's current state ...
        # ... (the reference adds gravity as an extra force of -m g on top of the world gravity, env/...:524-527: tvc_oracle_wrench
        # returns the applied force without the world's own gravity, exactly the applyExternalForce argument it was pinned to)
        bul.step(F, tau)
        orc.physics(a)
        errs.append(_rel(orc.state13(), bul.state13()))
    bul.close()
    return np.array(errs)


def test_free_flight_zero_action_100_steps_matches_pybullet():
    start = np.array([0, 0, 10.0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0], dtype=np.float64)  # high enough never to touch the ground
    errs = _run(np.zeros((100, 2)), start, plane=False, contact=False)
    print("zero-action free flight, worst relative error vs pybullet:", errs.max())
    assert errs.max() <= 1e-4, errs.max()


def test_free_flight_random_actions_100_steps_matches_pybullet():
    rng = np.random.default_rng(7)
    start = np.array([0, 0, 10.0, 0.02, -0.03, 0.01, 1, 0.1, -0.2, 0.3, 0.05, -0.04, 0.02], dtype=np.float64)
    start[3:7] /= np.linalg.norm(start[3:7])
    errs = _run(rng.uniform(-1, 1, (100, 2)), start, plane=False, contact=False)
    print("random-action free flight, worst relative error vs pybullet:", errs.max())
    assert errs.max() <= 1e-4, errs.max()


def test_nominal_fall_with_ground_contact_is_reported():
    """the reference's own start (1 m above the plane, zero action): free flight until the first touch (step 34 - 36 in the
    oracle), then the build-defined contact model against Bullet's manifold + sequential-impulse solver.  The free-flight segment
    must hold 1e-4; the contact segment is printed for the maintainer and only required to stay bounded (the rocket ends lying on
    the plane in both)."""
    start = np.array([0, 0, 1.0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0], dtype=np.float64)
    errs = _run(np.zeros((120, 2)), start, plane=True, contact=True)
    first = 30
    print("nominal fall: worst free-flight error", errs[:first].max(), "| contact segment median", np.median(errs[first:]),
          "max", errs[first:].max())
    assert errs[:first].max() <= 1e-4
    assert np.isfinite(errs).all()
