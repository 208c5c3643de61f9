"""ctypes binding of the CPU fp64 oracle (oracle/tvc_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by the product package (tvc_ai_amd/).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libtvc_oracle.so")
HIST_MAX = 1000

PHASE_NAMES = ["boost", "coast", "landing", "touchdown", "hover", "complete", "failed"]


def build(force=False):
    """Compile the oracle with gcc (seconds). Building the checker is not using it."""
    src = os.path.join(_HERE, "tvc_oracle.c")
    hdr = os.path.join(_HERE, "tvc_oracle.h")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libtvc_oracle.so"])
    return _LIB_PATH


class Params(C.Structure):
    _fields_ = [
        ("mass", C.c_double), ("inertia", C.c_double * 3), ("thrust", C.c_double),
        ("half_len", C.c_double), ("radius", C.c_double), ("lin_damp", C.c_double),
        ("ang_damp", C.c_double), ("gravity", C.c_double), ("dt_sub", C.c_double),
        ("n_sub", C.c_int), ("max_episode_steps", C.c_int), ("distinct_window", C.c_int),
        ("contact", C.c_int), ("auto_reset", C.c_int),
        ("cg_offset", C.c_double), ("wind", C.c_double * 3),
        ("mu", C.c_double), ("erp", C.c_double), ("cop_s0", C.c_double),
        ("init_pos", C.c_double * 3), ("init_quat", C.c_double * 4),
    ]


class Env(C.Structure):
    _fields_ = [
        ("pos", C.c_double * 3), ("quat", C.c_double * 4), ("vel", C.c_double * 3),
        ("omega", C.c_double * 3), ("fuel", C.c_double),
        ("step", C.c_int), ("phase", C.c_int), ("mission_successful", C.c_int),
        ("success_run", C.c_int),
        ("prev_action", C.c_double * 2), ("has_prev_action", C.c_int),
        ("hist", C.c_double * HIST_MAX), ("hist_len", C.c_int), ("hist_head", C.c_int),
        ("has_prev_obs", C.c_int), ("prev_obs8", C.c_double * 8),
        ("episodes", C.c_long),
    ]


class Scalars(C.Structure):
    _fields_ = [
        ("altitude", C.c_double), ("tilt", C.c_double), ("omega_mag", C.c_double),
        ("v_h", C.c_double), ("v_z_abs", C.c_double), ("x", C.c_double), ("y", C.c_double),
        ("crashed", C.c_int),
    ]


class Out(C.Structure):
    _fields_ = [
        ("obs", C.c_float * 10), ("reward", C.c_double), ("terminated", C.c_int),
        ("truncated", C.c_int), ("components", C.c_double * 12), ("sc", Scalars),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.tvc_oracle_default_params.argtypes = [C.POINTER(Params)]
        L.tvc_oracle_scale_mass.argtypes = [C.POINTER(Params), C.c_double]
        L.tvc_oracle_init.argtypes = [C.POINTER(Env), C.POINTER(Params)]
        L.tvc_oracle_reset.argtypes = [C.POINTER(Env), C.POINTER(Params)]
        L.tvc_oracle_observe.argtypes = [C.POINTER(Env), C.POINTER(Params), C.POINTER(C.c_float)]
        L.tvc_oracle_physics.argtypes = [C.POINTER(Env), C.POINTER(Params), C.POINTER(C.c_double)]
        L.tvc_oracle_wrench.argtypes = [C.POINTER(Env), C.POINTER(Params), C.POINTER(C.c_double),
                                        C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.tvc_oracle_scalars_from_state.argtypes = [C.POINTER(Env), C.POINTER(Scalars)]
        L.tvc_oracle_logic.argtypes = [C.POINTER(Env), C.POINTER(Params), C.POINTER(Scalars),
                                       C.POINTER(C.c_double), C.POINTER(Out)]
        L.tvc_oracle_step.argtypes = [C.POINTER(Env), C.POINTER(Params), C.POINTER(C.c_double), C.POINTER(Out)]
        L.tvc_oracle_quat_to_matrix.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.tvc_oracle_quat_to_euler.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.tvc_oracle_fuel_after.argtypes = [C.c_int]
        L.tvc_oracle_fuel_after.restype = C.c_double
        L.tvc_oracle_run.argtypes = [C.POINTER(Env), C.POINTER(Params), C.c_int, C.c_int,
                                     C.POINTER(C.c_float), C.POINTER(C.c_double)]
        L.tvc_oracle_run.restype = C.c_long
        L.tvc_oracle_sizeof_env.restype = C.c_int
        L.tvc_oracle_sizeof_params.restype = C.c_int
        assert L.tvc_oracle_sizeof_env() == C.sizeof(Env), (L.tvc_oracle_sizeof_env(), C.sizeof(Env))
        assert L.tvc_oracle_sizeof_params() == C.sizeof(Params)
        _lib = L
    return _lib


def default_params(**over):
    p = Params()
    lib().tvc_oracle_default_params(C.byref(p))
    for k, v in over.items():
        if k == "mass_scale":
            lib().tvc_oracle_scale_mass(C.byref(p), float(v))
        elif isinstance(v, (list, tuple, np.ndarray)):
            arr = getattr(p, k)
            for i, x in enumerate(v):
                arr[i] = float(x)
        else:
            setattr(p, k, v)
    return p


def _dvec(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


class OracleEnv:
    """One reference-semantics env object (N = 1)."""

    def __init__(self, params=None, **over):
        self.p = params if params is not None else default_params(**over)
        self.e = Env()
        lib().tvc_oracle_init(C.byref(self.e), C.byref(self.p))

    def reset(self):
        lib().tvc_oracle_reset(C.byref(self.e), C.byref(self.p))
        return self.observe()

    def observe(self):
        obs = (C.c_float * 10)()
        lib().tvc_oracle_observe(C.byref(self.e), C.byref(self.p), obs)
        return np.array(obs, dtype=np.float32)

    def step(self, action):
        a, ap = _dvec(action)
        out = Out()
        lib().tvc_oracle_step(C.byref(self.e), C.byref(self.p), ap, C.byref(out))
        return out

    def physics(self, action):
        a, ap = _dvec(action)
        lib().tvc_oracle_physics(C.byref(self.e), C.byref(self.p), ap)

    def wrench(self, action):
        """external (F, tau) about the COM for the CLIPPED action, as step() would assemble it now"""
        a, ap = _dvec(np.clip(np.asarray(action, dtype=np.float64), -1.0, 1.0))
        F, tau = (C.c_double * 3)(), (C.c_double * 3)()
        lib().tvc_oracle_wrench(C.byref(self.e), C.byref(self.p), ap, F, tau)
        return np.array(F), np.array(tau)

    def scalars(self):
        sc = Scalars()
        lib().tvc_oracle_scalars_from_state(C.byref(self.e), C.byref(sc))
        return sc

    def logic(self, sc, action):
        a, ap = _dvec(action)
        out = Out()
        lib().tvc_oracle_logic(C.byref(self.e), C.byref(self.p), C.byref(sc), ap, C.byref(out))
        return out

    # convenient numpy views
    def state13(self):
        e = self.e
        return np.array(list(e.pos) + list(e.quat) + list(e.vel) + list(e.omega), dtype=np.float64)

    def set_state13(self, s):
        e = self.e
        s = [float(x) for x in s]
        for i in range(3):
            e.pos[i] = s[i]
            e.vel[i] = s[7 + i]
            e.omega[i] = s[10 + i]
        for i in range(4):
            e.quat[i] = s[3 + i]


class OracleVec:
    """N independent oracle envs stepped in a C loop (used by parity tests and the CPU baseline)."""

    def __init__(self, n, params=None, **over):
        self.n = n
        self.p = params if params is not None else default_params(**over)
        self.envs = (Env * n)()
        for i in range(n):
            lib().tvc_oracle_init(C.byref(self.envs[i]), C.byref(self.p))
        self.per_env_params = None

    def set_per_env_params(self, plist):
        assert len(plist) == self.n
        self.per_env_params = plist

    def _p(self, i):
        return self.per_env_params[i] if self.per_env_params is not None else self.p

    def observe(self):
        obs = np.zeros((self.n, 10), dtype=np.float32)
        buf = (C.c_float * 10)()
        for i in range(self.n):
            lib().tvc_oracle_observe(C.byref(self.envs[i]), C.byref(self._p(i)), buf)
            obs[i] = np.frombuffer(buf, dtype=np.float32)
        return obs

    def step(self, actions):
        actions = np.asarray(actions, dtype=np.float64)
        n = self.n
        obs = np.zeros((n, 10), dtype=np.float32)
        rew = np.zeros(n, dtype=np.float64)
        term = np.zeros(n, dtype=np.uint8)
        trunc = np.zeros(n, dtype=np.uint8)
        out = Out()
        a = (C.c_double * 2)()
        for i in range(n):
            a[0], a[1] = actions[i, 0], actions[i, 1]
            lib().tvc_oracle_step(C.byref(self.envs[i]), C.byref(self._p(i)), a, C.byref(out))
            obs[i] = np.frombuffer(out.obs, dtype=np.float32)
            rew[i] = out.reward
            term[i] = out.terminated
            trunc[i] = out.truncated
        return obs, rew, term, trunc

    def run(self, actions_tn2):
        """Time-critical path for the CPU baseline: actions [T, n, 2] float32, auto-reset per params."""
        a = np.ascontiguousarray(actions_tn2, dtype=np.float32)
        T = a.shape[0]
        rs = C.c_double(0.0)
        steps = lib().tvc_oracle_run(self.envs, C.byref(self.p), self.n, T,
                                     a.ctypes.data_as(C.POINTER(C.c_float)), C.byref(rs))
        return steps, rs.value

    def state13(self):
        out = np.zeros((self.n, 13), dtype=np.float64)
        for i in range(self.n):
            e = self.envs[i]
            out[i] = list(e.pos) + list(e.quat) + list(e.vel) + list(e.omega)
        return out

    def set_state13(self, s):
        for i in range(self.n):
            e = self.envs[i]
            for k in range(3):
                e.pos[k] = float(s[i, k])
                e.vel[k] = float(s[i, 7 + k])
                e.omega[k] = float(s[i, 10 + k])
            for k in range(4):
                e.quat[k] = float(s[i, 3 + k])

    def aux(self):
        """Integer aux state per env: step, phase, mission_successful, success_run, hist_len."""
        return np.array([[e.step, e.phase, e.mission_successful, e.success_run, e.hist_len, e.has_prev_action]
                         for e in self.envs], dtype=np.int64)


def load_export(vec, dyn, aux, prev_action, hist):
    """Copy a tvc_env_export_state() snapshot (numpy arrays) into an OracleVec.
    aux columns: step, phase, mission_successful, success_run, hist_len, has_prev_action, distinct, episode."""
    L = lib()
    W = hist.shape[1]
    for i in range(vec.n):
        e = vec.envs[i]
        for k in range(3):
            e.pos[k] = float(dyn[i, k])
            e.vel[k] = float(dyn[i, 7 + k])
            e.omega[k] = float(dyn[i, 10 + k])
        for k in range(4):
            e.quat[k] = float(dyn[i, 3 + k])
        e.step = int(aux[i, 0])
        e.phase = int(aux[i, 1])
        e.mission_successful = int(aux[i, 2])
        e.success_run = int(aux[i, 3])
        e.has_prev_action = int(aux[i, 5])
        e.prev_action[0] = float(prev_action[i, 0])
        e.prev_action[1] = float(prev_action[i, 1])
        e.fuel = L.tvc_oracle_fuel_after(int(aux[i, 0]))
        hl = int(aux[i, 4])
        wl = min(hl, W)
        # the device keeps only the last W rewards; older entries (hl - wl of them) are unknown to it
        # and only their count matters when distinct_window == W.
        e.hist_head = 0
        e.hist_len = hl
        pad = hl - wl
        for k in range(pad):
            e.hist[k] = -12345.0 - k  # distinct fillers outside any window the oracle will look at
        for k in range(wl):
            e.hist[pad + k] = float(hist[i, k])


def params_from_export(base_over, par_row):
    """Per-env oracle params from one row of the exported DR params
    (mass_scale, thrust_scale, cg_offset, wind xyz)."""
    p = default_params(**base_over)
    lib().tvc_oracle_scale_mass(C.byref(p), float(par_row[0]))
    p.thrust = p.thrust * float(par_row[1])
    p.cg_offset = float(par_row[2])
    p.wind[0], p.wind[1], p.wind[2] = float(par_row[3]), float(par_row[4]), float(par_row[5])
    return p
