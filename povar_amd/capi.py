"""ctypes binding of libpovar_hip.so (include/povar_hip.h) -- used by tests and bench.py.

The binding mirrors the C ABI one to one; there is no Python or CPU fallback: if the shared
library is missing or no HIP device is present every call fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("POVAR_LIB", os.path.join(_PKG, "libpovar_hip.so"))
HEADER = os.path.join(os.path.dirname(_PKG), "include", "povar_hip.h")

NORM = {"NONE": 0, "HUBER": 1, "CAUCHY": 2}
POWER_VARPROJ, POWER_SCHUR_COMPLEMENT = 0, 1
E0_IMPLICIT, E0_TILES, E0_IMPLICIT_LDSACC, E0_TILES_LDSACC = 0, 1, 2, 3
NO_CONVERGENCE, SUCCESS, FAILURE = 0, 1, 2
NUMERIC_FAILURE = 1
(BUF_DIAG2, BUF_POSE_SCALING, BUF_JL_COL_SCALE, BUF_HLL_INV, BUF_B, BUF_B_INV, BUF_STORAGE,
 BUF_JL_COL_SCALE_H, BUF_B_JOINT, BUF_B_INV_JOINT, BUF_NC_HOUSEHOLDER, BUF_SC_PRECOND,
 BUF_SC_BLOCKDIAG) = range(13)
SC_PCG, SC_CHOLESKY = 0, 1


class Options(C.Structure):
    _fields_ = [("robust_norm", C.c_int32), ("huber_parameter", C.c_double),
                ("jacobi_scaling_eps", C.c_double), ("device", C.c_int32), ("e0_mode", C.c_int32), ("flags", C.c_uint32)]


# povar_options.flags (include/povar_hip.h: POVAR_FLAG_*)
FLAG_DETERMINISTIC, FLAG_DET_GATHER_TERMS, FLAG_NO_GRAPH, FLAG_NO_PACKED_ROWS = 1 << 0, 1 << 1, 1 << 2, 1 << 16


def flag_e0_kernel(k):
    return ((k + 1) & 0xF) << 4


def flag_series_kernel(m):
    return ((m + 1) & 0x3) << 8


def flag_placement(p):
    return (p & 0x3) << 12


class ResidualInfo(C.Structure):
    _fields_ = [("all_num_obs", C.c_int64), ("all_error", C.c_double),
                ("all_residual_sum", C.c_double), ("valid_num_obs", C.c_int64),
                ("valid_error", C.c_double), ("valid_residual_sum", C.c_double),
                ("is_numerically_valid", C.c_int32)]


class ProfileInfo(C.Structure):
    _fields_ = [("e0_ms", C.c_double), ("e0_launches", C.c_int64), ("binv_ms", C.c_double),
                ("binv_launches", C.c_int64), ("comm_ms", C.c_double), ("comm_launches", C.c_int64)]


class LayoutInfo(C.Structure):
    _fields_ = [("grid", C.c_int32), ("lds_slots", C.c_int32), ("n_global", C.c_int32), ("n_tail", C.c_int32),
                ("n_tiles", C.c_int64), ("n_rows", C.c_int64), ("n_cold", C.c_int64), ("n_obs", C.c_int64), ("lane_per_landmark", C.c_int64),
                ("create_ms", C.c_double), ("strategy", C.c_int32), ("hubs", C.c_int32), ("placement", C.c_int32),
                ("placement_ms", C.c_double), ("e0_kernel", C.c_int32), ("ck_ready", C.c_int32), ("ck_batches", C.c_int32),
                ("ck_slots", C.c_int32), ("ck_tiles_max", C.c_int32), ("ck_part_rec", C.c_int32), ("ck_rows", C.c_int64),
                ("ck_chunks", C.c_int64), ("ck_cold_chunks", C.c_int64), ("ck_build_ms", C.c_double), ("e0_auto", C.c_int32),
                ("tune_lpl_us", C.c_float), ("tune_ck_us", C.c_float), ("e0_kernel_h", C.c_int32), ("ckh_ready", C.c_int32),
                ("ckh_batches", C.c_int32), ("ckh_slots", C.c_int32), ("ckh_chunks", C.c_int64), ("ckh_cold_chunks", C.c_int64),
                ("e0_auto_h", C.c_int32), ("tune_lpl_h_us", C.c_float), ("tune_ck_h_us", C.c_float),
                ("res_ready", C.c_int32), ("res_active", C.c_int32), ("res_auto", C.c_int32), ("res_wgs", C.c_int32),
                ("res_waves", C.c_int32), ("res_rows", C.c_int32), ("res_rounds", C.c_int32), ("res_records", C.c_int32),
                ("res_max_cams", C.c_int32), ("res_max_lms", C.c_int32), ("res_max_chunks", C.c_int32),
                ("res_max_oq", C.c_int32), ("res_order", C.c_int32),
                ("res_lds_bytes", C.c_int32), ("res_build_ms", C.c_double), ("tune_terms_us", C.c_float),
                ("tune_res_us", C.c_float), ("res_failed", C.c_int32), ("ck_packed", C.c_int32), ("ck_cold_q", C.c_int32),
                ("ckh_stride", C.c_int32), ("ckh_accumulators", C.c_int32), ("ckh_capped_obs", C.c_int64)]


class TimingsInfo(C.Structure):
    _fields_ = [("linearize_ms", C.c_double), ("linearize_calls", C.c_int64), ("prepare_ms", C.c_double),
                ("prepare_calls", C.c_int64), ("solve_ms", C.c_double), ("solve_calls", C.c_int64),
                ("apply_ms", C.c_double), ("apply_calls", C.c_int64), ("other_ms", C.c_double),
                ("other_calls", C.c_int64)]


class PovarError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """Compile the gfx950 library in-tree (hipcc cross-compiles without a GPU)."""
    src_dir = os.path.join(_PKG, "csrc")
    srcs = [os.path.join(src_dir, f) for f in os.listdir(src_dir) if f.endswith((".hip", ".hpp", ".map")) or f == "Makefile"] + [HEADER]
    if force or not os.path.exists(LIB_PATH) or any(
            os.path.getmtime(LIB_PATH) < os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", src_dir], stdout=subprocess.DEVNULL)  # (make rebuilds the units whose sources changed)
    return LIB_PATH


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise PovarError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _LIB = C.CDLL(LIB_PATH)
        _LIB.povar_last_error.restype = C.c_char_p
        _LIB.povar_device_bytes.restype = C.c_int64
        _LIB.povar_destroy.restype = None
    return _LIB


def _p(a):
    return C.c_void_p(a.ctypes.data)


def shard_range(lm_off, world, rank):
    lm_off = np.ascontiguousarray(lm_off, dtype=np.int32)
    b, e = C.c_int32(), C.c_int32()
    rc = lib().povar_shard_range(C.c_int32(lm_off.shape[0] - 1), _p(lm_off), C.c_int32(world),
                                 C.c_int32(rank), C.byref(b), C.byref(e))
    if rc:
        raise PovarError(lib().povar_last_error().decode())
    return b.value, e.value


def comm_unique_id() -> bytes:
    buf = (C.c_uint8 * 128)()
    rc = lib().povar_comm_unique_id(buf)
    if rc:
        raise PovarError(lib().povar_last_error().decode())
    return bytes(buf)


class Context:
    """One povar_ctx: the device-resident linearizor state of one problem (or landmark shard)."""

    def __init__(self, n_cams, lm_off, cam_idx, obs, robust_norm="NONE", huber=1.0, eps=1e-5,
                 device=0, e0_mode=E0_IMPLICIT, flags=0):
        self.L = lib()
        self.lm_off = np.ascontiguousarray(lm_off, dtype=np.int32)
        self.cam_idx = np.ascontiguousarray(cam_idx, dtype=np.int32)
        self.obs = np.ascontiguousarray(obs, dtype=np.float64).reshape(-1, 2)
        self.n_cams = int(n_cams)
        self.n_lms = self.lm_off.shape[0] - 1
        self.n_obs = self.cam_idx.shape[0]
        opts = Options(NORM[robust_norm], huber, eps, device, e0_mode, flags)
        self.h = C.c_void_p()
        self._chk(self.L.povar_create(C.byref(self.h), C.c_int32(self.n_cams), C.c_int32(self.n_lms),
                                      C.c_int64(self.n_obs), _p(self.lm_off), _p(self.cam_idx),
                                      _p(self.obs), C.byref(opts)))

    def _chk(self, rc, allow_numeric=False):
        if rc < 0 or (rc > 0 and not allow_numeric):
            raise PovarError(f"povar_hip rc={rc}: {self.L.povar_last_error().decode()}")
        return rc

    def close(self):
        if self.h:
            self.L.povar_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # state
    def set_cameras(self, cams):
        cams = np.ascontiguousarray(cams, dtype=np.float64)
        assert cams.size == 12 * self.n_cams
        self._chk(self.L.povar_set_cameras(self.h, _p(cams)))

    def get_cameras(self):
        out = np.zeros((self.n_cams, 12))
        self._chk(self.L.povar_get_cameras(self.h, _p(out)))
        return out

    def set_landmarks(self, lms):
        lms = np.ascontiguousarray(lms, dtype=np.float64)
        assert lms.size == 3 * self.n_lms
        self._chk(self.L.povar_set_landmarks(self.h, _p(lms)))

    def get_landmarks(self):
        out = np.zeros((self.n_lms, 3))
        self._chk(self.L.povar_get_landmarks(self.h, _p(out)))
        return out

    def backup_pose(self):
        self._chk(self.L.povar_backup_pose(self.h))

    def restore_pose(self):
        self._chk(self.L.povar_restore_pose(self.h))

    # Linearizor surface
    def init_landmarks_pose(self, alpha):
        self._chk(self.L.povar_init_landmarks_pose(self.h, C.c_double(alpha)))

    def error_pose(self, alpha):
        ri = ResidualInfo()
        self._chk(self.L.povar_error_pose(self.h, C.c_double(alpha), C.byref(ri)))
        return ri

    def linearize_pose(self, alpha):
        return self._chk(self.L.povar_linearize_pose(self.h, C.c_double(alpha)), allow_numeric=True) == 0

    def solve_pose(self, lam, solver_type, m, q_tol=0.0, r_tol=-1.0):
        inc = np.zeros(12 * self.n_cams)
        it, st = C.c_int32(), C.c_int32()
        rc = self._chk(self.L.povar_solve_pose(self.h, C.c_double(lam), C.c_int32(solver_type),
                                               C.c_int32(m), C.c_double(q_tol), C.c_double(r_tol),
                                               _p(inc), C.byref(it), C.byref(st)), allow_numeric=True)
        return inc, it.value, st.value, rc

    def apply_pose(self, solver_type, alpha, inc):
        inc = np.ascontiguousarray(inc, dtype=np.float64)
        ld = C.c_double()
        self._chk(self.L.povar_apply_pose(self.h, C.c_int32(solver_type), C.c_double(alpha), _p(inc),
                                          C.byref(ld)))
        return ld.value

    # step 2
    def set_landmarks_homogeneous(self, lms_h):
        lms_h = np.ascontiguousarray(lms_h, dtype=np.float64)
        assert lms_h.size == 4 * self.n_lms
        self._chk(self.L.povar_set_landmarks_homogeneous(self.h, _p(lms_h)))

    def get_landmarks_homogeneous(self):
        out = np.zeros((self.n_lms, 4))
        self._chk(self.L.povar_get_landmarks_homogeneous(self.h, _p(out)))
        return out

    def backup_joint(self):
        self._chk(self.L.povar_backup_joint(self.h))

    def restore_joint(self):
        self._chk(self.L.povar_restore_joint(self.h))

    def error_homogeneous(self):
        ri = ResidualInfo()
        self._chk(self.L.povar_error_homogeneous(self.h, C.byref(ri)))
        return ri

    def linearize_homogeneous(self):
        return self._chk(self.L.povar_linearize_homogeneous(self.h), allow_numeric=True) == 0

    def prepare_joint(self, lam):
        self._chk(self.L.povar_prepare_joint(self.h, C.c_double(lam)))

    def solve_joint(self, lam, m, q_tol=0.0, r_tol=-1.0):
        inc = np.zeros(11 * self.n_cams)
        it, st = C.c_int32(), C.c_int32()
        rc = self._chk(self.L.povar_solve_joint(self.h, C.c_double(lam), C.c_int32(m), C.c_double(q_tol),
                                                C.c_double(r_tol), _p(inc), C.byref(it), C.byref(st)),
                       allow_numeric=True)
        return inc, it.value, st.value, rc

    # explicit-Schur-complement solvers (LinearizorSC: PCG / CHOLESKY / RIPCG)
    def set_jl_col_scaling(self, enable):
        self._chk(self.L.povar_set_jl_col_scaling(self.h, C.c_int32(1 if enable else 0)))

    def solve_pose_sc(self, lam, method=SC_PCG, min_iterations=0, max_iterations=500, eta=1e-2):
        inc = np.zeros(12 * self.n_cams)
        it, st = C.c_int32(), C.c_int32()
        rc = self._chk(self.L.povar_solve_pose_sc(self.h, C.c_double(lam), C.c_int32(method),
                                                  C.c_int32(min_iterations), C.c_int32(max_iterations),
                                                  C.c_double(eta), _p(inc), C.byref(it), C.byref(st)),
                       allow_numeric=True)
        return inc, it.value, st.value, rc

    def solve_joint_sc(self, lam, min_iterations=0, max_iterations=500, eta=1e-2):
        inc = np.zeros(11 * self.n_cams)
        it, st = C.c_int32(), C.c_int32()
        rc = self._chk(self.L.povar_solve_joint_sc(self.h, C.c_double(lam), C.c_int32(min_iterations),
                                                   C.c_int32(max_iterations), C.c_double(eta), _p(inc),
                                                   C.byref(it), C.byref(st)), allow_numeric=True)
        return inc, it.value, st.value, rc

    def apply_joint(self, inc):
        inc = np.ascontiguousarray(inc, dtype=np.float64)
        ld = C.c_double()
        self._chk(self.L.povar_apply_joint(self.h, _p(inc), C.byref(ld)))
        return ld.value

    def normalize_joint(self):
        self._chk(self.L.povar_normalize_joint(self.h))

    # finer grained
    def prepare_pose(self, lam, solver_type=POWER_VARPROJ):
        self._chk(self.L.povar_prepare_pose(self.h, C.c_double(lam), C.c_int32(solver_type)))

    def power_series_pose(self, m, q_tol=0.0, r_tol=-1.0):
        it, st = C.c_int32(), C.c_int32()
        self._chk(self.L.povar_power_series_pose(self.h, C.c_int32(m), C.c_double(q_tol),
                                                 C.c_double(r_tol), C.byref(it), C.byref(st)))
        return it.value, st.value

    def get_increment(self, dim=12):
        out = np.zeros(dim * self.n_cams)
        self._chk(self.L.povar_get_increment(self.h, _p(out)))
        return out

    def power_series_begin(self):
        self._chk(self.L.povar_power_series_begin(self.h))

    def power_series_step(self):
        self._chk(self.L.povar_power_series_step(self.h))

    def get_term(self, dim=12):
        out = np.zeros(dim * self.n_cams)
        self._chk(self.L.povar_get_term(self.h, _p(out)))
        return out

    def right_mul_e0_pose(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.zeros(12 * self.n_cams)
        self._chk(self.L.povar_right_mul_e0_pose(self.h, _p(x), _p(y)))
        return y

    def set_e0_mode(self, mode):
        self._chk(self.L.povar_set_e0_mode(self.h, C.c_int32(mode)))

    def synchronize(self):
        self._chk(self.L.povar_synchronize(self.h))

    def get_buffer(self, which, joint=False):
        if which in (BUF_SC_PRECOND, BUF_SC_BLOCKDIAG):
            n = (121 if joint else 144) * self.n_cams
            out = np.zeros(n)
            self._chk(self.L.povar_get_buffer(self.h, C.c_int32(which), _p(out), C.c_int64(n)))
            return out
        n = {BUF_DIAG2: 12 * self.n_cams, BUF_POSE_SCALING: 12 * self.n_cams,
             BUF_JL_COL_SCALE: 3 * self.n_lms, BUF_HLL_INV: 9 * self.n_lms, BUF_B: 12 * self.n_cams,
             BUF_B_INV: 144 * self.n_cams, BUF_STORAGE: 64 * self.n_obs,
             BUF_JL_COL_SCALE_H: 4 * self.n_lms, BUF_B_JOINT: 11 * self.n_cams,
             BUF_B_INV_JOINT: 121 * self.n_cams, BUF_NC_HOUSEHOLDER: 13 * self.n_cams}[which]
        out = np.zeros(n)
        self._chk(self.L.povar_get_buffer(self.h, C.c_int32(which), _p(out), C.c_int64(n)))
        return out

    def profile_enable(self, on=True):
        self._chk(self.L.povar_profile_enable(self.h, C.c_int32(1 if on else 0)))

    def profile_get(self):
        pi = ProfileInfo()
        self._chk(self.L.povar_profile_get(self.h, C.byref(pi)))
        return pi

    def device_bytes(self):
        return int(self.L.povar_device_bytes(self.h))

    def e0_model_bytes(self):
        """(landmark-major kernel, per-camera kernel) byte floor of one E0 application in the current mode."""
        a, b = C.c_int64(), C.c_int64()
        self._chk(self.L.povar_e0_model_bytes(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def timings_enable(self, on=True):
        self._chk(self.L.povar_timings_enable(self.h, C.c_int32(1 if on else 0)))

    def timings(self):
        ti = TimingsInfo()
        self._chk(self.L.povar_timings(self.h, C.byref(ti)))
        return ti

    def layout_info(self):
        li = LayoutInfo()
        self._chk(self.L.povar_get_layout_info(self.h, C.byref(li)))
        return li

    def layout_finalize(self, wait=True):
        """Swap the rows placed on the host thread in (wait: block until they are ready).  True when placed rows are
        in use afterwards.  A linearisation taken before the swap is dropped."""
        rc = self.L.povar_layout_finalize(self.h, C.c_int32(1 if wait else 0))
        if rc < 0:
            raise PovarError(f"povar_hip rc={rc}: {self.L.povar_last_error().decode()}")
        return rc == 1

    def set_e0_kernel(self, kernel):
        """Per-term E0 kernel of step 1: 0 = e0_lpl, 1..5 = e0_ck instantiations (include/povar_hip.h)."""
        self._chk(self.L.povar_set_e0_kernel(self.h, C.c_int32(int(kernel))))

    def set_series_kernel(self, mode):
        """-1: timed choice, 0: per-term kernels in a hipGraph, 1: the resident power series (include/povar_hip.h)."""
        self._chk(self.L.povar_set_series_kernel(self.h, C.c_int32(int(mode))))

    def comm_ranks(self):
        n = self.L.povar_comm_ranks(self.h)
        if n < 0:
            raise PovarError(f"povar_hip rc={n}: {self.L.povar_last_error().decode()}")
        return n

    def comm_init_host(self, world, rank, fn):
        """fn(buf: np.ndarray) sums buf in place over the ranks (host all-reduce hook)."""
        def _cb(ptr, n, user):
            fn(np.ctypeslib.as_array(ptr, shape=(n,)))
        self._cb = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_int64, C.c_void_p)(_cb)
        self._chk(self.L.povar_comm_init_host(self.h, C.c_int32(world), C.c_int32(rank), self._cb, None))

    def p2p_export(self, world) -> bytes:
        buf = (C.c_uint8 * 64)()
        self._chk(self.L.povar_p2p_export(self.h, C.c_int32(world), buf))
        return bytes(buf)

    def p2p_attach(self, world, rank, handles):
        blob = b"".join(handles)
        assert len(blob) == 64 * world
        buf = (C.c_uint8 * len(blob)).from_buffer_copy(blob)
        self._chk(self.L.povar_p2p_attach(self.h, C.c_int32(world), C.c_int32(rank), buf))

    def p2p_enable(self, on: bool):
        self._chk(self.L.povar_p2p_enable(self.h, C.c_int32(1 if on else 0)))

    def comm_init(self, world, rank, uid: bytes):
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        self._chk(self.L.povar_comm_init(self.h, C.c_int32(world), C.c_int32(rank), buf))
