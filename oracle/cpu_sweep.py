"""ctypes binding and build recipe of ``oracle/cpu_sweep.cpp`` - the CPU baseline of SURVEY 8d at host
strength (the reference's batch loop ``lyapunov.py:517-529`` in C++ / OpenMP over all cores).

TEST / MEASUREMENT INFRASTRUCTURE: only ``bench.py``'s ``cpu_baseline`` leg, ``__graft_entry__`` (which
builds it) and ``tests/`` may use this module.  The NumPy oracle checks it
(``tests/test_cpu_sweep.py``); it checks nothing itself.
"""

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "cpu_sweep.cpp")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "libslcpu.so")
MAXD, MAXP = 4, 5


class CpuModel(C.Structure):
    _fields_ = [("d", C.c_int32), ("n", C.c_int32), ("num_points", C.c_int64 * MAXD),
                ("offset", C.c_double * MAXD), ("unit_maxes", C.c_double * MAXD),
                ("K", C.c_double * MAXD), ("saturate", C.c_int32), ("reserved", C.c_int32),
                ("lower", C.c_double), ("upper", C.c_double),
                ("X", C.c_void_p), ("L", C.c_void_p), ("alpha", C.c_void_p),
                ("variance", C.c_double), ("lengthscales", C.c_double * MAXP), ("beta", C.c_double),
                ("prior", (C.c_double * MAXP) * MAXD), ("P", (C.c_double * MAXD) * MAXD),
                ("G", (C.c_double * MAXD) * MAXD), ("lf", C.c_double), ("tau", C.c_double)]


def _isa():
    """Widest x86-64 level of this host that g++ can target without -march=native (a library built
    in one container must not die on another machine's missing extensions: the stamp below rebuilds
    it when the host's level differs)."""
    try:
        flags = open("/proc/cpuinfo").read()
    except OSError:
        flags = ""
    if all(f in flags for f in ("avx512f", "avx512bw", "avx512vl", "avx512dq", "avx512cd")):
        return "x86-64-v4"
    if "avx2" in flags and "fma" in flags:
        return "x86-64-v3"
    return "x86-64"


def build(force=False):
    """g++ -O3 -fopenmp -> oracle/_build/libslcpu.so (git-ignored; rebuilt when the source or the
    host's instruction-set level changed)."""
    isa = _isa()
    stamp = LIB + ".stamp"
    want = "%s %d" % (isa, int(os.path.getmtime(SRC)))
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == want:
        return LIB
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = ["g++", "-O3", "-march=" + isa, "-fopenmp", "-fno-math-errno", "-fno-trapping-math", "-std=c++17",
           "-shared", "-fPIC", SRC, "-o", LIB]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("g++ failed on oracle/cpu_sweep.cpp:\n" + res.stdout)
    with open(stamp, "w") as f:
        f.write(want)
    return LIB


_lib = None


def load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.sl_cpu_lyap_check.argtypes = [C.POINTER(CpuModel), C.c_void_p, C.c_int64, C.c_int, C.c_void_p,
                                           C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
        _lib.sl_cpu_lyap_check.restype = C.c_int
        _lib.sl_cpu_max_threads.restype = C.c_int
    return _lib


class CpuSweep(object):
    """The decrease check of one ``benchmarks.make_case`` GP workload on the host cores.

    ``gp`` is the oracle's ``GPRCached`` of the same case (its Cholesky factor and ``alpha`` are the
    reference's cache, ``functions.py:395-415``: computed once, not part of the timed loop)."""

    def __init__(self, case, gp):
        d = case["d"]
        dyn = case["dynamics"]
        if dyn.get("kind") != "gp" or case.get("stack") or case["m"] != 1 or d > MAXD or case["lv"][0] != "abs_linear":
            raise ValueError("cpu_sweep.cpp restates the shared-kernel RBF GP configurations only")
        self.d = d
        self._keep = [np.ascontiguousarray(gp.X, dtype=np.float64),
                      np.ascontiguousarray(gp.cholesky, dtype=np.float64),
                      np.ascontiguousarray(gp.alpha, dtype=np.float64)]
        m = CpuModel()
        m.d, m.n = d, len(gp.X)
        limits = np.asarray(case["limits"], dtype=np.float64)
        num = np.asarray(case["num_points"], dtype=np.int64)
        unit = (limits[:, 1] - limits[:, 0]) / (num - 1)
        for k in range(d):
            m.num_points[k], m.offset[k], m.unit_maxes[k] = int(num[k]), limits[k, 0], unit[k]
            m.K[k] = float(np.asarray(case["K"]).reshape(-1)[k])
        m.saturate = int(case["saturate"] is not None)
        if case["saturate"] is not None:
            m.lower, m.upper = float(case["saturate"][0]), float(case["saturate"][1])
        m.X, m.L, m.alpha = (a.ctypes.data for a in self._keep)
        m.variance, m.beta = float(dyn["variance"]), float(dyn["beta"])
        ls = np.broadcast_to(np.asarray(dyn["lengthscales"], dtype=np.float64), (d + 1,))
        for q in range(d + 1):
            m.lengthscales[q] = float(ls[q])
        for k in range(d):
            for q in range(d + 1):
                m.prior[k][q] = float(dyn["prior"][k, q])
            for i in range(d):
                m.P[k][i] = float(case["P"][k, i])
                m.G[k][i] = float(case["lv"][1][k, i])
        m.lf, m.tau = float(case["lf"]), float(case["tau"])
        self.model = m
        self.flops_per_check = m.n * (4 * (d + 1) + 2) + 2 * m.n * d + m.n * m.n + 2 * m.n

    def check(self, indices, threads=0, records=False):
        """-> (negative bool[count], records or None, seconds, threads used)."""
        lib = load()
        idx = np.ascontiguousarray(indices, dtype=np.int64)
        neg = np.zeros(len(idx), dtype=np.uint8)
        rec = np.zeros((len(idx), 2 + 2 * self.d)) if records else None
        seconds, used = C.c_double(0.0), C.c_int(0)
        rc = lib.sl_cpu_lyap_check(C.byref(self.model), idx.ctypes.data, len(idx), int(threads),
                                   neg.ctypes.data, rec.ctypes.data if records else None,
                                   C.byref(seconds), C.byref(used))
        if rc != 0:
            raise RuntimeError("sl_cpu_lyap_check failed (%d)" % rc)
        return neg.astype(bool), rec, seconds.value, used.value


def max_threads():
    return int(load().sl_cpu_max_threads())
