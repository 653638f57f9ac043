"""Child process of tests/test_sanitizers_cpu.py (started with libasan preloaded): builds the two host-side C++ restatements --
oracle/cpu_native/minco_cpu.cpp and tests/host_harness/lbfgs_host.cpp, which instantiate the product's optimiser headers
(csrc/neo_lbfgs*.hpp, neo_linesearch.hpp) for the host -- with -fsanitize=address,undefined and runs a G1 / G3 subset
through them.  Any report of the sanitizers aborts the process (halt_on_error / -fno-sanitize-recover)."""
import ctypes
import os
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np

CSRC = os.path.join(REPO, "neo-planner_amd", "csrc")
SAN = ["-g", "-O1", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]
tmp = tempfile.mkdtemp(prefix="neo_san_")

# ---- oracle/cpu_native, sanitized, behind the package's own ctypes wrapper
from oracle import cpu_native as cn
from oracle import minco_np as onp
from helpers import golden, load, rel_err
san_lib = os.path.join(tmp, "libminco_cpu_san.so")
subprocess.check_call(["g++", "-std=c++17", "-shared", "-fPIC", "-pthread", "-ffp-contract=off", "-I", CSRC] + SAN +
                      [cn.SRC, "-o", san_lib])
cn.build = lambda force=False: san_lib      # load() then opens the sanitized build
cn._lib = None

checked = 0
for path in golden("g1_eval_s[0-2].npz"):
    d = load(path)
    g = onp.GridESDF(d["occ"], float(d["res"]), d["occ"].shape[1], d["occ"].shape[0], d["origin"])
    nm = cn.NativeMap.from_grid2d(g)
    v_max, T_min, T_max, safe_dis, delta_t = d["params"]
    prm = onp.PlannerParams(v_max=v_max, T_min=T_min, T_max=T_max, safe_dis=safe_dis, delta_t=delta_t, weights=list(d["weights"]))
    for M in (3, 21, 41):
        t = f"M{M}_"
        pl = cn.NativePlanner(prm)
        x = d[t + "x"]
        pl.read_planning_conditions(nm, d[t + "head"], d[t + "tail"], x[:2 * (M - 1)].reshape(2, M - 1), d[t + "ts"])
        cost = pl.get_cost(x)
        grad = pl.get_grad(x)
        assert abs(cost - d[t + "cost"]) <= 1e-9 * abs(d[t + "cost"])
        assert rel_err(grad, d[t + "grad"]) < 1e-8
        checked += 1
print(f"cpu_native under ASan + UBSan: {checked} G1 evaluations")

# the batched native optimiser (threads, the product's L-BFGS headers on the host) on a recorded reference run
runs = 0
for path in golden("g3_trace_once_*.npz")[:3]:
    d = load(path)
    occ = d["occ"]
    g = onp.GridESDF(occ, float(d["res"]), occ.shape[1], occ.shape[0], d["origin"])
    nm = cn.NativeMap.from_grid2d(g)
    wp, ts = d["init_wpts"], d["init_ts"]
    M = len(ts)
    P = onp.PlannerParams()
    tsv = np.asarray(ts, float)
    x0 = np.concatenate([np.asarray(wp, float).reshape(-1), -np.log((P.T_max - P.T_min) / (tsv - P.T_min) - 1.0)])   # map_T2tau (:468-475)
    B = 6
    X0 = np.tile(x0, (B, 1)) + 1e-3 * np.arange(B)[:, None]
    head = np.tile(np.asarray(d["head"], float)[None], (B, 1, 1)); tail = np.tile(np.asarray(d["tail"], float)[None], (B, 1, 1))
    # pad head / tail to (3, D) as read_planning_conditions does
    def pad(a):
        out = np.zeros((B, 3, a.shape[2])); out[:, :a.shape[1]] = a; return out
    out = cn.optimize_batch(nm, X0, pad(head), pad(tail), M, 2, threads=3)
    assert np.isfinite(out["x"]).all() and (out["nfev"] > 0).all()
    runs += B
print(f"cpu_native.optimize_batch under ASan + UBSan: {runs} runs on 3 threads")

# ---- tests/host_harness (the product's optimiser headers, both forms, double and float line search)
hh = os.path.join(tmp, "lbfgs_host_san.so")
subprocess.check_call(["g++", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", "-I", CSRC] + SAN +
                      [os.path.join(REPO, "tests", "host_harness", "lbfgs_host.cpp"), "-o", hh])
import test_lbfgs_host as tl
L = ctypes.CDLL(hh)
L.dcsrch_host.restype = ctypes.c_int
L.dcsrch_host.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_double, ctypes.c_double,
                          ctypes.POINTER(ctypes.c_double), ctypes.c_int] + [ctypes.c_double] * 5


def rosen(x):
    f = float(np.sum(100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2))
    g = np.zeros_like(x)
    g[:-1] = -400 * x[:-1] * (x[1:] - x[:-1] ** 2) - 2 * (1 - x[:-1])
    g[1:] += 200 * (x[1:] - x[:-1] ** 2)
    return f, g, np.zeros(4)


entries = [e for e in ("lbfgs_host_minimize", "lbfgs_host_minimize_sm") if hasattr(L, e)]
for entry in entries:
    for n in (2, 7, 61, 130):
        r = tl.host_minimize(L, np.linspace(-1.2, 1.0, n), rosen, ftol=1e-10, gtol=1e-8, entry=entry)
        assert r["f"] < 1e-6 * n, (entry, n, r["f"], r["status"])
print(f"host harness under ASan + UBSan: {entries} on Rosenbrock n = 2, 7, 61, 130")
print("SANITIZERS CLEAN")
