"""The C ABI from a C program (not ctypes): tests/c_abi/abi_smoke.c is compiled with gcc -std=c99 against
include/neo_planner.h, linked with -lneo_planner_hip, and run on a recorded reference scenario.
The compile-and-link half runs on CPU (`not gpu`); running it needs the MI355X."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import REPO
from helpers import golden, load

LIBDIR = os.path.join(REPO, "neo-planner_amd", "neo_planner_amd")
SRC = os.path.join(REPO, "tests", "c_abi", "abi_smoke.c")


def _build(tmp):
    exe = os.path.join(tmp, "abi_smoke")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(REPO, "include"), SRC,
                           "-L", LIBDIR, "-lneo_planner_hip", "-lm", f"-Wl,-rpath,{LIBDIR}",
                           "-Wl,--allow-shlib-undefined", "-o", exe])
    return exe


def _fixture(tmp):
    from oracle import minco_np as onp
    d = load(golden("g3_trace_plan_s0.npz")[0])
    occ = d["occ"].astype(np.int8)
    H, W = occ.shape
    pl = onp.OraclePlanner(onp.PlannerParams())
    wp, ts = pl.generate_init_variables(d["head"], d["tail"])
    pl.M = len(ts)
    D, M = d["head"].shape[1], len(ts)
    x0 = np.concatenate([wp.reshape(-1), pl.map_T2tau(ts)])
    head = np.zeros((3, D)); tail = np.zeros((3, D))
    head[:d["head"].shape[0]] = d["head"]; tail[:d["tail"].shape[0]] = d["tail"]
    path = os.path.join(tmp, "fixture.bin")
    with open(path, "wb") as f:
        f.write(struct.pack("<6i", W, H, M, D, int(d["r0_nit"]), int(d["r0_nfev"])))
        f.write(struct.pack("<3d", float(d["res"]), float(d["origin"][0]), float(d["origin"][1])))
        f.write(occ.tobytes())
        for a in (head, tail, x0, d["r0_x"]):
            f.write(np.ascontiguousarray(a, dtype=np.float64).tobytes())
        f.write(struct.pack("<d", float(d["final_cost"])))
    return path


def test_c_program_compiles_and_links_against_the_abi(tmp_path):
    exe = _build(str(tmp_path))
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_c_program_reproduces_a_reference_run(tmp_path):
    exe = _build(str(tmp_path))
    fx = _fixture(str(tmp_path))
    env = dict(os.environ)
    # torch's bundled HIP runtime is not involved here: the program links the library alone
    p = subprocess.run([exe, fx], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, (p.returncode, p.stdout, p.stderr)
    assert "C ABI ok" in p.stdout
