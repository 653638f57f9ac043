"""Builds libneo_planner_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

The library is several translation units -- the C ABI plus one unit per kernel family (neo_disp_*.hip) -- compiled
in parallel and linked into one shared object.  Objects are cached under csrc/build/ by CONTENT: an object's name carries a
hash of its command line, its source and every header; the library's stamp file the hash of all of them.  Time stamps
play no part (a checkout that restores a file, or a copy of the tree that does not keep them, rebuilds nothing)."""
import concurrent.futures
import hashlib
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(PKG), "csrc")
INCLUDE = os.path.join(os.path.dirname(os.path.dirname(PKG)), "include")
# NEO_BUILD_OUT: where an experiment build (NEO_BUILD_DEFS) is written; the product is always the in-tree path
LIB = os.environ.get("NEO_BUILD_OUT") or os.path.join(PKG, "libneo_planner_hip.so")
OBJDIR = os.path.join(CSRC, "build")
SOURCES = ["neo_abi.hip", "neo_disp_eval.hip", "neo_disp_sample.hip", "neo_disp_opt2d.hip", "neo_disp_opt2d_x.hip", "neo_disp_opt3d_f32.hip",
           "neo_disp_opt3d_f64.hip", "neo_disp_opt3d_w2.hip", "neo_disp_opt3d_x.hip", "neo_disp_group.hip", "neo_disp_opt3d_b.hip"]
HEADERS = ["neo_device.hpp", "neo_kernels.hpp", "neo_host.hpp", "neo_launch_opt.hpp", "neo_lbfgs.hpp",
           "neo_linesearch.hpp", "neo_lbfgs_sm.hpp", "neo_lbfgs_dir.hpp", "neo_group_kernel.hpp"]
STAMP = LIB + ".stamp"    # hash of command lines + sources + headers the library was built from (travels with the library)


def _hipcc():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libneo_planner_hip.so")
    return hipcc


def _flags():
    # NEO_BUILD_DEFS="-DNEO_STAMPS ..." : experiment builds only (tools/); the product is built without
    # -fno-slp-vectorize: packed fp32 operations (v_pk_*) want aligned register pairs; in these register-bound kernels
    # they cost moves and spills (8 v_mov per joint of the factor recurrence).  Measured on the MI355X with and without:
    # cfg2 all-fp32 943 k -> 995 k traj/s, cfg3 12.5 M -> 15.2 M, cfg4 888 k -> 945 k, cfg5 244 k -> 259 k; the
    # mixed-precision and the ESDF sample kernels unchanged.
    # -fno-strict-aliasing: the kernels view one LDS staging buffer as doubles, floats, ints and 16-byte vectors in turn; under
    # type-based alias analysis the compiler may (and in round 5 did, in one fp64 instantiation) move a load of one view
    # across a store of another.  The hand-overs are fenced as well; this keeps an unfenced one from becoming a silent
    # wrong result.
    # -ffp-contract=on (every unit since round 5): a multiply-add is fused where it is written as one expression and nowhere
    # else, so the bits a kernel computes follow from its source, not from what the optimiser happened to schedule next to
    # what (HIP's default `fast` fuses across statements as the surrounding code allows: in round 4 an unrelated change
    # re-rounded the all-fp32 kernels, in round 5 adding -fno-strict-aliasing re-rounded the fp64 ones).  Measured with the
    # fp64 / mixed units switched over: cfg2 fp64 633 -> 635 k traj/s, mixed 770 -> 771 k; recorded reference runs within 1e-9 of the
    # reference's finals 19 -> 21 of 22 (the two replan runs that used to end 1.3e-4 away), G6 finals within 1e-4 of the
    # reference's 89.2 -> 90.0 % (the reference against itself: 89.6 %).
    return ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-fno-strict-aliasing",
            "-ffp-contract=" + os.environ.get("NEO_FP_CONTRACT", "on"), "-Wno-unused-value", "-I", INCLUDE] + \
        os.environ.get("NEO_BUILD_DEFS", "").split()


# headers only one unit includes: a change there recompiles that unit alone
UNIT_HEADERS = {}


def _headers(src=None):
    """the headers every unit sees (src None) or the ones only `src` includes.  (The experiment bodies of rounds 3 - 5 are
    patches under tools/probe/ -- sample_experiment_hooks.patch re-adds their hooks -- and no longer part of any build.)"""
    if src is not None:
        return [os.path.join(CSRC, h) for h in UNIT_HEADERS.get(src, [])]
    return [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(INCLUDE, "neo_planner.h")]


def _headers_hash(src=None):
    h = hashlib.sha256()
    for d in _headers(src):
        h.update(os.path.basename(d).encode() + b"\0" + open(d, "rb").read() + b"\0")
    return h.hexdigest()


# per-unit compiler options: the all-fp32 optimiser kernels are allocated for three wavefronts per SIMD (168 registers);
# LLVM's alternative register-pressure tracker spills fewer registers there (25 instead of 33; 19 without SLP)
# The fp64 kernels in the two-wavefronts-per-SIMD allocation (256 registers): machine LICM hoists loop-invariant values --
# the fp64 coefficients of exp(), LDS addresses -- out of the optimiser loop and the allocator then spills them (31
# registers in the cfg2 parity kernel, reloaded one after the other at every evaluation); letting the sinking pass move
# such instructions back into the loop leaves 7 and is worth 13 % (549 k -> 620 k traj/s; the all-fp32 kernels do not
# care: 1.26 M either way).
_SINK = ["-mllvm", "-sink-insts-to-avoid-spills"]
UNIT_FLAGS = {"neo_disp_opt3d_x.hip": ["-mllvm", "-amdgpu-use-amdgpu-trackers"],
              "neo_disp_opt2d_x.hip": ["-mllvm", "-amdgpu-use-amdgpu-trackers"],
              "neo_disp_opt3d_b.hip": ["-mllvm", "-amdgpu-use-amdgpu-trackers"],
              # (the fp64 unit also without machine LICM: no spills at all in its two-waves kernels, 618 k -> 640 k; the
              #  same pair costs the mixed mode 6 % and the all-fp32 mode 2 %, so only there)
              "neo_disp_opt3d_f64.hip": _SINK + ["-mllvm", "-disable-machine-licm"],
              "neo_disp_opt3d_w2.hip": _SINK, "neo_disp_opt2d.hip": _SINK}


def _unit_flags(src):
    return list(UNIT_FLAGS.get(src, []))


def _compile(src, obj, verbose):
    cmd = [_hipcc()] + _flags() + _unit_flags(src) + ["-c", os.path.join(CSRC, src), "-o", obj + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(obj + ".tmp", obj)


def _key(src=None, headers_hash=None):
    """hash of everything that decides what the compiler produces: this file (options, per-unit options, source list), the
    experiment definitions, the headers and the source(s) -- of unit `src`, or of the whole library (src None).  An object
    or library built from anything else is not reused (options; round 5: contents instead of time stamps)"""
    h = hashlib.sha256()
    h.update(open(os.path.abspath(__file__), "rb").read())
    h.update(" ".join(_flags() + (_unit_flags(src) if src else [])).encode())
    h.update((headers_hash or _headers_hash()).encode())
    for s in ([src] if src else SOURCES):
        h.update(s.encode() + b"\0" + open(os.path.join(CSRC, s), "rb").read() + b"\0")
        h.update(_headers_hash(s).encode())
    return h.hexdigest()[:12]


def build(force=False, verbose=False, jobs=None):
    """compile the translation units that are missing or older than their sources (in parallel), link; returns the
    library's path"""
    if os.environ.get("NEO_BUILD_DEFS", "").strip() and not os.environ.get("NEO_BUILD_OUT"):
        # an experiment build must never replace the in-tree product library: _lib.load() would silently use it
        raise RuntimeError("NEO_BUILD_DEFS is for experiment builds: set NEO_BUILD_OUT=<path of the experiment library> as well "
                           "(and NEO_PLANNER_LIB=<that path> when running it)")
    hh = _headers_hash()
    lib_key = _key(None, hh)
    stamp = open(STAMP).read().strip() if os.path.exists(STAMP) else ""
    if not force and os.path.exists(LIB) and stamp == lib_key:
        return LIB          # (the GPU box receives the built library and its stamp without the object cache)
    os.makedirs(OBJDIR, exist_ok=True)
    todo, objs = [], []
    for s in SOURCES:
        obj = os.path.join(OBJDIR, s.replace(".hip", "." + _key(s, hh) + ".o"))
        objs.append(obj)
        if force or not os.path.exists(obj):
            todo.append((s, obj))
    jobs = jobs or int(os.environ.get("NEO_BUILD_JOBS", "0")) or min(len(todo) or 1, os.cpu_count() or 1, 8)
    with concurrent.futures.ThreadPoolExecutor(max_workers=max(jobs, 1)) as ex:
        for f in [ex.submit(_compile, s, o, verbose) for s, o in todo]:
            f.result()
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    with open(STAMP, "w") as f:
        f.write(lib_key + "\n")
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose=True))
