"""Builds libaerialgym_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m aerial_gym_simulator_amd._build [--force]

hipcc cross-compiles without a GPU.  The library links only against the HIP runtime;
nothing from torch crosses the C ABI (include/aerial_gym_hip.h).
"""
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libaerialgym_hip.so")
INCLUDE = os.path.join(os.path.dirname(PKG_DIR), "include")

SOURCES = ["agx_api.cpp", "agx_math_eval.hip", "agx_dynamics.hip", "agx_scene.hip", "agx_raycast.hip", "agx_lidar_nav.hip", "agx_imu.hip", "agx_task_glue.hip", "agx_exchange.hip", "agx_strict.hip"]
# -ffp-contract=off: every + - * / sqrt is one IEEE operation (bit-exact predicates, see
# DESIGN.md "numerics"); correctly rounded fp32 divide / sqrt is hipcc's default.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
         "-Wno-comment"]
# agx_raycast.hip: no SLP vectorisation.  The vectoriser packs the object node's dot products into v_pk_mul / v_pk_add_f32 on
# SCALAR operands, which need aligned register pairs: six s_mov per node visit hoisted in front of the node-kind branch, 20 more
# scalars spilled to lanes, and vector spills to scratch -- the frame is 6 % faster without it (profiles/r05_raycast_variants.txt).
PER_SOURCE_FLAGS = {"agx_raycast.hip": ["-fno-slp-vectorize"]}


def source_hash():
    """sha256 over the kernel sources, the C header and the compile flags: identifies the code a counter file under
    profiles/ was measured on (bench.py marks `traffic` stale when it differs).  The same string is compiled INTO the
    library (`agx_build_id()`, -DAGX_BUILD_ID), so `binary_is_current()` / `_lib.load()` compare sources and binary
    directly instead of trusting file times."""
    import hashlib

    h = hashlib.sha256((" ".join(FLAGS) + repr(sorted(PER_SOURCE_FLAGS.items()))).encode())
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".cpp")))
    for path in files + [os.path.join(INCLUDE, "aerial_gym_hip.h")]:
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def binary_build_id(path=None):
    """the build id embedded in a built library (None: no library, or one from before the id existed).  Read from the file's
    bytes: dlopen would pin a stale library in this process, and the rebuilt one of the same path could not be loaded after it"""
    path = path or LIB_PATH
    if not os.path.exists(path):
        return None
    tag = b"agx-build-id:"
    with open(path, "rb") as f:
        data = f.read()
    at = data.find(tag)
    if at < 0:
        return None
    end = data.find(b"\0", at)
    return data[at + len(tag):end].decode("ascii", "replace")


def binary_is_current():
    return binary_build_id() == source_hash()


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _run_parallel(cmds, verbose=False):
    """the translation units are independent: compile them side by side (agx_dynamics.hip alone takes ~50 s)"""
    if not cmds:
        return
    from concurrent.futures import ThreadPoolExecutor

    def one(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=min(len(cmds), os.cpu_count() or 4)) as ex:
        list(ex.map(one, cmds))


def build_library(force=False, verbose=False, extra_flags=(), lib_path=None):
    """extra_flags / lib_path: build an experimental variant next to the default library."""
    os.makedirs(LIB_DIR, exist_ok=True)
    if lib_path is not None:
        return _build_variant(extra_flags, lib_path, verbose)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(INCLUDE, "aerial_gym_hip.h"))
    objs, cmds = [], []
    build_id = source_hash()
    # the binary says which sources it was built from: anything else than the current hash is rebuilt, whatever the file
    # times are (a library copied from elsewhere, a checkout that reset mtimes)
    relink = force or binary_build_id() != build_id
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            raise RuntimeError(f"missing source {path}")
        obj = os.path.join(LIB_DIR, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        is_api = src == "agx_api.cpp"  # carries the build id: recompiled (1 s) with every relink
        if force or _stale(obj, [path] + headers) or (is_api and relink):
            extra = ['-DAGX_BUILD_ID="%s"' % build_id] if is_api else []
            cmds.append([hipcc] + FLAGS + PER_SOURCE_FLAGS.get(src, []) + extra + ["-I", INCLUDE, "-x", "hip", "-c", path, "-o", obj])
    _run_parallel(cmds, verbose)
    if relink or _stale(LIB_PATH, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIB_PATH


def _build_variant(extra_flags, lib_path, verbose):
    hipcc = _hipcc()
    objs, cmds = [], []
    tag = os.path.splitext(os.path.basename(lib_path))[0]
    for src in SOURCES:
        obj = os.path.join(LIB_DIR, f"{tag}_{os.path.splitext(src)[0]}.o")
        objs.append(obj)
        ident = ['-DAGX_BUILD_ID="%s+%s"' % (source_hash(), tag)] if src == "agx_api.cpp" else []
        cmds.append([hipcc] + FLAGS + PER_SOURCE_FLAGS.get(src, []) + list(extra_flags) + ident + ["-I", INCLUDE, "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj])
    _run_parallel(cmds, verbose)
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs, check=True)
    for o in objs:
        os.remove(o)
    return lib_path


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
