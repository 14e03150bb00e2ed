"""Builds libspegnet_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libspegnet_hip.so")
SOURCES = ["core.hip", "gemm.hip", "conv_halo.hip", "tn_block.hip", "attention.hip", "norm.hip", "elem.hip", "head.hip", "easpp.hip", "loss.hip", "optim.hip"]
# dev builds only (`--dev`): experiment kernels that are not part of the product library (csrc/dev/*.inc are included by gemm.hip under
# SPG_DEV_KERNELS at the places where those families used to stand)
DEV_SOURCES = ["dev/nt_wide.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-inline-asm", "-munsafe-fp-atomics"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, dev: bool = False) -> str:
    """dev=True (tools/ only, `python spegnet_amd/build.py --dev`): also compiles the superseded / experimental kernel families and the
    SPG_* ablation switches of csrc/gemm.hip (-DSPG_DEV_KERNELS) into libspegnet_hip_dev.so; the product library has neither."""
    global LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if dev:
        return _build_dev(hipcc, verbose)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.join(CSRC, "common.h"), os.path.join(HERE, "..", "include", "spegnet_hip.h")]
    objdir = os.path.join(HERE, "build")
    # A/B builds of the PRODUCT sources (tools/ only): SPG_VARIANT_TAG=_x SPG_VARIANT_FLAGS="-DTB_VARIANT=1" -> libspegnet_hip_x.so
    vtag, vflags = os.environ.get("SPG_VARIANT_TAG", ""), os.environ.get("SPG_VARIANT_FLAGS", "").split()
    lib = LIB
    if vtag:
        objdir = os.path.join(objdir, "variant" + vtag)
        lib = os.path.join(HERE, f"libspegnet_hip{vtag}.so")
    os.makedirs(objdir, exist_ok=True)

    def cc(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        path = os.path.join(CSRC, src)
        if force or _stale(obj, [path] + hdrs):
            cmd = [hipcc] + FLAGS + vflags + ["-c", path, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(cc, srcs))
    if force or _stale(lib, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    return lib


def _build_dev(hipcc, verbose):
    objdir = os.path.join(HERE, "build", "dev")
    os.makedirs(objdir, exist_ok=True)
    extra = os.environ.get("SPG_DEV_FLAGS", "").split()        # e.g. "-DSPG_V3_KS=1 -DSPG_V3_NS=4" -> its own library name
    tag = os.environ.get("SPG_DEV_TAG", "")
    lib = os.path.join(HERE, f"libspegnet_hip_dev{tag}.so")
    if tag:
        objdir = os.path.join(objdir, tag.strip("_"))
        os.makedirs(objdir, exist_ok=True)
    objs = []
    for src in [s_ for s_ in SOURCES + DEV_SOURCES if os.path.exists(os.path.join(CSRC, s_))]:
        obj = os.path.join(objdir, src.replace("/", "_").replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + ["-DSPG_DEV_KERNELS"] + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
        objs.append(obj)
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, dev="--dev" in sys.argv))
