"""Builds libescx.so (HIP, gfx950) in-tree: efficient-speech-codec_amd/esc/lib/libescx.so.

    python efficient-speech-codec_amd/build.py [--force]

hipcc cross-compiles for gfx950 without a GPU.  The .so is git-ignored but travels with the repo snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "esc", "lib")
# tuning builds: ESCX_BUILD_TAG=foo (with ESCX_EXTRA_CXXFLAGS) writes libescx_foo.so next to the product library; ESCX_LIB_TAG=foo makes esc/_native.py load it
TAG = os.environ.get("ESCX_BUILD_TAG", "")
OBJ_DIR = os.path.join(HERE, "build" + ("_" + TAG if TAG else ""))
LIB = os.path.join(OUT_DIR, "libescx" + ("_" + TAG if TAG else "") + ".so")
SOURCES = ["escx_api.cpp", "escx_params.cpp", "escx_profile.cpp", "collective.cpp", "train.hip", "disc.hip", "gemm_swin.hip", "gemm_misc.hip", "kernels_misc.hip", "fused_swin.hip"]
HEADERS = ["tune_env.h", "launch_prof.h", "split_terms.h", "fused_pvq.h", "fused_mlp_x3.h", "train_kernels.h", "train_mlp_fused.h", "disc_kernels.h", "gemm_bf16.h", "conv32_halo.h", "gemm_engine.h", "kernels.h", "launchers.h", "escx_internal.h", "fused_mlp.h", "fused_attn.h", "fused_rowgemm.h", "fused_deembed.h", os.path.join("..", "..", "include", "escx.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"] + os.environ.get("ESCX_EXTRA_CXXFLAGS", "").split()     # tuning builds only


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def build(force: bool = False, verbose: bool = True) -> str:
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _newest(deps):
        return LIB
    os.makedirs(OUT_DIR, exist_ok=True)
    os.makedirs(OBJ_DIR, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
        cmd = [HIPCC] + FLAGS + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=len(srcs)) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {LIB} ({os.path.getsize(LIB) / 1e6:.1f} MB)")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
