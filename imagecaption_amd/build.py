"""Build libcapdec.so (hand-written gfx950 HIP kernels + C ABI) in-tree with hipcc.

    python -m imagecaption_amd.build [--force]

Each ``csrc/*.hip`` is compiled to an object under ``csrc/_build/`` (skipped when up to date), then
linked into ``imagecaption_amd/libcapdec.so``.  hipcc cross-compiles for gfx950 without a GPU.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_build")
LIB = os.path.join(HERE, "libcapdec.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]
# Per-file extras.  -amdgpu-mfma-vgpr-form: MFMA accumulators live in ordinary VGPRs (gfx950 has one unified register
# file) instead of AGPRs -- kernels that touch their accumulators with vector instructions between products (softmax
# rescale, masking, bf16 packing, stores) otherwise pay a v_accvgpr_read / _write pair around every such touch.
FILE_FLAGS = {
    "attention_bf16.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    "attention_f32_ring.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    # no SLP vectorisation in the files that hold the vector epilogue (gemm_common.h, epilogue_store_vec)
    "gemm_bf16s_dma.hip": ["-fno-slp-vectorize"],
    "gemm_bf16s_p8.hip": ["-fno-slp-vectorize"],
    "gemm_bf16s.hip": ["-fno-slp-vectorize"],
    "gemm_bf16.hip": ["-fno-slp-vectorize"],
    "gemm_f32x3.hip": ["-fno-slp-vectorize"],
    "gemm_f32.hip": ["-fno-slp-vectorize"],
}


EXTRA = os.environ.get("CAPDEC_EXTRA_FLAGS", "").split()  # diagnostic builds (-DCAPDEC_...); never set by the package


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "capdec.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hmt = _deps_mtime()
    hipcc = _hipcc()
    jobs = []
    for s in srcs:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s[:-4] + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hmt):
            jobs.append((src, obj))

    def run(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + EXTRA + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
        if verbose and r.stderr.strip():
            print(r.stderr[-2000:], file=sys.stderr)
        return obj

    if jobs:
        if verbose:
            print(f"[capdec build] compiling {len(jobs)} file(s): {[os.path.basename(j[0]) for j in jobs]}")
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, s[:-4] + ".o") for s in srcs]
    if jobs or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(o) for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
        if verbose:
            print(f"[capdec build] linked {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
