"""Kernel-experiment builds of libcapdec.so with parts of the f32x3 GEMM knocked out (timing probes
only: results are wrong by construction).  Output: tools/_dbg/libcapdec_<variant>.so, selected with
CAPDEC_LIBRARY=...  Not part of the product."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "imagecaption_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_dbg")
src = open(os.path.join(CSRC, "gemm_f32x3.hip")).read()

def variant(name):
    s = src
    if name == "nomfma":
        s = re.sub(r"__builtin_amdgcn_mfma_f32_32x32x16_bf16\((\w+\[\d\]\[\w\]), (\w+\[\d\]\[\w\]), ([\w\[\]]+), 0, 0, 0\)",
                   r"probe_fake(\1, \2, \3)", s)
        s = s.replace("namespace capdec {", "namespace capdec {\ntypedef float f32x16p __attribute__((ext_vector_type(16)));\ntypedef __bf16 bf16x8p __attribute__((ext_vector_type(8)));\n__device__ __forceinline__ f32x16p probe_fake(bf16x8p a, bf16x8p b, f32x16p c) { c[0] += (float)a[0] * (float)b[7]; return c; }\n", 1)
    elif name == "nostore":
        s = s.replace("            const Split4 s = split4(r[i]);\n            *reinterpret_cast<bf16x4*>(img + off) = s.t0;\n            *reinterpret_cast<bf16x4*>(img + PLANE + off) = s.t1;\n            *reinterpret_cast<bf16x4*>(img + 2 * PLANE + off) = s.t2;",
                      "            asm volatile(\"\" ::\"v\"(r[i].x), \"v\"(r[i].y), \"v\"(r[i].z), \"v\"(r[i].w), \"v\"(off));")
    elif name == "nosplit":
        s = s.replace("            const Split4 s = split4(r[i]);", "            Split4 s; s.t0 = __builtin_bit_cast(bf16x4, make_float2(r[i].x, r[i].y)); s.t1 = __builtin_bit_cast(bf16x4, make_float2(r[i].z, r[i].w)); s.t2 = s.t0;")
    elif name == "noload":
        s = re.sub(r"r\[i\] = INTERIOR \? \*reinterpret_cast<const float4\*>\(ptr\) : xload4<VEC>\(ptr, [^;]*;",
                   "r[i] = make_float4((float)(size_t)ptr, 1.f, 2.f, 3.f);", s)
    elif name == "full":
        pass
    else:
        raise SystemExit(name)
    return s

objs = [os.path.join(CSRC, "_build", f) for f in os.listdir(os.path.join(CSRC, "_build")) if f.endswith(".o") and f != "gemm_f32x3.o"]
for name in sys.argv[1:]:
    tmp = os.path.join(CSRC, f"_probe_{name}.hip")   # beside its headers
    open(tmp, "w").write(variant(name))
    obj = os.path.join(OUT, f"x3_{name}.o")
    try:
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-c", tmp, "-o", obj], check=True)
    finally:
        os.remove(tmp)
    lib = os.path.join(OUT, f"libcapdec_{name}.so")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj] + objs, check=True)
    print("built", lib)
