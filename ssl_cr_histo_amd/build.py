"""Build libsslcr.so (HIP kernels + C-ABI) in-tree for gfx950 with hipcc.  No JIT cache: the built
.so travels with the repo snapshot to the GPU box."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsslcr.so")
SOURCES = ["conv_igemm.hip", "conv_halo.hip", "conv_halo256.hip", "conv_h16.hip", "conv_pp64.hip", "conv_dma.hip", "conv_s2.hip", "conv_s2d.hip", "conv_fp8.hip", "conv_wgrad.hip", "wgrad_halo.hip", "wgrad_dma.hip", "wgrad_s2.hip", "stem.hip", "stem_pool.hip", "augment.hip", "bn_eltwise.hip", "heads.hip", "optim.hip",
           "engine.cpp", "capi.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-fvisibility=hidden"]
# No SLP vectoriser where a wave's VALU work runs BESIDE its SIMD partner's MFMA stream (the ping-pong conv, the role-split stem
# backward): the vectoriser turns pairs of fp32 operations into v_pk_add_f32 / v_pk_fma_f32, and next to a full-rate MFMA stream a
# packed fp32 instruction gets issued once per ~140 cycles where a plain VALU instruction gets 1.3 per MFMA
# (profiles/r04_partner_instruction_cost.txt).  Same box: conv3x3_pp64 -2.4 ... -4 % per launch, stem_wgrad_pool2 -2 %, and the
# kernels fit their registers without spilling (256 + 12-32 B of scratch -> 228-236).  Neutral on the barrier-locked kernels
# (conv3x3_h16, conv3x3_halo256, conv_dma: left alone), harmful on wgrad3x3_halo (+57 % on <16,1>: left alone).
PER_FILE_FLAGS = {"conv_pp64.hip": ["-fno-slp-vectorize"], "stem.hip": ["-fno-slp-vectorize"]}


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def header_symbols():
    """the C-ABI entry points, read from include/sslcr.h"""
    import re
    with open(os.path.join(os.path.dirname(HERE), "include", "sslcr.h")) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(sslcr_[a-z0-9_]+)\s*\(", text)))


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "sslcr.h"))
    objs, procs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        if not os.path.exists(src):
            continue
        obj = os.path.join(objdir, s.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            cmd = [hipcc] + FLAGS + PER_FILE_FLAGS.get(s, []) + ["-x", "hip", "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out}")
        # a .hip object without device code means the HOST pass dropped the kernels without a diagnostic (seen in round 4: a
        # device-only type in a kernel body; hipcc returned 0 and the launches compiled to nothing)
        if s.endswith(".hip"):
            obj = os.path.join(objdir, s.rsplit(".", 1)[0] + ".o")
            with open(obj, "rb") as f:
                if b".hip_fatbin" not in f.read():
                    os.remove(obj)
                    raise RuntimeError(f"{s}: the object carries no device code (.hip_fatbin missing) -- kernels dropped by the host pass")
    # exactly the entry points include/sslcr.h declares are exported: -fvisibility=hidden + the header's visibility push covers
    # the functions, the version script also makes the compiler-generated kernel handles and template instances local
    vmap = os.path.join(objdir, "exports.map")
    with open(vmap, "w") as f:
        f.write("{ global: " + " ".join(n + ";" for n in header_symbols()) + " local: *; };\n")
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-L/opt/rocm/lib", "-lrccl",
                                                                                      f"-Wl,--version-script={vmap}"]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
