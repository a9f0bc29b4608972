"""Build libdruglamp_hip.so (gfx950) in-tree with hipcc.

    python -m druglamp_amd.build        # incremental
    python -m druglamp_amd.build -f     # force rebuild
    python -m druglamp_amd.build --study   # additionally libdruglamp_hip_study.so (-DDL_STUDY: reads the tile-study /
                                           # timing-decomposition switches from the environment; tools/ only, loaded with
                                           # DL_USE_STUDY_LIB=1 — the product library never reads the environment)

The shared object lands in druglamp_amd/lib/ (git-ignored; it travels to the GPU box with the
gpurun snapshot).  No cmake, no torch extension machinery: the C ABI has no torch types.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libdruglamp_hip.so")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"] + os.environ.get("DL_EXTRA_HIPCC_FLAGS", "").split()


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".cuh")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "druglamp_hip.h"))
    return hs


def csrc_hash() -> str:
    """sha1 over the kernel sources (csrc/*.hip, *.cuh and the C-ABI header), by name and content: what a recorded counter
    measurement (profiles/r*_pmc_summary.json) belongs to — bench.py reports `traffic` only from a summary taken at the same
    sources (measurement hygiene: .git does not travel to the GPU box, so this is a content hash)."""
    import hashlib
    h = hashlib.sha1()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".cuh"))]
    files.append(os.path.join(os.path.dirname(HERE), "include", "druglamp_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


# Per-file flags.  norm.hip: no SLP vectorisation.  hipcc (ROCm 7.2) turned the per-element fp32 arithmetic of the 16-byte
# LayerNorm backward into v_pk_mul/add/fma_f32 pairs whose LOW-half operands are written by the directly preceding
# v_lshlrev_b32 / v_and_b32 (bf16 unpacking); on MI355X those packed ops then read stale low halves in lanes 48-63 in a few
# launches per hundred WHEN ANOTHER PROCESS LOADS THE GPU (tools/contention_repeat.py: 8-27 mismatching launches of 80,
# always pass 1, lanes 48-63, even elements; 0 of 80 without the packed forms) — a timing-dependent forwarding hazard the
# compiler does not pad.  Found through the two-rank graph-vs-eager bit-identity test.
# elementwise.hip carries the same unpack-then-fp32 pattern (251 packed fp32 ops with the vectoriser on) and is HBM-bound: it
# gets the flag too (same-box step time unchanged).  bn.hip does NOT need it and stays on the vectoriser.
# Round 4 recorded that a no-SLP build of bn.hip fails tests/test_model_gpu.py::test_gcn_compact_padding_equals_the_512_row_computation
# (fp32, 5e-3 gradient error) although every dl_bn_* call agreed between the builds, and left it unexplained.  Round 5 resolved it
# (tools/bn_bisect.py records every ops call of both builds; profiles/r5_bn_bisect.txt): the builds' BatchNorm outputs differ in
# the last bit (another FMA contraction), which moves ONE pre-activation of the following ReLU-epilogue GEMM across zero (0 vs
# 1.7e-7 at a tensor maximum of 10.8); the backward then differs in that one row.  A ReLU kink in the test data, not a kernel
# fault: against the fp64 oracle both forms of the product build are at 4e-6 ... 1.2e-5 (test_gcn_full_and_compact_forms_
# against_the_fp64_oracle).  The contention test covers the BatchNorm statistics / apply / backward kernels (wide and generic,
# with and without row weights) for the packed-fp32 forwarding hazard the flag exists for.  gemm / attention keep the vectoriser as well
# (their epilogues rely on the packed forms: 15.70 -> 16.36 ms without) and are covered by tests/test_contention_gpu.py.
FILE_FLAGS = {"norm.hip": ["-fno-slp-vectorize"], "elementwise.hip": ["-fno-slp-vectorize"]}


def _compile(src, force, objdir=OBJDIR, extra=()):
    obj = os.path.join(objdir, src[:-4] + ".o")
    path = os.path.join(CSRC, src)
    if force or _stale(obj, [path] + _headers() + [os.path.abspath(__file__)]):
        keep = os.environ.get("DL_BUILD_NOSLP_FILES")           # (A/B builds, tools: comma-separated files that get their FILE_FLAGS)
        ff = FILE_FLAGS.get(src, []) if (keep is None or src in keep.split(",")) else []
        if src in os.environ.get("DL_BUILD_NOSLP_EXTRA", "").split(","):   # (A/B builds: further files without the SLP vectoriser)
            ff = ff + ["-fno-slp-vectorize"]
        cmd = [HIPCC] + FLAGS + ff + list(extra) + ["-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    return obj


def build(force: bool = False, verbose: bool = True, study: bool = False, variant: str = "", defines=()) -> str:
    """variant (tools only): a product-flavoured library under another name built with extra -D flags, loaded with
    DL_USE_STUDY_LIB=libdruglamp_hip_<variant>.so — same-box A/B of compile-time choices."""
    objdir = OBJDIR + ("_study" if study else "") + ("_" + variant if variant else "")
    lib = LIB.replace(".so", "_study.so") if study else LIB
    if variant:
        lib = LIB.replace(".so", "_%s.so" % variant)
    extra = (("-DDL_STUDY",) if study else ()) + tuple(defines)
    os.makedirs(objdir, exist_ok=True)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, objdir, extra), srcs))
    return _link(lib, objs, force, verbose)


def _link(LIB, objs, force, verbose):
    if force or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("built", LIB)
    elif verbose:
        print("up to date", LIB)
    return LIB


if __name__ == "__main__":
    if "--variant" in sys.argv:              # python -m druglamp_amd.build --variant nt7 -DDL_NT_MASK=7
        build(force="-f" in sys.argv, variant=sys.argv[sys.argv.index("--variant") + 1], defines=[a for a in sys.argv if a.startswith("-D")])
        sys.exit(0)
    build(force="-f" in sys.argv)
    if "--study" in sys.argv:
        build(force="-f" in sys.argv, study=True)
