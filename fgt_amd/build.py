"""Build libfgt_hip.so (gfx950) in-tree with hipcc.  `python -m fgt_amd.build` or `fgt_amd.build.build()`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libfgt_hip.so")
# (the eight long translation units first: with 8 jobs they all start at once and the build takes as long as the longest of them — 4 min 46 s -> ~3 min)
SOURCES = ["conv_igemm.hip", "conv_split.hip", "conv_f16.hip", "conv_taps.hip", "conv_taps_il.hip", "conv_taps_il_256x128.hip", "conv_wide.hip", "conv_taps_il_256x256.hip", "runtime.hip", "conv_c4.hip", "conv_direct.hip", "attention.hip", "attention_split.hip", "pointwise.hip", "flow_ops.hip", "laplace_fill.hip", "propagate.hip", "poisson_blend.hip", "solve_onchip.hip"]
# diagnostic builds only (build(variant=...)): measured-and-not-adopted schedule variants, trace instrumentation.  The product library never contains them.
DIAG_SOURCES = ["diag/conv_split_variants.hip", "diag/conv_taps_breg.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, variant=None, extra_flags=(), swap=None):
    """variant / extra_flags: diagnostic builds (lib/libfgt_hip_<variant>.so: the product sources + DIAG_SOURCES compiled with -DFGT_DIAG, e.g.
    `build(variant="diag")` for tools/split_sweep.py --diag, `build(variant="trace", extra_flags=["-DFGT_CONV_TRACE"])` for tools/conv_trace.py;
    selected at run time with FGT_HIP_LIB); the product library has neither.
    swap: {product source: diagnostic source} — a diagnostic build may replace a translation unit by its instrumented twin, e.g.
    {"attention_split.hip": "diag/attention_split_trace.hip"} (tools/attn_trace.py, tools/attn_ablate.py)."""
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj" + (f"_{variant}" if variant else ""))
    lib = os.path.join(LIBDIR, f"libfgt_hip_{variant}.so") if variant else LIB
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "conv_params.h"), os.path.join(CSRC, "conv_tile.h"), os.path.join(CSRC, "conv_taps_il.hip"), os.path.join(HERE, "..", "include", "fgt_hip.h")]
    hipcc = _hipcc()
    jobs = []
    objs = []
    if variant:
        extra_flags = list(extra_flags) + ["-DFGT_DIAG"]
    assert not swap or variant, "swap= is for diagnostic builds"
    for s in SOURCES + (DIAG_SOURCES if variant else []):
        s = (swap or {}).get(s, s)
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace("/", "_").replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + headers) or not os.path.exists(obj + ".usage"):
            # (-Rpass-analysis: the per-kernel register / scratch report goes to stderr; kept beside the object for tests/test_build_resources.py)
            jobs.append([hipcc] + FLAGS + list(extra_flags) + ["-c", src, "-o", obj, "-Rpass-analysis=kernel-resource-usage"])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
        if "-c" in cmd:
            with open(cmd[cmd.index("-o") + 1] + ".usage", "w") as f:
                f.write(r.stdout + r.stderr)

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        list(ex.map(run, jobs))
    if force or jobs or not os.path.exists(lib):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


def usage_report(src):
    """The compiler's kernel-resource-usage remarks of a product source as the last build() recorded them, or None when the object is missing
    or older than the source / headers (the caller compiles then)."""
    obj = os.path.join(LIBDIR, "obj", src.replace("/", "_").replace(".hip", ".o"))
    headers = [os.path.join(CSRC, h) for h in ("common.h", "conv_params.h", "conv_tile.h", "conv_taps_il.hip")] + [os.path.join(HERE, "..", "include", "fgt_hip.h")]
    if not os.path.exists(obj + ".usage") or _stale(obj, [os.path.join(CSRC, src)] + headers) or os.path.getmtime(obj + ".usage") < os.path.getmtime(obj):
        return None
    return open(obj + ".usage").read()





def csrc_hash():
    """sha1 over the kernel sources and the ABI header (what a PMC traffic profile depends on): profiles/kernel_traffic.json stores it, bench.py compares it
    with the running tree's — a profile stays valid across commits that do not touch a kernel."""
    import hashlib
    h = hashlib.sha1()
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))) 
    for f in files:
        h.update(f.encode())
        h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(HERE, "..", "include", "fgt_hip.h"), "rb").read())
    return h.hexdigest()[:12]


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
