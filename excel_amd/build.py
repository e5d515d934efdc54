"""Build libexcel_hip.so (gfx950) in-tree with hipcc.  `python -m excel_amd.build [--force]`."""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libexcel_hip.so")
SOURCES = ["gemm.hip", "gemm_bf16x3.hip", "norm.hip", "attn.hip", "attn_strip.hip", "cam.hip", "aff.hip", "par.hip", "attr.hip", "lvc.hip", "decoder.hip", "train.hip", "crf.hip", "abi.hip"]
# the translation units that depend on the 16-bit type of the split operand planes are compiled twice: bf16 (namespace excel_bf16) and,
# with -DEXCEL_SPLIT_F16, IEEE half (namespace excel_f16, objects *_f16.o) - the "f16x3" matrix-core mode (common.h, excel_internal.h)
SPLIT_SOURCES = ["gemm.hip", "gemm_bf16x3.hip", "norm.hip", "attn.hip", "attn_strip.hip", "cam.hip"]
HEADERS = ["common.h", "excel_internal.h", "excel_split_api.inc", "decoder_internal.h", os.path.join("..", "..", "include", "excel_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value",
         # fully unroll the big register-tile epilogues (a partially unrolled loop indexes the accumulator array
         # dynamically and sends it to scratch)
         "-mllvm", "-pragma-unroll-threshold=100000"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def source_id(dev=None):
    """sha256 (first 16 hex digits) of every kernel source and header the library is compiled from (file names + bytes, sorted;
    `-dev` appended for an EXCEL_DEV build).  build() embeds it in the library (`excel_build_id()`), bench.py prints both: a stale
    `.so` whose time stamps happen to look fresh cannot be benchmarked unnoticed."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".inc")))
    for f in files:
        h.update(f.encode())
        h.update(open(os.path.join(CSRC, f), "rb").read())
    pub = os.path.join(CSRC, "..", "..", "include", "excel_hip.h")
    h.update(b"include/excel_hip.h")
    h.update(open(pub, "rb").read())
    if dev is None:
        dev = os.environ.get("EXCEL_DEV") == "1"
    return h.hexdigest()[:16] + ("-dev" if dev else "")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """EXCEL_DEV=1 in the environment adds -DEXCEL_DEV: ablation / experiment switches read from environment variables are compiled
    in (development only; the shipped library has none)."""
    hipcc = _hipcc()
    force = force or os.environ.get("EXCEL_BUILD_FORCE") == "1"      # the driver-side "does it build from source" check
    flags = FLAGS + (["-DEXCEL_DEV"] if os.environ.get("EXCEL_DEV") == "1" else [])
    sid = source_id()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    # objects built with other flags (a dev build, another compiler) are stale whatever their time stamps say
    stamp = os.path.join(CSRC, ".build_flags")
    sig = " ".join([hipcc] + flags)
    if not force and (not os.path.exists(stamp) or open(stamp).read() != sig):
        force = os.path.exists(LIB) or any(os.path.exists(os.path.join(CSRC, s.replace(".hip", ".o"))) for s in SOURCES)
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
        extra = []
        stale = force or _stale(obj, [src] + hdrs)
        if s == "abi.hip":
            # abi.hip carries the id of ALL sources (excel_build_id): it is rebuilt whenever any of them changed
            extra = ['-DEXCEL_BUILD_ID="%s"' % sid]
            idstamp = os.path.join(CSRC, ".build_id")
            stale = stale or not os.path.exists(idstamp) or open(idstamp).read() != sid
        if stale:
            jobs.append([hipcc] + flags + extra + ["-c", src, "-o", obj])
        if s in SPLIT_SOURCES:
            obj16 = os.path.join(CSRC, s.replace(".hip", "_f16.o"))
            objs.append(obj16)
            if force or _stale(obj16, [src] + hdrs):
                jobs.append([hipcc] + flags + ["-DEXCEL_SPLIT_F16", "-c", src, "-o", obj16])
    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for cmd, res in zip(jobs, ex.map(lambda c: subprocess.run(c, capture_output=True, text=True), jobs)):
                if res.returncode != 0:
                    raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), res.stderr))
                if verbose:
                    print("[excel_amd.build] compiled", os.path.basename(cmd[-1]))
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("link failed: %s\n%s" % (" ".join(cmd), res.stderr))
        if verbose:
            print("[excel_amd.build] linked", LIB)
    with open(stamp, "w") as f:
        f.write(sig)
    with open(os.path.join(CSRC, ".build_id"), "w") as f:
        f.write(sid)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
