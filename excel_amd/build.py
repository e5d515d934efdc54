"""Build libexcel_hip.so (gfx950) in-tree with hipcc.  `python -m excel_amd.build [--force]`."""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libexcel_hip.so")
SOURCES = ["gemm.hip", "gemm_bf16x3.hip", "gemm_w4.hip", "norm.hip", "attn.hip", "attn_strip.hip", "cam.hip", "aff.hip", "par.hip", "attr.hip", "lvc.hip", "decoder.hip", "train.hip", "crf.hip", "abi.hip"]
# the translation units that depend on the 16-bit type of the split operand planes are compiled twice: bf16 (namespace excel_bf16) and,
# with -DEXCEL_SPLIT_F16, IEEE half (namespace excel_f16, objects *_f16.o) - the "f16x3" matrix-core mode (common.h, excel_internal.h)
SPLIT_SOURCES = ["gemm.hip", "gemm_bf16x3.hip", "gemm_w4.hip", "norm.hip", "attn.hip", "attn_strip.hip", "cam.hip"]
# compiled for the IEEE-half split type only (object *_f16.o): the two-product GEMM for fp16-valued weights ("f16x2")
F16_ONLY_SOURCES = ["gemm_w4x2.hip"]
HEADERS = ["common.h", "excel_internal.h", "excel_split_api.inc", "gemm_w4_body.inc", "decoder_internal.h", os.path.join("..", "..", "include", "excel_hip.h")]
# kernels that must never touch scratch memory: a spill or a dynamically indexed accumulator array inside these turns a matrix-core loop
# into a memory loop (round 5: one `break` in an unrolled epilogue loop sent the 320x256 GEMM's accumulators to scratch, 3.5x slower,
# all tests green).  build() reads hipcc's kernel-resource-usage remarks and refuses to link a library that violates this.
NO_SCRATCH = ("gemm_bf16x3_kernel", "gemm_w4_kernel", "gemm_w4x2_kernel", "attn_strip_kernel", "par_iterate_guide_kernel", "par_stats_tile_kernel")
# the guard must see what it guards: every object that is supposed to hold one of these kernels has to report it in hipcc's remarks (an
# empty or re-formatted remark stream would otherwise pass vacuously - advisor, round 5)
NO_SCRATCH_EXPECTED = {"gemm_bf16x3": "gemm_bf16x3_kernel", "gemm_w4": "gemm_w4_kernel", "gemm_w4x2": "gemm_w4x2_kernel", "attn_strip": "attn_strip_kernel",
                       "par": "par_iterate_guide_kernel"}
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


def _parse_resources(stderr):
    """hipcc -Rpass-analysis=kernel-resource-usage remarks -> {mangled kernel name: {"vgprs", "agprs", "sgprs", "scratch", "occupancy", "lds"}}"""
    import re
    out, cur = {}, None
    keys = {"VGPRs": "vgprs", "AGPRs": "agprs", "SGPRs": "sgprs", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occupancy",
            "LDS Size [bytes/block]": "lds", "VGPRs Spill": "vgpr_spill", "SGPRs Spill": "sgpr_spill"}
    for line in stderr.splitlines():
        m = re.search(r"remark:\s+(.*?)\s*\[-Rpass-analysis", line)
        if not m:
            continue
        body = m.group(1)
        if body.startswith("Function Name:"):
            cur = body.split(":", 1)[1].strip()
            out[cur] = {}
        elif cur and ":" in body:
            k, v = body.rsplit(":", 1)
            if k.strip() in keys:
                try:
                    out[cur][keys[k.strip()]] = int(v)
                except ValueError:
                    pass
    return out


def _object_sig(src, cmd_flags):
    """Content signature of ONE object: its source, every header / .inc under csrc (any .hip may include any of them: a superset is
    cheap and never stale), the public header and the exact command-line flags.  Objects are rebuilt when this changes, whatever the
    time stamps say - so the library's build id (source_id over the same files) really describes every object linked into it."""
    import hashlib
    h = hashlib.sha256()
    h.update(" ".join(cmd_flags).encode())
    h.update(open(src, "rb").read())
    for f in sorted(f for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))):
        h.update(f.encode())
        h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(CSRC, "..", "..", "include", "excel_hip.h"), "rb").read())
    return h.hexdigest()[:16]


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
    import json
    sigfile = os.path.join(CSRC, ".build_objs")
    try:
        sigs = json.load(open(sigfile))
    except Exception:
        sigs = {}
    resfile = os.path.join(CSRC, ".build_resources.json")      # per object: hipcc's register / scratch / occupancy figures of every kernel
    try:
        resources = json.load(open(resfile))
    except Exception:
        resources = {}
    objs, jobs = [], []           # jobs: (command, object name, signature)
    for s in SOURCES + F16_ONLY_SOURCES:
        src = os.path.join(CSRC, s)
        variants = [("_f16", ["-DEXCEL_SPLIT_F16"])] if s in F16_ONLY_SOURCES else [("", [])] + ([("_f16", ["-DEXCEL_SPLIT_F16"])] if s in SPLIT_SOURCES else [])
        for suffix, vflags in variants:
            obj = os.path.join(CSRC, s.replace(".hip", suffix + ".o"))
            objs.append(obj)
            # abi.hip carries the id of ALL sources (excel_build_id): its flags change whenever any of them changed
            extra = ['-DEXCEL_BUILD_ID="%s"' % sid] if s == "abi.hip" else []
            sig_o = _object_sig(src, [hipcc] + flags + vflags + extra)
            name = os.path.basename(obj)
            if force or not os.path.exists(obj) or sigs.get(name) != sig_o:
                jobs.append(([hipcc] + flags + vflags + extra + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", obj], name, sig_o))
    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for (cmd, name, sig_o), res in zip(jobs, ex.map(lambda j: subprocess.run(j[0], capture_output=True, text=True), jobs)):
                if res.returncode != 0:
                    sigs.pop(name, None)
                    json.dump(sigs, open(sigfile, "w"))
                    raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), res.stderr))
                resources[name] = _parse_resources(res.stderr)
                want = NO_SCRATCH_EXPECTED.get(name.replace("_f16.o", "").replace(".o", ""))
                if want and not any(want in k and "scratch" in v for k, v in resources[name].items()):
                    sigs.pop(name, None)
                    json.dump(sigs, open(sigfile, "w"))
                    raise RuntimeError("%s: hipcc reported no resource usage for %s - the no-scratch guard cannot see the kernel it guards "
                                       "(remark format changed?)" % (name, want))
                bad = {k: v for k, v in resources[name].items() if any(n in k for n in NO_SCRATCH) and (v.get("scratch", 0) or v.get("vgpr_spill", 0))}
                if bad and "-DEXCEL_DEV" in flags:
                    # development builds carry ablation arms that raise the register pressure: report, do not refuse
                    print("[excel_amd.build] WARNING (dev build): scratch in hot kernels of %s: %s" % (name, {k[:60]: v.get("scratch") for k, v in bad.items()}))
                elif bad:
                    sigs.pop(name, None)
                    json.dump(sigs, open(sigfile, "w"))
                    raise RuntimeError("%s: hot kernels use scratch memory (spill / dynamically indexed register array): %s" % (name, bad))
                sigs[name] = sig_o
                if verbose:
                    print("[excel_amd.build] compiled", name)
        json.dump(sigs, open(sigfile, "w"), indent=0, sort_keys=True)
        json.dump(resources, open(resfile, "w"), indent=0, sort_keys=True)
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
