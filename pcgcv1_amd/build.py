"""In-tree build of the two C-ABI libraries (include/pcgc.h).

    python -m pcgcv1_amd.build          # hipcc --offload-arch=gfx950 + g++

libpcgc_hip.so  : csrc/*.hip  (hipcc cross-compiles gfx950 without a GPU)
libpcgc_host.so : csrc/host.cpp (g++; no HIP dependency)
Both land in pcgcv1_amd/lib/ (git-ignored, shipped to the GPU box with the
snapshot).  Objects are rebuilt only when a source or header is newer.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib")
OBJ = os.path.join(HERE, "lib", "obj")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

HIP_SOURCES = {
    "conv_direct.hip": [],
    "conv_mfma.hip": [],
    "vrn_mfma.hip": [],
    "conv_valu.hip": [],
    "vrn_valu.hip": [],
    "vrn_row.hip": [],
    "vrn_seg.hip": [],
    "vrn_row32.hip": [],
    "vrn_row16.hip": [],
    "net.hip": [],
    "entropy.hip": ["-ffp-contract=off"],
    "tail.hip": ["-ffp-contract=off"],
    "train.hip": ["-ffp-contract=off"],
    "train_dw.hip": [],
    "train_plan.hip": [],
    "hyper_row.hip": [],
}
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
             "-fno-gpu-rdc"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + r.stderr)
        raise RuntimeError("build failed: " + os.path.basename(cmd[-1]))
    return r.stdout + r.stderr


def build(verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # every header a .hip file can include: editing any of them rebuilds all objects (they are few and small)
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join(INCLUDE, "pcgc.h")]
    jobs, objs = [], []
    # PCGC_EXPERIMENTS=1: the memory-ablation switches of the 64^3 row kernels (tools/exp/t_ablate.py); a stamp file makes a
    # change of the setting rebuild the one file that looks at it
    exp = os.environ.get("PCGC_EXPERIMENTS", "0") == "1"
    stamp = os.path.join(LIB, "experiments.stamp")
    exp_changed = (open(stamp).read().strip() if os.path.exists(stamp) else "0") != ("1" if exp else "0")
    # -fno-honor-nans for the row kernels' files (PCGC_NNAN=0: off): a ReLU on an MFMA result is then ONE v_max_f32 instead of a
    # canonicalising v_max_f32 v, v, v plus the maximum — every vector instruction between MFMAs costs about an MFMA issue slot
    # (tools/isa_mix.py: kernel BC 577 -> 457 vector instructions per three planes; round trip 39.5 -> 38.2 ms, same box, interleaved:
    # profiles/r06_vC_nnan_ab.txt).  Same bits for every input that is not a NaN (and max(NaN, 0) = 0 either way).
    nnan = os.environ.get("PCGC_NNAN", "1") == "1"
    for src, extra in HIP_SOURCES.items():
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(OBJ, src + ".o")
        objs.append(obj)
        if src in ("vrn_row.hip", "net.hip") and exp:
            extra = extra + ["-DPCGC_EXPERIMENTS"]
        if nnan and src in ("vrn_row.hip", "vrn_seg.hip", "vrn_row32.hip", "vrn_row16.hip", "hyper_row.hip"):
            extra = extra + ["-fno-honor-nans"]
        if _newer(obj, [path] + headers) or (src in ("vrn_row.hip", "net.hip") and exp_changed):
            jobs.append([hipcc] + HIP_FLAGS + extra + ["-I", INCLUDE, "-c", path, "-o", obj])
    with open(stamp, "w") as f:
        f.write("1" if exp else "0")
    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        for out in ex.map(_run, jobs):
            if verbose and out.strip():
                print(out)
    hip_so = os.path.join(LIB, "libpcgc_hip.so")
    if _newer(hip_so, objs):
        # Linked WITHOUT a DT_NEEDED on libamdhip64: the HIP runtime is whichever one the host process
        # already uses (PyTorch bundles its own libamdhip64.so; loading /opt/rocm's next to it would put two
        # HIP runtimes, with unrelated streams, in one process).  pcgcv1_amd/_lib.py makes the process's
        # runtime global before dlopen; a C/C++ host links libamdhip64 itself.
        _run(["g++", "-shared", "-fPIC", "-o", hip_so] + objs)
    host_so = os.path.join(LIB, "libpcgc_host.so")
    host_src = os.path.join(CSRC, "host.cpp")
    if _newer(host_so, [host_src] + headers):
        _run(["g++", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-pthread", "-ffp-contract=off", "-I", INCLUDE,
              host_src, "-o", host_so])
    return hip_so, host_so


if __name__ == "__main__":
    print(build(verbose=True))
