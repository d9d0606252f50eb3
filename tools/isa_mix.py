"""Static instruction mix of one kernel of a .hip translation unit: prologue / loops that contain MFMAs / epilogue.

    python tools/isa_mix.py pytorch-3dunet_amd/csrc/u3d_bf16.hip 'conv3d_bf16_kernelILi2ELi1ELi3ELi0EDF16b' [--top 12]

Compiles the file to gfx950 assembly with the build's flags, cuts out the first kernel whose mangled name contains the pattern and
counts VALU / SALU / DS / VMEM / MFMA instructions before the first MFMA loop, inside each loop that holds MFMAs, and after the last.
This is how round 3 found where the 7.4 VALU instructions per MFMA of the bf16-storage convolution kernel lived (DESIGN_HISTORY.md 4.7b):
static counts — multiply a loop's by its trip count, and read the epilogue's as an upper bound (it holds every mode's variant).
"""
import argparse
import collections
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-mllvm", "-pragma-unroll-threshold=200000"]


def classify(op):
    if op.startswith("v_mfma"):
        return "MFMA"
    for prefix, cls in (("v_", "VALU"), ("s_", "SALU"), ("ds_", "DS"), ("global_", "VMEM"), ("buffer_", "VMEM"), ("scratch_", "SCRATCH")):
        if op.startswith(prefix):
            return cls
    return None


def count(lines, top):
    total, ops = collections.Counter(), collections.Counter()
    for line in lines:
        line = line.strip()
        if not line or line[0] in ";.":
            continue
        op = line.split()[0]
        cls = classify(op)
        if cls:
            total[cls] += 1
            ops[f"{cls}:{op}"] += 1
    out = "  ".join(f"{k} {v}" for k, v in sorted(total.items()))
    if top:
        out += "\n" + "\n".join(f"      {v:5d} {k}" for k, v in ops.most_common(top))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("source")
    ap.add_argument("pattern")
    ap.add_argument("--top", type=int, default=0)
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "k.s")
        cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), *FLAGS, "-I", os.path.join(ROOT, "include"), "-I",
               os.path.join(ROOT, "pytorch-3dunet_amd", "csrc"), "-S", "--cuda-device-only", args.source, "-o", asm]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        text = open(asm).read().split("\n")
    start = next((i for i, l in enumerate(text) if l.endswith(":") is False and l.split(":")[0].find(args.pattern) >= 0 and l.startswith("_Z")), None)
    if start is None:
        sys.exit(f"no kernel matching {args.pattern!r}")
    end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
    body = text[start:end + 1]
    print(text[start].split(":")[0])
    mfma = [i for i, l in enumerate(body) if "v_mfma" in l]
    if not mfma:
        print("whole kernel:", count(body, args.top))
        return
    headers = [i for i, l in enumerate(body) if "Loop Header" in l]
    loops = []
    for k, h in enumerate(headers):
        stop = headers[k + 1] if k + 1 < len(headers) else len(body)
        inside = [i for i in mfma if h <= i < stop]
        if inside:
            back = next((i for i in range(inside[-1], len(body)) if "s_cbranch" in body[i]), stop)
            loops.append((h, back))
    first = loops[0][0] if loops else mfma[0]
    last = loops[-1][1] if loops else mfma[-1]
    print(f"prologue  [{0}:{first}]  ", count(body[:first], args.top))
    for h, b in loops:
        print(f"MFMA loop [{h}:{b}]  ", count(body[h:b], args.top))
    print(f"epilogue  [{last}:{len(body)}]  ", count(body[last:], args.top))


if __name__ == "__main__":
    main()
