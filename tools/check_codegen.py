#!/usr/bin/env python3
"""Developer tool (no GPU needed): compiles the integrator kernels to gfx950 assembly and reports, per kernel
variant, registers, spills, and the scratch (spill) instructions INSIDE the innermost traversal loop — a spill
reload there stalls every step on a scratch load (measured 0.75-0.8x on the dense configs), and whether the
register allocator puts one there changes with unrelated edits of the shading code.
usage: tools/check_codegen.py [--strict]   (--strict: exit 1 when a product kernel's traversal loop touches scratch)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "yocto-hair_amd", "csrc")
FLAGS = "-O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fno-vectorize -std=c++17 --cuda-device-only -S".split()
FLAGS += os.environ.get("YH_EXTRA_FLAGS", "").split()  # e.g. YH_EXTRA_FLAGS=-DYH_SUSPEND=8 to look at a developer variant


def kernels(src):
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, os.path.join(CSRC, src), "-o", f.name],
                       check=True, stderr=subprocess.DEVNULL)
        text = open(f.name).read()
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.sgpr_spill_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", text):
        meta[m.group(1)] = dict(scratch_bytes=int(m.group(2)), sgpr_spills=int(m.group(3)), vgprs=int(m.group(4)), vgpr_spills=int(m.group(5)))
    out = {}
    for name, info in meta.items():
        i = text.find("\n" + name + ":")
        if i < 0:
            continue
        body = text[i:text.find(".end_amdhsa_kernel", i)].split("\n")
        # loops: a header label and the last branch back to it; the traversal loop = the deepest loop of a few
        # hundred vector instructions that loads 16-byte records (the small loops inside it push scene entries)
        best, best_key = (0, 0, 0), (-1, -1)
        for k, l in enumerate(body):
            m = re.search(r"Loop Header: Depth=(\d+)", l)
            if not m:
                continue
            depth = int(m.group(1))
            lm = None
            for up in range(0, 8):  # the label sits on the header line or a few "Parent Loop" comment lines above it
                if k - up >= 0:
                    lm = re.match(r"(\.LBB\d+_\d+):", body[k - up])
                    if lm:
                        break
            if not lm:
                continue
            label = lm.group(1)
            end = max((j for j in range(k, len(body)) if re.search(r"s_c?branch\S*\s+" + re.escape(label) + r"\b", body[j])), default=k)
            seg = body[k:end + 1]
            valu = sum(1 for x in seg if re.match(r"\s+v_", x))
            if not (250 <= valu <= 2000) or sum(1 for x in seg if "global_load_dwordx4" in x) < 2:
                continue
            if (depth, valu) > best_key:
                best_key, best = (depth, valu), (valu, sum(1 for x in seg if "scratch_" in x), end - k)
        info.update(loop_valu=best[0], loop_scratch=best[1])
        out[name] = info
    return out


if __name__ == "__main__":
    bad = 0
    for src, pat in (("kernels.hip", r"k_traceILb0ELb[01]E"), ("stream.hip", r"k_streamILb[01]ELi\dELb0"), ("wavefront.hip", r"k_wavefront")):
        for name, k in sorted(kernels(src).items()):
            if not re.search(pat, name):
                continue
            flag = ""
            product = ("k_traceILb0ELb0" in name or "k_streamILb0" in name)  # the plain variants every BASELINE config runs
            if product and (k["loop_scratch"] or not k["loop_valu"]):
                flag, bad = "   <-- spill traffic in the traversal loop", bad + 1
            print(f"{name[:64]:64s} vgprs {k['vgprs']:3d} spilled {k['vgpr_spills']:3d} (sgpr {k['sgpr_spills']:3d}) scratch {k['scratch_bytes']:4d} B   "
                  f"traversal loop: {k['loop_valu']} VALU, {k['loop_scratch']} scratch ops{flag}")
    sys.exit(1 if (bad and "--strict" in sys.argv) else 0)
