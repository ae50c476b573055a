#!/usr/bin/env python3
"""Developer tool (no GPU needed): compiles the integrator kernels to gfx950 assembly and reports, per kernel
variant, registers, spills, and the instruction mix of its TRAVERSAL loops — with the scratch (spill) instructions
inside them: a spill reload there stalls every step on a scratch load (measured 0.75-0.8x on the dense configs),
and whether the register allocator puts one there changes with unrelated edits of the shading code.

Loop membership comes from LLVM's own block annotations ("in Loop: Header=BBn_m Depth=d", "Parent Loop ..."), not
from the layout: the block placement may put a loop's latch above its header.
usage: tools/check_codegen.py [--strict] [--loops]
  --strict: exit 1 when a traversal loop of a product kernel touches scratch, when k_stream's step makes two dependent memory round trips
            (a wait between its buffer loads that an earlier one of them has to satisfy), or when an inner loop (>= 100 vector instructions) of the
            GENERAL k_stream or of an out-of-line device function holds a scratch instruction
  --loops:  list every other loop of >= 100 vector instructions too
  env YH_EXTRA_FLAGS="-D...": look at a developer variant"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "yocto-hair_amd", "csrc")
import shutil


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def makefile_flags():
    """The device flags of the product build (yocto-hair_amd/Makefile: HIPFLAGS), so that the code looked at is the code shipped."""
    for line in open(os.path.join(ROOT, "yocto-hair_amd", "Makefile")):
        m = re.match(r"HIPFLAGS\s*:=\s*(.*)", line)
        if m:
            flags = [f.replace("$(ARCH)", "gfx950") for f in m.group(1).split()]
            return [f for f in flags if f != "-fPIC" and not f.startswith("-I")]
    raise SystemExit("HIPFLAGS not found in yocto-hair_amd/Makefile")


FLAGS = makefile_flags() + ["--cuda-device-only", "-S"]
FLAGS += os.environ.get("YH_EXTRA_FLAGS", "").split()


def loops_of(body):
    """body: the lines of one kernel. Returns {header: counts}, every instruction counted in its innermost loop
    and in all the loops around it."""
    blocks, cur, parent, depth = [], None, {}, {}
    k = 0
    while k < len(body):
        l = body[k]
        m = re.match(r"(\.LBB\d+_\d+):", l) or re.match(r"; %bb\.(\d+):", l)
        if m:
            ann, j = [l], k + 1
            while j < len(body) and re.match(r"\s+;", body[j]):
                ann.append(body[j])
                j += 1
            a, inner = " ".join(ann), None
            hm = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", a)
            if hm:
                inner = m.group(1).replace(".L", "")
                depth[inner] = int(hm.group(1))
                ps = re.findall(r"Parent Loop (BB\d+_\d+) Depth=\d+", a)
                parent[inner] = ps[-1] if ps else None
            else:
                im = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=\d+", a)
                inner = im.group(1) if im else None
            cur = dict(loop=inner, ins=[])
            blocks.append(cur)
        elif cur is not None and re.match(r"\s+[a-z]", l):
            cur["ins"].append(l.strip())
        k += 1
    res = {}
    for b in blocks:
        h = b["loop"]
        while h:
            r = res.setdefault(h, dict(valu=0, salu=0, lds=0, vmem=0, x4=0, scratch=0, depth=depth.get(h, 0)))
            for x in b["ins"]:
                if x.startswith("v_"):
                    r["valu"] += 1
                elif x.startswith("s_"):
                    r["salu"] += 1
                elif x.startswith("ds_"):
                    r["lds"] += 1
                elif x.startswith("scratch_"):
                    r["scratch"] += 1
                elif x.startswith(("global_", "buffer_", "flat_")):
                    r["vmem"] += 1
                    r["x4"] += 1 if "dwordx4" in x else 0
            h = parent.get(h)
    return res


def loads_in_one_round_trip(body):
    """k_stream's step (csrc/dev_lane.h: lane_step, COOP) issues its loads as BUFFER loads, the lane's own node first, the segment it tests for
    the wave behind the exchange: ONE memory round trip. body: the lines of the kernel. Returns (number of buffer loads, offending line or None): a
    wait between the first and the last of them that lets fewer loads stay in flight than have been issued makes the later loads wait for the
    earlier ones — two dependent round trips per step, which is what the first form of the cooperative leaves did (profiles/r05/coop_line_leaves.txt)."""
    idx = [k for k, l in enumerate(body) if re.match(r"\s+buffer_load_dword", l)]
    if not idx:
        return 0, None
    issued = 0
    for k in range(idx[0], idx[-1] + 1):
        l = body[k]
        if re.match(r"\s+buffer_load_dword", l):
            issued += 1
            continue
        m = re.match(r"\s+s_waitcnt\b(.*)", l)
        if m:
            v = re.search(r"vmcnt\((\d+)\)", m.group(1))
            if v and int(v.group(1)) < issued:
                return len(idx), l.strip()
    return len(idx), None


def kernels(src):
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        subprocess.run([hipcc(), *FLAGS, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, os.path.join(CSRC, src), "-o", f.name],
                       check=True, stderr=subprocess.DEVNULL)
        text = open(f.name).read()
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.sgpr_spill_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", text):
        meta[m.group(1)] = dict(scratch_bytes=int(m.group(2)), sgpr_spills=int(m.group(3)), vgprs=int(m.group(4)), vgpr_spills=int(m.group(5)))
    out = {}
    # the out-of-line device functions of the file (noinline callees: their loops have a register budget of their own)
    funcs = {}
    for m in re.finditer(r"^(_ZN3yhd\w+):\s*; @", text, re.M):
        end = text.find(".Lfunc_end", m.end())
        funcs[m.group(1)] = loops_of(text[m.start():end].split("\n"))
    out["__functions__"] = funcs
    for name, info in meta.items():
        i = text.find("\n" + name + ":")
        if i < 0:
            continue
        body  = text[i:text.find(".end_amdhsa_kernel", i)].split("\n")
        loops = loops_of(body)
        info["buffer_loads"], info["serialising_wait"] = loads_in_one_round_trip(body)
        # traversal loops = the step loops of the BVH walk: a few hundred vector instructions around the 16-byte loads
        # of a node / leaf record (the sample and item loops around them hold thousands)
        info["loops"] = loops
        info["trav"] = {h: r for h, r in loops.items() if 250 <= r["valu"] <= 2500 and r["x4"] >= 2}
        out[name] = info
    return out


if __name__ == "__main__":
    bad = 0
    for src, pat in (("kernels.hip", r"k_traceILb0ELb[01]E"), ("wide.hip", r"k_trace_sbsILb[01]E"), ("exact.hip", r"k_trace_exact"), ("stream.hip", r"k_streamILb[01]ELi\dELb0")):
        ks = kernels(src)
        funcs = ks.pop("__functions__", {})
        for name, k in sorted(ks.items()):
            if not re.search(pat, name):
                continue
            product = ("k_traceILb0ELb0" in name or "k_trace_sbsILb0" in name or "k_streamILb0" in name or "k_trace_exactILb0" in name)  # the plain variants every BASELINE config runs
            print(f"{name[:64]:64s} vgprs {k['vgprs']:3d} spilled {k['vgpr_spills']:3d} (sgpr {k['sgpr_spills']:3d}) scratch {k['scratch_bytes']:4d} B")
            for h, r in k["trav"].items():
                flag = ""
                # k_stream's step loop also holds the refill and the retire blocks (once per ray, not per step): one reload
                # in each is the known state; k_trace's traversal loops are pure step loops
                allowed = 2 if "k_stream" in name else 0
                if product and r["scratch"] > allowed:
                    flag, bad = "   <-- spill traffic in a traversal loop", bad + 1
                print(f"    loop {h:10s} depth {r['depth']}: {r['valu']:4d} VALU {r['salu']:4d} SALU {r['lds']:3d} LDS {r['vmem']:3d} VMEM, {r['scratch']} scratch ops{flag}")
            if "k_stream" in name:
                if k["buffer_loads"] == 0:
                    print("    the step's buffer loads were not found   <-- the check needs a look")
                    bad += 1 if product else 0
                elif k["serialising_wait"]:
                    print(f"    {k['buffer_loads']} buffer loads, and `{k['serialising_wait']}` between them   <-- the step makes two dependent memory round trips")
                    bad += 1 if product else 0
                else:
                    print(f"    {k['buffer_loads']} buffer loads of the step in one round trip (no wait between them that an earlier one has to satisfy)")
            if "k_streamILb1" in name:
                # the GENERAL k_stream (round 6): no scratch instruction inside any INNER loop of 100 vector instructions or more
                # (depth >= 2: the stages' loops; the depth-1 loop is the scheduler around the whole kernel, where the stages' few spills live)
                inner = {h: r for h, r in k["loops"].items() if r["valu"] >= 100 and r["depth"] >= 2}
                dirty = {h: r["scratch"] for h, r in inner.items() if r["scratch"]}
                print(f"    {len(inner)} inner loops of >= 100 vector instructions, scratch instructions inside them: {dirty if dirty else 'none'}")
                if dirty:
                    print("    <-- the GENERAL k_stream has spill traffic inside a loop again")
                    bad += 1
            if product and not k["trav"]:
                print("    no traversal loop found   <-- the heuristic needs a look")
                bad += 1
            if "--loops" in sys.argv:
                for h, r in k["loops"].items():
                    if r["valu"] >= 100 and h not in k["trav"]:
                        print(f"    (loop {h:10s} depth {r['depth']}: {r['valu']:5d} VALU {r['salu']:5d} SALU, {r['scratch']} scratch ops)")
        for fname, loops in sorted(funcs.items()):  # the noinline callees of this file: every loop of 100 vector instructions or more
            big = {h: r for h, r in loops.items() if r["valu"] >= 100}
            dirty = {h: r["scratch"] for h, r in big.items() if r["scratch"]}
            print(f"{fname[:64]:64s} out of line: {len(big)} loops of >= 100 vector instructions, scratch instructions inside them: {dirty if dirty else 'none'}")
            if dirty and "lane_trace_exact" not in fname:  # (lane_trace_exact: the compare-and-select form for axis-parallel rays, one ray in millions: known, not on any config's path)
                bad += 1
    sys.exit(1 if (bad and "--strict" in sys.argv) else 0)
