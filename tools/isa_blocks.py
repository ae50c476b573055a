#!/usr/bin/env python3
"""Developer tool (no GPU needed): basic blocks of one kernel in a gfx950 assembly file, with the instruction mix
of each — what a per-branch budget of a traversal step is made from (profiles/r04/k_stream_branch_budget.txt).

usage: tools/isa_blocks.py file.s kernel-name-substring [--loop BBn_m] [--dump BBn_m] [--marks]
  --marks    the file was compiled with -DYH_ISA_MARKS: instructions per "; YHMARK name" section, in layout order
             (a section = from one mark to the next; the same name met twice adds up)
  --loop H   only the blocks LLVM annotates as inside the loop with header H (any depth below it)
  --dump B   print the instructions of block B
A block's cost line: VALU (of which transcendental = quarter rate, v_div_* = the IEEE division's helpers), SALU, VMEM, LDS,
and the issue cycles of its vector instructions on a SIMD shared by two or more waves (2 per instruction, 4 per
transcendental / 64-bit one: MI355X_MICROARCH.md, "vector-instruction ISSUE cost", halved for a shared SIMD)."""
import re, sys


def kernel_body(text, pat):
    for m in re.finditer(r"^(_Z\S+):\s*;\s*@", text, re.M):
        if pat in m.group(1):
            j = text.find(".end_amdhsa_kernel", m.end())
            k = text.find(".Lfunc_end", m.end())
            return m.group(1), text[m.start():min(x for x in (j, k) if x > 0)].split("\n")
    raise SystemExit("no kernel matching " + pat)


TRANS = re.compile(r"v_(rcp|sqrt|rsq|exp|log|sin|cos)_")
DP = re.compile(r"v_\w+_(f64|u64|i64|b64)\b|v_mad_u64|v_mul_hi|v_mul_lo_u32")


def blocks_of(body):
    blocks, cur = [], None
    i = 0
    while i < len(body):
        l = body[i]
        m = re.match(r"\.L(BB\d+_\d+):", l) or re.match(r"; %bb\.(\d+):", l)
        if m:
            cur = dict(name=m.group(1), ins=[], ann=l[m.end():] + " ")
            blocks.append(cur)
            j = i + 1
            while j < len(body) and re.match(r"\s+;", body[j]):
                cur["ann"] += body[j].strip() + " "
                j += 1
        elif cur is not None and re.match(r"\s+[a-z]", l):
            cur["ins"].append(l.strip())
        i += 1
    for b in blocks:
        a = b["ann"]
        b["loops"] = re.findall(r"(?:Header=|Parent Loop )(BB\d+_\d+)", a)
        if re.search(r"Loop Header", a):
            b["loops"].append(b["name"])
        c = dict(valu=0, trans=0, div=0, dp=0, salu=0, vmem=0, lds=0, scratch=0, cyc=0)
        for x in b["ins"]:
            op = x.split()[0]
            if op.startswith("v_"):
                c["valu"] += 1
                t = bool(TRANS.match(op))
                d = bool(DP.match(op))
                c["trans"] += t
                c["dp"] += d
                c["div"] += op.startswith("v_div_")
                c["cyc"] += 4 if (t or d) else 2
            elif op.startswith("s_"):
                c["salu"] += 1
            elif op.startswith("scratch_"):
                c["scratch"] += 1
            elif op.startswith(("global_", "buffer_", "flat_")):
                c["vmem"] += 1
            elif op.startswith("ds_"):
                c["lds"] += 1
        b.update(c)
        # where control can go next: branch targets
        b["succ"] = [t for x in b["ins"] for t in re.findall(r"\.L(BB\d+_\d+)", x) if x.startswith(("s_cbranch", "s_branch"))]
    return blocks


def count(ins, c=None):
    c = c or dict(valu=0, trans=0, div=0, dp=0, salu=0, vmem=0, lds=0, scratch=0, cyc=0)
    for x in ins:
        op = x.split()[0]
        if op.startswith("v_"):
            t, d = bool(TRANS.match(op)), bool(DP.match(op))
            c["valu"] += 1; c["trans"] += t; c["dp"] += d; c["div"] += op.startswith("v_div_"); c["cyc"] += 4 if (t or d) else 2
        elif op.startswith("s_"):
            c["salu"] += 1
        elif op.startswith("scratch_"):
            c["scratch"] += 1
        elif op.startswith(("global_", "buffer_", "flat_")):
            c["vmem"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
    return c


def marks_of(body):
    """{mark: counts} and the order of first appearance."""
    sec, order, cur = {}, [], None
    for l in body:
        m = re.search(r"; YHMARK (\S+)", l)
        if m:
            cur = m.group(1)
            if cur not in sec:
                sec[cur] = count([]); order.append(cur)
        elif cur is not None and re.match(r"\s+[a-z]", l):
            count([l.strip()], sec[cur])
    return sec, order


if __name__ == "__main__":
    text = open(sys.argv[1]).read()
    name, body = kernel_body(text, sys.argv[2])
    if "--marks" in sys.argv:
        sec, order = marks_of(body)
        print("#", name)
        for k in order:
            c = sec[k]
            print(f"{k:16s} valu {c['valu']:4d} (trans {c['trans']:2d} div {c['div']:2d} dp {c['dp']:2d}) salu {c['salu']:3d} vmem {c['vmem']:2d} lds {c['lds']:2d} scr {c['scratch']:2d} cyc {c['cyc']:4d}")
        sys.exit(0)
    blocks = blocks_of(body)
    loop = sys.argv[sys.argv.index("--loop") + 1] if "--loop" in sys.argv else None
    dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
    print("#", name)
    tot = dict(valu=0, salu=0, vmem=0, lds=0, cyc=0)
    for b in blocks:
        if loop and loop not in b["loops"]:
            continue
        if dump:
            if b["name"] == dump:
                print("\n".join(b["ins"]))
            continue
        for k in tot:
            tot[k] += b[k]
        print(f"{b['name']:10s} valu {b['valu']:4d} (trans {b['trans']:2d} div {b['div']:2d} dp {b['dp']:2d}) salu {b['salu']:3d} vmem {b['vmem']:2d} lds {b['lds']:2d} scr {b['scratch']:2d} "
              f"cyc {b['cyc']:4d}  -> {','.join(b['succ']):24s} loops {','.join(b['loops'])}")
    if not dump:
        print("# total", tot)
