#!/usr/bin/env python3
"""Developer tool: the per-branch budget of k_stream's trace step — vector instructions issued x lanes active —
from (a) the static instruction counts between the YH_ISA_MARKS of a marked build (tools/isa_blocks.py --marks) and
(b) the dynamic per-branch counters of a YHAIR_ST_PROF=1 run (stderr of tools/shape_check.py SCENE RES SPP 3).

usage: tools/branch_budget.py marked.s prof_scene.txt [prof_scene2.txt ...]
       (marked.s: hipcc -S --cuda-device-only -DYH_ISA_MARKS csrc/stream.hip with the Makefile's flags)"""
import re, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_blocks

# which marked sections make up a branch of the step (a section = from its mark to the next one in layout order)
BRANCH_SECTIONS = {
    "step":      ["step_begin", "step_head", "after_enter", "step_end", "trace_lists"],  # every busy lane, every step
    "pop":       ["pop"],
    "scene":     ["scene"],
    "enter":     ["enter"],
    "fetch":     ["fetch"],
    "node":      ["node", "node_order"],
    "line-leaf": ["leaf", "line_leaf"],
    "tri-leaf":  ["tri_leaf"],
}


def static_counts(path):
    text = open(path).read()
    _, body = isa_blocks.kernel_body(text, "k_streamILb0ELi4ELb0E")
    sec, _ = isa_blocks.marks_of(body)
    out = {}
    for b, names in BRANCH_SECTIONS.items():
        out[b] = dict(valu=sum(sec[n]["valu"] for n in names if n in sec), cyc=sum(sec[n]["cyc"] for n in names if n in sec),
                      trans=sum(sec[n]["trans"] for n in names if n in sec), div=sum(sec[n]["div"] for n in names if n in sec))
    out["retire"] = dict(valu=sec.get("trace_retire", {}).get("valu", 0), cyc=sec.get("trace_retire", {}).get("cyc", 0), trans=0, div=0)
    out["refill"] = dict(valu=sec.get("trace_refill", {}).get("valu", 0), cyc=sec.get("trace_refill", {}).get("cyc", 0), trans=0, div=0)
    return out


def dynamic_counts(path):
    """Last k_stream profile block of the file: {branch: (share of wave steps, lanes when it ran)} + header numbers."""
    txt = open(path).read()
    blocks = txt.split("[yhair] k_stream ")[1:]
    if not blocks:
        raise SystemExit(path + ": no YHAIR_ST_PROF output")
    b = blocks[-1]
    dyn = {}
    for m in re.finditer(r"branch (\S+)\s+ran in\s+([\d.]+) % of the wave steps \((\d+) times\), ([\d.]+) lanes", b):
        dyn[m.group(1)] = (float(m.group(2)) / 100.0, float(m.group(4)))
    m = re.search(r"trace: (\d+) wave steps, ([\d.]+) lanes busy on average, (\d+) cycles per step", b)
    stages = {mm.group(1): float(mm.group(2)) for mm in re.finditer(r"^\[yhair\]\s+(\w+)\s+([\d.]+) % of wave time", b, re.M)}
    return dyn, int(m.group(1)), float(m.group(2)), stages


if __name__ == "__main__":
    st = static_counts(sys.argv[1])
    print(f"# static: vector instructions between the marks of {os.path.basename(sys.argv[1])} (k_stream<false, 4, false>; cyc = issue cycles on a shared SIMD: 2 per instruction, 4 per transcendental / 64-bit)")
    for b, c in st.items():
        print(f"#   {b:10s} {c['valu']:4d} VALU ({c['trans']} transcendental, {c['div']} v_div_*)  {c['cyc']:4d} cyc")
    for prof in sys.argv[2:]:
        dyn, steps, busy, stages = dynamic_counts(prof)
        print(f"\n## {os.path.basename(prof)}: {steps} wave steps, {busy:.1f} lanes busy on average; stages (% of wave time): " +
              ", ".join(f"{k} {v:.1f}" for k, v in stages.items()))
        print(f"{'branch':10s} {'ran in':>8s} {'lanes':>6s} {'VALU':>5s} {'issued/step':>12s} {'useful lane-instr/step':>23s} {'lane use':>9s}")
        tot_issued = tot_useful = 0.0
        for b in ("step", "pop", "scene", "enter", "fetch", "node", "line-leaf", "tri-leaf"):
            share, lanes = dyn.get(b, (0.0, 0.0))
            issued = share * st[b]["valu"]
            useful = issued * lanes
            tot_issued += issued
            tot_useful += useful
            print(f"{b:10s} {100 * share:7.1f}% {lanes:6.1f} {st[b]['valu']:5d} {issued:12.1f} {useful:23.0f} {lanes / 64:9.2f}")
        print(f"{'total':10s} {'':8s} {'':6s} {'':5s} {tot_issued:12.1f} {tot_useful:23.0f} {tot_useful / max(1e-9, tot_issued) / 64:9.2f}"
              f"   (+ retire {st['retire']['valu']} and refill {st['refill']['valu']} VALU when a lane finishes / the wave refills)")
        if "push" in dyn:
            print(f"pushes: {dyn['push'][0]:.2f} executions of a push block per wave step, {dyn['push'][1]:.1f} lanes each; "
                  f"second segment of a line-leaf step real for {dyn.get('2nd-seg', (0, 0))[1]:.1f} of its {dyn.get('line-leaf', (0, 0))[1]:.1f} lanes")
