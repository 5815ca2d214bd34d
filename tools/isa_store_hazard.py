#!/usr/bin/env python3
"""isa_store_hazard.py — static guard for the gfx950 wide-store data hazard.

Found in round 2: `buffer_store_dwordx4 v[2:5], ...` directly followed by `ds_read_b128 v[2:5], ...` stored
wrong values in lanes 12-15 of every row of 16 lanes under memory back-pressure: a 12/16-byte vector-memory store
reads its data registers over several passes, and nothing in the hardware holds back an LDS (or vector-memory)
RETURN into those registers.  hipcc's hazard recogniser pads only VALU writers (1-2 wait states).  The kernels
therefore keep the stored registers live to the end of their 4-row group; this tool checks the RESULT of that —
the instruction stream the compiler actually emitted — for every kernel in the library:

  for each >8-byte vector-memory store, the smallest number of instructions on ANY control-flow path to the next
  instruction that writes one of its data VGPRs, by writer class (valu / lds / vmem return).

Usage:  python tools/isa_store_hazard.py [libchanvese_hip.so] [--min-async N] [--json]
Exit status 1 if an LDS- or vector-memory-return writer sits closer than --min-async (default 16) instructions
behind a wide store, or a VALU writer closer than 1 instruction without the s_nop the recogniser owes.
The CPU test tests/test_isa_hazard.py runs it on the built library.
"""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
WIDE_STORE = re.compile(r"^(buffer_store_dwordx[34]|global_store_dwordx[34]|flat_store_dwordx[34]|"
                        r"buffer_store_format_xyzw?|scratch_store_dwordx[34])\b")
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
SEARCH_DEPTH = 400     # instructions followed on every path behind a store


def vregs(operand):
    out = set()
    for m in REG.finditer(operand):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def disassemble(lib):
    """{kernel symbol: [(addr, mnemonic, [operands])]} for every gfx950 code object bundled in `lib`."""
    tmp = tempfile.mkdtemp(prefix="isa_hazard_")
    try:
        local = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, local)
        subprocess.run([OBJDUMP, "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        funcs = {}
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", "--mcpu=gfx950", os.path.join(tmp, f)],
                                 check=True, capture_output=True, text=True).stdout
            cur = None
            for line in txt.splitlines():
                m = re.match(r"^([0-9a-f]+) <(.+)>:$", line)
                if m:
                    cur = funcs.setdefault(m.group(2), [])
                    continue
                m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
                if m and cur is not None:
                    ops = [o.strip() for o in m.group(2).split(",")] if m.group(2) else []
                    cur.append((int(m.group(3), 16), m.group(1), ops))
        return funcs
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def written_vgprs(mn, ops):
    """(set of VGPRs this instruction writes, writer class)."""
    if not ops:
        return set(), None
    if mn.startswith(("v_cmp", "v_readlane", "v_readfirstlane", "v_nop")) and not mn.startswith("v_cmpx"):
        return set(), None
    if mn.startswith(("v_swap", "v_permlane16_swap", "v_permlane32_swap")):
        return vregs(ops[0]) | vregs(ops[1] if len(ops) > 1 else ""), "valu"
    if mn.startswith("v_"):
        return vregs(ops[0]), "valu"
    if mn.startswith("ds_") and ("read" in mn or "rtn" in mn or "load" in mn or "bpermute" in mn or "permute" in mn or "swizzle" in mn):
        return vregs(ops[0]), "lds"
    if mn.startswith(("buffer_load", "global_load", "flat_load", "scratch_load", "buffer_atomic", "global_atomic", "flat_atomic",
                      "image_")):
        if "lds" in ops[-1].split():
            return set(), None
        if "atomic" in mn and not any("sc0" in o or "glc" in o for o in ops):
            return set(), None          # no return value
        return vregs(ops[0]), "vmem"
    return set(), None


def analyse(insts):
    """[(index, mnemonic, data regs, {class: min distance}, detail)] for the wide stores of one function."""
    by_addr = {a: i for i, (a, _, _) in enumerate(insts)}

    def successors(i):
        a, mn, ops = insts[i]
        nxt = [i + 1] if i + 1 < len(insts) else []
        if mn in ("s_endpgm", "s_setpc_b64", "s_swappc_b64"):
            return []
        if mn.startswith(("s_branch", "s_cbranch")):
            # SOPP branch: target = address of the next instruction + 4 * simm16 (printed unsigned by llvm-objdump)
            off = int(ops[-1], 0) & 0xffff
            off -= 0x10000 if off >= 0x8000 else 0
            tgt = a + 4 + 4 * off
            t = [by_addr[tgt]] if tgt in by_addr else []
            return t if mn.startswith("s_branch") else nxt + t
        return nxt

    out = []
    for i, (a, mn, ops) in enumerate(insts):
        if not WIDE_STORE.match(mn):
            continue
        data = vregs(ops[0])
        best, detail = {}, {}
        seen = {i}
        frontier = [(j, 1) for j in successors(i)]
        while frontier:
            nf = []
            for j, d in frontier:
                if j in seen or d > SEARCH_DEPTH:
                    continue
                seen.add(j)
                w, cls = written_vgprs(insts[j][1], insts[j][2])
                hit = w & data
                if hit:
                    if cls not in best or d < best[cls]:
                        best[cls] = d
                        detail[cls] = "%x: %s %s" % (insts[j][0], insts[j][1], ", ".join(insts[j][2]))
                    if hit == data:
                        continue       # every data register is dead beyond this writer on this path
                nf.extend((k, d + 1) for k in successors(j))
            frontier = nf
        out.append((i, mn, sorted(data), best, detail))
    return out


def nop_states_between(insts, i, d):
    """wait states supplied by s_nop between store i and the instruction d behind it (linear only)."""
    n = 0
    for j in range(i + 1, min(i + d, len(insts))):
        if insts[j][1] == "s_nop":
            n += int(insts[j][2][0], 0) + 1
    return n


def main():
    ap = argparse.ArgumentParser()
    here = os.path.dirname(os.path.abspath(__file__))
    ap.add_argument("lib", nargs="?", default=os.path.join(here, "..", "chan_vese_amd", "csrc", "libchanvese_hip.so"))
    ap.add_argument("--min-async", type=int, default=16,
                    help="smallest allowed instruction distance from a wide store to an LDS / vector-memory RETURN into its data registers")
    ap.add_argument("--json", action="store_true")
    args = ap.parse_args()
    funcs = disassemble(os.path.abspath(args.lib))
    report, bad = [], []
    for name, insts in sorted(funcs.items()):
        stores = analyse(insts)
        if not stores:
            continue
        mins = {}
        for i, mn, data, best, detail in stores:
            for cls, d in best.items():
                if cls not in mins or d < mins[cls][0]:
                    mins[cls] = (d, "%x: %s v%s -> %s" % (insts[i][0], mn, data, detail[cls]))
                if cls in ("lds", "vmem") and d < args.min_async:
                    bad.append((name, cls, d, "%x: %s -> %s" % (insts[i][0], mn, detail[cls])))
                if cls == "valu" and d == 1:
                    pass     # hipcc's recogniser inserts the s_nop itself when it is owed (soffset in an SGPR); nothing to flag statically
        report.append({"kernel": name, "wide_stores": len(stores),
                       "min_distance": {c: v[0] for c, v in mins.items()}, "where": {c: v[1] for c, v in mins.items()}})
    if args.json:
        print(json.dumps({"min_async": args.min_async, "kernels": report, "violations": bad}, indent=1))
    else:
        for r in report:
            print("%-110s stores %3d  min distance to a writer of the store data: %s" % (r["kernel"][:110], r["wide_stores"], r["min_distance"]))
        for b in bad:
            print("VIOLATION %s: %s return %d instructions behind the store (%s)" % b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
