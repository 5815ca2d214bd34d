#!/usr/bin/env python3
"""isa_store_hazard.py — static guard for the gfx950 wide-store data hazard.

The hazard (settled in round 3 with tools/store_hazard_probe.hip on MI355X, and from the ISA of the round-2 build that
failed): a vector-memory store of MORE than 64 bits reads its data registers over two passes; a VALU instruction that
writes one of those registers in the very next issue slot (no wait state in between) changes what lanes 12-15 of
every row of 16 lanes store -- only under memory back-pressure (0.3 % of the elements at 8 waves/SIMD streaming
stores, none at 64 waves), the dword that is overwritten first.  ONE wait state (any instruction, or `s_nop 0`)
between store and writer removes it.  This is the documented ">64-bit store data followed by a VALU write of the
data VGPRs" hazard -- but hipcc's recogniser (GCNHazardRecognizer::createsVALUHazard) pads it only when the store's
soffset is NOT a register, and the wave kernels address rows through an SGPR soffset: the round-2 build had
`buffer_store_dwordx4 v[2:5], v148, s[68:71], s14 offen` directly followed by `v_fma_f64 v[2:3], ...`.
LDS returns (ds_read_b128) and vector-memory returns (buffer_load_dwordx4) into the store's data registers right
behind it are NOT a hazard (0 bad of 6e8 elements in every load / guard combination of the probe), and 8-byte
stores are immune.

The kernels keep the stored registers live to the end of their 4-row group, so no writer of any class comes near;
this tool checks the RESULT — the instruction stream hipcc actually emitted — for every kernel in the library:

  for each >8-byte vector-memory store, the smallest number of instructions on ANY control-flow path to the next
  instruction that writes one of its data VGPRs, by writer class (valu / lds / vmem return).

The probe also shows the IMMEDIATE-soffset form to be worse (25 % of the elements even on an idle chip with no wait
state, and 0.3 % left under load with ONE): that form, and global_/flat_ stores, need two -- hipcc pads them itself.

Usage:  python tools/isa_store_hazard.py [libchanvese_hip.so] [--min-states N] [--json]
Exit status 1 if, on any control-flow path, a VALU write of a wide store's data registers follows it with fewer wait
states in between (each instruction is one, `s_nop N` is N + 1) than required: --min-states (default 1) for buffer
stores with a register soffset, one more for every other wide store.  LDS / vector-memory returns are reported only.
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

    def states(j):
        """wait states instruction j puts between its predecessor and its successor"""
        mn, ops = insts[j][1], insts[j][2]
        return int(ops[0], 0) + 1 if mn == "s_nop" and ops else 1

    out = []
    for i, (a, mn, ops) in enumerate(insts):
        if not WIDE_STORE.match(mn):
            continue
        data = vregs(ops[0] if mn.startswith("buffer_store") else ops[1])   # global_/flat_/scratch_: (vaddr, vdata, saddr)
        # soffset of a MUBUF store is the operand behind the 4-dword resource; a register there (sN, m0, ttmpN) is the form
        # hipcc's recogniser does NOT pad
        reg_soffset = False
        if mn.startswith("buffer_store"):
            so = [o.split()[0] for o in ops if re.match(r"^(s\d+|m0|ttmp\d+)\b", o)]
            reg_soffset = len(so) > 0
        best, detail = {}, {}
        # (instructions behind the store, wait states between store and instruction): breadth-first by instruction count; the
        # wait states are tracked per path and the minimum is kept
        seen = {}
        frontier = [(j, 1, 0) for j in successors(i)]
        while frontier:
            nf = []
            for j, d, ws in frontier:
                if d > SEARCH_DEPTH or (j in seen and seen[j] <= ws):
                    continue
                seen[j] = ws
                w, cls = written_vgprs(insts[j][1], insts[j][2])
                hit = w & data
                if hit:
                    if cls not in best or (ws, d) < best[cls]:
                        best[cls] = (ws, d)
                        detail[cls] = "%x: %s %s" % (insts[j][0], insts[j][1], ", ".join(insts[j][2]))
                    if hit == data:
                        continue       # every data register is dead beyond this writer on this path
                nf.extend((k, d + 1, ws + states(j)) for k in successors(j))
            frontier = nf
        out.append((i, mn, sorted(data), best, detail, reg_soffset))
    return out


def main():
    ap = argparse.ArgumentParser()
    here = os.path.dirname(os.path.abspath(__file__))
    ap.add_argument("lib", nargs="?", default=os.path.join(here, "..", "chan_vese_amd", "csrc", "libchanvese_hip.so"))
    ap.add_argument("--min-states", type=int, default=1,
                    help="wait states required between a wide buffer store with a REGISTER soffset and a VALU write of its data "
                         "registers (probe: 1 suffices, 0 corrupts); every other wide store (immediate soffset, global_/flat_/"
                         "scratch_) needs one more (probe: 1 still corrupts the immediate form) -- hipcc pads those itself")
    ap.add_argument("--json", action="store_true")
    args = ap.parse_args()
    funcs = disassemble(os.path.abspath(args.lib))
    report, bad = [], []
    for name, insts in sorted(funcs.items()):
        stores = analyse(insts)
        if not stores:
            continue
        mins, mins_reg = {}, {}
        for i, mn, data, best, detail, reg_soffset in stores:
            need = args.min_states if reg_soffset else args.min_states + 1
            for cls, (ws, d) in best.items():
                if cls not in mins or (ws, d) < mins[cls][:2]:
                    mins[cls] = (ws, d, "%x: %s v%s -> %s" % (insts[i][0], mn, data, detail[cls]))
                if reg_soffset and (cls not in mins_reg or ws < mins_reg[cls]):
                    mins_reg[cls] = ws
                if cls == "valu" and ws < need:
                    bad.append((name, ws, need, "%x: %s %s -> %s" % (insts[i][0], mn, "(register soffset)" if reg_soffset else "", detail[cls])))
        report.append({"kernel": name, "wide_stores": len(stores),
                       "register_soffset_stores": sum(1 for st in stores if st[5]),
                       "min_wait_states": {c: v[0] for c, v in mins.items()},
                       "min_wait_states_register_soffset": mins_reg,
                       "min_distance": {c: v[1] for c, v in mins.items()}, "where": {c: v[2] for c, v in mins.items()}})
    if args.json:
        print(json.dumps({"min_states": args.min_states, "kernels": report, "violations": bad}, indent=1))
    else:
        for r in report:
            print("%-104s stores %3d (reg soffset %3d)  wait states to the next writer of the data: %s; register-soffset stores only: %s" % (
                r["kernel"][:104], r["wide_stores"], r["register_soffset_stores"], r["min_wait_states"], r["min_wait_states_register_soffset"]))
        for b in bad:
            print("VIOLATION %s: VALU write %d wait state(s) behind a wide store, %d required (%s)" % b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
