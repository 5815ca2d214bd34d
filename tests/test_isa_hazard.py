"""Build check on the shipped binary: the gfx950 wide-store data hazard (tools/isa_store_hazard.py).

A >64-bit vector-memory store followed IMMEDIATELY by a VALU write of one of its data registers stores wrong values in
lanes 12-15 of every row of 16 under memory back-pressure (measured: tools/store_hazard_probe.hip); hipcc pads that
only for stores without a register soffset, and the wave kernels use one.  The kernels avoid it by construction (the
stored registers stay live to the end of the 4-row group); this test disassembles every code object of
libchanvese_hip.so and fails if any instantiation has a VALU writer in the issue slot right behind a wide store."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "chan_vese_amd", "csrc", "libchanvese_hip.so")
TOOL = os.path.join(ROOT, "tools", "isa_store_hazard.py")


@pytest.fixture(scope="module")
def report():
    if not os.path.exists(LIB):
        pytest.fail(f"{LIB} is missing: run __graft_entry__.build() first")
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("llvm-objdump of the ROCm toolchain not found")
    out = subprocess.run([sys.executable, TOOL, LIB, "--json"], capture_output=True, text=True, timeout=300)
    assert out.returncode in (0, 1), out.stderr[-2000:]
    return json.loads(out.stdout)


def test_no_valu_write_in_the_issue_slot_behind_a_wide_store(report):
    assert report["violations"] == [], report["violations"]


def test_every_streaming_kernel_with_wide_stores_was_seen(report):
    """The check is only worth something if it saw the kernels it is about: the 2-pixel CSV kernels (16-byte level-set
    stores: >= 8 per instantiation, one per row of the loop bodies)."""
    names = {k["kernel"]: k for k in report["kernels"]}
    # (the FP32-state instantiations, last template argument true = "Lb1E", store 8 bytes per lane: outside the hazard)
    wave2 = [k for n, k in names.items() if "csv_wave2_kernel" in n and "ELb0EEEv" in n]
    assert len(wave2) >= 6, sorted(names)            # 1-channel: strict, three cache policies; 3-channel: two cache policies
    assert len([n for n in names if "csv_wave2_kernel" in n and "ELb1EEEv" in n]) >= 0
    for k in wave2:
        assert k["wide_stores"] >= 8, k
        assert k["register_soffset_stores"] >= 8, k                  # the form hipcc does not pad
        assert k["min_wait_states_register_soffset"].get("valu", 1 << 30) >= 6, k     # by construction: live to the end of the group
    # (the 2-pixel Perona-Malik kernel, the other kernel with 16-byte stores, was pruned in round 4: tools/experiments/pruned_flavours/)
