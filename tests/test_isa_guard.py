"""What the headline speed rests on, read from the ISA.

The whole-set kernels live in a translation unit compiled without the machine scheduler (nb_nbody_sl.inc "Issue order"), and
STRICT's 13 % over the LDS-tiled kernel comes from the compiler turning the wave-uniform record loads into SCALAR loads -- which
it silently stops doing when anything in the kernel might clobber memory (a store ahead of the loop, an `asm volatile`, a real
function call).  Nothing else in the suite would notice: the results stay bit-identical, only the time changes.  So: build the
unit's assembly (`make asm-sl`, hipcc cross-compiles without a GPU) and hold the kernels to the properties DESIGN.md section 4
quotes.  The -m gpu suite holds the times themselves (test_gpu_parity.py::test_perf_floor_of_the_whole_set_kernels)."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

ASM = os.path.join(ROOT, "build", "asm", "nb_kernels_sl.s")

STRICT_SL = "_ZN3nbk21step_strict_sl_kernelILi4ELi16EEEvNS_8StepArgsEPKjjPKfS5_S5_"
FAST_SL = "_ZN3nbk19step_fast_sl_kernelILi4ELi8EEEvNS_8StepArgsEPKjjPKfS5_S5_"
FAST_PAIRS = "_ZN3nbk22step_fast_pairs_kernelILi4ELi4EEEvNS_8StepArgsEPKjjPKfS5_S5_NS_6PrTileE"
FAST_RING = "_ZN3nbk21step_fast_ring_kernelILi4EEEvNS_8RingArgsE"


@pytest.fixture(scope="module")
def asm():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "nenbody_amd", "csrc"), "asm-sl"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return open(ASM).read()


def kernel(asm, name):
    """(instructions of the kernel, its resource comment block)"""
    i = asm.index("\n" + name + ":")
    j = asm.index(".Lfunc_end", i)
    k = asm.index("; Occupancy", j)
    meta = dict((a, int(b)) for a, b in re.findall(r"; (\w+): (\d+)", asm[j:k + 40]))
    return asm[i:j], meta


def count(body, pattern):
    return len(re.findall(pattern, body))


def loops(body):
    """the bodies of the kernel's loops: a label .. the first backward branch to it"""
    out = []
    for m in re.finditer(r"^(\.LBB\d+_\d+):", body, re.M):
        label, start = m.group(1), m.end()
        b = re.search(r"s_cbranch_\w+ " + re.escape(label) + r"\b", body[start:])
        if b:
            out.append(body[start:start + b.start()])
    return out


def test_strict_headline_kernel_reads_its_records_through_scalar_loads(asm):
    body, meta = kernel(asm, STRICT_SL)
    # the planar main loop: two s_load_dwordx16 (x and y planes) per sixteen records, nothing else touches memory
    main = [lp for lp in loops(body) if count(lp, r"s_load_dwordx16") >= 2]
    assert main, "no loop of step_strict_sl_kernel<4,16> holds two s_load_dwordx16: the records are no longer scalar loads"
    lp = min(main, key=len)      # the innermost one: the planar fold itself (the others enclose it)
    assert count(lp, r"\b(global|flat|buffer)_load") == 0 and count(lp, r"\tds_") == 0 and count(lp, r"s_barrier") == 0
    assert count(body, r"\tds_") == 0 and count(body, r"scratch_") == 0 and count(body, r"buffer_load") == 0
    assert count(body, r"global_load") <= 2            # the body's own position and velocity record, once
    assert meta["NumVgprs"] <= 128 and meta["ScratchSize"] == 0 and meta["Occupancy"] >= 2
    assert meta.get("LDSByteSize", 0) == 0
    # the arithmetic of the loop is packed, with the reciprocals and the ordered additions between the rows
    assert count(lp, r"v_pk_(fma|mul|add)_f32") >= 100 and count(lp, r"v_rcp_f32") >= 16 and count(lp, r"v_add_f32") >= 32


def test_fast_pairs_kernel_keeps_its_registers_and_its_rotating_sums(asm):
    body, meta = kernel(asm, FAST_PAIRS)
    assert meta["ScratchSize"] == 0 and meta["NumVgprs"] <= 168 and meta["Occupancy"] >= 3      # three waves per SIMD
    assert count(body, r"v_sub_f32_dpp") >= 128 and count(body, r"scratch_") == 0
    assert count(body, r"wave_rol:1") >= 128         # the b-side sums turn by one lane per step, inside the subtraction


def test_fast_ring_kernel_is_the_same_sweep(asm):
    body, meta = kernel(asm, FAST_RING)
    assert meta["ScratchSize"] == 0 and meta["NumVgprs"] <= 168 and meta["Occupancy"] >= 3
    assert count(body, r"v_sub_f32_dpp") >= 128 and count(body, r"scratch_") == 0
    # no barrier inside the sweep loops: the four waves of a workgroup are independent until their a-side sums meet at the end
    assert all(count(lp, r"s_barrier") == 0 for lp in loops(body))
    assert count(body, r"s_barrier") >= 1


def test_fast_ordered_fold_reads_scalar_loads_too(asm):
    body, meta = kernel(asm, FAST_SL)
    assert count(body, r"s_load_dwordx16") >= 2 and meta["ScratchSize"] == 0 and meta["NumVgprs"] <= 128
    main = [lp for lp in loops(body) if count(lp, r"s_load_dwordx16") >= 2]
    assert main and count(min(main, key=len), r"\b(global|flat|buffer)_load") == 0


def test_the_guard_would_notice(asm):
    """the control arm: the LDS-free, scalar-load property is a property of THIS kernel, not of the parser -- the pairs form, which
    stages its b side through LDS and reads its records with vector loads, fails the same checks"""
    body, _ = kernel(asm, FAST_PAIRS)
    assert count(body, r"\tds_") > 0 and count(body, r"global_load") > 2
    assert not [lp for lp in loops(body) if count(lp, r"s_load_dwordx16") >= 2]
