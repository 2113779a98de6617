"""CPU: facts about the generated gfx950 code of the hot kernels that the measured speed depends on ("where hipcc
put the waits").  Read from the ISA listing that emsar_amd/_build.py leaves in build/ (-save-temps); skipped where that listing is
absent (the GPU box receives the built library without build/)."""
import os
import re

import pytest

from emsar_amd import _build

ASM = os.path.join(_build.BUILD, "emsar_hip-hip-amdgcn-amd-amdhsa-gfx950.s")


@pytest.fixture(scope="module")
def listing():
    _build.build_hip()
    if not os.path.exists(ASM):
        pytest.skip("no ISA listing (build/ is not shipped)")
    return open(ASM).read()


def _meta(txt):
    out = {}
    for blk in txt.split("  - .agpr_count:")[1:]:
        g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, None])[1]
        out[g("name")] = {k: int(g(k)) for k in ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size")}
    return out


def _body(txt, mangled):
    i = txt.index("\n" + mangled + ":")
    return txt[i:txt.index("\n.Lfunc_end", i)]


UNIT = "_ZN12_GLOBAL__N_117k_pass_tiled_unitILb%dELi%dELb0EEE"        # <WEIGHTED, MODE, STAMP = false>


def test_unit_kernels_keep_four_workgroups_per_cu_and_no_scratch(listing):
    """128 VGPRs = four waves per SIMD; a spilt register is a scratch access in the tile loop (and the s_waitcnt vmcnt(0) in front of its
    reload waits for every index load in flight)."""
    meta = _meta(listing)
    seen = 0
    for w in (0, 1):
        for mode in (0, 1):
            names = [n for n in meta if n and n.startswith(UNIT % (w, mode))]
            assert len(names) == 1, names
            m = meta[names[0]]
            seen += 1
            assert m["vgpr_count"] <= 128, (names[0], m)
            assert m["vgpr_spill_count"] == 0 and m["sgpr_spill_count"] == 0 and m["private_segment_fixed_size"] == 0, (names[0], m)
    assert seen == 4


def test_tile_descriptors_are_read_by_scalar_loads(listing):
    """emsar::Tile's 8- and 16-bit fields must not be fetched with vector loads (global_load_ushort / _ubyte + vmcnt(0) + readfirstlane in the
    middle of the tile loop): kernels_tiled.hpp reads a descriptor as sixteen dwords (tile_load)."""
    meta = _meta(listing)
    for n in meta:
        if n and "k_pass_tiled" in n:
            body = _body(listing, n)
            assert "global_load_ushort" not in body and "global_load_ubyte" not in body and "global_load_sbyte" not in body, n


def test_the_steps_of_the_unit_kernel_wait_for_their_own_loads_only(listing):
    """The first batch of the E-step is consumed while the eight backward loads requested just before are in flight, the first batch of
    the M-step while the next tile's eight forward loads are: their waits count down from vmcnt(15), not from vmcnt(7) or vmcnt(0)."""
    names = [n for n in _meta(listing) if n and n.startswith(UNIT % (0, 0))]
    body = _body(listing, names[0])
    loop = body[body.index("Loop Header: Depth=1"):]
    waits = [int(x) for x in re.findall(r"s_waitcnt vmcnt\((\d+)\)", loop)]
    # (the columns of a batch are not consumed in the order they were requested, so the counts do not fall one by one)
    assert waits[0] == 15, ("something waits for more than its own column before the E-step's first batch", waits[:12])
    assert waits.count(15) >= 2, waits[:60]                       # E-step and M-step
    first_batch = waits[:waits.index(15) + 8]
    assert min(first_batch) >= 8, first_batch                     # none of the eight loads that run ahead is waited for in the first batch
