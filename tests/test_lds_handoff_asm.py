"""The unfenced lane-to-lane LDS hand-offs of k_raster_c, checked in the code the compiler actually emitted.

SWR_WAVE_LDS_SYNC() expands to nothing in the product build (swr_raster_c.hip.h: every fence form costs 25 %), so correctness
rests on (a) one wave's LDS instructions executing in issue order and (b) the compiler keeping program order between a store
and a later load of the same LDS array.  (b) is checked here: the product source is compiled to gfx950 assembly and, in every
k_raster_c instantiation,
  * election: the 16-byte stores that clear the `touched` bitmap precede the returning ds_or that claims a pixel in it;
  * staging -> stream: the staging lanes' stores of the per-pair rows (`stage`) precede, in layout order and outside the chunk
    loop, the chunk loop's loads of those rows;
and k_cover's one hand-off (widest box of the wave: ds_max -> ds_read -> v_readfirstlane) carries a real wavefront fence (the
compiler's `; wave barrier` between the atomic and the read).  hipcc cross-compiles without a GPU, so this runs in the CPU suite."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "softwarerenderer_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc on this machine")
    out = tmp_path_factory.mktemp("asm")
    # the product build's own flags, from its Makefile (`make print-flags`): an optimisation level, -fno-slp-vectorize or a new -m flag
    # changed there changes the code checked here; only the link step's flags are dropped
    mk = subprocess.run(["make", "-s", "-C", CSRC, "print-flags"], check=True, capture_output=True, text=True).stdout.split()
    flags = [f for f in mk if f != "-shared" and not f.startswith("-Wl,")] + ["--cuda-device-only", "-S"]
    assert "-O3" in flags and "-ffp-contract=off" in flags and "--offload-arch=gfx950" in flags, flags
    asm = os.path.join(out, "swr_api.s")
    subprocess.run([HIPCC] + flags + ["swr_api.hip", "-o", asm], check=True, capture_output=True, cwd=CSRC)
    text = open(asm).read()
    found = {}
    for m in re.finditer(r"^(_ZN3swr(?:10k_raster_c|7k_cover)\w+):", text, flags=re.M):
        name = m.group(1)
        end = text.index(".amdhsa_kernel " + name, m.end())               # the kernel descriptor follows the function body
        lds = int(re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", text[end:end + 400]).group(1))
        found[name] = (text[m.end():end].split("\n"), lds)
    return found


def _offset(line):
    m = re.search(r"offset:(\d+)", line)
    return int(m.group(1)) if m else 0


def test_there_is_something_to_check(kernels):
    assert sum("k_raster_c" in k for k in kernels) >= 7 and sum("k_cover" in k for k in kernels) == 2


def test_election_clear_precedes_the_claiming_atomic(kernels):
    for name, (lines, lds) in kernels.items():
        if "k_raster_c" not in name:
            continue
        atom = [i for i, ln in enumerate(lines) if "ds_or_rtn_b32" in ln]
        assert len(atom) == 1, (name, atom)                       # one election per chunk iteration
        base = _offset(lines[atom[0]])
        # walk back to the chunk loop's header: the clearing store of the same array must be met on the way
        clear = None
        for i in range(atom[0] - 1, -1, -1):
            if "ds_write_b128" in lines[i] and _offset(lines[i]) == base:
                clear = i
                break
            assert not ("Loop Header: Depth=2" in lines[i]), f"{name}: reached the chunk loop's header without meeting the bitmap's clear"
        assert clear is not None, name
        between = lines[clear + 1:atom[0]]
        assert not any("ds_read" in ln and _offset(ln) == base for ln in between), name


def test_staging_stores_precede_the_stream_reads(kernels):
    STAGE = 4096 + 1024                                           # WaveLdsC: col[256] float4, z[256], then stage[][]
    for name, (lines, lds) in kernels.items():
        if "k_raster_c" not in name:
            continue
        depth2 = [i for i, ln in enumerate(lines) if "Loop Header: Depth=2" in ln]
        assert depth2, name
        loop = depth2[0]                                          # the chunk loop of the batch loop
        writes = [i for i, ln in enumerate(lines) if "ds_write_b128" in ln and _offset(ln) == STAGE]
        reads = [i for i, ln in enumerate(lines) if "ds_read_b128" in ln and _offset(ln) == STAGE]
        assert writes and reads, name
        assert max(writes) < loop, f"{name}: a staging store of row 0 sits inside the chunk loop"
        assert min(reads) > max(writes), f"{name}: a stream read of row 0 is laid out ahead of the staging store"


def test_k_cover_hand_off_is_fenced(kernels):
    for name, (lines, lds) in kernels.items():
        if "k_cover" not in name:
            continue
        mx = [i for i, ln in enumerate(lines) if "ds_max_u32" in ln or "ds_max_rtn_u32" in ln]
        assert mx, name
        i = mx[0]
        tail = lines[i + 1:i + 40]
        rd = next(k for k, ln in enumerate(tail) if "ds_read_b32" in ln)
        # a wavefront-scope fence needs no s_waitcnt (one wave's LDS instructions execute in order): what it pins is the ORDER, and
        # the compiler leaves its `; wave barrier` marker where the fence stood
        assert any("wave barrier" in ln for ln in tail[:rd]), f"{name}: no wave barrier between the atomic max and the read"
        assert any("v_readfirstlane_b32" in ln for ln in tail[rd:rd + 12]), name


def test_chain_replay_keeps_its_predicate_in_exec(kernels):
    """replay_chain (swr_raster_c.hip.h): one v_cmpx per step narrows EXEC, the three adds run under it, EXEC is saved before the first
    step and restored after the last inside the same asm statement.  As predicated C++ the compiler if-converted every step into three
    adds, a compare and three v_cndmask (profiles/r04_valu_issue_model.md); this keeps the shape from coming back unnoticed."""
    for name, (lines, lds) in kernels.items():
        if "k_raster_c" not in name:
            continue
        blocks, cur = [], None
        for ln in lines:
            if "#ASMSTART" in ln:
                cur = []
            elif "#ASMEND" in ln:
                if cur is not None and any("v_cmpx_lt_i32" in x for x in cur):
                    blocks.append(cur)
                cur = None
            elif cur is not None:
                cur.append(ln)
        # a row replay (3 steps with staged varyings, 7 with the varyings in HBM) and a column replay (15 steps) per instantiation
        steps = sorted(sum("v_cmpx_lt_i32" in x for x in b) for b in blocks)
        assert steps in ([3, 15], [7, 15]), (name, steps)
        for b in blocks:
            body = [x.strip() for x in b if x.strip()]
            assert body[0].startswith("s_mov_b64") and body[0].endswith("exec"), (name, body[0])          # save
            assert body[-1].startswith("s_mov_b64 exec,"), (name, body[-1])                              # restore
            assert sum(x.startswith("v_add_f32") for x in body) == 3 * sum("v_cmpx_lt_i32" in x for x in body), name
            assert not any("v_cndmask" in x for x in body), name
