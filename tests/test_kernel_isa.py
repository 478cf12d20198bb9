"""CPU: static checks on the gfx950 ISA hipcc emits for the body / conv_last kernel (reve_amd/csrc/kernels.hip).

The kernel leaves the last epilogue stores of a tile in flight across the tile barrier with a COUNTED `s_waitcnt vmcnt(N)`:
N must equal the number of vector-memory instructions younger than the tile's last LDS-DMA piece, or the barrier could be
passed before this wave's share of the next tile has landed (a race no parity test is guaranteed to hit).  hipcc is free to
move instructions, so the emitted stream itself is checked, for every instantiation; so are the register budget (weights
parked in AGPRs, no spills, no scratch) and the MFMA count per tile."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "reve_amd", "csrc")


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not present")
    d = tmp_path_factory.mktemp("isa")
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-save-temps",
                        "-I" + CSRC, "-c", os.path.join(CSRC, "kernels.hip"), "-o", "k.o"], cwd=d, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return open(os.path.join(d, "kernels-hip-amdgcn-amd-amdhsa-gfx950.s")).read()


def kernel_bodies(text):
    """name -> assembly text of each k_body instantiation"""
    out = {}
    for m in re.finditer(r"^(_ZN4reve6k_bodyILi(\d)ELi(\d)ELb(\d)EEEvNS_8ConvArgsEPKNS_9PlaneDescEPKj):\s*;", text, re.M):
        end = text.index("s_endpgm", m.end())
        out[(int(m.group(2)), int(m.group(3)), int(m.group(4)))] = text[m.end():end]
    return out


def test_counted_vmcnt_matches_the_emitted_stream(isa):
    bodies = kernel_bodies(isa)
    # (last != 0 with the third parameter set = the conv_last parity probes of reve_debug_run_layers: not product kernels)
    assert len(bodies) == 18, sorted(bodies)
    bodies = {k: v for k, v in bodies.items() if not (k[1] and k[2])}
    assert len(bodies) == 15, sorted(bodies)
    mfma_per_tile = {0: 576, 2: 144, 3: 288, 4: 432}          # 4 rows x 2 px-blocks x co-blocks x 18 k-steps
    for (order, last, unit), asm in sorted(bodies.items()):
        lines = [l.strip() for l in asm.split("\n")]
        bars = [i for i, l in enumerate(lines) if l.startswith("s_barrier")]
        # the tile loop's barrier; the x2 conv_last (two single-buffered workgroups per CU) has a second one at the end of the
        # tile, in front of the next tile's DMA pieces, and waits for everything (vmcnt(0)) behind them
        single = re.search(r"#define KB_LAST2_SINGLE (\d)", open(os.path.join(CSRC, "kernels.hip")).read()).group(1) == "1"
        bar = bars[-2] if (last == 2 and single) else bars[-1]
        loop = []
        for l in lines[bar + 1:]:
            if l.startswith("s_cbranch_scc"):
                break
            loop.append(l)
        ops = []
        for l in loop:
            if re.match(r"buffer_load_dwordx4 .* lds", l):
                ops.append("D")
            elif l.startswith("buffer_store"):
                ops.append("S")
            elif l.startswith("buffer_load"):
                ops.append("L")
            elif l.startswith("s_waitcnt vmcnt") and "lgkmcnt" not in l:
                ops.append(("W", int(re.search(r"vmcnt\((\d+)\)", l).group(1))))
        what = f"k_body<{order}, {last}, {unit}>"
        assert ops.count("D") == 20, what                                               # this wave's share of the 77 pieces
        last_dma = max(i for i, o in enumerate(ops) if o == "D")
        tail = ops[last_dma + 1:]
        assert isinstance(tail[-1], tuple), (what, tail)                                # the loop ends with the counted wait
        younger = sum(1 for o in tail if o in ("S", "L"))
        assert tail[-1][1] == younger, f"{what}: s_waitcnt vmcnt({tail[-1][1]}) but {younger} vector-memory instructions follow the last DMA"
        assert "L" not in tail, what                                                    # residual loads sit ahead of the tile's DMA
        assert sum(l.startswith("v_mfma_f32_16x16x32_f16") for l in loop) == mfma_per_tile[last], what
        assert not any(l.startswith(("scratch_", "v_accvgpr_read", "v_accvgpr_write")) for l in loop), what


def test_register_budget(isa):
    """one wave per SIMD: up to 512 registers, 256 of them AGPRs holding the weights (216 for the 3 co-blocks of x4); nothing
    spilled, no scratch."""
    meta = isa[isa.index("amdhsa.kernels:"):]
    n = 0
    for blk in meta.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        if "k_body" not in name or re.search(r"k_bodyILi\dELi[234]ELb1", name):     # (the conv_last probes are not product kernels)
            continue
        n += 1
        agpr = int(blk.split()[0])
        vgpr = int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1))
        assert int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1)) == 0, name
        # (a few SGPRs of the x3 instantiation are parked in VGPR lanes: no memory involved, private_segment stays 0)
        assert int(re.search(r"\.sgpr_spill_count:\s+(\d+)", blk).group(1)) <= 8, name
        assert int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1)) == 0, name
        last = int(re.search(r"k_bodyILi\dELi(\d)", name).group(1))
        assert agpr == {0: 256, 2: 72, 3: 144, 4: 216}[last], (name, agpr)
        assert vgpr <= 512, (name, vgpr)
    assert n == 15


def test_product_library_carries_no_diagnostic_code(tmp_path):
    """The timing-only instrumentation (in-kernel stamps, ablations: wrong outputs by design) is kept apart from the product:
    k_pair's, k_last_strip's and k_wino's live in kernels_*_diag.inc, included only under -DREVE_DIAGNOSTIC_BUILD
    (scripts/ablate_pair.sh); k_body's was removed.  So: the shipped library holds no stamp buffer or reader; the product
    translation units compile with the .inc files ABSENT and contain no conditional compilation inside the kernels beyond the
    include itself; and a build that names one of the switches without -DREVE_DIAGNOSTIC_BUILD stops."""
    lib = os.path.join(ROOT, "reve_amd", "libreve_hip.so")
    if not os.path.exists(lib):
        pytest.skip("libreve_hip.so not built")
    blob = open(lib, "rb").read()
    for name in (b"g_stamps2", b"reve_debug_read_stamps2", b"g_stamps_pair", b"g_stamps_wino", b"reve_debug_read_stamps_pair"):
        assert name not in blob, name
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not present")
    src = tmp_path / "csrc"
    src.mkdir()
    for f in os.listdir(CSRC):
        if f.endswith((".hip", ".h")):
            shutil.copy(os.path.join(CSRC, f), src / f)
    assert sorted(f for f in os.listdir(CSRC) if f.endswith(".inc")) == ["kernels_last_diag.inc", "kernels_pair_diag.inc", "kernels_wino_diag.inc"]
    base = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "--cuda-device-only", "-fsyntax-only"]
    for f in ("kernels_pair.hip", "kernels_last.hip", "kernels_wino.hip", "kernels.hip", "kernels_first.hip"):
        r = subprocess.run(base + [str(src / f)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (f, r.stderr[-1500:])
        # inside the kernels: no #if / #ifdef — the only conditionals of a file are its tuning defaults (#ifndef X / #define X) and
        # the guard + include of the instrumentation in front of the code
        text = open(os.path.join(CSRC, f)).read()
        body = text[text.index("__global__"):]
        assert not re.search(r"^\s*#\s*(if|ifdef|ifndef|elif|else)\b", body, re.M), f
    for f, flag, word in (("kernels.hip", "-DSTAMPS", "no longer exist"), ("kernels.hip", "-DABL2_NO_STORE", "no longer exist"),
                          ("kernels_pair.hip", "-DSTAMPS", "diagnostic switches"), ("kernels_pair.hip", "-DABLP_NO_EPI", "diagnostic switches"),
                          ("kernels_last.hip", "-DKL_ABL_NO_EPI", "timing-only"), ("kernels_wino.hip", "-DKWD_NO_DMA", "diagnostic switches")):
        r = subprocess.run(base + [flag, "-I" + CSRC, os.path.join(CSRC, f)], capture_output=True, text=True, timeout=600)
        assert r.returncode != 0 and word in r.stderr, (f, flag, r.stderr[-500:])
    # ... and with it, the instrumented variants still build
    for f, flag in (("kernels_pair.hip", "-DSTAMPS"), ("kernels_last.hip", "-DKL_ABL_NO_EPI"), ("kernels_wino.hip", "-DSTAMPS")):
        r = subprocess.run(base + ["-DREVE_DIAGNOSTIC_BUILD", flag, "-I" + CSRC, os.path.join(CSRC, f)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (f, flag, r.stderr[-1500:])


@pytest.fixture(scope="module")
def isa_pair(tmp_path_factory):
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not present")
    d = tmp_path_factory.mktemp("isa_pair")
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-save-temps",
                        "-I" + CSRC, "-c", os.path.join(CSRC, "kernels_pair.hip"), "-o", "k.o"], cwd=d, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return open(os.path.join(d, "kernels_pair-hip-amdgcn-amd-amdhsa-gfx950.s")).read()


def test_fused_pair_kernel_stream(isa_pair):
    """k_pair (two body layers per launch): per step a wave of the first layer issues 5 LDS-DMA pieces and waits with vmcnt(5), a
    wave of the second 4 pieces + 8 activation stores and waits with vmcnt(12) — the counts its end-of-step `s_waitcnt` relies on
    to know that the PREVIOUS step's pieces have landed; 288 MFMAs per wave and step; the first layer's waves write 8 pieces to
    the LDS ring; weights in 256 AGPRs, nothing spilled, no scratch, no MFMA result round trip through AGPRs."""
    n = 0
    for m in re.finditer(r"^(_ZN4reve6k_pairILb(\d)ELb(\d)EEEvNS_8PairArgsE):\s*;", isa_pair, re.M):
        n += 1
        asm = isa_pair[m.end():isa_pair.index("s_endpgm", m.end())]
        lines = [l.strip() for l in asm.split("\n")]
        # the two step bodies are the two longest runs of straight-line code that contain MFMAs
        runs, cur = [], []
        for l in lines:
            if l.startswith(("s_cbranch", "s_branch", "s_barrier")) or re.match(r"^\.LBB\d+_\d+:", l):
                if cur:
                    runs.append(cur)
                cur = []
            else:
                cur.append(l)
        runs = [r for r in runs if sum(x.startswith("v_mfma_f32_16x16x32_f16") for x in r) > 0]
        steps = sorted(runs, key=lambda r: -sum(x.startswith("v_mfma") for x in r))[:2]
        kinds = set()
        for r in steps:
            assert sum(x.startswith("v_mfma_f32_16x16x32_f16") for x in r) == 288
            dma = sum(bool(re.match(r"buffer_load_dwordx4 .* lds", x)) for x in r)
            stores = sum(x.startswith("buffer_store_dwordx4") for x in r)
            lds_writes = sum(x.startswith("ds_write_b128") for x in r)
            assert (dma, stores, lds_writes) in ((5, 0, 8), (4, 8, 0)), (dma, stores, lds_writes)
            kinds.add("B" if stores else "A")
            assert not any(x.startswith(("scratch_", "v_accvgpr_read", "v_accvgpr_write", "v_readlane", "v_writelane")) for x in r)
            # The step's instruction budget.  A wave alone on its SIMD issues one instruction of any class per 4 cycles (an MFMA takes
            # two turns): 8 x 288 + 4 x everything else must stay below the MFMA pipe's 16 x 288 with room for what does not overlap
            # (679 with roles and phases outside the step loop, ~820 when they were branches inside it).
            n_instr = sum(bool(re.match(r"[a-z]\w+", x)) for x in r)
            assert n_instr <= (700 if (m.group(2), m.group(3)) == ("1", "0") else 760), (m.group(1), n_instr)
            assert sum(x.startswith("v_mov_b32") for x in r) <= 4, "register copies inside a step: paths are merging in the step loop again"
        assert kinds == {"A", "B"}
        # the whole kernel: round 5 took the two lab variants out of it (strips rolled bottom-up, per-XCD segment tables with in-kernel
        # clocks): 3,142 -> 2,883 instructions for the shipped instantiation, 3,518 -> 3,200 for the canvas one; a bound that a
        # variant creeping back in would break
        total = sum(bool(re.match(r"[a-z]\w+", x)) for x in lines)
        assert total <= {("1", "0"): 2950, ("0", "0"): 3050, ("1", "1"): 3280, ("0", "1"): 3380}[(m.group(2), m.group(3))], (m.group(1), total)
        waits = [int(x) for x in re.findall(r"s_waitcnt vmcnt\((\d+)\) lgkmcnt\(0\)", asm)]
        # end-of-step waits: B idle, A, B active (the canvas instantiations also wait for their gutter-column bytes at the start of a unit)
        assert sorted(w for w in set(waits) if w < 30) == [4, 5, 12] and (m.group(3) == "1" or max(waits) == 12), waits
    assert n == 4          # unit slopes or the general PReLU form x whole frame or a canvas of planes with gutters
    meta = isa_pair[isa_pair.index("amdhsa.kernels:"):]
    for blk in meta.split("  - .agpr_count:")[1:]:
        assert int(blk.split()[0]) == 256
        assert int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1)) <= 512
        # the unit-slope instantiation (every model seen so far) spills nothing; the general PReLU form may park up to 16 registers
        # around the weights prologue and the final flush — never inside a step (checked above: no scratch_ in the step bodies)
        general = "k_pairILb0E" in re.search(r"\.name:\s+(\S+)", blk).group(1)
        assert int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1)) <= (16 if general else 0)
        assert int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1)) <= (128 if general else 0)
        assert int(re.search(r"\.sgpr_spill_count:\s+(\d+)", blk).group(1)) == 0          # (6 and 23 before the lab fields left PairArgs)
    # the diagnostic switches stop a build that does not ask for them
    for flag in ("-DSTAMPS", "-DABLP_NO_EPI"):
        r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form=1", flag,
                            "--cuda-device-only", "-fsyntax-only", "-I" + CSRC, os.path.join(CSRC, "kernels_pair.hip")],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode != 0 and "diagnostic" in r.stderr, (flag, r.stderr[-500:])


def test_conv_last_strip_kernel_stream(tmp_path):
    """k_last_strip<2 / 3 / 4> (kernels_last.hip, option "strip_last"): each of the two written-out steps (even / odd) carries
    72 MFMAs per co-block, 36 operand reads, 8 LDS-DMA pieces, 4 residual loads and the rows' stores (x2: a dword and a short per
    row, x3: 8 bytes and three single bytes, x4: 12 bytes), and ends with the counted `s_waitcnt vmcnt(N)` that relies on exactly
    those N vector-memory instructions being younger than the previous step's pieces; nothing spilled, no accumulator-file round
    trips, no waterfall loops inside a step; the timing-only KL_ABL_* switches stop a build that does not ask for them."""
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not present")
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-save-temps",
                        "-I" + CSRC, "-c", os.path.join(CSRC, "kernels_last.hip"), "-o", "k.o"], cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    text = open(os.path.join(tmp_path, "kernels_last-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    for sc, ncob, stores in ((2, 1, {"buffer_store_dword ": 4, "buffer_store_short": 4}), (3, 2, {"buffer_store_dwordx2": 4, "buffer_store_byte": 12}),
                             (4, 3, {"buffer_store_dwordx3": 4})):
        # (both instantiations: whole frames, and — CANVAS — the interiors of a tiled frame's planes)
        for canvas in (0, 1):
            m = re.search(r"^_ZN4reve12k_last_stripILi%dELb%dEEEvNS_13LastStripArgsE:\s*;" % (sc, canvas), text, re.M)
            asm = text[m.end():text.index("s_endpgm", m.end())]
            runs, cur = [], []
            for l in (x.strip() for x in asm.split("\n")):
                if l.startswith(("s_cbranch", "s_branch", "s_barrier")) or re.match(r"^\.LBB\d+_\d+:", l):
                    if cur:
                        runs.append(cur)
                    cur = []
                else:
                    cur.append(l)
            steps = [r for r in runs if sum(x.startswith("v_mfma_f32_16x16x32_f16") for x in r) == 72 * ncob]
            assert len(steps) == 2, sc
            n_vmem = 4 + 8 + sum(stores.values())
            for r in steps:
                assert sum(x.startswith("ds_read_b128") for x in r) == 36
                assert sum(bool(re.match(r"buffer_load_dwordx4 .* lds", x)) for x in r) == 8
                assert sum(bool(re.match(r"buffer_load_dword v", x)) for x in r) == 4
                for kind, n in stores.items():
                    assert sum(x.startswith(kind) for x in r) == n, (sc, kind)
                assert sum(x.startswith("buffer_store") for x in r) == sum(stores.values())
                assert [x for x in r if x.startswith("s_waitcnt vmcnt")][-1] == "s_waitcnt vmcnt(%d)" % n_vmem, sc
                assert not any(x.startswith(("scratch_", "v_readlane", "v_writelane", "v_accvgpr_read", "v_accvgpr_write")) for x in r), sc
                assert sum(x.startswith("v_readfirstlane") for x in r) <= 4, sc        # (loop control between the steps; none per row)
    for blk in text[text.index("amdhsa.kernels:"):].split("  - .agpr_count:")[1:]:
        assert int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1)) == 0
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-DKL_ABL_NO_EPI", "--cuda-device-only", "-fsyntax-only",
                        "-I" + CSRC, os.path.join(CSRC, "kernels_last.hip")], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "timing-only" in r.stderr



@pytest.fixture(scope="module")
def isa_wino(tmp_path_factory):
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not present")
    d = tmp_path_factory.mktemp("isa_wino")
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-fno-slp-vectorize", "-save-temps",
                        "-I" + CSRC, "-c", os.path.join(CSRC, "kernels_wino.hip"), "-o", "k.o"], cwd=d, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return open(os.path.join(d, "kernels_wino-hip-amdgcn-amd-amdhsa-gfx950.s")).read()


def test_winograd_pair_kernel_stream(isa_wino):
    """k_wino (option "winograd"): per step and wave 192 MFMAs, 64 operand reads, k_pair's DMA pieces / stores / ring writes and the
    counted waits that rely on them; its U fragments in 192 AGPRs, nothing spilled; no accumulator-file copies, no register copies
    and no packed fp32 sums inside a step (each was measured as lost issue turns: docs/LAB_NOTES.md R4-6), and an instruction
    budget: the kernel is bound by instruction issue, 8 cycles an MFMA + 4 anything else."""
    n = 0
    for m in re.finditer(r"^(_ZN4reve6k_winoILb(\d)ELb(\d)EEEvNS_8PairArgsE):\s*;", isa_wino, re.M):
        n += 1
        asm = isa_wino[m.end():isa_wino.index("s_endpgm", m.end())]
        lines = [l.strip() for l in asm.split("\n")]
        runs, cur = [], []
        for l in lines:
            if l.startswith(("s_cbranch", "s_branch", "s_barrier")) or re.match(r"^\.LBB\d+_\d+:", l):
                if cur:
                    runs.append(cur)
                cur = []
            else:
                cur.append(l)
        steps = sorted(runs, key=lambda r: -sum(x.startswith("v_mfma") for x in r))[:2]
        kinds = set()
        for r in steps:
            assert sum(x.startswith("v_mfma_f32_16x16x32_f16") for x in r) == 192
            assert sum(x.startswith("ds_read_b128") for x in r) == 64
            dma = sum(bool(re.match(r"buffer_load_dwordx4 .* lds", x)) for x in r)
            stores = sum(x.startswith("buffer_store_dwordx4") for x in r)
            lds_writes = sum(x.startswith("ds_write_b128") for x in r)
            assert (dma, stores, lds_writes) in ((5, 0, 8), (4, 8, 0)), (dma, stores, lds_writes)
            kinds.add("B" if stores else "A")
            assert not any(x.startswith(("scratch_", "v_accvgpr_read", "v_accvgpr_write", "v_readlane", "v_writelane", "v_pk_add_f32", "v_fma_mix")) for x in r)
            assert sum(x.startswith("v_mov_b32") for x in r) <= 4
            n_instr = sum(bool(re.match(r"[a-z]\w+", x)) for x in r)
            assert n_instr <= (930 if (m.group(2), m.group(3)) == ("1", "0") else 990), (m.group(1), n_instr)       # (the first shipped form: ~1,200)
            # the MFMAs are not left bare: between two of them stand at most ~8 other instructions and rarely none
            gaps, g = [], 0
            for x in r:
                if x.startswith("v_mfma"):
                    gaps.append(g)
                    g = 0
                elif re.match(r"[a-z]\w+", x):
                    g += 1
            assert sum(1 for g in gaps[1:] if g == 0) <= 40, sum(1 for g in gaps[1:] if g == 0)
        assert kinds == {"A", "B"}
        waits = [int(x) for x in re.findall(r"s_waitcnt vmcnt\((\d+)\) lgkmcnt\(0\)", asm)]
        assert sorted(w for w in set(waits) if w < 30 and w > 0) == [4, 5, 12], waits
    assert n == 4          # unit slopes or the general PReLU form x whole frame or a canvas of planes with gutters
    meta = isa_wino[isa_wino.index("amdhsa.kernels:"):]
    for blk in meta.split("  - .agpr_count:")[1:]:
        assert int(blk.split()[0]) == 192
        assert int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1)) == 0
        assert int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1)) == 0


def test_winograd_ring_layout_has_no_bank_conflicts():
    """The Winograd kernel's tile reads: lane (t, g) = t + 16 g reads the 16-byte chunk 4 hf + g of pixel 32 q + 2 t + i, one i per
    ds_read_b128.  The LDS serves a b128 read in four groups of sixteen lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same
    + 32: MI355X_MICROARCH.md, LDS); a group is free of conflicts when its sixteen addresses fall into sixteen different 16-byte slots
    of the 256-byte bank row.  With pixels in order that cannot be (one parity of pixels = one half of the row: measured, half of
    the LDS cycles were conflicts); the layout of kw_ring_off is checked here for every read the kernel issues, through the library
    (no GPU needed)."""
    from reve_amd import _lib
    lib = _lib.load()
    off = lib.reve_debug_wino_ring_offset
    assert off(66, 0) < 0 and off(0, 8) < 0
    g0 = [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27]
    g1 = [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]
    groups = [g0, g1, [l + 32 for l in g0], [l + 32 for l in g1]]
    seen = set()
    for q in range(2):
        for hf in range(2):
            for i in range(4):
                for grp in groups:
                    slots = [(off(32 * q + 2 * (l & 15) + i, 4 * hf + (l >> 4)) // 16) % 16 for l in grp]
                    assert len(set(slots)) == 16, (q, hf, i, grp, slots)
    # and it is a layout: the 66 x 8 chunks of a row land on 66 x 8 different 16-byte places inside the row's 72 pixel slots
    for j in range(66):
        for c in range(8):
            o = off(j, c)
            assert 0 <= o < 72 * 128 and o % 16 == 0 and o not in seen
            seen.add(o)

