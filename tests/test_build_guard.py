"""CPU (hipcc cross-compiles gfx950 here): the hand-scheduled scan kernels must not spill.

The main loops of k_scan_f32 / k_scan_bf16_qs / k_scan_bf16_qs2 / k_scan_f16_qs4 issue their loads by hand (inline asm, counted waits) and pin
operands to register classes; their correctness and speed both assume that hipcc keeps every staging / fragment / stationary
register where it was put.  A spill would (a) reload the stationary query fragments from scratch inside the MFMA loop behind a
`vmcnt(0)` that drains the LDS-DMA queue (seen while building k_scan_bf16_qs2: DESIGN.md section 4) and (b) let the compiler
touch registers an in-flight asm load still owns.  So the production instantiations are checked for zero VGPR spills and
zero scratch on every build of the test suite (round-2 advisor finding)."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lemon_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


_ASM_CACHE = {}


def _kernel_asm(src):
    """The gfx950 assembly hipcc generates for a kernel source (-save-temps), once per test session."""
    if src in _ASM_CACHE:
        return _ASM_CACHE[src]
    tmp = tempfile.mkdtemp(prefix="lemon_guard_")
    try:
        base = os.path.splitext(src)[0]
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-inline-asm",
                               "-save-temps=obj", "-c", os.path.join(CSRC, src), "-o", os.path.join(tmp, base + ".o")],
                              stderr=subprocess.DEVNULL)
        asm = open(os.path.join(tmp, f"{base}-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    _ASM_CACHE[src] = asm
    return asm


def _kernel_meta(src):
    asm = _kernel_asm(src)
    meta = {}
    for blk in asm.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        meta[name] = {k_: int(re.search(rf"\.{k_}:\s+(\d+)", blk).group(1))
                      for k_ in ("vgpr_spill_count", "private_segment_fixed_size", "vgpr_count")}
    return meta


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("src,must_be_clean", [
    # (mangled-name fragments of the PRODUCTION instantiations: PROF = false)
    ("knn_f32.hip", ["k_scan_f32ILb0ELb0ELb0E", "k_scan_f32ILb1ELb0ELb0E"]),
    ("knn_bf16.hip", ["k_scan_bf16_qsILi12ELb0ELb0E", "k_scan_bf16_qsILi8ELb0ELb0E", "k_scan_bf16_qsILi4ELb0ELb0E",
                      "k_scan_bf16_qs2ILi12ELi16ELb0ELb0ELb1E", "k_scan_bf16_qs2ILi12ELi20ELb1ELb0ELb1E",
                      "k_scan_bf16_qs2ILi8ELi0ELb0ELb0ELb1E", "k_bf16_finalILb0ELb1E", "k_bf16_finalILb1ELb1E",
                      # round 5: the same scan on v_mfma_f32_16x16x32_f16 (IP at pitches 768 / 512, squared L2 at pitch 512)
                      "k_scan_f16_qs4ILi12ELi16ELb0E", "k_scan_f16_qs4ILi8ELi0ELb0E", "k_scan_f16_qs4ILi8ELi0ELb1E"]),
    # the hand-written GEMM (asm LDS-DMA / ds_read / MFMA with pinned accumulators: two workgroups per CU need <= 256 registers)
    # and the attention kernels (three waves per SIMD)
    ("gemm_f16x3.hip", ["k_gemm_f16x3tILi0E", "k_gemm_f16x3tILi1E"]),
    ("attention.hip", ["k_attention_hd64_shortILi2ELi2ELb1E", "k_attention_hd64_shortILi1ELi2ELb1E", "k_attention_hd64_shortILi2ELi0ELb1E",
                       "k_attention_hd64ILi2ELb1E", "k_attention_hd64ILi0ELb1E"]),
])
def test_scan_kernels_do_not_spill(src, must_be_clean):
    meta = _kernel_meta(src)
    for frag in must_be_clean:
        hits = [n for n in meta if frag in n]
        assert hits, f"{frag} not found among {sorted(meta)[:6]}..."
        for n in hits:
            assert meta[n]["vgpr_spill_count"] == 0 and meta[n]["private_segment_fixed_size"] == 0, (n, meta[n])
            if "k_gemm_f16x3t" in n:
                assert meta[n]["vgpr_count"] <= 256, (n, meta[n])          # 128 AccVGPRs + 128 VGPRs: two waves per SIMD
            if "k_attention_hd64_short" in n:
                assert meta[n]["vgpr_count"] <= 168, (n, meta[n])          # three waves per SIMD


# ---- the hand-issued asm and the compiler around it (round-3 verdict item 7, advisor finding on M0) --------------------------
# (a) `s_mov_b32 m0, ...; global_load_lds_dwordx4` is issued from inline asm that cannot declare its M0 write (M0 is a reserved
#     register: hipcc warns that a clobber "may not be preserved" and ignores it), so correctness rests on hipcc itself never
#     using M0 in these kernels -- asserted here on the generated ISA: no instruction OUTSIDE an asm block mentions m0.
# (b) the asm ds_read / global_load statements deliver their destination registers late (the counted s_waitcnt that retires
#     them is a separate asm statement naming those registers); between issue and wait no compiler-generated instruction may
#     read or write such a register (a copy inserted there would read stale data).  Walked in text order over each kernel
#     (straight-line approximation: pending registers are dropped at unconditional branches and at the end of the function).
_ASM_LOAD = re.compile(r"^(ds_read_b(?:32|64|96|128)|ds_read_b64_tr_b16|global_load_dword(?:x[234])?|global_load_ushort|buffer_load_dword(?:x[234])?)\s+(v\[\d+:\d+\]|v\d+)")
_VREG = re.compile(r"\bv(?:\[(\d+):(\d+)\]|(\d+))")


def _vregs(text):
    out = set()
    for m in _VREG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def _kernel_bodies(asm):
    """{mangled name: [lines]} of every kernel function in a -save-temps .s file."""
    bodies, name, cur = {}, None, None
    for ln in asm.splitlines():
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", ln)
        if m:
            name, cur = m.group(1), []
            continue
        if name is not None:
            if ln.startswith(".Lfunc_end"):
                bodies[name] = cur
                name = None
            else:
                cur.append(ln)
    return bodies


def _check_asm_discipline(name, lines):
    in_asm = False
    pend = {"lgkm": [], "vm": []}          # issue-ordered [set of dest regs] per counter
    problems = []
    valu_written = {}                      # VGPR -> [instructions, MFMAs] issued since a compiler-generated VALU wrote it
    for i, raw in enumerate(lines):
        ln = raw.split(";")[0].strip() if ";;#" not in raw else raw.strip()
        if ";;#ASMSTART" in raw:
            in_asm = True
            continue
        if ";;#ASMEND" in raw:
            in_asm = False
            continue
        if not ln or ln.endswith(":") or ln.startswith("."):
            continue
        if in_asm:
            # (c) hipcc pads no wait states in front of an asm MFMA: a compiler-generated VALU write of one of its A / B
            #     operand registers must be at least one other MFMA or three instructions old.  (Measured on gfx950 with the
            #     16x16x32 kernel: `v_pk_mul_f16 v100; s_nop 0; v_mfma ... v[100:103]` used the OLD v100 -- one wait state is
            #     not enough --, while v101 / v102, written two and three instructions ahead, were current: the hardware
            #     needs two wait states, as LLVM pads for an MFMA it can see; three are asked for here.)
            if ln.startswith("v_mfma"):
                ops = [o.strip() for o in ln.split(None, 1)[1].split(",")]
                for r in _vregs(" ".join(ops[1:3])):
                    if r in valu_written and valu_written[r][1] < 1 and valu_written[r][0] < 3:
                        problems.append(f"{name}: asm `{ln}` reads v{r} {valu_written[r][0]} instruction(s) after a compiler-generated VALU wrote it (line {i})")
                for v in valu_written.values():
                    v[0] += 1; v[1] += 1
            else:
                for v in valu_written.values():
                    v[0] += 1
            m = _ASM_LOAD.match(ln)
            if m and " lds" not in ln:
                pend["lgkm" if ln.startswith("ds_") else "vm"].append(_vregs(m.group(2)))
            elif "lds" in ln and ln.startswith("global_load_lds"):
                pend["vm"].append(set())             # counts in vmcnt, no register destination
            for cnt, key in (("lgkmcnt", "lgkm"), ("vmcnt", "vm")):
                w = re.search(cnt + r"\((\d+)\)", ln) if ln.startswith("s_waitcnt") else None
                if w:
                    keep = int(w.group(1))
                    pend[key] = pend[key][len(pend[key]) - keep:] if keep else []
            continue
        # compiler-generated instruction
        if re.search(r"\bm0\b", ln):
            problems.append(f"{name}: compiler-generated use of M0 at line {i}: {ln}")
        if ln.startswith(("s_branch", "s_endpgm", "s_setpc")):
            pend = {"lgkm": [], "vm": []}
            continue
        if ln.startswith("s_waitcnt"):
            # the compiler's own waits retire too (vmcnt(0) / lgkmcnt(0) forms)
            for cnt, key in (("lgkmcnt", "lgkm"), ("vmcnt", "vm")):
                w = re.search(cnt + r"\((\d+)\)", ln)
                if w and int(w.group(1)) == 0:
                    pend[key] = []
            continue
        for v in valu_written.values():
            v[0] += 1
        if ln.startswith("v_") and not ln.startswith(("v_cmp", "v_accvgpr_write", "v_readfirstlane", "v_readlane")):
            dst = ln.split(None, 1)[1].split(",")[0]
            for r in _vregs(dst):
                valu_written[r] = [0, 0]
        live = set().union(*pend["lgkm"], *pend["vm"]) if (pend["lgkm"] or pend["vm"]) else set()
        hit = _vregs(ln) & live
        if hit:
            problems.append(f"{name}: compiler-generated `{ln}` touches v{sorted(hit)} while an asm load to it is in flight (line {i})")
    return problems


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("src,kernels", [
    ("gemm_f16x3.hip", ["k_gemm_f16x3tILi0E", "k_gemm_f16x3tILi1E", "k_gemm_f16x3t16ILi0E", "k_gemm_f16x3t16ILi1E"]),
    ("knn_bf16.hip", ["k_scan_bf16_qs2ILi12ELi16ELb0ELb0ELb1E", "k_scan_bf16_qs2ILi12ELi20ELb1ELb0ELb1E", "k_scan_bf16_qs2ILi8ELi0ELb0ELb0ELb1E",
                      "k_scan_f16_qs4ILi12ELi16ELb0E", "k_scan_f16_qs4ILi8ELi0ELb0E", "k_scan_f16_qs4ILi8ELi0ELb1E"]),
    ("knn_f32.hip", ["k_scan_f32ILb0ELb0ELb0E", "k_scan_f32ILb1ELb0ELb0E"]),
])
def test_hand_issued_asm_is_left_alone_by_the_compiler(src, kernels):
    asm = _kernel_asm(src)
    bodies = _kernel_bodies(asm)
    for frag in kernels:
        hits = [n for n in bodies if frag in n]
        assert hits, f"{frag} not found among {sorted(bodies)[:6]}..."
        for n in hits:
            problems = _check_asm_discipline(n, bodies[n])
            assert not problems, "\n".join(problems[:10])


# ---- k_attention_hd64_f16 (64 < L <= 288): built for four waves per SIMD (two workgroups of seven waves per CU) ----------------
# hipcc keeps within 128 registers by spilling a few per-thread staging addresses AROUND the key-tile loops (written once before
# the first loop, read back once at the second key block's staging): tolerated -- what is asserted is that no scratch access
# sits inside a loop that issues MFMAs, and that the register count really allows the fourth wave.
def _loops_with_mfma_and_scratch(lines, _probe=False):
    """Loop headers (asm labels) whose loop body holds both an MFMA and a scratch access.  hipcc annotates every basic block of a
    loop with `in Loop: Header=BBx_y` / `Loop Header`, on the label line or on a comment line right below it."""
    kinds = {}
    cur, last_label = None, None
    for raw in lines:
        lab = re.match(r"^(?:\.L(BB\d+_\d+):|; %bb\.\d+:)", raw)
        if lab:
            cur, last_label = None, lab.group(1)
        if lab or raw.lstrip().startswith(";"):
            m = re.search(r"in Loop: Header=(BB\d+_\d+)", raw)
            if m:
                cur = m.group(1)
            elif "Loop Header" in raw and last_label:
                cur = last_label
            continue
        if cur is None:
            continue
        ins = raw.split(";")[0].strip()
        if ins.startswith("v_mfma"):
            kinds.setdefault(cur, set()).add("mfma")
        if ins.startswith("scratch_"):
            kinds.setdefault(cur, set()).add("scratch")
    if _probe:                                   # (self-check of the parser: loops that issue MFMAs were found at all)
        return [h for h, k in kinds.items() if "mfma" in k]
    return [h for h, k in kinds.items() if {"mfma", "scratch"} <= k]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_long_sequence_attention_kernel_fits_four_waves_and_keeps_scratch_out_of_its_loops():
    meta = _kernel_meta("attention.hip")
    bodies = _kernel_bodies(_kernel_asm("attention.hip"))
    hits = [n for n in meta if "k_attention_hd64_f16ILi" in n]
    assert len(hits) == 4, sorted(meta)
    for n in hits:
        assert meta[n]["vgpr_count"] <= 128, (n, meta[n])
        assert meta[n]["vgpr_spill_count"] <= 20, (n, meta[n])
        assert _loops_with_mfma_and_scratch(bodies[n]) == [], n
        assert _loops_with_mfma_and_scratch(bodies[n], _probe=True), "the parser must see the MFMA loops"


# ---- k_gemm_f16x3t16 and its LayerNorm-fold variants -------------------------------------------------------------------------
# (1) the two plain instantiations (the measured product kernels) hold everything in registers; the fold variants may spill a few
#     epilogue values, never inside the MFMA loop;
# (2) no packed fp32 instruction takes the HIGH dword of a register pair for its LOW lane (`op_sel:[..1..]`): with
#     `v_pk_mul_f32 ... op_sel:[0,1]` on a (rstd, -mean rstd) pair that had just arrived from LDS / memory the low results of
#     lanes 48-63 were wrong now and then on the GPU (round 4: DESIGN.md "Hardware facts", profiles/r4/ln_fold_opsel_fault.txt); the epilogues keep such
#     products in single registers, and this test keeps hipcc from quietly re-pairing them.
@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_gemm16_variants_register_discipline():
    meta = _kernel_meta("gemm_f16x3.hip")
    bodies = _kernel_bodies(_kernel_asm("gemm_f16x3.hip"))
    names = [n for n in bodies if "k_gemm_f16x3t16ILi" in n]
    assert len(names) == 10, sorted(names)       # fp32; SiLU, GELU operand; fold x (fp32, SiLU, GELU); emit x four chain forms (LEAN)
    for n in names:
        assert meta[n]["vgpr_count"] <= 256, (n, meta[n])            # (arch + accumulation registers: two waves per SIMD)
        if "Lb0ELb0E" in n:
            assert meta[n]["vgpr_spill_count"] == 0 and meta[n]["private_segment_fixed_size"] == 0, (n, meta[n])
        assert _loops_with_mfma_and_scratch(bodies[n]) == [], n
        assert _loops_with_mfma_and_scratch(bodies[n], _probe=True), "the parser must see the MFMA loop"
        bad = [ln.strip() for ln in bodies[n] if re.search(r"\bv_pk_\w+_f32\b", ln) and re.search(r"op_sel:\[[01,]*1", ln)]
        assert not bad, (n, bad[:4])


# ---- every kernel of the library: no packed fp32 instruction whose LOW lane takes the HIGH dword of a pair -------------------
# tools/micro/pk_opsel_war.hip (variants 10-12, nothing hand-issued): with MFMA waves of the same workgroup on the SIMD, hipcc's own
# `v_pk_mul_f32 d, a, b op_sel:[0,1]` + `v_pk_fma_f32` give wrong LOW results in lanes 48-63 a few thousand times per 6.5e8; with
# the multiplier broadcast into its own pair first (no op_sel; op_sel_hi forms allowed) never.  Any kernel here can share a SIMD
# with matrix instructions, so the pattern is refused everywhere.
@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_no_kernel_uses_a_packed_fp32_op_with_op_sel():
    from lemon_amd import build
    for src in build.SOURCES:
        if not os.path.exists(os.path.join(CSRC, src)):
            continue
        for name, lines in _kernel_bodies(_kernel_asm(src)).items():
            bad = [ln.strip() for ln in lines if re.search(r"\bv_pk_\w+_f32\b", ln) and re.search(r"op_sel:\[[01,]*1", ln)]
            assert not bad, (src, name, bad[:3])
