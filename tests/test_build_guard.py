"""CPU (hipcc cross-compiles gfx950 here): the hand-scheduled scan kernels must not spill.

The main loops of k_scan_f32 / k_scan_bf16_qs / k_scan_bf16_qs2 issue their loads by hand (inline asm, counted waits) and pin
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


def _kernel_meta(src):
    tmp = tempfile.mkdtemp(prefix="lemon_guard_")
    try:
        base = os.path.splitext(src)[0]
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-inline-asm",
                               "-save-temps=obj", "-c", os.path.join(CSRC, src), "-o", os.path.join(tmp, base + ".o")],
                              stderr=subprocess.DEVNULL)
        asm = open(os.path.join(tmp, f"{base}-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
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
                      "k_scan_bf16_qs2ILi8ELi0ELb0ELb0ELb1E", "k_bf16_finalILb0ELb1E", "k_bf16_finalILb1ELb1E"]),
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
