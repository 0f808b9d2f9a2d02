"""Guards on the generated gfx950 code (CPU: hipcc cross-compiles; no GPU needed).

The weight-gradient kernel stages its operands with LDS-DMA (`global_load_lds`) and orders DMA and fragment reads with
its own counted `s_waitcnt vmcnt(N)`.  clang puts a full `s_waitcnt vmcnt(0)` in front of LDS reads it cannot prove
independent of outstanding LDS-DMA (it did for the `ds_read_tr16_b64` intrinsic: 515 instead of 396 us per layer,
DESIGN.md section 4.2), which silently drains the prefetch pipeline.  This test fails if such a wait comes back.
"""
import os
import re
import shutil
import subprocess

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "clg-vqa_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _kernel_bodies(asm, name_part):
    """{mangled name: [lines]} of the kernels whose name contains name_part"""
    out, cur = {}, None
    for line in asm.split("\n"):
        m = re.match(r"^(_Z\S+):", line)
        if m:
            cur = m.group(1) if name_part in m.group(1) else None
            if cur:
                out[cur] = []
        elif cur is not None:
            out[cur].append(line)
            if "s_endpgm" in line:
                cur = None
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_dw_kernel_main_loop_has_no_compiler_inserted_dma_drain(tmp_path):
    asm_file = tmp_path / "dw.s"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-S", "--cuda-device-only",
                    "-o", str(asm_file), os.path.join(CSRC, "dw.hip")], check=True, capture_output=True, timeout=600)
    kernels = _kernel_bodies(asm_file.read_text(), "dw_grouped_kernel")
    assert len(kernels) == 8, sorted(kernels)  # the four operand-layout modes x {tile per workgroup, stream-K}
    for name, body in kernels.items():
        in_asm, in_loop, n_reads, n_dma = False, False, 0, 0
        for i, line in enumerate(body):
            t = line.strip()
            if "#ASMSTART" in t:
                in_asm = True
            elif "#ASMEND" in t:
                in_asm = False
            elif re.match(r"^\.LBB\d+_\d+:", t):
                in_loop = "Loop" in t
            n_reads += t.startswith("ds_read_b64_tr_b16") or t.startswith("ds_read_b128")
            n_dma += t.startswith("global_load_lds_dwordx4")
            if in_loop and not in_asm and t.startswith("s_waitcnt") and "vmcnt" in t:
                # a wait the compiler added inside a loop: allowed only in the epilogue's read-modify-write loop
                # (mask / accumulate: plain global loads), never in front of an LDS fragment read
                nxt = next((b.strip() for b in body[i + 1:] if b.strip() and not b.strip().startswith(";")), "")
                assert not nxt.startswith("ds_read"), f"{name}: '{t}' before '{nxt}'"
        assert n_reads >= 24 and n_dma >= 16, (name, n_reads, n_dma)  # the loop really is LDS-DMA + fragment reads
        if "Lb1E" in name:
            # stream-K form: the segment loop around the K loop spills a few per-segment values (addresses of the hand-off
            # slots) -- never inside the K loop: nothing between the first and the last MFMA may touch scratch
            mf = [i for i, b in enumerate(body) if b.strip().startswith("v_mfma")]
            inner = "\n".join(body[mf[0]:mf[-1] + 1])
            assert "scratch_" not in inner, f"{name}: register spills inside the K loop"
        else:
            meta = "\n".join(body)
            assert "scratch_" not in meta, f"{name}: register spills"
