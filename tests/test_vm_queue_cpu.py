"""The post-attention kernel counts its loads in flight by hand (`s_waitcnt vmcnt(N)` from a compile-time model of the
queue).  This test compiles the kernel to assembly and replays the queue: every wait of the output projection must
cover exactly the loads its piece needs (tools/check_vm_queue.py).  No GPU involved: hipcc cross-compiles."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc to generate the kernel's assembly")
def test_output_projection_waits_match_the_issue_order():
    from tools import check_vm_queue as cq

    path = cq.generate_asm()
    try:
        ev = cq.events(path)
        # the tiled kernel really is built the way the checker assumes: loads only through inline assembly in the pass
        assert sum(e[0] == "X" for e in ev) == 12 + 13 * 4
        assert cq.check(path) == []
    finally:
        shutil.rmtree(os.path.dirname(path), ignore_errors=True)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc to generate the kernel's assembly")
def test_output_projection_waits_of_the_16x16x32_kernel():
    """The same queue in the kernel's 16x16x32 form (32 MFMAs per weight piece)."""
    from tools import check_vm_queue as cq

    path = cq.generate_asm("post_attn16.hip")
    try:
        ev = cq.events(path, cq.KERNEL16)
        assert sum(e[0] == "X" for e in ev) == 12 + 13 * 4
        assert sum(1 for l in open(path) if "v_mfma_f32_16x16x32_f16" in l) >= 2048
        assert cq.check(path, cq.KERNEL16, 32) == []
    finally:
        shutil.rmtree(os.path.dirname(path), ignore_errors=True)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc to generate the kernels' assembly")
@pytest.mark.parametrize("source", ["gemm.hip", "attention.hip", "post_attn.hip", "post_attn16.hip", "vocoder_kernels.hip"])
def test_every_lds_dma_load_keeps_its_distance_from_the_m0_write(source):
    """An LDS-DMA load takes its LDS address from M0, and on gfx9 it must not issue in the cycle after the scalar write
    of M0 (the loads are inline assembly, so the compiler's hazard padding does not apply): in the generated assembly of
    every kernel that uses them, the nearest M0 write in front of each load is a scalar move at least two instructions
    away, with no label in between."""
    from tools import check_vm_queue as cq

    path = cq.generate_asm(source)
    try:
        assert cq.check_m0(path) == []
    finally:
        shutil.rmtree(os.path.dirname(path), ignore_errors=True)
