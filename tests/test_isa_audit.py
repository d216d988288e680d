"""tools/isa_asm_audit.py on the compiled kernels (CPU; hipcc cross-compiles): no compiler instruction may touch a register an
inline-asm LDS read is still writing (cdna_hip_programming.md §5.7 item 1), and the audit itself must see a planted violation."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_asm_audit as audit  # noqa: E402

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.isfile("/opt/rocm/bin/hipcc"), reason="hipcc not available")


def test_audit_flags_a_copy_of_a_register_an_asm_read_is_still_writing(tmp_path):
    bad = tmp_path / "bad.s"
    bad.write_text("_Zkernel:\n\t;;#ASMSTART\n\tds_read_b128 v[2:5], v18 offset:0\n\t;;#ASMEND\n\tv_mov_b64_e32 v[8:9], v[4:5]\n"
                   "\t;;#ASMSTART\n\ts_waitcnt lgkmcnt(0)\n\t;;#ASMEND\n\tv_add_f32_e32 v1, v2, v3\n\ts_endpgm\n")
    kernels, reads, problems = audit.audit(str(bad))
    assert (kernels, reads) == (1, 1) and len(problems) == 1 and "v_mov_b64" in problems[0]
    good = tmp_path / "good.s"
    good.write_text("_Zkernel:\n\t;;#ASMSTART\n\tds_read_b128 v[2:5], v18 offset:0\n\t;;#ASMEND\n\t;;#ASMSTART\n\ts_waitcnt lgkmcnt(0)\n"
                    "\t;;#ASMEND\n\tv_mov_b64_e32 v[8:9], v[4:5]\n\ts_endpgm\n")
    assert audit.audit(str(good))[2] == []


@pytest.mark.parametrize("src", ["linear_mfma.hip", "thin_layer.hip", "mlp3.hip", "small_rollout.hip"])
def test_no_kernel_touches_a_register_with_an_asm_lds_read_in_flight(src):
    path = os.path.join(ROOT, "neural_inventory_control_amd", "csrc", src)
    extra = ["-ffp-contract=off"] if src == "small_rollout.hip" else []
    _, _, problems = audit.audit(audit.compile_to_asm(path, extra))
    assert problems == [], problems[:5]
