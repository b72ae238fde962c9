"""The C-ABI library builds, loads and exports every symbol include/tacorl_hip.h declares
(no compute calls: there is no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "tacorl_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tacorl_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from tacorl_amd import _lib, build

    build.build(verbose=False)
    L = _lib.lib()
    syms = _header_symbols()
    assert len(syms) >= 20
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing
    # the ctypes table binds exactly the header's symbols
    assert sorted(_lib.declared_symbols()) == syms


def test_layout_queries_run_without_gpu():
    from tacorl_amd import ops

    offs, total = ops.encoder_param_layout()
    assert total == 117188 and offs[7] % 4 == 0  # 117185 reference params + 3 pad floats
    _, act = ops.encoder_act_layout(2, 84, 84)
    assert act == 2 * (400 * 32 + 81 * 64 + 49 * 64 + 128 + 256)
    w, b, tot = ops.mlp_param_layout([71, 256, 1])
    assert w[0] == 0 and b[0] == 71 * 256 and tot >= 71 * 256 + 256 + 256 + 1


def test_launch_batch_state_machine_without_gpu():
    """tacorl_prep_batch_* / tacorl_reduce_batch_*: begin / end pair up per thread, nesting and a stray end are refused, and an
    EMPTY batch launches nothing (so this runs without a GPU); supported-shape queries of the round-5 entry points."""
    import ctypes as C

    from tacorl_amd import _lib

    L = _lib.lib()
    for begin, end in ((L.tacorl_prep_batch_begin, L.tacorl_prep_batch_end), (L.tacorl_reduce_batch_begin, L.tacorl_reduce_batch_end)):
        assert end(None) != 0          # no open batch
        assert begin() == 0
        assert begin() != 0            # no nesting
        assert end(None) == 0          # empty: nothing to launch
        assert end(None) != 0
        assert begin() == 0 and end(None) == 0
    assert L.tacorl_rnn_linear_supported(3840, 2048, 192) == 1 and L.tacorl_rnn_linear_supported(256, 100, 64) == 0
    assert L.tacorl_encoder_fused_supported(84, 84) == 1 and L.tacorl_encoder_fused_supported(150, 200) == 1  # (round 6: encoder_ring.hip)
    assert L.tacorl_encoder_fused_supported(200, 200) == 0 and L.tacorl_encoder_bwd_fused_ws_bytes(1, (C.c_int * 1)(8), 200, 200) == 0
    assert L.tacorl_encoder_bwd_fused_ws_bytes(1, (C.c_int * 1)(8), 150, 200) > 0 and L.tacorl_encoder_fused_act_format(150, 200) == 1


def test_rnn_wgrad_batch_refuses_bad_arguments_without_gpu():
    """tacorl_rnn_wgrad_batch validates its arguments before it touches the device: problem counts outside 1..4, row counts the
    kernel does not take, misaligned operands and leading dimensions are refused (nothing is launched, so this runs here)."""
    import ctypes as C

    from tacorl_amd import _lib

    L = _lib.lib()
    assert L.tacorl_rnn_wgrad_supported(3840, 2048, 2048) == 1 and L.tacorl_rnn_wgrad_supported(3840 + 32, 2048, 2048) == 0
    P4, I4 = C.c_void_p * 4, C.c_int * 4
    ok_ptr = P4(4096, 8192, 12288, 16384)  # (never dereferenced: every call below is refused in the argument checks)
    rows = I4(3840, 3840, 3584, 3584)
    call = lambda n, dz, x, R, dw, ld=2048: L.tacorl_rnn_wgrad_batch(n, dz, ld, x, ld, R, 2048, 2048, dw, None, 0, None)  # noqa: E731
    assert call(0, ok_ptr, ok_ptr, rows, ok_ptr) != 0 and call(5, ok_ptr, ok_ptr, rows, ok_ptr) != 0
    assert call(4, ok_ptr, ok_ptr, I4(3840, 3840, 3584, 3584 + 32), ok_ptr) != 0      # a row count that is not a whole stage
    assert call(2, P4(4096, 8200, 0, 0), ok_ptr, rows, ok_ptr) != 0                     # 8-byte aligned operand
    assert call(2, ok_ptr, ok_ptr, rows, ok_ptr, ld=2044) != 0 and call(2, ok_ptr, ok_ptr, rows, ok_ptr, ld=1024) != 0


def test_missing_library_fails_loudly(monkeypatch):
    from tacorl_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libtacorl_hip.so")
    with pytest.raises(_lib.TacorlHipError):
        _lib.lib()


def test_encoder_fused_code_object_leaves_m0_to_the_dma_pieces(tmp_path):
    """Round 6: the fused encoder's in-loop LDS-DMA pieces set M0 (the LDS destination) once per group of four and rely on it
    STAYING there between asm statements - legitimate only while hipcc itself never reads or writes M0 in these kernels.
    Checked on the built gfx950 code object: inside every encoder_fused_kernel<H, W> the only instructions that mention m0
    are `s_mov_b32 m0, <sgpr>` (a piece's set / restore) and `s_mov_b32 <sgpr>, m0` (the old-style pieces' save)."""
    import re
    import shutil
    import subprocess

    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    obj = os.path.join(here, "tacorl_amd", "lib", "obj", "encoder_fused.o")
    tools = "/opt/rocm/lib/llvm/bin"
    need = [os.path.join(tools, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")]
    if not os.path.exists(obj) or not all(os.path.exists(t) or shutil.which(os.path.basename(t)) for t in need):
        pytest.skip("needs the in-tree object file (python -m tacorl_amd.build) and the ROCm llvm tools")
    fat, co = str(tmp_path / "fat.bin"), str(tmp_path / "dev.co")
    subprocess.run([need[0], "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
    subprocess.run([need[1], "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    f"--output={co}"], check=True)
    dis = subprocess.run([need[2], "-d", co], check=True, capture_output=True, text=True).stdout
    ok = re.compile(r"^s_mov_b32 (m0, (s\d+|vcc_lo|vcc_hi)|(s\d+|vcc_lo|vcc_hi), m0)$")
    kernel, seen, bad = None, 0, []
    for ln in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
        if m:
            kernel = m.group(1)
            continue
        if kernel is None or "encoder_fused_kernel" not in kernel or "m0" not in ln:
            continue
        ins = re.sub(r"\s+", " ", ln.split("//")[0]).strip()
        seen += 1
        if not ok.match(ins):
            bad.append(f"{kernel}: {ins}")
    assert seen > 0, "no M0 write found at all: has the kernel's DMA changed?"
    assert not bad, bad[:10]


def _kernel_disassembly(obj_name, tmp_path):
    """{kernel name: [instruction text]} of an in-tree gfx950 object file (None when it or the ROCm llvm tools are missing)."""
    import re
    import shutil
    import subprocess

    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    obj = os.path.join(here, "tacorl_amd", "lib", "obj", obj_name)
    tools = "/opt/rocm/lib/llvm/bin"
    need = [os.path.join(tools, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")]
    if not os.path.exists(obj) or not all(os.path.exists(t) or shutil.which(os.path.basename(t)) for t in need):
        return None
    fat, co = str(tmp_path / (obj_name + ".fat")), str(tmp_path / (obj_name + ".co"))
    subprocess.run([need[0], "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
    subprocess.run([need[1], "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    f"--output={co}"], check=True)
    dis = subprocess.run([need[2], "-d", co], check=True, capture_output=True, text=True).stdout
    out, kernel = {}, None
    for ln in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
        if m:
            kernel = m.group(1)
            out.setdefault(kernel, [])
            continue
        ins = re.sub(r"\s+", " ", ln.split("//")[0]).strip()
        if kernel is not None and ins and not ins.endswith(":"):
            out[kernel].append(ins)
    return out


@pytest.mark.parametrize("obj,kernel", [("encoder_fused.o", "encoder_fused_kernel"), ("encoder_ring.o", "encoder_ring_kernel")])
def test_encoder_code_objects_wait_for_mfma_results(obj, kernel, tmp_path):
    """The fused encoder kernels issue their MFMAs from inline asm, which hipcc's hazard recogniser does not see: the
    read-after-MFMA wait states (an XDL result needs ~12 before a non-MFMA reader, cdna_hip_programming.md section 5.7) are the
    source's own `s_nop`s and the distance its schedules keep.  What the source cannot see are the register COPIES hipcc adds
    at a branch merge or a loop edge for a live accumulator - round 6 met `v_mov_b64` copies three instructions behind a chain's
    last MFMA (pre-MFMA values, wrong pixels).  Checked on the built code object, in program order: no non-MFMA instruction
    reads a VGPR that an MFMA wrote fewer than 10 wait states earlier (s_nop N = N + 1, an MFMA in between = 4, anything else = 1)."""
    import re

    dis = _kernel_disassembly(obj, tmp_path)
    if dis is None:
        pytest.skip("needs the in-tree object file (python -m tacorl_amd.build) and the ROCm llvm tools")
    vreg = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")

    def regs(tok):
        r = set()
        for m in vreg.finditer(tok):
            if m.group(1) is not None:
                r.add(int(m.group(1)))
            else:
                r.update(range(int(m.group(2)), int(m.group(3)) + 1))
        return r

    checked, bad = 0, []
    for name, body in dis.items():
        if kernel not in name:
            continue
        recent = []  # [set of destination VGPRs, wait states since]
        for ins in body:
            op, _, rest = ins.partition(" ")
            ops_ = [t.strip() for t in rest.split(",")]
            if op.startswith("v_mfma"):
                for r in recent:
                    r[1] += 4
                recent.append([regs(ops_[0]), 0])
                checked += 1
            else:
                # sources: every operand of a store / DS write, every operand but the first of anything else that names VGPRs
                is_store = op.startswith(("ds_write", "global_store", "scratch_store", "buffer_store", "flat_store"))
                src = set().union(*[regs(t) for t in (ops_ if is_store else ops_[1:])]) if ops_ else set()
                for dst, age in recent:
                    if age < 10 and src & dst:
                        bad.append(f"{name}: `{ins}` reads v{sorted(src & dst)} {age} wait states behind the MFMA that writes them")
                step = int(ops_[0], 0) + 1 if op == "s_nop" else 1
                for r in recent:
                    r[1] += step
            recent = [r for r in recent if r[1] < 16]
    assert checked > 100, f"no {kernel} MFMAs found: has the kernel changed?"
    assert not bad, bad[:8]

