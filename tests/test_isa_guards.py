"""Compile-time guards on the generated gfx950 ISA (no GPU needed: hipcc cross-compiles).

Two kernels rely on properties the compiler does not promise:
* attn_enc_dma_kernel keeps LDS-DMA loads in flight across barriers and fetches the next head's Q rows / table entry with
  inline-asm loads that the waitcnt pass does not track; the destination registers must not be read, copied or spilled
  before the explicit `s_waitcnt vmcnt(0)`, and no compiler-placed vmcnt wait (a tracked load, a spill) may drain the queue;
* gemm_pp2_kernel keeps DMA loads in flight across barriers with counted vmcnt: its K loop must contain no
  `vmcnt(0)` drain and no scratch traffic.
A compiler upgrade that breaks either would give timing-dependent garbage or a silent 2x slowdown; this test fails instead.
"""
import os
import re
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "llm-rankers_amd", "csrc", "rk_engine.hip")


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "rk.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Wno-unused-value",
                    "-w", SRC, "-o", str(out)], check=True, timeout=600)
    return out.read_text().split("\n")


def kernel_body(lines, mangled):
    start = next(i for i, l in enumerate(lines) if l.startswith(mangled + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))   # (a kernel may hold several s_endpgm)
    return lines[start:end + 1]


def vregs(text):
    out = set()
    for a, b in re.findall(r"v\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(a) for a in re.findall(r"\bv(\d+)\b", text))
    return out


@pytest.mark.parametrize("ng", [1, 2])
def test_dma_attention_keeps_its_dma_queue_and_its_asm_destinations(isa, ng):
    """attn_enc_dma_kernel: K/V rows travel by LDS-DMA across barriers and Q rows / table entries by inline-asm loads.  The
    only vmcnt waits may be the kernel's own (inside ASM blocks) - a compiler-placed one means the waitcnt pass saw a
    tracked load or a scratch access and drains the DMA queue; no spills; no register written by an asm load is touched
    between its issue and the asm wait that follows it."""
    body = kernel_body(isa, f"_Z19attn_enc_dma_kernelILi{ng}EEv11AttnEncArgs")
    assert not any("scratch_" in l for l in body), "DMA attention kernel spills"
    m = re.search(r"NumVgprs: (\d+)", "\n".join(isa[isa.index(body[-1]):isa.index(body[-1]) + 400]))
    assert m and int(m.group(1)) <= 168, "more than 168 VGPRs: three waves per SIMD (two groups per CU) no longer fit"
    # four item bodies (1 / 2 / 3 key tiles, and the waves without query rows), each as {V, next K} + last item's V; prologue K
    assert sum("global_load_lds_dwordx4" in l for l in body) == 4 + 4 * 12
    assert sum("ds_read_b64_tr_b16" in l for l in body) == 2 * 16 * (1 + 2 + 3)
    assert sum("v_mfma_f32_32x32x16_f16" in l for l in body) == 2 * 16 * (1 + 2 + 3)
    pending, n_waits = set(), 0
    for i, l in enumerate(body):
        in_asm = i > 0 and "ASMSTART" in body[i - 1]
        code = l.split(";")[0]
        if "vmcnt" in code:
            assert in_asm, f"compiler-placed vmcnt wait in the DMA attention kernel: {l.strip()}"
            if re.search(r"vmcnt\(0\)", code):
                pending.clear()
            n_waits += 1
            continue
        m = re.match(r"\s*global_load_dword(?:x4)?\s+(v\[\d+:\d+\]|v\d+),", code)
        if m and in_asm:
            pending |= vregs(m.group(1))
            continue
        if not code.strip() or code.strip().startswith(".") or "ASM" in l:
            continue
        assert not (vregs(code) & pending), f"asm load destination touched before its wait: {l.strip()}"
    assert n_waits >= 4
    # the token offsets of the next-but-one item: explicit s_load_dwordx2 (a compiler-issued vector load would put a vmcnt(0) in
    # the middle of a head); its destination pair is not read or copied before the asm lgkmcnt(0) that follows it
    n_sloads = 0
    for i, l in enumerate(body):
        m = re.match(r"\s*s_load_dwordx2 s\[(\d+):(\d+)\]", l)
        if not (m and "ASMSTART" in body[i - 1]):
            continue
        n_sloads += 1
        a, b = int(m.group(1)), int(m.group(2))
        for j in range(i + 1, len(body)):
            if "s_waitcnt lgkmcnt(0)" in body[j] and "ASMSTART" in body[j - 1]:
                break
            code = body[j].split(";")[0]
            assert not re.search(rf"\bs{a}\b|\bs{b}\b|s\[{a}:{b}\]", code), f"offset pair touched before its wait: {code.strip()}"
    assert n_sloads == 3                                  # two in the prologue, one per item in the loop
    assert not any(re.match(r"\s*global_load_dwordx2", l) for l in body), "vector load of the token offsets"


# (epilogue kind, folded-RMSNorm row factors, K split over workgroups): every product instantiation of the ping-pong kernel
@pytest.mark.parametrize("epi,rs,split", [(0, 0, 0), (0, 1, 0), (1, 0, 0), (2, 0, 0), (2, 1, 0), (3, 0, 0), (3, 1, 0), (4, 0, 0), (1, 0, 1), (4, 0, 1)])
def test_pingpong_gemm_k_loop_keeps_loads_in_flight(isa, epi, rs, split):
    body = kernel_body(isa, f"_Z15gemm_pp2_kernelILi{epi}ELi0ELb{rs}ELb{split}EEv8GemmArgs")
    assert not any("scratch_" in l for l in body), "ping-pong GEMM spills"
    # the K loop = the innermost loop that holds MFMAs
    heads = [i for i, l in enumerate(body) if "Inner Loop Header" in l]
    assert heads, "no inner loop found"
    loops = []
    for h in heads:
        label = next(body[j].split(":")[0] for j in range(h, h - 3, -1) if body[j].startswith(".LBB"))
        back = max(j for j, l in enumerate(body) if re.search(r"s_cbranch\w+\s+" + re.escape(label) + r"\b", l))
        loops.append(body[h:back + 1])
    kloop = max(loops, key=lambda b: sum("v_mfma" in l for l in b))
    assert sum("v_mfma" in l for l in kloop) == 32, "two super-phases of 16 MFMA per K tile"
    assert sum("global_load_lds_dwordx4" in l for l in kloop) == 8
    # round 4: the DMA instructions are written out in the saddr form (SGPR base pair + one 32-bit lane offset register): no
    # 64-bit address pairs, no v_lshl_add_u64 among the MFMAs - worth 8 VGPRs of offsets and ~2 % of the grouped pipeline
    assert all(re.search(r"global_load_lds_dwordx4 v\d+, s\[\d+:\d+\]", l) for l in kloop if "global_load_lds_dwordx4" in l)
    assert not any("v_lshl_add_u64" in l for l in kloop)
    assert not any(re.search(r"vmcnt\(0\)", l) for l in kloop), "the K loop drains the DMA queue"
    assert sum("s_barrier" in l for l in kloop) == 4
    # M0 belongs to the hand-written DMA statements (it is on their clobber list): nothing else in the kernel reads or writes it
    for i, l in enumerate(body):
        code = l.split(";")[0]
        if re.search(r"\bm0\b", code):
            assert re.match(r"\s*s_mov_b32 m0, s\d+", code) and "ASMSTART" in body[i - 1], f"M0 used outside the DMA statements: {l.strip()}"


@pytest.mark.parametrize("name", ["_Z19dec_cross_qk_kernel9DecQKArgs", "_Z19dec_cross_cv_kernel9DecCVArgs"])
def test_fused_decoder_kernels_fit_two_workgroups_per_cu(isa, name):
    """decoder_kernels.h (round 4): eight-wave workgroups whose load latency is hidden by a second (third) workgroup on the CU -
    at most 128 VGPRs (four waves per SIMD) and no scratch."""
    body = kernel_body(isa, name)
    assert not any("scratch_" in l for l in body), "fused decoder kernel spills"
    tail = "\n".join(isa[isa.index(body[-1]):isa.index(body[-1]) + 400])
    m = re.search(r"NumVgprs: (\d+)", tail)
    assert m and int(m.group(1)) <= 128, m and m.group(1)
    assert sum("v_mfma_f32_32x32x16_f16" in l for l in body) >= 8


def _instructions(body):
    """(index, text) of the machine instructions of a kernel body (labels, directives and comments dropped)."""
    out = []
    for i, l in enumerate(body):
        code = l.split(";")[0].strip()
        if code and not code.startswith(".") and not code.endswith(":"):
            out.append((i, code))
    return out


def test_no_valu_written_sgpr_feeds_an_inline_asm_memory_instruction_without_wait_states(isa):
    """gfx9 / CDNA hazard: a VALU instruction that writes an SGPR (v_readlane / v_readfirstlane - which is how the compiler brings
    a spilled SGPR back - , v_cmp with an SGPR destination) needs FIVE wait states before a VMEM instruction reads that SGPR.
    The compiler's hazard recognizer inserts them for its own instructions but does not look inside inline asm: round 5's chained GEMM
    launch (removed in round 6) restored the row-factor base from a spill lane right in front of a hand-written load and
    faulted on its first launch.  Every hand-written VMEM instruction with an SGPR base (loads, LDS-DMA, atomics) in every kernel
    of the library: no VALU write of its base registers within the five preceding wait states (s_nop N counts N + 1)."""
    text = isa
    starts = [i for i, l in enumerate(text) if re.match(r"^_Z\w+:", l)]
    checked = 0
    for k, s in enumerate(starts):
        end = starts[k + 1] if k + 1 < len(starts) else len(text)
        body = text[s:end]
        ins = _instructions(body)
        in_asm = set()
        inside = False
        for i, l in enumerate(body):
            if "ASMSTART" in l:
                inside = True
            elif "ASMEND" in l:
                inside = False
            elif inside:
                in_asm.add(i)
        for pos, (i, code) in enumerate(ins):
            if i not in in_asm:
                continue
            m = re.match(r"(global_load_\w+|global_atomic_\w+|global_store_\w+)\s.*\bs\[(\d+):(\d+)\]", code)
            if not m:
                continue
            base = set(range(int(m.group(2)), int(m.group(3)) + 1))
            checked += 1
            waits, back = 0, pos - 1
            while back >= 0 and waits < 5:
                prev = ins[back][1]
                n = re.match(r"s_nop (\d+)", prev)
                if n:
                    waits += int(n.group(1)) + 1
                else:
                    w = re.match(r"v_(readlane|readfirstlane)_b32 s(\d+)", prev) or re.match(r"v_cmp\w* s\[(\d+):(\d+)\]", prev)
                    if w:
                        regs = {int(w.group(2))} if w.re.pattern.startswith("v_(read") else set(range(int(w.group(1)), int(w.group(2)) + 1))
                        assert not (regs & base), f"{text[s][:60]}: `{prev}` {waits} wait state(s) in front of hand-written `{code}`"
                    waits += 1
                back -= 1
    assert checked > 50        # the ping-pong GEMM instantiations and the DMA attention kernels hold dozens of such instructions each


@pytest.mark.parametrize("nw,max_vgprs", [(12, 168), (4, 256)])
def test_long_sequence_attention_keeps_its_dma_in_flight(isa, nw, max_vgprs):
    """attn_enc_long_kernel: the next chunk's K / V rows travel by LDS-DMA while the current chunk is computed; the only vmcnt
    waits inside the chunk loop may be the kernel's own (a compiler-placed one - for a tracked load whose first use slipped
    into the loop - would drain the DMA queue of every chunk); no spills; three waves per SIMD at twelve waves per workgroup."""
    body = kernel_body(isa, f"_Z20attn_enc_long_kernelILi{nw}EEv11AttnEncArgs")
    assert not any("scratch_" in l for l in body), "long-sequence attention kernel spills"
    m = re.search(r"NumVgprs: (\d+)", "\n".join(isa[isa.index(body[-1]):isa.index(body[-1]) + 400]))
    assert m and int(m.group(1)) <= max_vgprs, m and m.group(1)
    assert sum("v_mfma_f32_32x32x16_f16" in l for l in body) == 4 * 32       # four chunk bodies (mask x near / far), 16 + 16 MFMAs each
    first_mfma = next(i for i, l in enumerate(body) if "v_mfma" in l)
    last_mfma = max(i for i, l in enumerate(body) if "v_mfma" in l)
    for i in range(first_mfma, last_mfma):
        code = body[i].split(";")[0]
        if "vmcnt" in code:
            assert "ASMSTART" in body[i - 1], f"compiler-placed vmcnt wait between the chunk bodies: {code.strip()}"


@pytest.mark.parametrize("nw", [4, 8])
def test_llama_dma_attention_keeps_its_dma_in_flight(isa, nw):
    """attn_causal128_dma_kernel: the next 64-key chunk (K and V, two 64-column images each) travels by LDS-DMA while the current one
    is computed; no compiler-placed vmcnt wait between the chunk bodies, no spills, two workgroups per CU (<= 256 VGPRs, 64 KiB)."""
    body = kernel_body(isa, f"_Z25attn_causal128_dma_kernelILi{nw}EEv14AttnCausalArgs")
    assert not any("scratch_" in l for l in body), "Llama DMA attention kernel spills"
    m = re.search(r"NumVgprs: (\d+)", "\n".join(isa[isa.index(body[-1]):isa.index(body[-1]) + 400]))
    assert m and int(m.group(1)) <= 256, m and m.group(1)
    assert sum("global_load_lds_dwordx4" in l for l in body) == 2 * 32 // nw         # chunk 0 + the loop's next chunk, 32 / NW 1-KiB pieces per wave
    assert sum("v_mfma_f32_32x32x16_f16" in l for l in body) == 2 * 32            # two chunk bodies (diagonal / below), 16 + 16 MFMAs each
    assert sum("ds_read_b64_tr_b16" in l for l in body) == 2 * 2 * 16
    first_mfma = next(i for i, l in enumerate(body) if "v_mfma" in l)
    last_mfma = max(i for i, l in enumerate(body) if "v_mfma" in l)
    for i in range(first_mfma, last_mfma):
        code = body[i].split(";")[0]
        if "vmcnt" in code:
            assert "ASMSTART" in body[i - 1], f"compiler-placed vmcnt wait between the chunk bodies: {code.strip()}"
