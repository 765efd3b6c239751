"""Static audit of the gfx950 ISA for ONE hazard class: a VGPR that is the destination of an LDS read still in flight is touched
(read, copied, overwritten) before an `s_waitcnt lgkmcnt(n)` has retired that read.

Why: the tap / attention kernels issue their fragment reads as inline `ds_read_b128` with hand-counted waits (conv_taps_il.hip, attention_split.hip).
hipcc treats an asm statement's VGPR destination as written at ;;#ASMEND: under register pressure it may place a `v_mov` of that register (a PHI copy on
a loop back-edge, a live-range split) between the read and the wait that covers it.  Nothing interlocks a VGPR read against an outstanding LDS
return on gfx9: the copy takes the OLD register contents whenever the LDS is slower than the distance to the copy — correct in a lone process,
wrong in a few launches per hundred when other work shares the CU (NOTEBOOK §12.7b / §13).  A passing bit-equality test is no evidence; this is.

Input: the device assembly of a translation unit (`hipcc -S --offload-device-only --offload-arch=gfx950 -O3 ...`), or a source file to compile.
Model: per kernel a CFG of basic blocks; the state is the in-order queue of outstanding LGKM operations (LDS reads with their destination
registers, LDS writes / SMEM loads without).  `s_waitcnt lgkmcnt(n)` retires every LDS operation but the n youngest LDS operations (LDS returns in
order; SMEM entries only make the wait stricter) and everything when n = 0.  Every (block, state) pair is walked once.

  python tools/asm_hazard_audit.py fgt_amd/csrc/conv_taps_il.hip [-DFGT_IL_A_EARLY=1 ...]     # compiles to a temp .s first
  python tools/asm_hazard_audit.py file.s
  python tools/asm_hazard_audit.py fgt_amd/lib/obj/conv_taps_il.o                               # the built object: disassembled, seconds
Exit code 1 when a hazard is found.  tests/test_build_resources.py runs it over the kernels with asm reads."""
import os
import re
import subprocess
import sys
import tempfile

REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")
LGKM = re.compile(r"lgkmcnt\((\d+)\)")
LABEL = re.compile(r"^([.\w$]+):")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def parse_kernels(asm):
    """-> {kernel: [(label or None, mnemonic, operands, lineno)]} for every .amdhsa kernel of the file."""
    names = set(re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", asm, re.M))
    kernels, cur, pending_label = {}, None, None
    for no, raw in enumerate(asm.splitlines(), 1):
        line = raw.split(";")[0].rstrip() if not raw.lstrip().startswith(";") else ""
        if not line.strip():
            continue
        m = LABEL.match(line)
        if m:
            lab = m.group(1)
            if lab in names:
                cur = kernels.setdefault(lab, [])
                pending_label = None
            elif cur is not None:
                pending_label = lab if pending_label is None else pending_label
                cur.append((lab, "<label>", "", no))
            continue
        if cur is None or line.lstrip().startswith("."):
            if line.strip().startswith(".Lfunc_end") or line.strip().startswith(".size"):
                cur = None
            continue
        parts = line.strip().split(None, 1)
        cur.append((None, parts[0], parts[1] if len(parts) > 1 else "", no))
        if parts[0] == "s_endpgm":
            pass
    return kernels


OBJ_SYM = re.compile(r"^[0-9a-f]+ <([^>]+)>:")


def parse_objdump(dis):
    """The same structure from `llvm-objdump -d --symbolize-operands` of a code object: symbols `addr <name>:`, local labels `addr <L12>:`,
    instructions followed by `// addr: encoding`."""
    kernels, cur = {}, None
    for no, raw in enumerate(dis.splitlines(), 1):
        m = OBJ_SYM.match(raw)
        if m:
            lab = m.group(1)
            if re.fullmatch(r"L\d+", lab):
                if cur is not None:
                    cur.append((lab, "<label>", "", no))
            else:
                cur = kernels.setdefault(lab, [])
            continue
        if cur is None or not raw.startswith("\t"):
            continue
        line = raw.split("//")[0].strip()
        if not line:
            continue
        parts = line.split(None, 1)
        cur.append((None, parts[0], parts[1] if len(parts) > 1 else "", no))
    return {k: v for k, v in kernels.items() if any(mn == "s_endpgm" for _, mn, _, _ in v)}


def disassemble_object(obj):
    """Fat object (hipcc -c) -> text disassembly of its gfx950 code object."""
    llvm = "/opt/rocm/lib/llvm/bin/"
    d = tempfile.mkdtemp()
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
    subprocess.run([llvm + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
    subprocess.run([llvm + "clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co, "--unbundle"], check=True)
    out = subprocess.run([llvm + "llvm-objdump", "-d", "--symbolize-operands", co], capture_output=True, text=True, check=True).stdout
    for f in (fat, co):
        os.remove(f)
    os.rmdir(d)
    return out


def build_blocks(ins):
    """Basic blocks: [(start, end)] over the instruction list, label -> block index, successors per block."""
    starts = {0}
    label_at = {}
    for i, (lab, mn, ops, _) in enumerate(ins):
        if mn == "<label>":
            label_at[lab] = i
            starts.add(i)
        elif mn.startswith("s_cbranch") or mn == "s_branch" or mn == "s_endpgm" or mn.startswith("s_setpc"):
            starts.add(i + 1)
    order = sorted(s for s in starts if s < len(ins))
    blocks = [(s, order[k + 1] if k + 1 < len(order) else len(ins)) for k, s in enumerate(order)]
    index_of = {s: k for k, (s, _) in enumerate(blocks)}
    succ = []
    for k, (s, e) in enumerate(blocks):
        last = None
        for i in range(e - 1, s - 1, -1):
            if ins[i][1] != "<label>":
                last = ins[i]
                break
        nxt = [k + 1] if k + 1 < len(blocks) else []
        if last is None:
            succ.append(nxt)
        elif last[1] == "s_endpgm" or last[1].startswith("s_setpc"):
            succ.append([])
        elif last[1] == "s_branch":
            succ.append([index_of[label_at[last[2].strip()]]] if last[2].strip() in label_at else [])
        elif last[1].startswith("s_cbranch"):
            t = last[2].strip().split(",")[-1].strip()
            succ.append(nxt + ([index_of[label_at[t]]] if t in label_at else []))
        else:
            succ.append(nxt)
    return blocks, succ


def returns_data(mn):
    return (mn.startswith("ds_read") or mn.startswith("ds_load") or "permute" in mn or mn.startswith("ds_swizzle") or "_rtn" in mn
            or mn.startswith("ds_consume") or mn.startswith("ds_append"))


def audit_kernel(name, ins, max_states=200000):
    blocks, succ = build_blocks(ins)
    hazards = {}
    seen = set()
    stack = [(0, ())]
    n_asm_reads = 0
    while stack:
        b, state = stack.pop()
        if (b, state) in seen:
            continue
        seen.add((b, state))
        if len(seen) > max_states:
            raise RuntimeError(f"{name}: state explosion")
        q = list(state)                       # entries: (kind, frozenset(regs), lineno)
        s, e = blocks[b]
        for i in range(s, e):
            _, mn, ops, no = ins[i]
            if mn == "<label>":
                continue
            touched = regs_of(ops)
            if touched:
                for kind, regs, rno in q:
                    if regs and (regs & touched) and rno != no:
                        hazards.setdefault((rno, no), (ins_text(ins, rno), f"{mn} {ops}"))
            if mn.startswith("ds_"):
                dst = frozenset(regs_of(ops.split(",")[0])) if returns_data(mn) else frozenset()
                q.append(("lds", dst, no))
            elif mn.startswith("s_load") or mn.startswith("s_buffer_load") or mn.startswith("s_scratch_load"):
                q.append(("smem", frozenset(), no))
            elif mn == "s_waitcnt":
                m = LGKM.search(ops)
                n = None
                if m:
                    n = int(m.group(1))
                elif re.fullmatch(r"\s*(0x)?[0-9a-fA-F]+\s*", ops or ""):
                    v = int(ops.strip(), 0)
                    n = (v >> 8) & 0xF
                if n is not None:
                    if n == 0:
                        q = []
                    else:
                        lds_idx = [k for k, ent in enumerate(q) if ent[0] == "lds"]
                        retire = set(lds_idx[:-n]) if len(lds_idx) > n else set()
                        q = [ent for k, ent in enumerate(q) if k not in retire]
            elif mn == "s_endpgm":
                q = []
        st = tuple(q)
        for nb in succ[b]:
            stack.append((nb, st))
    return hazards, len(seen)


def ins_text(ins, lineno):
    for _, mn, ops, no in ins:
        if no == lineno:
            return f"{mn} {ops}"
    return "?"


def compile_to_asm(src, flags):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    here = os.path.dirname(os.path.abspath(__file__))
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--offload-device-only", "-Wno-unused-result",
           "-I", os.path.join(here, "..", "include"), src, "-o", out] + list(flags)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stderr[-4000:])
    return out


def audit_file(path, flags=(), only=None, verbose=True):
    if path.endswith(".o"):
        asm_path = path
        kernels = parse_objdump(disassemble_object(path))
    else:
        asm_path = path if path.endswith(".s") else compile_to_asm(path, flags)
        kernels = parse_kernels(open(asm_path).read())
    total = 0
    report = {}
    for name, ins in kernels.items():
        if only and only not in name:
            continue
        n_ds_asm = sum(1 for _, mn, _, _ in ins if mn.startswith("ds_read"))
        hz, nstates = audit_kernel(name, ins)
        report[name] = hz
        total += len(hz)
        if verbose:
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
            print(f"{'HAZARD' if hz else 'ok    '} {len(ins):6d} instructions, {n_ds_asm:4d} ds_read, {nstates:5d} (block, state) pairs: {dem[:150]}")
            for (rno, no), (rd, use) in sorted(hz.items())[:12]:
                print(f"        line {rno}: {rd}   is still in flight at line {no}: {use}")
    return total, report, asm_path


if __name__ == "__main__":
    args = sys.argv[1:]
    flags = [a for a in args[1:] if a.startswith("-")]
    only = next((a.split("=", 1)[1] for a in args if a.startswith("--only=")), None)
    flags = [f for f in flags if not f.startswith("--only=")]
    total, _, asm_path = audit_file(args[0], flags, only)
    print(f"{total} hazard site(s); assembly: {asm_path}")
    sys.exit(1 if total else 0)
