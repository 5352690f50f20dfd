#!/usr/bin/env python3
"""ISA lint of libldt_hip.so — run by build.sh after the link; a failure fails the build.

The hot kernels carry instruction streams whose correctness rests on things hipcc neither checks nor promises to keep across a
toolchain bump: hand-counted `s_waitcnt vmcnt(N)` immediates (VMEM retires in order: N = the number of younger requests), asm-issued
loads hipcc does not count (their destination VGPRs must stay untouched until the hand-written wait), LDS data returning into
registers an MFMA in flight still reads as its C operand (profiles/r04_mid_epilogue_hazard.txt).  This tool disassembles the gfx950
code objects embedded in the library (`.hip_fatbin` -> clang offload bundles -> ELF -> llvm-objdump -d) and checks, per kernel:

  R1  scratch / register spills (`.private_segment_fixed_size`, `.vgpr_spill_count`, `.sgpr_spill_count` of the code object's metadata)
      of every kernel of the GUARDED families: none in a kernel with asm-issued register loads (a spill / reload beside them moves
      registers whose data has not arrived), and elsewhere no more than the audited amount recorded in isa_signatures.json ("scratch":
      a few multi-tile 256-tile forms spill 3-20 VGPRs around their epilogues).  (Rounds 4-5 read these keys without their leading dot,
      so the rule never fired: round 6 found 13 spilled VGPRs in the headline's dominant kernel that way and removed them.)
  R2  the VMEM / wait FINGERPRINT of every guarded kernel — the ordered sequence of [load, LDS-DMA load, store, `s_waitcnt vmcnt(N)`
      with its immediate, s_barrier, branch] events — equals the audited one in isa_signatures.json.  The counted waits were derived
      from exactly that sequence; any change (a compiler that reorders, merges, splits or adds a VMEM op or a wait) must be re-audited
      by a human: read the diff printed here against the kernel's comments, then `python isa_lint.py --update libldt_hip.so`.
  R3  asm-load safety, the kernels of ASM_LOADS (register loads issued by `asm volatile`): between such a load and the first `s_waitcnt vmcnt(N)` that covers it (N <= the
      number of VMEM ops issued after it on every path), no instruction names one of its destination registers (a copy, spill or reuse
      of a register whose data is still in flight).  Walks the control-flow graph, loops included (check_asm_load_safety).
  R4  MFMA-C hazard, the mid-tile GEMM family: the fence between the main loop and the epilogue (4 x `s_nop 15` between two scheduling
      barriers: LDS data returning into registers that MFMAs in flight still name, profiles/r04_mid_epilogue_hazard.txt) is present and
      sits where the source put it: directly downstream of the main loop's last MFMA, no LDS read in between.

Usage:  isa_lint.py <libldt_hip.so>            check (exit 1 on a violation)
        isa_lint.py --update <libldt_hip.so>   rewrite isa_signatures.json from the present binary (after a human audit)
        isa_lint.py --dump REGEX <lib>         print the fingerprints of the kernels matching REGEX
Reference for what the kernels compute: model/layers.py:110-133,183-229 (the GEMMs and attention of the Score block)."""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
LLVM = os.environ.get("LDT_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
SIG_FILE = os.path.join(HERE, "isa_signatures.json")

# kernel families with hand-counted waits / asm-issued loads (demangled-name regexes)
GUARDED = [r"gemm_bf16_nt_256f_kernel<", r"gemm_qkv_attn256_kernel<", r"gemm_bf16_nt_mid_kernel<", r"attn_fwd_head_kernel<",
           r"gemm_bf16_nt_kernel<", r"attn_fwd_kernel<", r"attn_fwd_resident_kernel<", r"attn_oproj_resident_kernel<"]
MFMA_C_FAMILIES = [r"gemm_bf16_nt_mid_kernel<"]
# register-destination loads issued by `asm volatile` (hipcc neither counts nor waits for them): (kernel regex, mnemonic, source site)
ASM_LOADS = [(r"attn_fwd_head_kernel<", "global_load_dwordx4", "attention.hip: the Q fragments"),
             (r"gemm_bf16_nt_256f_kernel<\d+, \d+, 1, 1, \d+>", "global_load_dwordx4", "gemm_bf16.hip WREG: the W fragment half-sets"),
             (r"gemm_bf16_nt_mid_kernel<\d+, \d+, \d+, 2, \d+>", "global_load_dwordx2", "gemm_mid.hip mid_loader: the row-statistics partials")]


def code_objects(lib):
    """gfx950 ELF images embedded in `lib` (one clang offload bundle per linked object file)."""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib], check=True)
        data = open(fat, "rb").read()
    magic, out, pos = b"__CLANG_OFFLOAD_BUNDLE__", [], 0
    while True:
        i = data.find(magic, pos)
        if i < 0:
            break
        n = struct.unpack_from("<Q", data, i + 24)[0]
        off = i + 32
        for _ in range(n):
            o, s, ts = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + ts].decode()
            off += ts
            if "gfx950" in triple and s:
                out.append(data[i + o:i + o + s])
        pos = i + 24
    return out


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
    return dict(zip(names, r.stdout.split("\n")))


def kernel_meta(elf_path):
    """{mangled kernel name: its `amdhsa.kernels` metadata entry} from the code object's note (YAML)."""
    import yaml
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", elf_path], capture_output=True, text=True, check=True).stdout
    m = re.search(r"^\s*---\s*$(.*?)^\s*\.\.\.\s*$", txt, flags=re.S | re.M)
    if not m:
        return {}
    doc = yaml.safe_load(m.group(1))
    return {k[".name"]: k for k in doc.get("amdhsa.kernels", [])}


REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), r) for r in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


class Ins:
    __slots__ = ("op", "text", "addr", "label", "target")

    def __init__(self, op, text, addr, label):
        self.op, self.text, self.addr, self.label, self.target = op, text, addr, label, None


def disassemble(elf_bytes):
    """-> ({mangled: [Ins]}, {mangled: metadata entry}); Ins.label marks branch targets (the start of a straight-line region)"""
    with tempfile.NamedTemporaryFile(suffix=".elf", delete=False) as f:
        f.write(elf_bytes)
        path = f.name
    try:
        txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", path], capture_output=True, text=True, check=True).stdout
        meta = kernel_meta(path)
    finally:
        os.unlink(path)
    kernels, cur, start, targets = {}, None, {}, {}
    for line in txt.split("\n"):
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", line)
        if m:
            cur = kernels.setdefault(m.group(2), [])
            start[m.group(2)] = int(m.group(1), 16)
            targets[m.group(2)] = set()
            cur_name = m.group(2)
            continue
        if cur is None or not line.strip():
            continue
        body = line.split("//")[0].strip()
        if not body:
            continue
        am = re.search(r"//\s*([0-9A-Fa-f]+):", line)
        op = body.split()[0]
        cur.append(Ins(op, body, int(am.group(1), 16) if am else -1, False))
        if op.startswith("s_cbranch") or op == "s_branch":
            tm = re.search(r"<.+\+0x([0-9a-fA-F]+)>\s*$", line)
            if tm:
                cur[-1].target = start[cur_name] + int(tm.group(1), 16)
                targets[cur_name].add(cur[-1].target)
    for name, ins_list in kernels.items():
        for i in ins_list:
            if i.addr in targets[name]:
                i.label = True
    return kernels, meta


def is_vmem(op):
    return op.startswith(("global_load", "global_store", "global_atomic", "buffer_load", "buffer_store", "buffer_atomic", "flat_load",
                          "flat_store", "flat_atomic", "scratch_load", "scratch_store"))


def is_lds_dma(ins):
    return "_lds_" in ins.op or ins.text.rstrip().endswith(" lds")


def vmcnt_of(ins):
    if ins.op != "s_waitcnt":
        return None
    m = re.search(r"vmcnt\((\d+)\)", ins.text)
    return int(m.group(1)) if m else None


def fingerprint(ins_list):
    ev = []
    for i in ins_list:
        if is_vmem(i.op):
            ev.append("D" if is_lds_dma(i) else ("L" if "load" in i.op else ("A" if "atomic" in i.op else "S")))
        elif i.op == "s_waitcnt":
            n = vmcnt_of(i)
            if n is not None:
                ev.append("W%d" % n)
        elif i.op == "s_barrier":
            ev.append("|")
        elif i.op.startswith("s_cbranch") or i.op == "s_branch":
            ev.append("b")
    return " ".join(ev)


def check_asm_load_safety(name, ins_list, tally=None, mnemonic=None):
    """R3: destination registers of a register load stay unnamed until a wait that covers it, on EVERY path.  The instruction list is
    walked as a control-flow graph (fall-through + branch targets, back edges included): a state is (instruction, number of VMEM ops
    issued behind the load so far); `s_waitcnt vmcnt(N)` with N <= that number ends a path as covered; any instruction that names a
    destination register before that is a violation (a copy, spill or reuse of a register whose data is still in flight).  The count only
    grows along a path and a smaller count is the harder case, so an instruction is revisited only with a smaller count than before:
    loops terminate.  Branch conditions are not correlated (every successor of a conditional branch is followed): conservative.
    `tally` counts the loads by verdict: "covered" (on every path), "open" (some path reaches the end of the kernel without a covering
    wait: an error for the asm-issued families), "violation"."""
    errs = []
    n = len(ins_list)
    tally = tally if tally is not None else {}
    idx = {ins.addr: k for k, ins in enumerate(ins_list)}
    for k, ins in enumerate(ins_list):
        if not (is_vmem(ins.op) and "load" in ins.op) or is_lds_dma(ins) or (mnemonic is not None and ins.op != mnemonic):
            continue
        first = ins.text.split(None, 1)[1].split(",")[0]
        dst = regs_of(first)
        if not dst:
            continue
        best = {}                                           # instruction -> smallest count it was entered with
        stack = [(k + 1, 0)]
        verdict = "covered"
        while stack and verdict != "violation":
            j, after = stack.pop()
            while True:
                if j >= n:
                    verdict = "open"
                    break
                if best.get(j, 1 << 30) <= after:
                    break
                best[j] = after
                x = ins_list[j]
                if x.op.startswith(("s_endpgm", "s_setpc")):
                    verdict = "open" if verdict == "covered" else verdict
                    break
                w = vmcnt_of(x)
                if w is not None and w <= after:
                    break                                   # covered on this path
                if is_vmem(x.op):
                    # a younger LOAD into the same registers is ordered behind this one by the hardware (VMEM returns in order; exec-masked
                    # if / else arms load disjoint lanes of one register): only its address operands are checked
                    named = regs_of(x.text.split(",", 1)[1] if ("load" in x.op and "," in x.text) else x.text)
                    if named & dst:
                        errs.append("%s: `%s` names a register of `%s` before the wait that covers the load" % (name, x.text, ins.text))
                        verdict = "violation"
                        break
                    after += 1
                elif x.op.startswith(("s_cbranch", "s_branch")):
                    t = idx.get(x.target) if x.target is not None else None
                    if t is None:
                        errs.append("%s: R3: branch `%s` behind `%s` has no resolvable target" % (name, x.text, ins.text))
                        verdict = "violation"
                        break
                    if x.op == "s_branch":
                        j = t
                        continue
                    stack.append((t, after))                # conditional: the taken arm later, the fall-through now
                elif x.op not in ("s_nop", "s_waitcnt", "s_barrier", "s_sleep", "s_setprio") and regs_of(x.text) & dst:
                    errs.append("%s: `%s` names a register of `%s` before the wait that covers the load" % (name, x.text, ins.text))
                    verdict = "violation"
                    break
                j += 1
        tally[verdict] = tally.get(verdict, 0) + 1
        if verdict == "open" and mnemonic is not None:
            errs.append("%s: R3: a path behind the asm-issued `%s` reaches the end of the kernel without a covering `s_waitcnt vmcnt`" % (name, ins.text))
    return errs


def check_mfma_c_hazard(name, ins_list):
    """R4: the main-loop -> epilogue fence of the mid-tile kernels is in the binary where the source put it: a run of >= 4 consecutive
    `s_nop 15` (64 wait states) whose nearest matrix / LDS-read neighbour upstream is an MFMA (no LDS read has slipped in between the last
    MFMA of the main loop and the fence) and downstream of which the first LDS read or MFMA follows the run."""
    errs, runs, k, n = [], [], 0, len(ins_list)
    while k < n:
        if ins_list[k].text.split() == ["s_nop", "15"]:
            j = k
            while j < n and ins_list[j].text.split() == ["s_nop", "15"]:
                j += 1
            if j - k >= 4:
                runs.append((k, j))
            k = j
        else:
            k += 1
    if not runs:
        return ["%s: R4: no run of 4 x `s_nop 15` found (the main-loop -> epilogue fence of gemm_mid.hip is gone)" % name]
    for (k, j) in runs:
        up = next((ins_list[i] for i in range(k - 1, max(-1, k - 200), -1) if ins_list[i].op.startswith(("v_mfma", "ds_read"))), None)
        if up is None or not up.op.startswith("v_mfma"):
            errs.append("%s: R4: the instruction stream above the `s_nop 15` fence ends in `%s`, not in the main loop's last MFMA"
                        % (name, up.text if up else "<nothing within 200 instructions>"))
    return errs


def toolchain_id():
    """The compiler the fingerprints depend on: first line of `clang --version` of the LLVM the library was built with."""
    try:
        out = subprocess.run([os.path.join(LLVM, "clang"), "--version"], capture_output=True, text=True, check=True).stdout
        return out.strip().splitlines()[0]
    except (OSError, subprocess.CalledProcessError, IndexError):
        return None


def load_gold():
    try:
        return json.load(open(SIG_FILE))
    except (OSError, ValueError):
        return None


def analyse(lib):
    fps, errs, nk = {}, [], 0
    analyse.tally = {}
    analyse.scratch = {}
    gold_doc = load_gold() or {}
    analyse.audited_scratch = gold_doc.get("scratch", {})
    for elf in code_objects(lib):
        kernels, meta = disassemble(elf)
        names = demangle(list(kernels))
        for mangled, ins_list in kernels.items():
            name = names[mangled]
            nk += 1
            guarded = any(re.search(p, name) for p in GUARDED)
            if guarded:
                m = meta.get(mangled)
                if m is None:
                    errs.append("%s: no metadata entry found (R1 cannot be checked)" % name)
                else:
                    has_asm_loads = any(re.search(pat, name) for pat, _m, _s in ASM_LOADS)
                    used = {key: int(m.get("." + key, 0)) for key in ("private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count")}
                    analyse.scratch[name] = used["private_segment_fixed_size"]
                    allowed = 0 if has_asm_loads else analyse.audited_scratch.get(name, 0)
                    if used["private_segment_fixed_size"] > allowed or (has_asm_loads and (used["vgpr_spill_count"] or used["sgpr_spill_count"])):
                        errs.append("%s: scratch %d B, %d VGPRs / %d SGPRs spilled (R1: audited %d B%s)" % (
                            name, used["private_segment_fixed_size"], used["vgpr_spill_count"], used["sgpr_spill_count"], allowed,
                            "; kernels with asm-issued register loads must not spill at all" if has_asm_loads else ""))
                fps[name] = fingerprint(ins_list)
            for pat, mnem, _site in ASM_LOADS:
                if re.search(pat, name):
                    t = {}
                    errs += check_asm_load_safety(name, ins_list, t, mnem)
                    if not t:
                        errs.append("%s: R3: no `%s` found (the asm-issued loads of this kernel family are gone or changed form: update ASM_LOADS)" % (name, mnem))
                    for k_, v_ in t.items():
                        analyse.tally[k_] = analyse.tally.get(k_, 0) + v_
            if any(re.search(p, name) for p in MFMA_C_FAMILIES):
                errs += check_mfma_c_hazard(name, ins_list)
    return fps, errs, nk


def main(argv):
    if len(argv) >= 2 and argv[0] == "--dump":
        fps, _, _ = analyse(argv[2])
        for k, v in sorted(fps.items()):
            if re.search(argv[1], k):
                print(k + "\n    " + v + "\n")
        return 0
    update = "--update" in argv
    lib = [a for a in argv if not a.startswith("--")][0]
    if not os.path.exists(os.path.join(LLVM, "llvm-objdump")):
        # without the disassembler nothing can be checked: say so loudly, do not fail a correct library's build
        print("isa_lint: WARNING: %s/llvm-objdump not found — the hand-counted kernels of %s were NOT checked (set LDT_LLVM_BIN)" % (LLVM, lib), file=sys.stderr)
        return 0
    fps, errs, nk = analyse(lib)
    warns = []
    tc = toolchain_id()
    if update:
        with open(SIG_FILE, "w") as f:
            json.dump({"what": "audited VMEM / wait fingerprints of the guarded kernels (isa_lint.py R2); L load, D LDS-DMA load, S store, A atomic, "
                               "Wn s_waitcnt vmcnt(n), | s_barrier, b branch; `scratch`: audited scratch bytes of the guarded kernels that spill (R1); "
                               "`toolchain`: the compiler these were audited on",
                       "toolchain": tc, "scratch": {k: v for k, v in sorted(analyse.scratch.items()) if v}, "kernels": fps}, f, indent=1, sort_keys=True)
        print("isa_lint: wrote %d fingerprints to %s (toolchain: %s)" % (len(fps), SIG_FILE, tc))
    else:
        doc = load_gold()
        gold = doc.get("kernels") if doc else None
        if gold is None:
            errs.append("R2: %s is missing or unreadable (run isa_lint.py --update after auditing the kernels)" % SIG_FILE)
        else:
            # the fingerprints are a function of the compiler's schedule: on ANOTHER toolchain than the audited one a difference is a
            # prompt to re-audit (warning), not a broken build — the hazard rules R1 / R3 / R4 are checked on the binary itself and stay fatal
            same_tc = doc.get("toolchain") is None or tc is None or doc.get("toolchain") == tc
            sink = errs if same_tc else warns
            for k in sorted(set(gold) | set(fps)):
                if k not in fps:
                    sink.append("R2: audited kernel %s is no longer in the library (re-audit, then --update)" % k)
                elif k not in gold:
                    sink.append("R2: guarded kernel %s has no audited fingerprint (audit it, then --update)" % k)
                elif gold[k] != fps[k]:
                    a, b = gold[k].split(), fps[k].split()
                    d = next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), min(len(a), len(b)))
                    sink.append("R2: VMEM / wait fingerprint of %s changed at event %d: audited `... %s` now `... %s` (re-audit the hand-counted "
                                "waits of this kernel, then --update)" % (k, d, " ".join(a[max(0, d - 6):d + 6]), " ".join(b[max(0, d - 6):d + 6])))
            if warns:
                print("isa_lint: WARNING: built with `%s`, fingerprints audited on `%s`: %d fingerprint difference(s) NOT treated as errors.  Re-audit: "
                      "`python3 isa_lint.py --dump REGEX lib` against the kernels' comments, run the -m gpu suite (bit-equality and soak tests), then "
                      "`python3 isa_lint.py --update lib`." % (tc, doc.get("toolchain"), len(warns)), file=sys.stderr)
                for w in warns[:10]:
                    print("  " + w, file=sys.stderr)
    if errs:
        print("isa_lint: %d violation(s) in %s" % (len(errs), lib), file=sys.stderr)
        for e in errs[:60]:
            print("  " + e, file=sys.stderr)
        return 1
    t = analyse.tally
    print("isa_lint: %d kernels checked, %d guarded fingerprints match, %d asm-issued register loads covered by their hand-written wait with no "
          "destination named before it; no violations" % (nk, len(fps), t.get("covered", 0)))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
