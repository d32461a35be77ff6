"""Static check of the built gfx950 code objects (called by build.sh after linking, and by tests/test_isa_lint.py).

  python dsf_amd/csrc/isa_lint.py dsf_amd/lib/libdsf_hip.so

Disassembles every device code object bundled in the shared library and fails on
  * scratch memory (`scratch_*` instructions or a non-zero private segment): a spill or a stack object in a kernel of this library
    is a performance bug (round 4: `cond ? *ptr : zero4` compiled into a select between pointers with the zero in scratch);
  * packed-FP32 arithmetic (`v_pk_fma_f32`, `v_pk_mul_f32`, `v_pk_add_f32`) outside ALLOW_PACKED_FP32: round 4 found the
    auto-vectorised MANO backward returning wrong bits in lanes 48-63 beside conv_x6 workgroups (DESIGN.md section 2,
    tools/platform/war_bisect.py), so the library is built with -fno-slp-vectorize -fno-vectorize and this check proves that
    the flags (or a later toolchain) really left no such instruction behind -- round 4's build claimed it and four kernels
    still had them, emitted by the loop vectoriser;
  * a write-after-read pair behind a packed-FP32 instruction if one is ever allow-listed (the narrowed trigger, see war_pairs()).
Everything here is text processing on `llvm-objdump -d` output; no GPU."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM_BIN = "/opt/rocm/lib/llvm/bin"
PACKED_FP32 = re.compile(r"\bv_pk_(fma|mul|add)_f32\b")
SCRATCH = re.compile(r"\bscratch_(load|store)\w*")
# kernel-name substrings that may contain packed-FP32 arithmetic (none today; an entry needs a beside-convolution GPU test)
ALLOW_PACKED_FP32 = ()


def disassemble(so_path):
    """{code object name: [(kernel symbol, [instruction text, ...]), ...]} of every gfx950 bundle in the library."""
    tmp = tempfile.mkdtemp(prefix="dsf_isa_")
    try:
        local = os.path.join(tmp, os.path.basename(so_path))
        shutil.copy(so_path, local)
        subprocess.check_call([os.path.join(LLVM_BIN, "llvm-objdump"), "--offloading", local], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out = {}
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            text = subprocess.check_output([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", "--no-show-raw-insn", os.path.join(tmp, f)]).decode()
            notes = subprocess.check_output([os.path.join(LLVM_BIN, "llvm-readelf"), "--notes", os.path.join(tmp, f)]).decode()
            kernels, cur = [], None
            for line in text.split("\n"):
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    cur = (m.group(1), [])
                    kernels.append(cur)
                elif cur is not None and line.startswith("\t"):
                    ins = line.split("//")[0].strip()
                    if ins:
                        cur[1].append(ins)
            out[f] = {"kernels": kernels, "notes": notes}
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _vregs(op):
    m = re.fullmatch(r"v(\d+)", op)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", op)
    return set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else set()


def _operands(ins):
    parts = ins.split(None, 1)
    if len(parts) == 1:
        return parts[0], []
    return parts[0], [o.strip().split()[0] for o in re.split(r",\s*(?![^\[]*\])", parts[1]) if o.strip()]


def war_pairs(instructions):
    """(packed instruction, overwriting instruction) pairs: a packed-FP32 op whose next vector instruction (waits and nops
    skipped) writes one of its source registers."""
    hits = []
    for i, ins in enumerate(instructions):
        if not PACKED_FP32.search(ins):
            continue
        _, ops = _operands(ins)
        src = set().union(*[_vregs(o) for o in ops[1:]]) if len(ops) > 1 else set()
        for nxt in instructions[i + 1:]:
            mn, nops = _operands(nxt)
            if mn in ("s_waitcnt", "s_nop"):
                continue
            if mn.startswith("v_") and nops and (_vregs(nops[0]) & src):
                hits.append((ins, nxt))
            break
    return hits


def lint(so_path):
    """list of violation strings (empty = clean) and a summary dict."""
    objs = disassemble(so_path)
    bad, n_kernels, n_ins = [], 0, 0
    for name, o in objs.items():
        for m in re.finditer(r"\.name:\s+(\S+)[^}]*?\.private_segment_fixed_size:\s+(\d+)", o["notes"], flags=re.S):
            if int(m.group(2)) != 0:
                bad.append("%s: kernel %s has a %s-byte private (scratch) segment" % (name, m.group(1), m.group(2)))
        for sym, ins in o["kernels"]:
            n_kernels += 1
            n_ins += len(ins)
            sc = [i for i in ins if SCRATCH.search(i)]
            if sc:
                bad.append("%s: %s uses scratch memory (%d instructions, first: %s)" % (name, sym, len(sc), sc[0]))
            pk = [i for i in ins if PACKED_FP32.search(i)]
            if pk and not any(a in sym for a in ALLOW_PACKED_FP32):
                bad.append("%s: %s contains %d packed-FP32 instructions (first: %s); build with -fno-slp-vectorize -fno-vectorize "
                           "or add it to ALLOW_PACKED_FP32 together with a beside-convolution GPU test" % (name, sym, len(pk), pk[0]))
            for a, b in war_pairs(ins):
                bad.append("%s: %s overwrites a source of a packed-FP32 instruction with the next vector instruction: %s -> %s" % (name, sym, a, b))
    return bad, {"code_objects": len(objs), "kernels": n_kernels, "instructions": n_ins}


if __name__ == "__main__":
    violations, summary = lint(sys.argv[1])
    print("isa_lint: %(code_objects)d code objects, %(kernels)d symbols, %(instructions)d instructions" % summary)
    for v in violations:
        print("isa_lint: " + v, file=sys.stderr)
    sys.exit(1 if violations else 0)
