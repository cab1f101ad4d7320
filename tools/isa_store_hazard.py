"""Scan the gfx950 code objects of esvit_amd/csrc/build/*.o for a 96/128-bit VMEM store whose data VGPRs are overwritten within the next few
instructions (measured on MI355X, tools/probe/diag_attn64b.py: with every CU fully occupied the head_dim-64 attention forward stored the
constant a `v_mov_b32` wrote into the first data register of a `buffer_store_dwordx4 ... offen` with an SGPR offset, issued right before it --
the case LLVM's hazard recogniser exempts from the "VMEM store > 64 bits, then write of its data VGPRs" wait state).

    python tools/isa_store_hazard.py [window=6] [object name filter]

Also lists (and tests/test_isa_cpu.py asserts there are none) the 96/128-bit MUBUF stores that carry an SGPR offset at all: common.h's
buffer_store_b128 keeps that field zero, which puts the store under the compiler's own wait-state rule."""
import glob, os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"
window = int(sys.argv[1]) if len(sys.argv) > 1 else 6
flt = sys.argv[2] if len(sys.argv) > 2 else ""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


total = 0
sgpr_offset_stores = []
for obj in sorted(glob.glob(os.path.join(root, "esvit_amd/csrc/build/*.o"))):
    if flt and flt not in obj:
        continue
    with tempfile.TemporaryDirectory() as t:
        if subprocess.run([LLVM + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, t + "/fat.bin"]).returncode:
            continue
        if subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + t + "/fat.bin",
                           "--output=" + t + "/k.co"], stderr=subprocess.DEVNULL).returncode:
            continue
        asm = subprocess.run([LLVM + "llvm-objdump", "-d", t + "/k.co"], capture_output=True, text=True).stdout.splitlines()
    kern, ins = None, []
    found = {}
    for line in asm + ["0 <end>:"]:
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            for i, (op, args) in enumerate(ins):
                if not re.match(r"(buffer|global|flat|scratch)_store_dwordx[34]", op):
                    continue
                if op.startswith("buffer") and len(args) > 3 and re.match(r"(s\d+|m0|vcc_lo|vcc_hi|ttmp\d+)\b", args[3]):
                    sgpr_offset_stores.append((os.path.basename(obj), kern, op + " " + ", ".join(args)))
                data = regs(args[1] if op.startswith(("global", "flat", "scratch")) and len(args) > 1 else args[0])
                if op.startswith("buffer"):
                    data = regs(args[0])
                for j in range(i + 1, min(len(ins), i + 1 + window)):
                    op2, a2 = ins[j]
                    if op2.startswith(("s_", "buffer_store", "global_store", "flat_store", "ds_write", "ds_store", "scratch_store")) or not a2:
                        if op2.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_barrier")):
                            break
                        continue
                    if regs(a2[0]) & data:
                        found.setdefault(kern, []).append((op, " ".join(args[:2]), j - i, op2 + " " + a2[0]))
                        break
            kern, ins = m.group(1), []
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//", line)
        if m:
            ins.append((m.group(1), [a.strip() for a in m.group(2).split(",")] if m.group(2) else []))
    for k, v in found.items():
        total += len(v)
        print("%s: %s" % (os.path.basename(obj), k[:110]))
        for f in v[:6]:
            print("     %s %s  <- +%d: %s" % f)
        if len(v) > 6:
            print("     ... %d more" % (len(v) - 6))
print("sites:", total)
print("96/128-bit buffer stores with an SGPR offset:", len(sgpr_offset_stores))
for o_, k_, i_ in sgpr_offset_stores[:20]:
    print("   ", o_, (k_ or "")[:90], i_)
