"""Per-kernel ISA statistics of a compiled translation unit (no GPU needed).

    python tools/isa_stats.py build/obj/<unit>.o [kernel-name-substring] [--classes] [--dump]

Extracts the gfx950 code object from the object's .hip_fatbin, prints for every kernel (or those
whose demangled name contains the substring) the register / LDS / scratch figures of the kernel
descriptor's metadata and a STATIC count of its instructions by opcode (straight-line count of the
emitted code, loops counted once).  --classes groups the opcodes the way DESIGN.md's instruction
accounts do; --dump writes the disassembly of the selected kernels to stdout.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"

CLASSES = [
    ("v_mad_u64_u32", "multiply-add 32x32+64"),
    ("v_mul_lo_u32", "multiply (low)"), ("v_mul_hi_u32", "multiply (high)"), ("v_mul_u32_u24", "multiply 24"),
    ("v_lshrrev_b64", "64-bit shift"), ("v_lshlrev_b64", "64-bit shift"), ("v_lshl_add_u64", "64-bit add"),
    ("v_add_co_u32", "carry add/sub"), ("v_addc_co_u32", "carry add/sub"), ("v_sub_co_u32", "carry add/sub"),
    ("v_subb_co_u32", "carry add/sub"), ("v_subbrev_co_u32", "carry add/sub"), ("v_subrev_co_u32", "carry add/sub"),
    ("v_and_b32", "mask / logic"), ("v_or_b32", "mask / logic"), ("v_xor_b32", "mask / logic"), ("v_not_b32", "mask / logic"),
    ("v_and_or_b32", "mask / logic"), ("v_bfe_u32", "mask / logic"), ("v_bfi_b32", "mask / logic"), ("v_or3_b32", "mask / logic"),
    ("v_lshrrev_b32", "32-bit shift"), ("v_lshlrev_b32", "32-bit shift"), ("v_alignbit_b32", "32-bit shift"),
    ("v_lshl_or_b32", "32-bit shift"), ("v_lshl_add_u32", "32-bit shift"), ("v_ashrrev_i32", "32-bit shift"),
    ("v_add_u32", "32-bit add/sub"), ("v_sub_u32", "32-bit add/sub"), ("v_subrev_u32", "32-bit add/sub"),
    ("v_add3_u32", "32-bit add/sub"), ("v_add_lshl_u32", "32-bit add/sub"),
    ("v_mov_b32", "move"), ("v_accvgpr_write_b32", "move"), ("v_accvgpr_read_b32", "move"), ("v_pk_mov_b32", "move"),
    ("v_mov_b64", "move"),
    ("v_cndmask_b32", "select"), ("v_cmp", "compare"), ("v_readfirstlane_b32", "lane op"), ("v_readlane_b32", "lane op"),
]


def klass(op):
    for prefix, name in CLASSES:
        if op.startswith(prefix):
            return name
    if op.startswith("v_"):
        return "other VALU"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_"):
        return "SALU / branch"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM (" + ("scratch" if op.startswith("scratch_") else "global") + ")"
    return "other"


def code_object(path):
    tmp = tempfile.mkdtemp(prefix="isa_")
    if open(path, "rb").read(4) == b"\x7fELF":
        sections = subprocess.run([LLVM + "/llvm-readelf", "-S", path], capture_output=True, text=True).stdout
        if ".hip_fatbin" not in sections:
            return path                                         # already a device code object
        fat = os.path.join(tmp, "fat.bin")
        subprocess.check_call([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, path])
    else:
        fat = path
    co = os.path.join(tmp, "dev.co")
    subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    return co


def metadata(co):
    """{kernel symbol: {field: value}} from the amdhsa.kernels metadata note (one '- .agpr_count' list item per kernel)"""
    txt = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    out, cur = {}, None
    keep = ("agpr_count", "vgpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size",
            "vgpr_spill_count", "sgpr_spill_count", "max_flat_workgroup_size", "symbol")
    for line in txt.splitlines():
        m = re.match(r"^(\s+)(- )?\.(\w+):\s+(.*)", line)
        if not m:
            continue
        indent, item, k, v = len(m.group(1)), m.group(2), m.group(3), m.group(4).strip().strip("'")
        if item and indent == 2:                                # a new kernel record (top-level list item)
            if cur and "symbol" in cur:
                out[cur["symbol"].replace(".kd", "")] = cur
            cur = {}
        if cur is not None and indent <= 4 and k in keep:
            cur[k] = v
    if cur and "symbol" in cur:
        out[cur["symbol"].replace(".kd", "")] = cur
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    path = args[0]
    want = args[1] if len(args) > 1 else ""
    co = code_object(path)
    meta = metadata(co)
    dis = subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
    kernels, cur = collections.OrderedDict(), None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = m.group(1); kernels[cur] = []; continue
        if cur is None:
            continue
        m = re.match(r"^\s+(\w+)\b(.*?)(?://.*)?$", line)
        if m and not line.lstrip().startswith("//"):
            kernels[cur].append((m.group(1), m.group(2)))
    for sym, ins in kernels.items():
        dem = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
        if want and want not in dem and want not in sym:
            continue
        md = meta.get(sym, {})
        print("== %s" % dem[:200])
        if md:
            print("   vgpr %s  agpr %s  sgpr %s  lds %s B  scratch %s B  vgpr spills %s" % (
                md.get("vgpr_count"), md.get("agpr_count"), md.get("sgpr_count"), md.get("group_segment_fixed_size"),
                md.get("private_segment_fixed_size"), md.get("vgpr_spill_count")))
        if "--dump" in sys.argv:
            for op, rest in ins:
                print("      %s%s" % (op, rest))
            continue
        cnt = collections.Counter(op for op, _ in ins)
        valu = sum(v for k, v in cnt.items() if k.startswith("v_"))
        print("   %d instructions, %d VALU" % (len(ins), valu))
        if "--classes" in sys.argv:
            cc = collections.Counter()
            for op, v in cnt.items():
                cc[klass(op)] += v
            for k, v in cc.most_common():
                print("   %8d  %s" % (v, k))
        else:
            for k, v in cnt.most_common(40):
                print("   %8d  %s" % (v, k))


if __name__ == "__main__":
    main()
