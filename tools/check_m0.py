"""Build-time check behind the LDS-DMA helpers that leave M0 clobbered (csrc/kzv_common.h glds16_asm_*_m0, glds16_s in the GEMM files;
ADVICE r03): hipcc cannot be told about the clobber (m0 is a reserved register: a clobber entry is ignored with a warning), so the
invariant "a kernel whose inline asm writes M0 contains no compiler-generated use of M0" is checked on the ISA instead (a kernel is
either all-asm or all-builtin in its LDS-DMA: the builtin's M0 is the compiler's own business): every mention of m0 outside an inline-asm
block (;;#ASMSTART .. ;;#ASMEND) in a kernel that also mentions m0 INSIDE one is reported.
   python tools/check_m0.py [file.hip ...]      (default: every csrc/*.hip that contains a *_m0 / glds16_s helper call)
Exit code 1 if any compiler-generated M0 use is found."""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kuzushiji-vision_amd", "csrc")


def users():
    out = []
    for f in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
        s = open(f, encoding="utf-8").read()
        if re.search(r"glds16_asm_m0\(|glds16_asm_soff_m0\(|glds16_s\(", s):
            out.append(f)
    return out


def check(src):
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast", "-Wno-unused-function",
                        "--save-temps=obj", "-c", src, "-o", os.path.join(tmp, "x.o")], check=True, cwd=CSRC, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        asm = glob.glob(os.path.join(tmp, "*gfx950.s"))[0]
        outside, asm_m0, inside, kernel = [], set(), False, ""
        for n, line in enumerate(open(asm, encoding="utf-8", errors="replace"), 1):
            t = line.strip()
            if t.startswith(";;#ASMSTART"):
                inside = True
            elif t.startswith(";;#ASMEND"):
                inside = False
            elif re.match(r"^_Z\w+:", t):
                kernel = t.split(":")[0]
            elif not t.startswith((";", ".")) and re.search(r"\bm0\b", t.split(";")[0]):
                if inside:
                    asm_m0.add(kernel)
                else:
                    outside.append((kernel, n, t))
        return [b for b in outside if b[0] in asm_m0]


if __name__ == "__main__":
    files = [os.path.abspath(a) for a in sys.argv[1:]] or users()
    total = 0
    for f in files:
        bad = check(f)
        print(f"{os.path.basename(f)}: {len(bad)} compiler-generated M0 uses")
        for k, n, t in bad[:10]:
            print(f"   {k} line {n}: {t}")
        total += len(bad)
    sys.exit(1 if total else 0)
