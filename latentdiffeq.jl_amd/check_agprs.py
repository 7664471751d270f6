"""Build-time guard for the hidden accumulator tiles of k_mlpb / k_mlpc (csrc/lde_mlpb.h).

Those kernels keep their weight-gradient tiles in AGPRs a[A0 : 256) that the compiler is not told about; nothing in LLVM makes
that a hard reservation (`amdgpu_num_vgpr` is a budget the allocator may exceed when it splits the file between VGPRs and AGPRs),
so every build is checked: in the device code of lde_mlp.o no instruction of those kernels may WRITE an AGPR at or above A0 except
the three forms the inline asm emits — `v_accvgpr_write_b32 aN, 0`, `v_mfma_f32_16x16x4_f32 a[R:R+3], v, v, a[R:R+3]` (same
range as C and D) — and nothing but `v_accvgpr_read_b32` and those MFMAs may read one. A violation fails the build."""
import os
import re
import subprocess
import sys
import tempfile

def llvm_bin() -> str:
    """llvm-objdump of the SAME toolchain hipcc drives: <rocm>/lib/llvm/bin, derived from hipcc's location or ROCM_PATH."""
    import shutil
    cands = []
    hipcc = shutil.which("hipcc")
    if hipcc:
        cands.append(os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc))), "lib", "llvm", "bin"))
    if os.environ.get("ROCM_PATH"):
        cands.append(os.path.join(os.environ["ROCM_PATH"], "lib", "llvm", "bin"))
    cands.append("/opt/rocm/lib/llvm/bin")
    for c in cands:
        if os.path.exists(os.path.join(c, "llvm-objdump")):
            return c
    raise FileNotFoundError("llvm-objdump not found next to hipcc, under ROCM_PATH or /opt/rocm: the hidden-AGPR check cannot run")


A0 = {"k_mlpb": 256 - 4 * 51, "k_mlpc": 256 - 4 * 26}     # mlpb::A0, mlpc::A0


def _regs(tok):
    m = re.fullmatch(r"a(\d+)", tok)
    if m:
        return range(int(m.group(1)), int(m.group(1)) + 1)
    m = re.fullmatch(r"a\[(\d+):(\d+)\]", tok)
    if m:
        return range(int(m.group(1)), int(m.group(2)) + 1)
    return range(0)


def check_object(obj: str) -> list:
    """Returns the list of violations ('kernel: instruction') in a host object with bundled gfx950 code."""
    LLVM = llvm_bin()
    with tempfile.TemporaryDirectory() as td:
        import glob
        import shutil
        cp = os.path.join(td, "o.o")
        shutil.copy(obj, cp)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", cp], check=True, capture_output=True)    # writes o.o.<n>.<target> next to it
        devs = glob.glob(cp + ".*gfx950*")
        if len(devs) != 1:
            return [f"{obj}: expected one gfx950 code object, found {len(devs)}"]
        dev = devs[0]
        dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", dev], check=True, capture_output=True, text=True).stdout
    bad, kern, lim, seen, scanned = [], None, None, set(), set()
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            name = m.group(1)
            kern, lim = None, None
            for k, a0 in A0.items():
                if f"6{k}I" in name:          # _ZN3lde6k_mlpbI…
                    seen.add(k)
                    # the kernels' template arguments end with <…, bool ADJ> (k_mlpb: SOLVER, DP, ACT, ADJ; k_mlpc: SOLVER, ACT, ADJ): the
                    # LAST bool of the mangled argument list is ADJ; only the adjoint instantiations own tiles
                    bools = re.findall(r"Lb([01])E", name)
                    if bools and bools[-1] == "1":
                        kern, lim = name, a0
                        scanned.add(k)
            continue
        if kern is None:
            continue
        ins = line.split("//")[0].strip()
        if not ins or " " not in ins:
            continue
        op, rest = ins.split(None, 1)
        toks = [t.strip() for t in rest.split(",")]
        hi = [t for t in toks if any(r >= lim for r in _regs(t))]
        if not hi:
            continue
        if op == "v_accvgpr_write_b32" and toks[1] == "0":
            continue
        if op == "v_accvgpr_read_b32" and hi == [toks[1]]:
            continue
        if op == "v_mfma_f32_16x16x4_f32" and toks[0] == toks[3] and hi == [toks[0], toks[3]] and min(_regs(toks[0])) >= lim:
            continue
        bad.append(f"{kern[:40]}: {ins}")
    for k in A0:
        if k not in seen:
            bad.append(f"{k}: no such kernel in {obj} (the check looks for the mangled name)")
        elif k not in scanned:   # (a changed template signature must not turn the check into a no-op)
            bad.append(f"{k}: no ADJ = true instantiation was scanned in {obj} — the name heuristic no longer matches the kernel's template arguments")
    return bad


if __name__ == "__main__":
    v = check_object(sys.argv[1])
    print("\n".join(v[:20]) if v else "hidden AGPR ranges respected")
    sys.exit(1 if v else 0)
