"""Registers / scratch / LDS of the kernels in an object file (from the code object's metadata notes).
    python abl/kernel_resources.py latentdiffeq.jl_amd/_obj/lde_mlp.o [name-substring …]"""
import glob, os, re, shutil, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
obj, pats = sys.argv[1], sys.argv[2:]
with tempfile.TemporaryDirectory() as td:
    cp = os.path.join(td, "o.o")
    shutil.copy(obj, cp)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", cp], check=True, capture_output=True)
    dev = glob.glob(cp + ".*gfx950*")[0]
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", dev], check=True, capture_output=True, text=True).stdout
cur = {}
rows = []
for line in notes.splitlines():
    m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2).strip()
    if k == "agpr_count" and cur.get("name"):
        pass
    if k in ("name", "vgpr_count", "agpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "symbol"):
        cur[k] = v
    if k == "wavefront_size":
        rows.append(cur)
        cur = {}
for r in rows:
    nm = r.get("name", "?")
    if pats and not any(p in nm for p in pats):
        continue
    cf = shutil.which("c++filt")
    dem = subprocess.run([cf, nm], capture_output=True, text=True).stdout.strip() if cf else nm
    print(f"{dem[:110]:110s} vgpr {r.get('vgpr_count','?'):>4} agpr {r.get('agpr_count','?'):>4} sgpr {r.get('sgpr_count','?'):>4} scratch {r.get('private_segment_fixed_size','?'):>5} lds {r.get('group_segment_fixed_size','?'):>6}")
