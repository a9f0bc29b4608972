"""Static audit of the compiled kernels (no GPU): per kernel the registers, scratch bytes, LDS bytes and how the code waits for
memory — `s_waitcnt vmcnt(0)` (a FULL wait: on gfx950 vmcnt counts loads AND stores in order, so a full wait in front of a use of
a loaded register is also a wait for every older store) against counted waits `vmcnt(k > 0)`, next to the numbers of global
loads / stores / LDS-DMA instructions.  Kernels whose epilogue or streaming loop has one full wait per store group are the
candidates of DESIGN section 8 next-list item 1a (tools/study_epilogue_order).
    python tools/wait_audit.py [file.hip ...]      (default: every druglamp_amd/csrc/*.hip; ~1 min per file)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "druglamp_amd", "csrc")
files = sys.argv[1:] or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
rows = []
for f in files:
    src = f if os.path.isabs(f) else os.path.join(CSRC, f)
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                            "-I" + CSRC, "--cuda-device-only", "-S", src, "-o", out], capture_output=True, text=True)
        if r.returncode != 0:
            print("compile failed:", f, r.stderr[-400:]); continue
        s = open(out).read()
    meta = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
        b = m.group(2)
        g = lambda k: int(re.search(r"\.amdhsa_" + k + r" (\d+)", b).group(1))
        meta[m.group(1)] = (g("next_free_vgpr"), g("private_segment_fixed_size"), g("group_segment_fixed_size"))
    for name, (vgpr, scratch, lds) in meta.items():
        i = s.index("\n" + name + ":"); j = s.index(".Lfunc_end", i)
        body = s[i:j]
        full = len(re.findall(r"s_waitcnt[^\n]*vmcnt\(0\)", body))
        counted = len(re.findall(r"s_waitcnt[^\n]*vmcnt\((?!0\))", body))
        ld = len(re.findall(r"\n\s*(?:global|buffer|flat)_load_(?!lds)", body))
        st = len(re.findall(r"\n\s*(?:global|buffer|flat)_store", body))
        dma = len(re.findall(r"global_load_lds|buffer_load[^\n]* lds", body))
        dem = subprocess.run(["/usr/bin/c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(anonymous namespace\)::", "", dem)
        dem = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", dem)[:78]
        rows.append((f, dem, vgpr, scratch, lds, ld, st, dma, full, counted))
print("%-16s %-78s %5s %7s %7s %5s %6s %4s %9s %8s" % ("file", "kernel", "vgpr", "scratch", "lds", "loads", "stores", "dma", "vmcnt(0)", "vmcnt(k)"))
for r in sorted(rows, key=lambda r: (r[0], -r[8])):
    print("%-16s %-78s %5d %7d %7d %5d %6d %4d %9d %8d" % r)
