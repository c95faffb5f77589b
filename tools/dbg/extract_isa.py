"""Development aid: disassemble the gfx950 code object inside a built library (python tools/dbg/extract_isa.py LIB OUT.s)."""
import struct, subprocess, sys, tempfile
lib, out = sys.argv[1], sys.argv[2]
d = open(lib, "rb").read()
i = d.find(b"__CLANG_OFFLOAD_BUNDLE__")
n = struct.unpack_from("<Q", d, i + 24)[0]; o = i + 32
for _ in range(n):
    off, size, tl = struct.unpack_from("<QQQ", d, o); o += 24
    t = d[o:o + tl].decode(); o += tl
    if "gfx950" in t:
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(d[i + off:i + off + size]); f.flush()
            open(out, "w").write(subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", f.name], capture_output=True, text=True).stdout)
