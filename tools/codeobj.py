#!/usr/bin/env python3
"""tools/codeobj.py <lib.so> [kernel substring]: register / scratch / LDS figures of the gfx950 code object inside an engine library
(the offload bundle is cut out by hand; llvm-readelf --notes prints the kernel descriptors' metadata)."""
import re
import struct
import subprocess
import sys
import tempfile

lib = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
data = open(lib, "rb").read()
i = data.find(b"__CLANG_OFFLOAD_BUNDLE__")
n = struct.unpack_from("<Q", data, i + 24)[0]
off = i + 32
co = None
for _ in range(n):
    o, s, tl = struct.unpack_from("<QQQ", data, off)
    off += 24
    t = data[off:off + tl]
    off += tl
    if b"gfx950" in t:
        co = data[i + o:i + o + s]
with tempfile.NamedTemporaryFile(suffix=".co") as f:
    f.write(co)
    f.flush()
    out = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], text=True)
cur = {}
rows = []
for line in out.splitlines():
    m = re.match(r"\s*-?\s*\.(\w+):\s*(\S+)", line)
    if not m:
        continue
    k, v = m.groups()
    if k == "agpr_count" and cur.get("name"):
        rows.append(cur)
        cur = {}
    if k in ("agpr_count", "group_segment_fixed_size", "name", "private_segment_fixed_size", "sgpr_count", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count"):
        cur[k] = v
if cur.get("name"):
    rows.append(cur)
for r in rows:
    if want in r.get("name", ""):
        print(f"{r['name'][:48]:48s} vgpr {r.get('vgpr_count'):>4s} (spill {r.get('vgpr_spill_count'):>3s})  sgpr {r.get('sgpr_count'):>4s} (spill {r.get('sgpr_spill_count'):>3s})  "
              f"scratch {r.get('private_segment_fixed_size'):>4s} B  static LDS {r.get('group_segment_fixed_size'):>5s} B")
