"""Registers / LDS / scratch of every kernel in the given objects (from the code object's metadata notes)."""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(__file__))
import isa_hazard_lint as lint

for path in sys.argv[1:]:
    with tempfile.TemporaryDirectory() as tmp:
        co = lint.device_code(path, tmp)
        text = subprocess.run([os.path.join(lint.LLVM, "llvm-readelf"), "--notes", co], stdout=subprocess.PIPE, text=True).stdout
    cur = {}
    for line in text.splitlines():
        m = re.match(r"\s*-?\s*\.(name|vgpr_count|agpr_count|sgpr_count|private_segment_fixed_size|group_segment_fixed_size|vgpr_spill_count):\s*(.+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "name" and cur.get("vgpr_count") is not None:
            pass
        cur[k] = v
        if all(x in cur for x in ("name", "vgpr_count", "agpr_count", "sgpr_count", "private_segment_fixed_size")) and k in ("vgpr_count", "name", "vgpr_spill_count"):
            pass
    # the notes list one map per kernel: re-parse by blocks
    for blk in re.split(r"\n\s*- \.agpr_count:", text)[1:]:
        blk = ".agpr_count:" + blk
        g = lambda key: (re.search(r"\." + key + r":\s*(\S+)", blk) or [None, "?"])[1]
        print(os.path.basename(path), g("name")[:60], "vgpr", g("vgpr_count"), "agpr", g("agpr_count"), "sgpr", g("sgpr_count"),
              "scratch", g("private_segment_fixed_size"), "spill", g("vgpr_spill_count"))
