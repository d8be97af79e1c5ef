"""Copy the profile summaries a `gpurun ... bash tools/profile_round.sh <tag>` call left under gpurun_out/ into profiles/ and record,
INSIDE every file, which sources they were measured on: the git commit of the tree the snapshot was taken from (HEAD at
collection time, '+dirty' if the worktree differed) and the library's source stamp (sha256 over csrc/ + include/fdn.h + flags,
4dflownet_amd/build.py) that the GPU box wrote next to the profiles.  JSON files get a "_meta" key, text / csv files a leading
'# ...' line.  Refuses when the stamp measured on the box differs from the tree's current stamp (profiles of other kernels).
    python tools/collect_profiles.py r3"""
import glob
import importlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag = sys.argv[1] if len(sys.argv) > 1 else "r3"
force = "--force" in sys.argv
build = importlib.import_module("4dflownet_amd.build")
src = os.path.join(ROOT, "gpurun_out")
stamp_file = os.path.join(src, "%s_source_stamp.txt" % tag)
box_stamp = open(stamp_file).read().split()[-1] if os.path.exists(stamp_file) else None
tree_stamp = build.source_stamp()
if box_stamp != tree_stamp and not force:
    sys.exit("profiles under gpurun_out/%s_* were measured on source stamp %s, the tree is at %s: re-run tools/profile_round.sh "
             "(or pass --force to collect them anyway, marked stale)" % (tag, str(box_stamp)[:16], tree_stamp[:16]))
head = subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
dirty = bool(subprocess.run(["git", "status", "--porcelain", "--", "4dflownet_amd", "include", "bench.py"], cwd=ROOT,
                            capture_output=True, text=True).stdout.strip())
meta = {"commit": head + ("+dirty" if dirty else ""), "lib_source_stamp": box_stamp, "stale": box_stamp != tree_stamp,
        "collected_by": "tools/collect_profiles.py %s" % tag}
line = "# commit %s lib_source_stamp %s%s\n" % (meta["commit"], box_stamp, " STALE (tree at %s)" % tree_stamp[:16] if meta["stale"] else "")
n = 0
for path in sorted(glob.glob(os.path.join(src, tag + "_*"))):
    name = os.path.basename(path)
    if name.endswith(("_prof_bench.log", "_source_stamp.txt")) or os.path.isdir(path):
        continue
    dst = os.path.join(ROOT, "profiles", name)
    if name.endswith(".json"):
        text = open(path).read().strip()
        if not text:
            continue
        d = json.loads(text.splitlines()[-1] if text.count("\n") and not text.lstrip().startswith("{\n") else text)
        d["_meta"] = meta
        with open(dst, "w") as f:
            json.dump(d, f, indent=1 if "pmc_traffic" in name else None)
            f.write("\n")
    else:
        body = open(path).read()
        with open(dst, "w") as f:
            f.write(line + body)
    n += 1
print("collected %d files into profiles/ (%s)" % (n, line.strip()))
