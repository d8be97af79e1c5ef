"""Copy the profile summaries a `gpurun ... bash tools/profile_round.sh <tag>` call left under gpurun_out/ into profiles/ and record,
INSIDE every file, which sources they were measured on: the git commit of the tree the snapshot was taken from (HEAD at
collection time, '+dirty' if the worktree differed) and the library's source stamp (sha256 over csrc/ + include/fdn.h + flags,
4dflownet_amd/build.py) that the GPU box wrote next to the profiles.  JSON files get a "_meta" key, text / csv files a leading
'# ...' line.  Refuses when the stamp measured on the box differs from the tree's current stamp (profiles of other kernels).
    python tools/collect_profiles.py r3
A `<tag>_gputest.txt` (tools/gputest_round.sh: the whole `-m gpu` suite + smoke()) is collected only when BOTH its library stamp and its
suite stamp (package python + GPU tests + oracle + entry points, build.suite_stamp) equal the tree's, and always as profiles/<round>_gputest.txt
(`r4c` -> `r4`): the tracked parity record of the round.   python tools/collect_profiles.py r4c --gputest-only"""
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
gputest_only = "--gputest-only" in sys.argv
build = importlib.import_module("4dflownet_amd.build")
src = os.path.join(ROOT, "gpurun_out")


def git_meta():
    head = subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    dirty = bool(subprocess.run(["git", "status", "--porcelain", "--", "4dflownet_amd", "include", "bench.py", "tests", "oracle",
                                 "__graft_entry__.py"], cwd=ROOT, capture_output=True, text=True).stdout.strip())
    return head + ("+dirty" if dirty else "")


def collect_gputest():
    """gpurun_out/<tag>_gputest.txt -> profiles/<round>_gputest.txt, refused unless it ran on the tree's kernels AND python/tests."""
    import re
    path = os.path.join(src, "%s_gputest.txt" % tag)
    if not os.path.exists(path):
        return False
    body = open(path).read()
    got = dict(re.findall(r"^(lib_source_stamp|suite_stamp) ([0-9a-f]{64})$", body, re.M))
    want = {"lib_source_stamp": build.source_stamp(), "suite_stamp": build.suite_stamp()}
    stale = [k for k in want if got.get(k) != want[k]]
    if stale and not force:
        sys.exit("gpurun_out/%s_gputest.txt ran on other sources than the tree's (%s differ): re-run tools/gputest_round.sh" % (tag, ", ".join(stale)))
    m = re.search(r"^(\d+) passed(?:, (\d+) skipped)?.* in [0-9.]+s", body, re.M)
    ok = m and "== pytest exit code 0" in body and "== smoke exit code 0" in body
    if not ok and not force:
        sys.exit("gpurun_out/%s_gputest.txt is not a green run (pytest / smoke exit codes, summary line)" % tag)
    rnd = re.match(r"r\d+", tag).group(0)
    dst = os.path.join(ROOT, "profiles", "%s_gputest.txt" % rnd)
    with open(dst, "w") as f:
        f.write("# commit %s lib_source_stamp %s suite_stamp %s%s\n# %s; collected by tools/collect_profiles.py %s\n" % (
            git_meta(), got.get("lib_source_stamp"), got.get("suite_stamp"), " STALE" if stale else "",
            m.group(0) if m else "NOT GREEN", tag))
        f.write(body)
    print("collected %s (%s)" % (os.path.relpath(dst, ROOT), m.group(0) if m else "not green"))
    return True


if gputest_only:
    sys.exit(0 if collect_gputest() else "no gpurun_out/%s_gputest.txt" % tag)
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
# only the files of the gpurun call that wrote the stamp (gpurun_out/ accumulates over calls: an older call's files were measured on other
# sources and must not be re-labelled with this commit); files merged back from one call carry times within seconds of each other
t_call = os.path.getmtime(stamp_file)
for path in sorted(glob.glob(os.path.join(src, tag + "_*"))):
    name = os.path.basename(path)
    if name.endswith(("_prof_bench.log", "_source_stamp.txt", "_gputest.txt")) or os.path.isdir(path):
        continue
    if abs(os.path.getmtime(path) - t_call) > 600 and "--all" not in sys.argv:
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
collect_gputest()
