"""Helper for keeping tools/probes/probe_switches.patch applicable after a product kernel changed.

    python tools/probes/regen_patch_hunks.py <file.hip> <edited copy>

Replaces the hunks of ml-unigen_amd/csrc/<file.hip> inside probe_switches.patch by `diff -u <product file> <edited copy>` -- the edited copy
being the product file with the probe switches re-inserted by hand (apply the old patch to a scratch copy of csrc, fix the rejects there).
The patch's header comment and the other files' hunks are kept as they are."""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
name, edited = sys.argv[1], sys.argv[2]
rel = f"ml-unigen_amd/csrc/{name}"
patch_path = os.path.join(ROOT, "tools", "probes", "probe_switches.patch")
text = open(patch_path).read()
new = subprocess.run(["diff", "-u", os.path.join(ROOT, rel), edited], capture_output=True, text=True).stdout
lines = new.split("\n")
assert lines[0].startswith("--- ") and lines[1].startswith("+++ "), "no differences?"
lines[0], lines[1] = f"--- a/{rel}", f"+++ b/{rel}"
new = "\n".join(lines)
# the section of this file in the patch: from its '--- a/<rel>' line to the next '--- a/' line (or the end)
parts = text.split("\n--- a/")
out = [parts[0]]
done = False
for sec in parts[1:]:
    if sec.startswith(rel[len(""):].replace("ml-unigen_amd/csrc/", "ml-unigen_amd/csrc/")) and sec.split("\n", 1)[0].strip() == rel:
        # keep a possible 'diff ...' line that precedes the next section (it belongs to the NEXT file): it is at the end of this section
        tail = ""
        body_lines = sec.split("\n")
        while body_lines and (body_lines[-1].startswith("diff ") or body_lines[-1] == ""):
            tail = body_lines.pop() + ("\n" + tail if tail else "")
        out.append(new[len("--- a/"):].rstrip("\n") + ("\n" + tail if tail.strip() else ""))
        done = True
    else:
        out.append(sec)
assert done, f"{rel} has no section in the patch"
open(patch_path, "w").write("\n--- a/".join(out).rstrip("\n") + "\n")
print("replaced the hunks of", rel)
