#!/usr/bin/env python3
"""Merge the scan_pmc.json files of two tools/profile_round.sh calls (the headline part and the `fullonly` part, each within gpurun's
20-minute limit) into profiles/scan_pmc.json: the union of their launch classes; sources and tag from the first.
    python tools/merge_pmc.py gpurun_out/r05a/scan_pmc.json gpurun_out/r05b/scan_pmc.json [-o profiles/scan_pmc.json]"""
import json
import sys

args = [a for a in sys.argv[1:] if a != "-o"]
out = "profiles/scan_pmc.json"
if "-o" in sys.argv:
    out = sys.argv[sys.argv.index("-o") + 1]
    args.remove(out)
base = json.load(open(args[0]))
for other in args[1:]:
    o = json.load(open(other))
    if o.get("sources") != base.get("sources"):
        print("WARNING: the two profiles were taken on different sources", file=sys.stderr)
    for k, v in o.get("classes", {}).items():
        base["classes"].setdefault(k, v)
    base["tag"] = base["tag"] + "+" + o.get("tag", "")
json.dump(base, open(out, "w"), indent=1)
print("wrote", out, "classes:", sorted(base["classes"]))
