#!/usr/bin/env python3
"""Conformance kit, checker side: diff the probe's output (probe.cpp built against the REAL Eigen / nanoGrid / nanoPCL /
fastdem) with what this repo's oracle assumes (expected.json).

    ./probe > probe.txt && python3 scripts/conformance/check.py probe.txt

    python3 scripts/conformance/check.py probe.txt --skip regions     # (this repo's own run of the probe against its mirror)

Per group: OK, or the first lines that differ.  For `move` it says WHICH of the two readings of GridMap::move() the real
library implements — all layers cleared in the vacated strips (this repo's default) or the basic layers only (engine
option `move_clear_basic` = 1, ElevationMap::setMoveClearBasic in the C++ mirror).  Exit code 0 iff every group matches
(move: either reading)."""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    skip = set()
    if "--skip" in sys.argv:
        skip = set(sys.argv[sys.argv.index("--skip") + 1].split(","))
    exp = json.load(open(os.path.join(HERE, "expected.json")))
    got = [l.rstrip("\n") for l in open(sys.argv[1]) if l.strip()]
    by = {}
    for l in got:
        by.setdefault(l.split(" ", 1)[0], []).append(l)
    groups = {"index": by.get("index", []) + by.get("position", []), "regions": by.get("region", []) + by.get("cells", []),
              "colors": by.get("color", []), "pre": by.get("pre", []) + by.get("map", [])}
    bad = 0

    def diff(name, g, e):
        ge, ee = sorted(g), sorted(e)
        if ge == ee:
            print(f"{name:10s} OK   ({len(e)} facts)")
            return True
        miss = [l for l in ee if l not in set(ge)]
        extra = [l for l in ge if l not in set(ee)]
        print(f"{name:10s} DIFF ({len(miss)} of {len(e)} facts differ)")
        for l in miss[:6]:
            key = " ".join(l.split(" ")[:2])
            other = [x for x in extra if x.startswith(key + " ")]
            print(f"    assumed : {l[:150]}")
            print(f"    library : {(other[0] if other else '(no such line)')[:150]}")
        return False

    for name in ("index", "regions", "colors", "pre"):
        if name in skip:
            print(f"{name:10s} skipped")
            continue
        bad += not diff(name, groups[name], exp[name])
    mv = by.get("move", [])
    if sorted(mv) == sorted(exp["move_all"]):
        print("move       OK   the vacated strips clear EVERY layer (this repo's default)")
    elif sorted(mv) == sorted(exp["move_basic"]):
        print("move       OK   the vacated strips clear the BASIC layers only: set the engine option move_clear_basic = 1 "
              "(ElevationMap::setMoveClearBasic(true) in the C++ mirror)")
    else:
        bad += 1
        diff("move", mv, exp["move_all"])
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
