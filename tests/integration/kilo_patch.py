#!/usr/bin/env python
"""The reference-side patch of INTEGRATION.md section 2 as something that can be applied and compiled.

tests/integration/kilo_hip.ed holds, for each of the four files of /root/reference/legkilo/src it touches, the sha256 of the file it was
made from and an ed-style edit script (the form of `diff -e`: line ranges + the NEW text only - no line of the reference is stored in this repo):

    core/slam/KILO.h       eskf_ / map_manager_  ->  one KiloPath (one device handle); the fused-scan switch
    core/slam/KILO.cc      initializeFromYaml creates the KiloPath; predictUpdatePoint / Imu / KinImu bodies (KILO.cc:108-314) become one call
                           each; first frame through the same StateInitial + BuildVoxelMap calls; optional one-call bucket loop
    core/slam/eskf.h       becomes a forwarding header to leg-kilo_amd/host/legkilo_host_eigen.hpp  (core/slam/eskf.cc leaves the build)
    core/slam/voxel_map.h  the same                                                                  (core/slam/voxel_map.cc leaves the build)

    python tests/integration/kilo_patch.py apply <reference src dir> <out dir>     patched copies of the four files under <out dir>
    python tests/integration/kilo_patch.py make  <orig dir> <new dir>              (re)generate kilo_hip.ed from an edited copy
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ED = os.path.join(HERE, "kilo_hip.ed")
FILES = ["core/slam/KILO.h", "core/slam/KILO.cc", "core/slam/eskf.h", "core/slam/voxel_map.h"]


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def parse(text):
    """-> {relative path: (sha256 of the original, [ed script lines])}"""
    out, cur = {}, None
    for line in text.split("\n"):
        if line.startswith("=== "):
            _, rel, digest = line.split()
            cur = out.setdefault(rel, (digest.split("=", 1)[1], []))[1]
        elif cur is not None:
            cur.append(line)
    return out


def apply_ed(lines, script):
    """Applies a `diff -e` script (commands a / c / d on 1-based line ranges, bottom-up; text blocks end with a lone '.')."""
    i = 0
    while i < len(script):
        cmd = script[i]
        i += 1
        if not cmd:
            continue
        op, rng = cmd[-1], cmd[:-1]
        assert op in "acd", cmd
        a, b = (rng.split(",") + [rng])[:2] if "," in rng else (rng, rng)
        a, b = int(a), int(b)
        text = []
        if op in "ac":
            while script[i] != ".":
                text.append(script[i])
                i += 1
            i += 1
        if op == "a":
            lines[a:a] = text
        elif op == "c":
            lines[a - 1:b] = text
        else:
            del lines[a - 1:b]
    return lines


def apply(ref_src, out_dir):
    spec = parse(open(ED, encoding="utf-8").read())
    assert sorted(spec) == sorted(FILES), sorted(spec)
    for rel, (digest, script) in spec.items():
        src = os.path.join(ref_src, rel)
        got = sha(src)
        if got != digest:
            raise SystemExit(f"{src}: sha256 {got[:16]}... is not the file kilo_hip.ed was made from ({digest[:16]}...)")
        lines = open(src, encoding="utf-8").read().split("\n")
        dst = os.path.join(out_dir, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        open(dst, "w", encoding="utf-8").write("\n".join(apply_ed(lines, script)))
    return [os.path.join(out_dir, rel) for rel in FILES]


def make(orig, new):
    """Edit scripts in `diff -e` form (bottom-up, so that line numbers stay those of the original), made with difflib on the same
    split("\n") line lists apply() works on - files without a final newline included."""
    import difflib

    parts = []
    for rel in FILES:
        a = open(os.path.join(orig, rel), encoding="utf-8").read().split("\n")
        b = open(os.path.join(new, rel), encoding="utf-8").read().split("\n")
        script = []
        for tag, i1, i2, j1, j2 in reversed(difflib.SequenceMatcher(None, a, b, autojunk=False).get_opcodes()):
            if tag == "equal":
                continue
            assert "." not in b[j1:j2], "a lone '.' line in the new text is not supported"
            rng = str(i1 + 1) if i2 - i1 == 1 else f"{i1 + 1},{i2}"
            if tag == "delete":
                script.append(rng + "d")
            elif tag == "insert":
                script += [f"{i1}a"] + b[j1:j2] + ["."]
            else:
                script += [rng + "c"] + b[j1:j2] + ["."]
        parts.append(f"=== {rel} sha256={sha(os.path.join(orig, rel))}\n" + "\n".join(script) + "\n")
    open(ED, "w", encoding="utf-8").write("".join(parts))


if __name__ == "__main__":
    if len(sys.argv) == 4 and sys.argv[1] == "apply":
        for p in apply(sys.argv[2], sys.argv[3]):
            print(p)
    elif len(sys.argv) == 4 and sys.argv[1] == "make":
        make(sys.argv[2], sys.argv[3])
    else:
        raise SystemExit(__doc__)
