#!/usr/bin/env python3
"""Compares the gfx950 instruction streams of kernels between two sets of device assembly files (hipcc -S --cuda-device-only).

    python tools/compare_kernel_isa.py old.s -- new1.s new2.s ...

Used when volume.hip was split into one translation unit per kernel family (round 5): every kernel of the old file must come out
of its new file instruction for instruction.  Function names are compared with the anonymous-namespace and namespace qualifiers
removed (shared types moved from an anonymous namespace to `opv`); labels, comments and directives are dropped, branch targets are
replaced by their distance in instructions."""
import re
import sys


def kernels(path):
    out, name, body = {}, None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            out[name] = body
            name = None
            continue
        s = line.split(";")[0].strip()
        if not s or s.startswith("."):
            if re.match(r"^\.LBB\d+_\d+:", s):
                body.append(("label", s[:-1]))
            continue
        body.append(("inst", s))
    return out


def normalise(body):
    labels, n = {}, 0
    for kind, s in body:
        if kind == "label":
            labels[s] = n
        else:
            n += 1
    res, i = [], 0
    for kind, s in body:
        if kind != "inst":
            continue
        s = re.sub(r"\.LBB\d+_\d+", lambda m: "L%+d" % (labels.get(m.group(0), 0) - i), s)
        res.append(s)
        i += 1
    return res


def key(mangled):
    # drop namespace qualifiers of the function and of its parameter types: compare by kernel name + template arguments
    m = re.search(r"(k_[a-z_0-9]+?)(I[A-Za-z0-9_]+?E)?(Ev|E)", mangled)
    k = re.search(r"\d+(k_[a-z_]+)", mangled).group(1)
    t = re.search(r"\d+k_[a-z_]+(I(?:L[bij][0-9]+E)+E)?", mangled).group(1) or ""
    return k + t


def main():
    sep = sys.argv.index("--")
    old = {}
    for p in sys.argv[1:sep]:
        for n, b in kernels(p).items():
            old[key(n)] = normalise(b)
    new = {}
    for p in sys.argv[sep + 1:]:
        for n, b in kernels(p).items():
            new[key(n)] = (p, normalise(b))
    bad = 0
    for k in sorted(old):
        if k not in new:
            print("%-44s MISSING in the new files" % k); bad += 1; continue
        p, b = new[k]
        same = b == old[k]
        print("%-44s %6d instructions  %s  (%s)" % (k, len(old[k]), "identical" if same else "DIFFERENT (%d)" % len(b), p.split("/")[-1]))
        bad += not same
    for k in sorted(set(new) - set(old)):
        print("%-44s only in the new files (%s)" % (k, new[k][0].split("/")[-1]))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
